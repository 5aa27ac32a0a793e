export PYTHONPATH=$PWD
O=gpurun_out/r05_c13; mkdir -p $O
timeout 600 python tools/build_ab_lib.py gemm_t32.hip > $O/build_ab.log 2>&1
D=$PWD/tensor-ops_amd/build_ab
for sk in 0 1 2; do
echo "== head skip $sk"
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp_step && TOPS_HIP_LIB=$D/libtensorops_hip.so LD_LIBRARY_PATH=$D TOPS_T32_HEAD_SKIP=$sk TOPS_T32_STAMPS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_step -o step -- python $GRAFT_REPO_ROOT/tools/step_bench.py 200 > $GRAFT_REPO_ROOT/$O/prof_$sk.log 2>&1; grep "t32\]" $GRAFT_REPO_ROOT/$O/prof_$sk.log; head -3 $(find /tmp/rp_step -name "*kernel_stats.csv" | head -1))
done
