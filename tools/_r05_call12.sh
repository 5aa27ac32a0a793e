export PYTHONPATH=$PWD
O=gpurun_out/r05_c12; mkdir -p $O
timeout 600 python tools/build_ab_lib.py gemm_t32.hip > $O/build_ab.log 2>&1
D=$PWD/tensor-ops_amd/build_ab
for pd in 2 1; do
echo "== pair on t32, PD $pd"; TOPS_HIP_LIB=$D/libtensorops_hip.so LD_LIBRARY_PATH=$D TOPS_GEMM_T32_PAIR=1 TOPS_T32_PD=$pd TOPS_T32_STAMPS=1 timeout 200 python tools/step_bench.py 400 2>&1 | tail -3
done
echo "== product"; timeout 200 python tools/step_bench.py 400 2>&1 | tail -1
(cd /tmp && export TMPDIR=/tmp && TOPS_HIP_LIB=$D/libtensorops_hip.so LD_LIBRARY_PATH=$D TOPS_GEMM_T32_PAIR=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_step -o step -- python $GRAFT_REPO_ROOT/tools/step_bench.py 400 > $GRAFT_REPO_ROOT/$O/prof_step.log 2>&1; cp $(find /tmp/rp_step -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/step_kernel_stats_pair_t32.csv)
head -4 $O/step_kernel_stats_pair_t32.csv
