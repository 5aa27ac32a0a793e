export PYTHONPATH=$PWD
O=gpurun_out/r05_c3
mkdir -p $O
timeout 300 python tools/t32_check.py --time > $O/t32_check.txt 2>&1; tail -5 $O/t32_check.txt
timeout 200 python tools/step_bench.py 400 > $O/step_bench.txt 2>&1; tail -1 $O/step_bench.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_step -o step -- python $GRAFT_REPO_ROOT/tools/step_bench.py 400 > $GRAFT_REPO_ROOT/$O/prof_step.log 2>&1; cp $(find /tmp/rp_step -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/step_kernel_stats.csv)
head -6 $O/step_kernel_stats.csv
timeout 600 python tools/build_ab_lib.py gemm_t32.hip > $O/build_ab.log 2>&1; tail -1 $O/build_ab.log
D=$PWD/tensor-ops_amd/build_ab; TOPS_HIP_LIB=$D/libtensorops_hip.so LD_LIBRARY_PATH=$D TOPS_T32_STAMPS=1 timeout 200 python tools/step_bench.py 400 > $O/step_stamps.txt 2>&1; tail -4 $O/step_stamps.txt
TOPS_HIP_LIB=$D/libtensorops_hip.so LD_LIBRARY_PATH=$D TOPS_GEMM_T32=0 timeout 200 python tools/step_bench.py 400 > $O/step_t32_off.txt 2>&1; tail -1 $O/step_t32_off.txt
timeout 200 python tools/c5_f64_probe.py > $O/c5_f64.txt 2>&1; tail -3 $O/c5_f64.txt
timeout 1200 python -m pytest tests/test_gpu_transfers.py tests/test_gpu_full_size.py tests/test_gpu_batch_rule.py tests/test_gpu_host_mirror.py tests/test_gpu_lazy.py tests/test_gpu_fuzz_gemm.py tests/test_gpu_btensor_route.py -m gpu -q > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 600 python -m pytest tests/test_gpu_f64.py -m gpu -q -k "short_k or mid_size or c5 or config5" > $O/tests_f64.log 2>&1; tail -3 $O/tests_f64.log
cp gpurun_out/r05_btensor_route.json $O/ 2>/dev/null
