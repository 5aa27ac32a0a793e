export PYTHONPATH=$PWD
O=gpurun_out/r05_c12; mkdir -p $O
timeout 200 python tools/step_bench.py 400 > $O/step_bench.txt 2>&1; tail -1 $O/step_bench.txt
TOPS_STEP_SEAM=0 timeout 200 python tools/step_bench.py 400 2>&1 | tail -1
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_step -o step -- python $GRAFT_REPO_ROOT/tools/step_bench.py 400 > $GRAFT_REPO_ROOT/$O/prof_step.log 2>&1; cp $(find /tmp/rp_step -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/step_kernel_stats.csv)
head -4 $O/step_kernel_stats.csv
timeout 200 python bench.py --no-aux --steps 20 --warmup 5 > $O/bench_noaux.json 2> $O/bench_noaux.err; python -c "
import json;d=json.load(open('$O/bench_noaux.json'));print(d['value'],d['ms_per_step'],d['step']['kernel_launches'])"
timeout 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_batch_rule.py tests/test_gpu_host_mirror.py tests/test_gpu_lazy.py tests/test_gpu_top_level.py tests/test_gpu_call_trace.py tests/test_gpu_online.py tests/test_gpu_fuzz_gemm.py -m gpu -q > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 900 python -m pytest tests/test_gpu_switches.py -m gpu -q -k "default or SEAM or everything_off or LAZY" > $O/tests_sw.log 2>&1; tail -3 $O/tests_sw.log
timeout 600 python tools/build_ab_lib.py gemm_t32.hip > $O/build_ab.log 2>&1
D=$PWD/tensor-ops_amd/build_ab
TOPS_HIP_LIB=$D/libtensorops_hip.so LD_LIBRARY_PATH=$D TOPS_T32_STAMPS=1 timeout 200 python tools/step_bench.py 400 2>&1 | tail -3
