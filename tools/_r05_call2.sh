export PYTHONPATH=$PWD
O=gpurun_out/r05_box2
mkdir -p $O
timeout 300 python tools/t32_check.py --time > $O/t32_check.txt 2>&1; tail -6 $O/t32_check.txt
timeout 200 python tools/step_bench.py 400 > $O/step_bench.txt 2>&1; tail -2 $O/step_bench.txt
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_step -o step -- python $GRAFT_REPO_ROOT/tools/step_bench.py 400 > $GRAFT_REPO_ROOT/$O/prof_step.log 2>&1; cd $GRAFT_REPO_ROOT
find $O/prof_step -name "*kernel_stats.csv" | head -1 | xargs -I{} head -8 {}
timeout 900 python -m pytest tests/test_gpu_transfers.py tests/test_gpu_btensor_route.py tests/test_gpu_batch_rule.py tests/test_gpu_top_level.py -m gpu -q -x > $O/new_tests.log 2>&1; tail -3 $O/new_tests.log
timeout 900 python -m pytest tests/test_gpu_switches.py -m gpu -q -x -k "default or KSPLIT or PINNED or SEAM or everything_off" > $O/switch_tests.log 2>&1; tail -3 $O/switch_tests.log
timeout 600 python -m pytest tests/test_gpu_multi.py -m gpu -q -x > $O/multi_tests.log 2>&1; tail -3 $O/multi_tests.log
timeout 600 python -u tools/platform_probe.py --procs 12 --seconds 30 --out $O > $O/platform_probe.log 2>&1
rm -rf $O/stress_staged
TOPS_DL_SENTINEL=1 timeout 900 python -u tools/stress_suite.py --loops 8 --parallel 8 --conditions corun,hot --out $O/stress_staged > $O/stress_staged.log 2>&1
tail -1 $O/stress_staged.log
rm -rf $O/stress_direct
TOPS_PINNED_STAGING=0 TOPS_DL_SENTINEL=1 timeout 900 python -u tools/stress_suite.py --loops 8 --parallel 8 --conditions corun,hot --out $O/stress_direct > $O/stress_direct.log 2>&1
tail -1 $O/stress_direct.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_box2/platform_probe.json"))
print(json.dumps(d["summary"]))
PY
