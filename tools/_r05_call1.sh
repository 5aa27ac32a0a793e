export PYTHONPATH=$PWD
mkdir -p gpurun_out/r05_box1
O=gpurun_out/r05_box1
nproc > $O/nproc.txt; cat /proc/sys/kernel/numa_balancing >> $O/nproc.txt 2>&1
timeout 900 python tools/platform_probe.py --procs 12 --seconds 40 --out $O > $O/platform_probe.log 2>&1
timeout 600 python -m pytest tests/test_gpu_transfers.py -m gpu -q -x > $O/transfers_test.log 2>&1; tail -3 $O/transfers_test.log
# A: the runtime's pageable path, every download into a sentinel-filled destination, 12 loops in flight
rm -rf $O/stress_direct
TOPS_PINNED_STAGING=0 TOPS_DL_SENTINEL=1 timeout 1500 python tools/stress_suite.py --loops 12 --parallel 12 --conditions corun,hot --out $O/stress_direct --budget-s 1200 > $O/stress_direct.log 2>&1
tail -2 $O/stress_direct.log
# B: the product (pinned staging), 8 in flight
rm -rf $O/stress_staged
TOPS_DL_SENTINEL=1 timeout 1500 python tools/stress_suite.py --loops 16 --parallel 8 --conditions corun,hot --out $O/stress_staged --budget-s 1100 > $O/stress_staged.log 2>&1
tail -2 $O/stress_staged.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_box1/platform_probe.json"))
print(json.dumps(d["summary"]))
PY
