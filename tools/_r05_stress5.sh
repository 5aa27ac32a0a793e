export PYTHONPATH=$PWD
O=gpurun_out/r05_box5
mkdir -p $O
nproc > $O/nproc.txt
timeout 500 python -u tools/platform_probe.py --procs 12 --seconds 25 --out $O > $O/platform_probe.log 2>&1
rm -rf $O/stress_staged
TOPS_DL_SENTINEL=1 timeout 2400 python -u tools/stress_suite.py --loops 24 --parallel 8 --conditions corun,hot --out $O/stress_staged --budget-s 1500 > $O/stress_staged.log 2>&1
tail -1 $O/stress_staged.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_box5/platform_probe.json"))
print(json.dumps(d["summary"]))
PY
