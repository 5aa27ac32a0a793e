export PYTHONPATH=$PWD
O=gpurun_out/r05_final; mkdir -p $O
python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; tail -3 $O/gputest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_final/bench_driver.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "steady", d["steady_state"])
print("roofline", d["roofline"]["frac"], d["roofline"]["achieved"])
print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value","dtype","gpu_over_cpu")}, d.get("gpu_over_cpu_same_precision"))
print("cpu_b", d["cpu_baseline"]["host"].get("cpu_b_all_cores"))
print("fp64 c5", d["fp64"]["gmul_c5a"])
PY
bash tools/collect_profiles.sh r05 > $O/collect.log 2>&1; tail -3 $O/collect.log
