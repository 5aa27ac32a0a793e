"""The JSON line bench.py prints is a contract with the driver (one line; metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload, plus the `roofline` and
`cpu_baseline` objects).  The lines committed under profiles/ for this round are held to it here, and to their own
arithmetic -- no GPU needed; tests/test_gpu_multi.py checks a freshly produced line the same way."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r05_bench_driver_protocol*.json")))


def _load(path):
    txt = open(path).read().strip().splitlines()
    return json.loads(txt[-1])


def test_there_is_a_committed_line_for_this_round():
    assert LINES, "profiles/r05_bench_driver_protocol*.json"


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_committed_line_keeps_the_contract(path):
    d = _load(path)
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None            # BASELINE.md holds no published number for this metric
    assert d["dtype"] == "f32" and "workload" in d["config"] and "model" not in d["config"]
    # BASELINE.json's metric, letter for letter up to the arrows (the line is ASCII)
    assert d["metric"] == base["metric"].replace("\u2192", "->")
    # value and ms_per_step are the same measurement
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 2e-3
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 2e-3 and 0 < r["frac"] <= 1.0
    # achieved = algorithmic flops of one launch / its measured duration (2 * 4096^3 per launch)
    if "ms_per_launch" in r:
        assert abs(2.0 * 4096 ** 3 / (r["ms_per_launch"] * 1e-3) / 1e12 - r["achieved"]) < 0.5
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["unit"] == d["unit"]


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_committed_line_is_consistent_with_itself(path):
    d = _load(path)
    t = d.get("timing") or {}
    if t.get("ms_per_step_by_region"):
        regs = sorted(t["ms_per_step_by_region"])
        assert abs(regs[len(regs) // 2] - d["ms_per_step"]) < 1e-4      # value comes from the median region
    st = d.get("steady_state")
    if st:
        assert st["steps_per_s"] >= d["value"] * 0.98                   # longer regions never cost more per step
    sv = d.get("step_variants") or {}
    for head in ("softmax_crossEntropy", "logistic_squaredError"):
        if head in sv:
            assert sv[head]["launches_step"] == 3 and sv[head]["ms_step"] > 0.01
    step = d.get("step") or {}
    if "device_ms_per_step" in step and "algorithmic_flops" in step:
        assert abs(step["algorithmic_flops"] / (step["device_ms_per_step"] * 1e-3) / 1e12 - step["tflops"]) < 0.2
