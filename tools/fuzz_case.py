"""Replay one case of tests/test_gpu_lazy_fuzz.py and print where recorded and eager execution differ."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_lazy_fuzz as F
from tensor_ops_amd.hipt import HipT
case = int(sys.argv[1]); dt = np.float32 if (len(sys.argv) < 3 or sys.argv[2] == "f32") else np.float64
T = HipT(0, dtype=dt)
fam2 = os.environ.get("FAMILY", "1") == "2"     # FAMILY=2: the training-step family
rng = np.random.default_rng(F.SEED + (1000003 if fam2 else 0) + case)
leaves, steps, kinds, sizes = (F.build_program2 if fam2 else F.build_program)(rng)
B = sizes["B"]
inputs = {}
for name, kind in leaves.items():
    shape = (((B,) if kind[2] else ()) + (kind[1],)) if kind[0] == "vec" else (kind[1], kind[2])
    inputs[name] = rng.uniform(0, 1, size=shape) if (fam2 and name == "y") else rng.uniform(-1, 1, size=shape)
produced = [st[1] for st in steps if st[0] not in ("copy", "copy_many")]
if fam2:
    produced = [n for n in produced if n.startswith(("nW", "nb", "gW", "gb", "dz", "e", "q", "dh"))] or produced
k = int(rng.integers(1, len(produced) + 1))
demand = [produced[i] for i in rng.permutation(len(produced))[:k]]
late = set(d for d in demand if rng.random() < 0.25)
print("sizes", sizes); 
for st in steps: print("  ", st)
print("demand", demand, "late", late)
eager = F.run_program(T, leaves, steps, inputs, demand, False, late)
os.environ["TOPS_LAZY_DEBUG"] = os.environ.get("DBG", "0")
lazy = F.run_program(T, leaves, steps, inputs, demand, True, late)
for name in demand:
    a, b = eager[name].astype(np.float64), lazy[name].astype(np.float64)
    bad = ~np.isclose(a, b, rtol=1e-5, atol=1e-5, equal_nan=True)
    print(name, a.shape, "max|a|", np.nanmax(np.abs(a)) if a.size else 0, "nonfinite eager/lazy", (~np.isfinite(a)).sum(), (~np.isfinite(b)).sum(), "mismatches", bad.sum())
    if bad.sum():
        idx = np.argwhere(bad)[:5]
        for i in idx: print("    at", tuple(i), a[tuple(i)], b[tuple(i)])
