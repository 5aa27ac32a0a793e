"""All four operand layouts of gmul on full-tile shapes against numpy fp64 (development check for kernel variants)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd.hipt import HipT
T = HipT(0)
rng = np.random.default_rng(3)
worst = 0.0
for (m, k, n) in ((512, 256, 768), (1024, 1024, 1024), (256, 16, 256), (4096, 4096, 4096)):
    a = rng.uniform(-1, 1, (m, k)).astype(np.float32); b = rng.uniform(-1, 1, (k, n)).astype(np.float32)
    want = a.astype(np.float64) @ b.astype(np.float64) if m < 4096 else None
    for ta in (0, 1):
        for tb in (0, 1):
            da = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
            db = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
            got = T.gmul(1, 1, 1, da, db).numpy()
            if want is None:
                rows = rng.integers(0, m, 8)
                ref = a[rows].astype(np.float64) @ b.astype(np.float64)
                err = np.linalg.norm(got[rows] - ref) / np.linalg.norm(ref)
            else:
                err = np.linalg.norm(got - want) / np.linalg.norm(want)
            worst = max(worst, err)
            print("m%d k%d n%d ta%d tb%d rel err %.2e" % (m, k, n, ta, tb, err), "OK" if err < 1e-5 else "FAIL")
print("worst", worst)
