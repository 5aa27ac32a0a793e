"""Stream-K route: whole-matrix bit-exact check on integer data (all layouts) and timing against the plain route."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd.hipt import HipT
T = HipT(0)
rng = np.random.default_rng(12)
bad = 0
for (m, k, n) in ((3072, 320, 3072), (2304, 1024, 2304), (3840, 256, 3840), (1280, 2048, 2560), (4352, 272, 4352)):
    a = rng.integers(-2, 3, size=(m, k)).astype(np.float32); b = rng.integers(-2, 3, size=(k, n)).astype(np.float32)
    want = (a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32)
    for ta in (0, 1):
        for tb in (0, 1):
            da = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
            db = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
            for rep in range(2):
                got = T.gmul(1, 1, 1, da, db).numpy()
                ok = np.array_equal(got, want)
                bad += not ok
                if not ok:
                    d = np.argwhere(got != want)
                    print(m, k, n, ta, tb, "WRONG", len(d), "first", d[:3].tolist())
    print("shape", m, k, n, "done")
print("bad =", bad)
