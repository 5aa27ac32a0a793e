"""Learn-layer GEMM shapes (a tall batch of rows through 784 -> 300 -> 100 -> 10): bit-exact integer check of whichever route
serves them, on the layouts a layer uses.  usage: learn_check.py [M K N ...]   (LEARN_DTYPE=f64: the fp64 instance)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tensor_ops_amd.hipt import HipT
F64 = os.environ.get("LEARN_DTYPE") == "f64"   # LEARN_DTYPE=f64: the fp64 instance
T = HipT(0, dtype=np.float64) if F64 else HipT(0)
DT = np.float64 if F64 else np.float32
v = [int(x) for x in sys.argv[1:]] or [60000, 784, 300, 60000, 300, 100, 60000, 100, 10, 8192, 300, 100, 8200, 300, 100, 4104, 100, 10, 300, 4096, 100]
bad = 0
for i in range(0, len(v), 3):
    m, k, n = v[i:i + 3]
    for ta in (0, 1):
        for tb in (0, 1):
            rng = np.random.default_rng(m + 3 * k + 7 * n + ta * 2 + tb)
            a = rng.integers(-2, 3, size=(m, k)).astype(DT)
            b = rng.integers(-2, 3, size=(k, n)).astype(DT)
            da = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
            db = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
            l0 = T.stats()["launches"]
            got = T.gmul(1, 1, 1, da, db).numpy()
            nl = T.stats()["launches"] - l0
            want = (a.astype(np.float64) @ b.astype(np.float64)).astype(DT)
            ok = np.array_equal(got, want)
            bad += not ok
            if not ok or os.environ.get("LEARN_VERBOSE"):
                print("%s %d x %d x %d ta %d tb %d launches %d" % ("ok " if ok else "BAD", m, k, n, ta, tb, nl), flush=True)
print("learn_check mismatches", bad)
