import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tensor_ops_amd.hipt import HipT
T = HipT(0)
np.set_printoptions(linewidth=250, suppress=True)
def run(m, k, n, ta, tb, a, b, tag):
    da = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
    db = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
    got = T.gmul(1, 1, 1, da, db).numpy()
    want = a @ b
    bad = got != want
    rows = np.unique(np.nonzero(bad)[0]); cols = np.unique(np.nonzero(bad)[1])
    print(tag, m, k, n, ta, tb, "bad", bad.sum(), "rows", rows[:24], "n", len(rows), "cols", cols[:24], "n", len(cols))
    if bad.any():
        i, j = np.argwhere(bad)[0]
        print("  first", i, j, got[i, j], want[i, j], " diff hist", np.unique((got - want)[bad])[:10])
for (m, k, n) in [(128, 64, 128), (128, 128, 128), (256, 256, 256), (512, 4096, 512)]:
    rng = np.random.default_rng(5)
    a = rng.integers(-2, 3, (m, k)).astype(np.float32); b = rng.integers(-2, 3, (k, n)).astype(np.float32)
    run(m, k, n, 0, 0, a, b, "rand")
    a1 = np.ones((m, k), np.float32)
    bc = (np.arange(k)[:, None] * 0 + np.arange(n)[None, :]).astype(np.float32)
    run(m, k, n, 0, 0, a1, bc, "A=1,B=col")
    ar = (np.arange(m)[:, None] + 0 * np.arange(k)[None, :]).astype(np.float32)
    b1 = np.ones((k, n), np.float32)
    run(m, k, n, 0, 0, ar, b1, "A=row,B=1")
