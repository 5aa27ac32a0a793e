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
    print(tag, m, k, n, ta, tb, "bad", bad.sum(), "rows", np.unique(np.nonzero(bad)[0])[:20], "cols", np.unique(np.nonzero(bad)[1])[:20])
    if bad.any():
        i, j = np.argwhere(bad)[0]
        print("  first", i, j, got[i, j], want[i, j])
        print("  got row ", got[i, :16]); print("  want row", want[i, :16])
    return got, want
for (m, k, n) in [(64, 64, 64), (64, 128, 64)]:
    for ta, tb in [(0, 0), (0, 1), (1, 1)]:
        a = np.ones((m, k), np.float32)
        b = (np.arange(k)[:, None] * 0 + np.arange(n)[None, :]).astype(np.float32)
        run(m, k, n, ta, tb, a, b, "b=col")
        b = (np.arange(k)[:, None] + 0 * np.arange(n)[None, :]).astype(np.float32)
        run(m, k, n, ta, tb, a, b, "b=k  ")
        # one-hot k: which k reach the output
        for kk in (0, 5, 17, k - 1):
            b = np.zeros((k, n), np.float32); b[kk, :] = 1
            run(m, k, n, ta, tb, a, b, "b=onehot k%d" % kk)
