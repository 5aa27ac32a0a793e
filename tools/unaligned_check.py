import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
from tensor_ops_amd.hipt import HipT
T = HipT(0)
rng = np.random.default_rng(3)
for (m, k, n) in ((1000, 1003, 1001), (4097, 4096, 4097), (4096, 4099, 4096)):
    a = rng.integers(-2, 3, size=(m, k)).astype(np.float32); b = rng.integers(-2, 3, size=(k, n)).astype(np.float32)
    for ta in (0, 1):
        for tb in (0, 1):
            da = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
            db = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
            got = T.gmul(1, 1, 1, da, db).numpy()
            want = (a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32)
            print(m, k, n, ta, tb, "exact" if np.array_equal(got, want) else "WRONG")
