import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
from tensor_ops_amd.hipt import HipT
T = HipT(0, dtype=np.float64)
rng = np.random.default_rng(2)
for (m, k, n) in ((1000, 992, 1001), (517, 64, 300), (2000, 3001, 2000)):
    a = rng.integers(-3, 4, size=(m, k)).astype(np.float64); b = rng.integers(-3, 4, size=(k, n)).astype(np.float64)
    for ta in (0, 1):
        for tb in (0, 1):
            da = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
            db = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
            print(m, k, n, ta, tb, "exact" if np.array_equal(T.gmul(1, 1, 1, da, db).numpy(), a @ b) else "WRONG")
for (m, k, n) in ((1000, 992, 1000), (2000, 3008, 2000), (4100, 4096, 4100)):
    A = T.genRand((m, k), "uniform", -1, 1, 1); B = T.genRand((k, n), "uniform", -1, 1, 2)
    for _ in range(5): T.gmul(1, 1, 1, A, B)
    T.sync(); T.timer_start()
    for _ in range(10): T.gmul(1, 1, 1, A, B)
    ms = T.timer_stop() / 10
    print("f64 %dx%dx%d: %.3f ms %.1f TF" % (m, k, n, ms, 2.0 * m * k * n / ms / 1e9))
