"""fp64 GEMM kernels: whole-matrix bit-exact check on integer data for all four operand layouts, and timing."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd.hipt import HipT
T = HipT(0, dtype=np.float64)
rng = np.random.default_rng(11)
bad = 0
for (m, k, n) in ((4096, 288, 4096), (4096, 32, 2048), (2048, 1024, 4096), (4352, 160, 4096), (2048, 800, 2048), (3072, 208, 3072), (1280, 1024, 1152), (4352, 1024, 4096)):
    a = rng.integers(-3, 4, size=(m, k)).astype(np.float64); b = rng.integers(-3, 4, size=(k, n)).astype(np.float64)
    want = a @ b
    for ta in (0, 1):
        for tb in (0, 1):
            da = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
            db = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
            got = T.gmul(1, 1, 1, da, db).numpy()
            ok = np.array_equal(got, want)
            bad += not ok
            print(m, k, n, "ta%d tb%d" % (ta, tb), "exact" if ok else "WRONG (max |d| %.3g)" % np.abs(got - want).max())
print("bad =", bad)
n = 4096
A = T.genRand((n, n), "uniform", -1, 1, 1); B = T.genRand((n, n), "uniform", -1, 1, 2)
for name, (x, y) in (("ta0 tb0", (A, B)), ("ta1 tb1", (T.transp(A), T.transp(B))), ("ta0 tb1", (A, T.transp(B))), ("ta1 tb0", (T.transp(A), B))):
    for _ in range(10): T.gmul(1, 1, 1, x, y)
    T.sync(); T.timer_start()
    for _ in range(20): T.gmul(1, 1, 1, x, y)
    ms = T.timer_stop() / 20
    print("f64 gmul 4096^3 %s: %.3f ms  %.1f TF (%.1f%% of 78.6)" % (name, ms, 2.0 * n**3 / ms / 1e9, 2.0 * n**3 / ms / 1e9 / 78.6 * 100))
for (m, k, n) in ((4100, 4096, 4096), (4352, 4096, 4480), (6000, 2048, 6000), (4097, 1024, 4097)):
    A = T.genRand((m, k), "uniform", -1, 1, 1); B = T.genRand((k, n), "uniform", -1, 1, 2)
    for _ in range(5): T.gmul(1, 1, 1, A, B)
    T.sync(); T.timer_start()
    for _ in range(10): T.gmul(1, 1, 1, A, B)
    ms = T.timer_stop() / 10
    print("f64 gmul %dx%dx%d: %.3f ms  %.1f TF" % (m, k, n, ms, 2.0 * m * k * n / ms / 1e9))
for (m, k, n) in ((2048, 2048, 2048), (3072, 3072, 3072), (1536, 4096, 1536), (4352, 4096, 4352)):
    A = T.genRand((m, k), "uniform", -1, 1, 1); B = T.genRand((k, n), "uniform", -1, 1, 2)
    for _ in range(5): T.gmul(1, 1, 1, A, B)
    T.sync(); T.timer_start()
    for _ in range(10): T.gmul(1, 1, 1, A, B)
    ms = T.timer_stop() / 10
    print("f64 gmul %dx%dx%d: %.3f ms  %.1f TF" % (m, k, n, ms, 2.0 * m * k * n / ms / 1e9))
