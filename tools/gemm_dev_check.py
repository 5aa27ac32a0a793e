"""Development check for the pinned 256x256 body: bit-exact on small integers for all four operand layouts and a few K
(1, 2, 3, many k-tiles), then timing at 4096^3.  Works with a TOPS_GEMM_DEV build of gemm_f32_mfma.hip."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd.hipt import HipT
T = HipT(0)
rng = np.random.default_rng(5)
bad = 0
shapes = [(4096, 16, 4096), (4096, 32, 4096), (4096, 48, 4096), (4096, 1024, 4096), (4096, 4096, 4096)]
if len(sys.argv) > 1 and sys.argv[1] == "edge":
    shapes += [(4000, 1024, 4000), (4088, 2048, 4092)]
for (m, k, n) in shapes:
    a = rng.integers(-3, 4, (m, k)).astype(np.float32); b = rng.integers(-3, 4, (k, n)).astype(np.float32)
    want = a @ b
    for ta in (0, 1):
        for tb in (0, 1):
            da = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
            db = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
            got = T.gmul(1, 1, 1, da, db).numpy()
            ok = np.array_equal(got, want)
            bad += not ok
            print("m%d k%d n%d ta%d tb%d %s" % (m, k, n, ta, tb, "exact" if ok else "WRONG (%d elements)" % int((got != want).sum())))
print("bad", bad)
for ta in (0, 1):
    for tb in (0, 1):
        m = k = n = 4096
        a = T.genRand((k, m) if ta else (m, k), "uniform", -1, 1, 1); b = T.genRand((n, k) if tb else (k, n), "uniform", -1, 1, 2)
        if ta: a = T.transp(a)
        if tb: b = T.transp(b)
        for _ in range(150): T.gmul(1, 1, 1, a, b)
        T.sync(); T.timer_start()
        for _ in range(100): T.gmul(1, 1, 1, a, b)
        ms = T.timer_stop() / 100
        print("ta%d tb%d 4096^3 %.4f ms %.2f TF" % (ta, tb, ms, 2.0 * m * k * n / ms / 1e9))
