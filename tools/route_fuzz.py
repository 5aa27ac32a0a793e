"""Routing-boundary fuzz (round 6, last): GEMM shapes drawn around the thresholds the routing rules of gemm_kwave.hip / gemm_kw16.hip /
gemm_kwave_f64.hip / gemv.hip test -- tile counts of 1,024 / 2,048 workgroup-rounds, K of 16 / 64 / 128 / 512 / 2,048 / 3,072 / 8,192,
extents of 1 / 8 / 16 / 96 / 128 / 256 / 512 -- in all four operand layouts, with and without `beta * C`, exact on small integers.
   usage: route_fuzz.py cases seed     (ROUTE_DTYPE=f64: the fp64 instance)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tensor_ops_amd.hipb import HipB
F64 = os.environ.get("ROUTE_DTYPE") == "f64"
DT = np.float64 if F64 else np.float32
B = HipB(0, dtype=DT); T = B.T
cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
small = [1, 2, 7, 8, 9, 15, 16, 17, 31, 32, 63, 64, 95, 96, 100, 127, 128, 129, 255, 256, 300, 511, 512, 784]
ks = [1, 15, 16, 17, 32, 63, 64, 65, 100, 127, 128, 129, 256, 300, 511, 512, 513, 1024, 2047, 2048, 2049, 3071, 3072, 3073, 8191, 8192, 8200]
bad = 0
for c in range(cases):
    kind = rng.integers(0, 4)
    if kind == 0:      # many tiles around the rounds of 256 / 2,048
        t = int(rng.choice([1000, 1024, 1030, 1800, 2040, 2048, 2060, 2304, 3072, 4096, 5024, 6400]))
        tm = int(rng.choice([d for d in range(1, t + 1) if t % d == 0 and 4 <= d <= t // 4] or [32]))
        m, n = 64 * tm - int(rng.integers(0, 3)) * int(rng.integers(0, 40)), 64 * (t // tm) - int(rng.integers(0, 3)) * int(rng.integers(0, 40))
        k = int(rng.choice([16, 48, 64, 100, 128, 256, 300, 512, 520, 784]))
    elif kind == 1:    # one small extent beside a large one
        m, n = int(rng.choice(small)), int(rng.choice([1024, 2048, 4096, 10000, 30000]))
        if rng.integers(0, 2): m, n = n, m
        k = int(rng.choice(ks))
    elif kind == 2:    # a few tiles under a long K
        m, n = int(rng.choice(small)), int(rng.choice(small))
        k = int(rng.choice([1536, 2048, 3072, 4096, 8192, 8200, 20000, 60000]))
    else:              # anything
        m, n, k = int(rng.choice(small + [1000, 1024, 1100])), int(rng.choice(small + [1000, 1024, 1100])), int(rng.choice(ks))
    m, n, k = max(m, 1), max(n, 1), max(k, 1)
    if 2.0 * m * n * k > 6e9 or max(m * k, k * n, m * n) > 1.5e8:
        continue
    ta, tb, beta = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 3)) == 0
    a = rng.integers(-2, 3, (m, k)).astype(DT); b = rng.integers(-2, 3, (k, n)).astype(DT)
    da = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
    db = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
    want = a.astype(np.float64) @ b.astype(np.float64)
    if beta:
        cc = rng.integers(-4, 5, (m, n)).astype(DT)
        got = B.gemm(-2.0, da, db, (3.0, T.put(cc))).numpy()
        want = -2.0 * want + 3.0 * cc
    else:
        got = T.gmul(1, 1, 1, da, db).numpy()
    ok = np.array_equal(got.astype(np.float64), want)
    bad += not ok
    if not ok:
        print("BAD %d x %d x %d ta %d tb %d beta %d  max err %g" % (m, k, n, ta, tb, beta, np.max(np.abs(got - want))), flush=True)
print("route_fuzz cases %d seed %d mismatches %d" % (cases, seed, bad))
