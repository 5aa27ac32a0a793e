"""Random large extents through gmul, bit-exact on small integers: the routes of the pinned 256x256 / 256x128 bodies (whole
rounds, edge tiles run whole, carved blocks + border strips, hybrid stream-K, short K, one / odd numbers of k-tiles).
usage: pinned_fuzz.py [cases] [seed]      (FUZZ_DTYPE=f64 for the fp64 body)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd.hipt import HipT
from tools.mismatch_report import same
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
DT = np.float64 if os.environ.get("FUZZ_DTYPE") == "f64" else np.float32
T = HipT(0, dtype=DT) if DT is np.float64 else HipT(0)
bad = 0


def extent():
    r = rng.random()
    if r < 0.35:
        return int(256 * rng.integers(12, 26))            # whole tiles: 3072 .. 6400
    if r < 0.7:
        return int(4 * rng.integers(768, 1600))           # multiples of 4: edge tiles on the pinned body
    return int(rng.integers(3000, 6000))                  # anything: carved block + border strips


for case in range(n_cases):
    M, N = extent(), extent()
    K = int(16 * rng.integers(1, 40)) if rng.random() < 0.8 else int(rng.integers(16, 700))
    if rng.random() < 0.15:
        K = int(16 * rng.integers(64, 130))               # long enough for stream-K shares
    ta, tb = bool(rng.integers(2)), bool(rng.integers(2))
    a = rng.integers(-2, 3, (M, K)).astype(DT)
    b = rng.integers(-2, 3, (K, N)).astype(DT)
    A = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
    B = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
    got = T.gmul(1, 1, 1, A, B).numpy()
    want = a @ b
    if not same(got, want, a=a, b=b, tool='pinned_fuzz', seed=seed, case=case, M=M, K=K, N=N, ta=ta, tb=tb, dtype=DT.__name__):
        bad += 1
        print("MISMATCH M%d K%d N%d ta%d tb%d: %d elements" % (M, K, N, ta, tb, int((got != want).sum())))
print("cases", n_cases, "mismatches", bad)
