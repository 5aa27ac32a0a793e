"""Random GEMM extents / layouts / epilogue scales through gmul, bit-exact against numpy on small integers
(every routing decision of run_gemm: tile shapes, carves, edge tiles, split-K, K tails, the short-K and small kernels).
usage: gemm_fuzz.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd.hipt import HipT
from tools.mismatch_report import same
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
DT = np.float64 if os.environ.get("FUZZ_DTYPE") == "f64" else np.float32   # (fp64: the same routes' fp64 kernels)
T = HipT(0, dtype=DT) if DT is np.float64 else HipT(0)
special = [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257, 384, 500, 511, 512, 513,
           640, 768, 1000, 1023, 1024, 1025, 1280, 1536, 2000, 2047, 2048, 2049, 2304, 2560, 3000, 4096, 4100]
def dim():
    r = rng.random()
    if r < 0.5:
        return int(special[int(rng.integers(len(special)))])
    if r < 0.8:
        return int(rng.integers(1, 600))
    return int(rng.integers(600, 3200))
bad = 0
for case in range(n_cases):
    M, K, N = dim(), dim(), dim()
    while M * K + K * N + M * N > 60e6 or M * N * K > 3e10:
        M, K, N = dim(), dim(), dim()
    ta, tb = bool(rng.integers(2)), bool(rng.integers(2))
    a = rng.integers(-2, 3, (M, K)).astype(DT)
    b = rng.integers(-2, 3, (K, N)).astype(DT)
    A = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
    B = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
    want = (a.astype(np.float64) @ b.astype(np.float64)).astype(DT)
    Cd = T.gmul(1, 1, 1, A, B)
    got = Cd.numpy()
    ok = same(got, want, a=a, b=b, tool='gemm_fuzz', seed=seed, case=case, M=M, K=K, N=N, ta=ta, tb=tb, dtype=DT.__name__)
    if not ok:
        bad += 1
        nz = np.argwhere(got != want)
        print("MISMATCH", case, (M, K, N), "ta", ta, "tb", tb, "count", len(nz), "first", nz[:3].tolist())
        if os.environ.get("FUZZ_DIAG"):
            # what kind of failure: are the operands intact on the device?  does the same launch give the right answer now?
            # is the host's own reference right (an exact integer product, no BLAS)?
            ah = A.numpy(); bh = B.numpy()
            a_bad = np.unique(np.nonzero(ah != a)[0]) if ah.shape == a.shape else "shape"
            b_bad = np.unique(np.nonzero(bh != b)[1 if tb else 0]) if bh.shape == b.shape else "shape"
            got2 = Cd.numpy()    # the SAME device result downloaded again: a transfer that went wrong, or a wrong result?
            again = T.gmul(1, 1, 1, A, B).numpy()
            exact = (a.astype(np.int64) @ b.astype(np.int64)).astype(DT) if M * N * K < 4e9 else None
            msg = ("DIAG gemm_fuzz seed %d case %d %s ta %s tb %s: rows %s cols %s | device A differs from host A in rows %s | device B in %s %s | "
                   "second download of the same result right %s, identical to the first %s | relaunch right %s, relaunch identical to first %s | host reference exact %s | first result exact %s"
                   % (seed, case, (M, K, N), ta, tb, np.unique(nz[:, 0])[:12].tolist(), np.unique(nz[:, 1])[:12].tolist(),
                      a_bad[:12].tolist() if not isinstance(a_bad, str) else a_bad, "cols" if not tb else "rows of B^T",
                      b_bad[:12].tolist() if not isinstance(b_bad, str) else b_bad,
                      bool(np.array_equal(got2, want)), bool(np.array_equal(got2, got)),
                      bool(np.array_equal(again, want)), bool(np.array_equal(again, got)),
                      None if exact is None else bool(np.array_equal(exact, want)), None if exact is None else bool(np.array_equal(exact, got))))
            print(msg, flush=True)
            d = os.environ.get("TOPS_MISMATCH_DIR")
            if d:
                os.makedirs(d, exist_ok=True)
                with open(os.path.join(d, "diag_%d.txt" % os.getpid()), "a") as f:
                    f.write(msg + "\n")
    del A, B, Cd
print("cases", n_cases, "mismatches", bad)
