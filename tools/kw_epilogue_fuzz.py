"""The wave-split GEMM kernels' fused epilogues on random extents: a recorded `W x + b` (batched matVec + sumT), alone,
under logistic and under tanh, against numpy -- integers, so the pre-activation is exact; fp32 or fp64 (FUZZ_DTYPE=f64).
FUZZ_SPLIT=1: fp32 extents of 10 .. 170 tiles with a long K (several workgroups per tile, partial tiles met in the L2).
usage: kw_epilogue_fuzz.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd import hipt
from tensor_ops_amd.hipt import HipT
from tools.mismatch_report import same
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
_print = print


def print(*a, **kw):   # (diagnostics also go where the stress harness keeps what failed)
    _print(*a, **kw)
    d = os.environ.get("TOPS_MISMATCH_DIR")
    if d and a and str(a[0]).startswith("DIAG"):
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "diag_%d.txt" % os.getpid()), "a") as f:
            f.write(" ".join(str(v) for v in a) + "\n")



rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DT = np.float64 if os.environ.get("FUZZ_DTYPE") == "f64" else np.float32
T = HipT(0, dtype=DT) if DT is np.float64 else HipT(0)
tol = 1e-12 if DT is np.float64 else 2e-6
bad = 0
for case in range(n_cases):
    M = int(rng.integers(640, 2600)); N = int(rng.integers(130, 1500)); K = int(rng.integers(128, 900))
    if DT is np.float64:
        M = int(rng.integers(640, 1400)); N = int(rng.integers(130, 700))
    elif os.environ.get("FUZZ_SPLIT"):   # few 64x64 tiles and a long K: several workgroups per tile (gemm_kwave.hip, KS > 1)
        M = int(rng.integers(130, 1100)); N = int(rng.integers(130, max(131, min(1100, 10000 * 64 // M // 64))))
        K = int(rng.integers(600, 4200))
    W = rng.integers(-2, 3, (N, K)).astype(DT); X = rng.integers(-2, 3, (M, K)).astype(DT); b = rng.integers(-3, 4, N).astype(DT)
    want = X.astype(np.float64) @ W.T.astype(np.float64) + b
    dW, dX, db = T.put(W), T.put(X, batched=True), T.put(b)
    l0 = T.stats()["launches"]
    with T.memo():
        z = T.force(T.sumT([T.matVec(dW, dX), db], (N,)))
    nl = T.stats()["launches"] - l0
    zh = z.numpy()
    if os.environ.get("FUZZ_DIAG") and not np.array_equal(zh.reshape(M, N), want.astype(DT)):
        # what kind of failure: operands intact on the device?  the same bits on a second download?  right when launched again?
        wr = np.unique(np.nonzero(zh.reshape(M, N) != want.astype(DT))[0])
        x_ok = np.array_equal(dX.numpy().reshape(M, K), X); w_ok = np.array_equal(dW.numpy(), W); b_ok = np.array_equal(db.numpy(), b)
        z_again = np.array_equal(z.numpy(), zh)
        with T.memo():
            z2 = T.force(T.sumT([T.matVec(dW, dX), db], (N,)))
        relaunch = np.array_equal(z2.numpy().reshape(M, N), want.astype(DT))
        runs = np.split(wr, np.nonzero(np.diff(wr) > 1)[0] + 1)
        g2 = zh.reshape(M, N); w2 = want.astype(DT)
        # is a wrong row some OTHER row's right answer (rows swapped / an operand row read from elsewhere)?
        perm = []
        for r in wr[:8]:
            hit = np.nonzero((w2 == g2[r]).all(axis=1))[0]
            perm.append((int(r), [int(v) for v in hit[:3]]))
        xd = dX.numpy().reshape(M, K)
        xbad = np.unique(np.nonzero(xd != X)[0])
        print("DIAG   wrong rows %s ...; each equals the right answer of rows %s; device X differs from host X in rows %s; bias-only rows (X row zero)? %s"
              % ([int(v) for v in wr[:24]], perm, [int(v) for v in xbad[:24]], [bool(np.array_equal(g2[r], b)) for r in wr[:4]]), flush=True)
        print("DIAG case %d %s: %d wrong rows in %d runs %s | X %s W %s b %s on device | second download identical %s | relaunch right %s | launches %d | ptrs z %x X %x"
              % (case, (M, K, N), len(wr), len(runs), [(int(r[0]), int(r[-1])) for r in runs[:6]], x_ok, w_ok, b_ok, z_again, relaunch, nl,
                 z.ptr or 0, dX.ptr or 0), flush=True)
    ok = same(zh, want.astype(DT), a=X, b=W.T, tool='kw_epilogue_fuzz', case=case, M=M, K=K, N=N, dtype=DT.__name__, note='want includes bias')
    with T.memo():
        h = T.force(T.liftT(hipt.logistic_closure, [T.sumT([T.matVec(dW, dX), db], (N,))], key="kwf-logistic"))
    ok = ok and np.max(np.abs(h.numpy() - 1 / (1 + np.exp(-want)))) < tol
    with T.memo():
        t = T.force(T.liftT(lambda v: hipt.tanh(v[0]), [T.sumT([T.matVec(dW, dX), db], (N,))], key="kwf-tanh"))
    ok = ok and np.max(np.abs(t.numpy() - np.tanh(want))) < 10 * tol
    if not ok:
        bad += 1
        print("MISMATCH", case, (M, K, N), "launches", nl)
    del dW, dX, db, z, h, t
print("cases", n_cases, "mismatches", bad)
