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
    ok = same(z.numpy(), want.astype(DT), a=X, b=W.T, tool='kw_epilogue_fuzz', case=case, M=M, K=K, N=N, dtype=DT.__name__, note='want includes bias')
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
