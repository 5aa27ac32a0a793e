"""The pinned 256x256 body's fused way out (bias, bias + logistic) on large layers: a recorded `W x + b` (batched matVec +
sumT), alone and under logistic, against numpy -- integers, so the pre-activation is exact.  Full tiles and edge tiles."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd import hipt
from tensor_ops_amd.hipt import HipT
from tools.mismatch_report import same
T = HipT(0)
rng = np.random.default_rng(9)
bad = 0
for (M, K, N) in ((4096, 64, 4096), (4096, 272, 4096), (4000, 288, 4000), (4352, 1024, 4352), (8192, 48, 2048)):
    W = rng.integers(-2, 3, (N, K)).astype(np.float32); X = rng.integers(-2, 3, (M, K)).astype(np.float32)
    b = rng.integers(-3, 4, N).astype(np.float32)
    want = X.astype(np.float64) @ W.T.astype(np.float64) + b
    dW, dX, db = T.put(W), T.put(X, batched=True), T.put(b)
    l0 = T.stats()["launches"]
    with T.memo():
        z = T.force(T.sumT([T.matVec(dW, dX), db], (N,)))
    nl = T.stats()["launches"] - l0
    ok = same(z.numpy(), want.astype(np.float32), tool='pinned_epilogue_check', M=M, K=K, N=N)
    l0 = T.stats()["launches"]
    with T.memo():
        h = T.force(T.liftT(hipt.logistic_closure, [T.sumT([T.matVec(dW, dX), db], (N,))], key="pe-logistic"))
    nl2 = T.stats()["launches"] - l0
    err = float(np.max(np.abs(h.numpy() - 1 / (1 + np.exp(-want)))))
    ok = ok and err < 2e-6
    print((M, K, N), "launches", nl, nl2, "logistic err %.2e" % err, "ok" if ok else "MISMATCH")
    bad += not ok
    del dW, dX, db, z, h
print("mismatches", bad)
# what the fused way out costs next to the plain product (4096^3, steady state)
M = K = N = 4096
W = T.genRand((N, K), "uniform", -1, 1, 1)
X = T.put(rng.uniform(-1, 1, (M, K)).astype(np.float32), batched=True)
b = T.put(rng.uniform(-1, 1, N).astype(np.float32))


def timed(fn, n=60, warm=80):
    for _ in range(warm): fn()
    T.sync(); T.timer_start()
    for _ in range(n): fn()
    return T.timer_stop() / n


def plain():
    with T.memo():
        T.force(T.matVec(W, X))


def fused():
    with T.memo():
        T.force(T.liftT(hipt.logistic_closure, [T.sumT([T.matVec(W, X), b], (N,))], key="pe-logistic"))


print("4096^3 batched matVec            %.4f ms" % timed(plain))
print("4096^3 logistic(matVec + bias)   %.4f ms (one launch)" % timed(fused))
