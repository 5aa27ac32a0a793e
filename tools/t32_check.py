"""gemm_t32.hip (four DMA-fed waves per 32x32 tile) on the shapes it takes from the small-GEMM route: exact on integers in
all four operand layouts, ragged K / M / N, the fused epilogues (bias + logistic / tanh through a recorded `W x + b`), and
its time beside the step's forward shape.  usage: t32_check.py [--time]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd import hipt
from tensor_ops_amd.hipt import HipT
T = HipT(0)
rng = np.random.default_rng(3)
bad = 0
shapes = [(1024, 784, 256), (256, 1024, 784), (1024, 256, 256), (512, 512, 512), (1000, 260, 300), (992, 788, 252), (320, 1024, 320),
          (1024, 4096, 128), (128, 2048, 1024), (640, 272, 640), (1024, 1000, 256), (704, 8192, 160)]
for (M, K, N) in shapes:
    for ta in (False, True):
        for tb in (False, True):
            a = rng.integers(-2, 3, (M, K)).astype(np.float32); b = rng.integers(-2, 3, (K, N)).astype(np.float32)
            A = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
            B = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
            l0 = T.stats()["launches"]
            got = T.gmul(1, 1, 1, A, B).numpy()
            nl = T.stats()["launches"] - l0
            want = (a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32)
            if not np.array_equal(got, want):
                bad += 1
                nz = np.argwhere(got != want)
                print("MISMATCH", (M, K, N), "ta", ta, "tb", tb, "count", len(nz), "first", nz[:4].tolist(), "launches", nl)
# the forward layer as the step records it: W x + b under logistic / tanh
for (B_, i, o) in [(1024, 784, 256), (1024, 300, 160), (992, 784, 256)]:
    W = rng.integers(-2, 3, (o, i)).astype(np.float32); X = rng.integers(-2, 3, (B_, i)).astype(np.float32); bb = rng.integers(-3, 4, o).astype(np.float32)
    want = X.astype(np.float64) @ W.T.astype(np.float64) + bb
    dW, dX, db = T.put(W), T.put(X, batched=True), T.put(bb)
    with T.memo():
        z = T.force(T.sumT([T.matVec(dW, dX), db], (o,)))
    if not np.array_equal(z.numpy().reshape(B_, o), want.astype(np.float32)):
        bad += 1; print("MISMATCH bias", (B_, i, o))
    with T.memo():
        h = T.force(T.liftT(hipt.logistic_closure, [T.sumT([T.matVec(dW, dX), db], (o,))], key="t32-logistic"))
    e = np.max(np.abs(h.numpy().reshape(B_, o) - 1 / (1 + np.exp(-want))))
    if e > 2e-6:
        bad += 1; print("MISMATCH logistic", (B_, i, o), e)
print("t32_check mismatches", bad)
if "--time" in sys.argv:
    for (M, K, N, ta, tb) in [(1024, 784, 256, False, True), (256, 1024, 784, True, False), (1024, 256, 256, False, True)]:
        a = rng.standard_normal((M, K)).astype(np.float32); b = rng.standard_normal((K, N)).astype(np.float32)
        A = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
        B = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
        for _ in range(50): T.gmul(1, 1, 1, A, B)
        T.sync(); T.timer_start()
        for _ in range(500): T.gmul(1, 1, 1, A, B)
        ms = T.timer_stop() / 500
        print("time %dx%dx%d ta %s tb %s: %.2f us  %.1f TF" % (M, K, N, ta, tb, ms * 1e3, 2.0 * M * K * N / ms / 1e9))
