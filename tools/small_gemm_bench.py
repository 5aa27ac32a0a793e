"""Shapes of the batched ffLayer step through the planner (for tuning the small-GEMM kernels)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd.hipt import HipT, Graph  # noqa: E402

T = HipT(0)


def bench(name, fn, reps=50):
    fn()
    with Graph() as g:          # graph replay: device time without host launch overhead
        for _ in range(reps):
            fn()
    g.launch()
    T.sync()
    T.timer_start()
    g.launch()
    ms = T.timer_stop() / reps
    print("%-46s %7.2f us" % (name, ms * 1e3))


B = 1024
X = T.genRand((784,), "uniform", 0, 1, 1, batch=B)
W1 = T.genRand((256, 784), "normal", 0, 0.5, 2)
H = T.genRand((256,), "uniform", 0, 1, 3, batch=B)
W2 = T.genRand((10, 256), "normal", 0, 0.5, 4)
dZ1 = T.genRand((256,), "uniform", -1, 1, 5, batch=B)
dZ2 = T.genRand((10,), "uniform", -1, 1, 6, batch=B)
bench("fwd1  X.W1^T   1024x256x784", lambda: T.gmul(1, 1, 0, W1, X))
bench("gW1   dZ1^T.X  256x784x1024", lambda: T.gmul_batch_sum(1, 0, 1, dZ1, X))
bench("fwd2  H.W2^T   1024x10x256", lambda: T.gmul(1, 1, 0, W2, H))
bench("gW2   dZ2^T.H  10x256x1024", lambda: T.gmul_batch_sum(1, 0, 1, dZ2, H))
bench("dH    dZ2.W2   1024x256x10", lambda: T.gmul(1, 1, 0, T.transp(W2), dZ2))
bench("colsum [1024,256]", lambda: T.batch_sum(dZ1))
bench("colsum [1024,10]", lambda: T.batch_sum(dZ2))
