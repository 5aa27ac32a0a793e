"""fp64 instance probe: gmul (f64 MFMA) and map logistic rates."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tensor_ops_amd.hipt import HipT, logistic_closure

T = HipT(0, dtype=np.float64)

def t(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    T.sync(); T.timer_start()
    for _ in range(iters): fn()
    return T.timer_stop() / iters

sizes = [int(v) for v in sys.argv[1:]] or [1024, 2048, 4096, 8192]
for n in sizes:
    a = T.genRand((n, n), "uniform", -1.0, 1.0, 11)
    b = T.genRand((n, n), "uniform", -1.0, 1.0, 12)
    ms = t(lambda: T.gmul(1, 1, 1, a, b))
    print("f64 gmul %d^3: %.3f ms  %.1f TF (%.1f%% of 78.6)" % (n, ms, 2.0 * n ** 3 / ms / 1e9, 2.0 * n ** 3 / ms / 1e9 / 78.6 * 100))
    if n == 1024:
        got = T.gmul(1, 1, 1, a, b).numpy()
        want = a.numpy() @ b.numpy()
        print("  rel err", np.linalg.norm(got - want) / np.linalg.norm(want))
    del a, b
