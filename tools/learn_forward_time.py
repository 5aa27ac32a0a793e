"""The reference's 784 -> 300 -> 100 -> 10 forward over a whole data set, recorded layer by layer (`logistic (W x + b)`): launches and
time per layer.  usage: learn_forward_time.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tensor_ops_amd.hipt import HipT, logistic_closure
T = HipT(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
rng = np.random.default_rng(1)
dims = [784, 300, 100, 10]
x = T.put(rng.uniform(0, 1, (B, dims[0])).astype(np.float32), batched=True)
for li in range(3):
    i, o = dims[li], dims[li + 1]
    W = T.put((0.1 * rng.standard_normal((o, i))).astype(np.float32)); b = T.put((0.1 * rng.standard_normal(o)).astype(np.float32))

    def layer():
        with T.memo():
            return T.force(T.liftT(logistic_closure, [T.sumT([T.matVec(W, x), b], (o,))], key="lf-logistic"))
    l0 = T.stats()["launches"]; y = layer(); nl = T.stats()["launches"] - l0
    for _ in range(50): layer()
    T.sync(); T.timer_start()
    for _ in range(200): layer()
    ms = T.timer_stop() / 200
    print("layer %d: %d x %d x %d  launches %d  %.1f us  %.1f TF" % (li + 1, B, i, o, nl, ms * 1e3, 2.0 * B * i * o / ms / 1e9), flush=True)
    x = y
