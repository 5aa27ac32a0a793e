import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tensor_ops_amd import tops as H
from tensor_ops_amd.hipt import HipT
T = HipT(0); H.hlib()
rng = np.random.default_rng(9)
i, h, o, n, B = 12, 20, 24, 2, 16
fc = tuple(T.put(v) for v in (0.5 * rng.standard_normal(h), 0.5 * rng.standard_normal((h, h)), 0.5 * rng.standard_normal((h, i)), 0.5 * rng.standard_normal(h)))
ff = tuple(T.put(v) for v in (0.5 * rng.standard_normal((o, h)), 0.5 * rng.standard_normal(o)))
net = H.rnn_genNet([(fc, "actLogistic", "actLogistic")], (ff, None), "actSoftmax")
xs = [T.put(rng.uniform(-1, 1, (B, i)), batched=True) for _ in range(n)]
ys = [T.put(rng.uniform(0.1, 0.9, (B, o)), batched=True) for _ in range(n)]
with T.memo():
    _, gs, gp = H.rnn_netGrad(net, "crossEntropy", xs, ys, want_inputs=False)
    l0 = T.stats()["launches"]
    T.force_many(gs + gp)
    print("launches", T.stats()["launches"] - l0)
