import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tensor_ops_amd import tops as H
from tensor_ops_amd.hipt import HipT
T = HipT(0)
rng = np.random.default_rng(1)
def c3(B, i, h, o):
    ws = [(0.5 * rng.standard_normal((h, i)), 0.5 * rng.standard_normal(h)), (0.5 * rng.standard_normal((o, h)), 0.5 * rng.standard_normal(o))]
    X = rng.uniform(0, 1, (B, i)); Y = np.zeros((B, o)); Y[np.arange(B), rng.integers(0, o, B)] = 1
    return ws, X, Y
ws, X, Y = c3(50, 30, 17, 20)
net = H.net_then(H.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapTanh", "actSoftmax"), H.scale(0.5))
tr = H.Trainer(net, "squaredError", 0.01, T.put(X, batched=True), T.put(Y, batched=True), use_graph=False)
print("tanh + softmax(20) + scale + squaredError: launches", tr.launches_per_step)
ws, X, Y = c3(64, 40, 24, 30)
net = H.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
tr = H.Trainer(net, "crossEntropy", 0.01, T.put(X, batched=True), T.put(Y, batched=True), use_graph=False)
print("logistic + softmax(30) + crossEntropy: launches", tr.launches_per_step)
