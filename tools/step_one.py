"""One stack / batch of tools/step_scan.py in a loop (for a kernel trace).  usage: step_one.py B d0 d1 ... dn"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tensor_ops_amd import tops as H
from tensor_ops_amd.hipt import HipT
T = HipT(0); H.hlib()
rng = np.random.default_rng(3)
B = int(sys.argv[1]); dims = [int(x) for x in sys.argv[2:]]
ws = [(0.5 * rng.standard_normal((o, i)) / np.sqrt(i), 0.5 * rng.standard_normal(o)) for i, o in zip(dims[:-1], dims[1:])]
X = rng.uniform(0, 1, (B, dims[0])); Y = np.zeros((B, dims[-1])); Y[np.arange(B), rng.integers(0, dims[-1], B)] = 1
net = H.genNet([(T.put(w.astype(np.float32)), T.put(b.astype(np.float32))) for w, b in ws], "actMapLogistic", "actSoftmax")
tr = H.Trainer(net, "crossEntropy", 0.01 / B, T.put(X.astype(np.float32), batched=True), T.put(Y.astype(np.float32), batched=True), use_graph=False)
for _ in range(300): tr.step()
T.sync()
print("launches", tr.launches_per_step, "fused", tr.fused)
