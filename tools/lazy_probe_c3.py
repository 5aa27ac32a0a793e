"""Print the plan (TOPS_LAZY_DEBUG=1) of the first batched config-3 gradient of a fresh process."""
import os
import sys
os.environ.setdefault("TOPS_LAZY_DEBUG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tensor_ops_amd import tops
from tensor_ops_amd.hipt import HipT

T = HipT(0)
rng = np.random.default_rng(1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ws = [(0.5 * rng.standard_normal((256, 784)), 0.5 * rng.standard_normal(256)),
      (0.5 * rng.standard_normal((10, 256)), 0.5 * rng.standard_normal(10))]
X = rng.uniform(0, 1, size=(B, 784))
Y = np.zeros((B, 10))
Y[np.arange(B), rng.integers(0, 10, size=B)] = 1.0
net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
tr = tops.Trainer(net, "crossEntropy", 0.02, T.put(X, batched=True), T.put(Y, batched=True), use_graph=False)
print("launches_per_step", tr.launches_per_step)
tr.step()
print("step_launches", tr.step_launches)
