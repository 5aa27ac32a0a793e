"""One config-3 step with the planner's dump on stderr (TOPS_LAZY_DEBUG=1)."""
import sys, os
os.environ["TOPS_LAZY_DEBUG"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd import tops
from tensor_ops_amd.hipt import HipT
T = HipT(0)
rng = np.random.default_rng(1)
i, h, o, B = 784, 256, 10, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
head, loss = (("actSoftmax", "crossEntropy"), ("actLogistic", "squaredError"))[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
ws = [(0.5 * rng.standard_normal((h, i)), 0.5 * rng.standard_normal(h)),
      (0.5 * rng.standard_normal((o, h)), 0.5 * rng.standard_normal(o))]
X = rng.uniform(0, 1, size=(B, i)); Y = np.zeros((B, o)); Y[np.arange(B), rng.integers(0, o, size=B)] = 1.0
net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", head)
print("== create (grad warm-up)", file=sys.stderr, flush=True)
tr = tops.Trainer(net, loss, 0.02, T.put(X, batched=True), T.put(Y, batched=True), use_graph=False, use_fused=True)
print("== step", file=sys.stderr, flush=True)
tr.step(); T.sync()
print("launches", tr.launches_per_step, tr.step_launches)
