"""One UNBATCHED trainNetwork step (the reference's own per-sample call, app/MNIST.hs:390-396) inside a scope, plan on stderr."""
import sys, os
os.environ["TOPS_LAZY_DEBUG"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd import tops
from tensor_ops_amd.hipt import HipT
T = HipT(0)
rng = np.random.default_rng(1)
sizes = [784, 300, 100, 10]
ws = [(rng.normal(0, 0.5, size=(o, i)) / np.sqrt(i), rng.normal(0, 0.5, size=o)) for i, o in zip(sizes, sizes[1:])]
x = rng.uniform(0, 1, 784); y = np.zeros(10); y[3] = 1
net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
dx, dy = T.put(x), T.put(y)
st = T.stats()["launches"]
with T.memo():
    n2 = tops.trainNetwork(net, "crossEntropy", 0.02, dx, dy)
    ps = n2.params
print("launches", T.stats()["launches"] - st, file=sys.stderr)
