"""Online SGD (`foldl' trainNetwork`, app/MNIST.hs:390-396) over a resident synthetic data set:
us per sample for the MNIST app's 784->300->100->10 stack.  --f64: the reference's precision."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from tensor_ops_amd import tops  # noqa: E402
from tensor_ops_amd.hipt import HipT  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20000
f64 = "--f64" in sys.argv
generic = "--generic" in sys.argv
dt = np.float64 if f64 else np.float32
if f64:
    tops.hlib()
    tops.set_elem_dtype(np.float64)
T = HipT(0, dtype=dt)
rng = np.random.default_rng(5)
sizes = [784, 300, 100, 10]
ws = [(rng.normal(0, 0.5, size=(o, i)) / np.sqrt(i), rng.normal(0, 0.5, size=o)) for i, o in zip(sizes, sizes[1:])]
X = rng.uniform(0, 1, size=(n, 784))
Y = np.zeros((n, 10))
Y[np.arange(n), rng.integers(0, 10, size=n)] = 1.0
net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
dX, dY = T.put(X, batched=True), T.put(Y, batched=True)
for rep in range(2):
    T.sync()
    t0 = time.perf_counter()
    out = tops.trainAll(net, "crossEntropy", 0.02, dX, dY, use_fused=not generic)
    T.sync()
    dt_s = time.perf_counter() - t0
print("%sonline SGD %s: %d samples, %.2f us/sample (%.0f samples/s)" % (
    "fp64 " if f64 else "", "generic (graph replay)" if generic else "pre-fused", n, dt_s / n * 1e6, n / dt_s))
