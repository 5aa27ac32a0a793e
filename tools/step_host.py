"""Host-side enqueue cost of one step vs its GPU time (is the step CPU-bound?)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tensor_ops_amd import tops  # noqa: E402
from tensor_ops_amd.hipt import HipT  # noqa: E402

T = HipT(0)
ws, X, Y = bench.synth(0, 1024)
net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
tr = tops.Trainer(net, "crossEntropy", bench.RATE, T.put(X, batched=True), T.put(Y, batched=True))
for name, f in (("step", tr.step), ("grad+apply", lambda: (tr.grad(), tr.apply()))):
    for _ in range(50):
        f()
    T.sync()
    n = 3000
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    t1 = time.perf_counter()
    T.sync()
    t2 = time.perf_counter()
    print("%-11s enqueue %.2f us/step, drained after %.2f us/step" % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
    # one step at a time, synchronised: pure latency of the chain
    t0 = time.perf_counter()
    for _ in range(500):
        f()
        T.sync()
    print("%-11s synchronised: %.2f us/step" % (name, (time.perf_counter() - t0) / 500 * 1e6))
