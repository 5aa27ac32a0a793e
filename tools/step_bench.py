"""Step-only timing loop (for rocprofv3): batched gradTOp + SGD of 784->256->10, B=1024."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bench  # noqa: E402
from tensor_ops_amd import tops  # noqa: E402
from tensor_ops_amd.hipt import HipT  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
graph = "--no-graph" not in sys.argv
fused = "--generic" not in sys.argv   # --generic: the library's fusion off, one launch per class-method call
f64 = "--f64" in sys.argv
if f64:
    tops.hlib()
    tops.set_elem_dtype(np.float64)
T = HipT(0, dtype=np.float64 if f64 else np.float32)
ws, X, Y = bench.synth(0, 1024)
net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
tr = tops.Trainer(net, "crossEntropy", bench.RATE, T.put(X, batched=True), T.put(Y, batched=True),
                  use_memo=True, use_graph=graph, use_fused=fused)
two_call = "--two-call" in sys.argv  # grad() then apply() (what a data-parallel rank runs); default: step()


def one():
    if two_call:
        tr.grad(); tr.apply()
    else:
        tr.step()


for _ in range(10):
    one()
T.sync()
T.timer_start()
for _ in range(iters):
    one()
ms = T.timer_stop() / iters
print(("fp64 " if f64 else "") + "step library_fusion=%s replay=%s launches: grad %d, step %d  %.4f ms/step  %.0f steps/s"
      % (tr.fused, graph, tr.launches_per_step, tr.step_launches, ms, 1e3 / ms))
