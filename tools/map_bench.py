"""map logistic over 512^3 fp32 (BASELINE config 5b), for rocprofv3 runs."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd.hipt import HipT, logistic_closure  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
T = HipT(0)
n = 512 ** 3
x = T.genRand((n,), "uniform", -4, 4, 3)
e = T.expr(logistic_closure, 1, key="logi")
for _ in range(2):
    T.liftT(e, [x])
T.sync()
T.timer_start()
for _ in range(iters):
    T.liftT(e, [x])
ms = T.timer_stop() / iters
print("map logistic 512^3 %.4f ms %.1f GB/s" % (ms, 8.0 * n / ms / 1e6))
