"""One HBM-bound form in a loop (for rocprofv3): usage gemv_bench.py matVec|vecMat|outerV|sumRows M K [iters]; WARM launches first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd.hipt import HipT
T = HipT(0)
op, m, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 50
warm = int(os.environ.get("WARM", "10"))
a = T.genRand((m, k), "uniform", -1, 1, 1); x = T.genRand((k,), "uniform", -1, 1, 2); y = T.genRand((m,), "uniform", -1, 1, 3)
f = {"matVec": lambda: T.matVec(a, x), "vecMat": lambda: T.vecMat(y, a), "outerV": lambda: T.outerV(y, x), "sumRows": lambda: T.sumRows(a)}[op]
for _ in range(warm): f()
T.sync(); T.timer_start()
for _ in range(iters): f()
ms = T.timer_stop() / iters
print("%s %d x %d: %.4f ms  %.1f GB/s (4 M K bytes)" % (op, m, k, ms, 4.0 * m * k / ms / 1e6))
