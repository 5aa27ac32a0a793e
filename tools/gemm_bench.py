"""GEMM-only timing loop (for rocprofv3 runs).  usage: gemm_bench.py M K N [iters]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd.hipt import HipT  # noqa: E402

m, k, n = (int(a) for a in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
T = HipT(0)
ta, tb = int(os.environ.get("TA", "0")), int(os.environ.get("TB", "0"))
a = T.genRand((k, m) if ta else (m, k), "uniform", -1, 1, 1)
b = T.genRand((n, k) if tb else (k, n), "uniform", -1, 1, 2)
if ta:
    a = T.transp(a)
if tb:
    b = T.transp(b)
for _ in range(int(os.environ.get("WARM", "2"))):
    T.gmul(1, 1, 1, a, b)
T.sync()
T.timer_start()
for _ in range(iters):
    T.gmul(1, 1, 1, a, b)
ms = T.timer_stop() / iters
print("ta%d tb%d " % (ta, tb) + "variant=%s gemm %dx%dx%d %.3f ms %.2f TF" % (os.environ.get("TOPS_GEMM_VARIANT", "-"), m, k, n, ms, 2.0 * m * k * n / ms / 1e9))
