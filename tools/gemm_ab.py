"""A/B of GEMM routes in steady state (60 ms warm-up, 40 ms timed): usage gemm_ab.py M K N [M K N ...]; the environment
(TOPS_GEMM_KW=0, TOPS_GEMM_STREAMK_HYBRID=0, ...) selects the route; GEMM_DTYPE=f64 times the fp64 instance."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd.hipt import HipT
F64 = os.environ.get("GEMM_DTYPE") == "f64"   # GEMM_DTYPE=f64: the fp64 instance
if F64:
    import numpy as np
    T = HipT(0, dtype=np.float64)
else:
    T = HipT(0)
v = [int(x) for x in sys.argv[1:]]
for i in range(0, len(v), 3):
    m, k, n = v[i:i + 3]
    a = T.genRand((m, k), "uniform", -1, 1, 1); b = T.genRand((k, n), "uniform", -1, 1, 2)

    def run(iters, warm):
        for _ in range(warm): T.gmul(1, 1, 1, a, b)
        T.sync(); T.timer_start()
        for _ in range(iters): T.gmul(1, 1, 1, a, b)
        return T.timer_stop() / iters
    est = max(run(20, 5), 1e-3)
    ms = run(max(20, int(40.0 / est)), max(20, int(60.0 / est)))
    print("%6d x %6d x %6d  %8.4f ms %7.2f TF" % (m, k, n, ms, 2.0 * m * k * n / ms / 1e9))
