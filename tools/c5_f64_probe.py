"""Config 5a in the reference's own element type (Double): '[512,512,64] x '[64,512] on the fp64 kernels."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from tensor_ops_amd.hipt import HipT

T64 = HipT(0, dtype=np.float64)
a = T64.genRand((512, 512, 64), "uniform", -1.0, 1.0, 5)
b = T64.genRand((64, 512), "uniform", -1.0, 1.0, 6)
ms = bench.time_launches(T64, lambda: T64.gmul(2, 1, 1, a, b), 100, warm=50)
flops = 17_179_869_184
byts = 2 * 604_110_848
print(json.dumps({"ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 2), "frac_mfma_f64": round(flops / ms / 1e9 / 78.6, 4),
                  "gbps": round(byts / ms / 1e6, 1), "frac_hbm": round(byts / ms / 1e6 / 8000, 4)}))
from tensor_ops_amd.hipt import logistic_closure
e = T64.expr(logistic_closure, 1, key="c5_f64_logistic")


def fused():
    with T64.memo():
        T64.force(T64.liftT(e, [T64.gmul(2, 1, 1, a, b)]))


l0 = T64.stats()["launches"]
fused()
nl = T64.stats()["launches"] - l0
msf = bench.time_launches(T64, fused, 100, warm=50)
print(json.dumps({"fused_map_logistic_ms": round(msf, 4), "launches": nl, "tflops": round(flops / msf / 1e9, 2)}))
c = T64.gmul(2, 1, 1, a, b)
msm = bench.time_launches(T64, lambda: T64.liftT(e, [c]), 50, warm=20)
print(json.dumps({"map_alone_ms": round(msm, 4)}))
