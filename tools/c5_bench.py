"""Config 5a experiments: gmul '[512,512,64] x '[64,512] and pure write bandwidth."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd.hipt import HipT  # noqa: E402

T = HipT(0)


def timeit(fn, iters=int(os.environ.get("ITERS", "20")), warm=int(os.environ.get("WARM", "3"))):
    for _ in range(warm):
        fn()
    T.sync()
    T.timer_start()
    for _ in range(iters):
        fn()
    return T.timer_stop() / iters


ms = timeit(lambda: T.konst((512, 512, 512), 1.0))
print("fill 512 MiB: %.3f ms  %.0f GB/s" % (ms, 512 * 2**20 / ms / 1e6))
a = T.genRand((512, 512, 64), "uniform", -1, 1, 1)
b = T.genRand((64, 512), "uniform", -1, 1, 2)
ms = timeit(lambda: T.gmul(2, 1, 1, a, b))
print("variant=%s gmul c5a: %.3f ms  %.1f TF  %.0f GB/s" % (os.environ.get("TOPS_GEMM_VARIANT", "-"), ms, 17.18 / ms, 604.11 / ms))
from tensor_ops_amd.hipt import logistic_closure  # noqa: E402
e = T.expr(logistic_closure, 1, key="c5b")


def fused():
    with T.memo():
        r = T.force(T.liftT(e, [T.gmul(2, 1, 1, a, b)]))
    return r


ms = timeit(fused)
print("variant=%s gmul+logistic fused: %.3f ms  %.1f TF  %.0f GB/s" % (os.environ.get("TOPS_GEMM_VARIANT", "-"), ms, 17.18 / ms, 604.11 / ms))
