"""Quick kernel timings through the C ABI (development aid, not the contract bench)."""
import sys
import os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd.hipt import HipT  # noqa: E402
from oracle import neuralnet as NN  # noqa: E402

T = HipT(0)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    T.sync()
    T.timer_start()
    for _ in range(iters):
        fn()
    return T.timer_stop() / iters


def gemm(m, k, n, ta=False, tb=False):
    a = T.genRand((k, m) if ta else (m, k), "uniform", -1, 1, 1)
    b = T.genRand((n, k) if tb else (k, n), "uniform", -1, 1, 2)
    A = T.transp(a) if ta else a
    B = T.transp(b) if tb else b
    ms = timeit(lambda: T.gmul(1, 1, 1, A, B))
    print("gemm %5dx%5dx%5d ta=%d tb=%d  %8.3f ms  %7.2f TF" % (m, k, n, ta, tb, ms, 2.0 * m * k * n / ms / 1e9))


for s in (1024, 2048, 4096, 8192):
    gemm(s, s, s)
gemm(4096, 4096, 4096, True, False)
gemm(4096, 4096, 4096, False, True)
gemm(4096, 4096, 4096, True, True)
gemm(262144, 64, 512)
gemm(1024, 784, 256, False, True)
gemm(256, 1024, 784, True, False)
gemm(1024, 256, 10, False, True)
gemm(1024, 10, 256)

n = 512 ** 3
x = T.genRand((n,), "uniform", -4, 4, 3)
e = T.expr(lambda v: NN.logistic(v[0]), 1, key="logi")
ms = timeit(lambda: T.liftT(e, [x]))
print("map logistic 512^3  %8.3f ms  %7.1f GB/s" % (ms, 8.0 * n / ms / 1e6))
e2 = T.expr(lambda v: v[0] * NN.logistic_prime(v[1]), 2, key="dlogi")
ms = timeit(lambda: T.liftT(e2, [x, x]))
print("d*logistic'(x) 512^3  %8.3f ms  %7.1f GB/s" % (ms, 12.0 * n / ms / 1e6))
e3 = T.expr(lambda v: v[0] * 2.0 + 1.0, 1, key="aff")
ms = timeit(lambda: T.liftT(e3, [x]))
print("affine 512^3  %8.3f ms  %7.1f GB/s" % (ms, 8.0 * n / ms / 1e6))
import oracle.ad as ad  # noqa: E402
e4 = T.expr(lambda v: ad.sin(v[0]) * v[0] + ad.exp(-v[0] * v[0]), 1, key="vm")
ms = timeit(lambda: T.liftT(e4, [x]), iters=5)
print("VM sin(x)x+exp(-x^2) 512^3  %8.3f ms  %7.1f GB/s  kind=%d" % (ms, 8.0 * n / ms / 1e6, e4.kind))
