"""What does a plain device copy / scale sustain on this box, next to map logistic? (ceiling check)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensor_ops_amd.hipt import HipT, logistic_closure
T = HipT(0)
n = 512 ** 3
x = T.genRand((n,), "uniform", -4, 4, 3)
def t(fn, iters=30, warm=10):
    for _ in range(warm): fn()
    T.sync(); T.timer_start()
    for _ in range(iters): fn()
    return T.timer_stop() / iters
e = T.expr(logistic_closure, 1, key="logi")
ms = t(lambda: T.liftT(e, [x])); print("map logistic      %.4f ms %6.0f GB/s" % (ms, 8.0 * n / ms / 1e6))
ms = t(lambda: T.scaleT(2.0, x)); print("scaleT (affine)   %.4f ms %6.0f GB/s" % (ms, 8.0 * n / ms / 1e6))
a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(10): b.copy_(a)
ev0.record()
for _ in range(30): b.copy_(a)
ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / 30; print("torch copy_       %.4f ms %6.0f GB/s" % (ms, 8.0 * n / ms / 1e6))
for _ in range(10): torch.mul(a, 2.0, out=b)
ev0.record()
for _ in range(30): torch.mul(a, 2.0, out=b)
ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / 30; print("torch mul         %.4f ms %6.0f GB/s" % (ms, 8.0 * n / ms / 1e6))
