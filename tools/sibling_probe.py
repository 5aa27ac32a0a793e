"""Config 5 through the inner boundary inside a scope (512 gemm + 512 liftB calls -> sibling batches, csrc/lazy.cpp): where
the time of the one flush goes -- planning, the whole flush (host), the device between events around to_force_many."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tensor_ops_amd import capi, hipt
from tensor_ops_amd.hipb import HipB
from oracle.btensor import BTensorOps

B = HipB(0, np.float32)
T = B.T
ops = BTensorOps(B)
rng = np.random.default_rng(55)
a = rng.integers(-2, 3, (512, 512, 64)).astype(np.float32)
b = rng.integers(-2, 3, (64, 512)).astype(np.float32)
A, Bm = ops.from_array(a), ops.from_array(b)


def leaves(t):
    return [t.val] if t.tag in "VM" else [h for x in t.val for h in leaves(x)]


def times():
    p, f = C.c_int64(), C.c_int64()
    capi.check(capi.lib().to_lazy_time(C.byref(p), C.byref(f)))
    return p.value, f.value


for rep in range(5):
    T.sync()
    p0, f0 = times()
    l0 = T.stats()["launches"]
    t0 = time.perf_counter()
    with T.memo():
        Cs = ops.gmul(2, 1, 1, A, Bm)
        Ls = ops.liftT(hipt.logistic_closure, [Cs])
        want = leaves(Cs) + leaves(Ls)
        t1 = time.perf_counter()
        T.timer_start()
        T.force_many(want)
        ms = T.timer_stop()
        t2 = time.perf_counter()
    t3 = time.perf_counter()
    p1, f1 = times()
    print("rep %d: record %.2f ms | force_many host %.2f ms (planning %.3f, flush total %.3f) | device between events %.3f ms | scope end %.2f ms | launches %d"
          % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (p1 - p0) / 1e6, (f1 - f0) / 1e6, ms, (t3 - t2) * 1e3, T.stats()["launches"] - l0), flush=True)
    del Cs, Ls, want
