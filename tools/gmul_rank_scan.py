"""gmul over rank classes (a : ms ++ os, b : Reverse os ++ ns) next to torch.tensordot on the same contraction, us; shapes chosen so
that the flattened GEMM is a few GFLOP.  usage: gmul_rank_scan.py"""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tensor_ops_amd.hipt import HipT
T = HipT(0)


def time_ours(f):
    for _ in range(5): f()
    T.sync(); T.timer_start()
    for _ in range(5): f()
    est = max(T.timer_stop() / 5, 1e-3)
    n = max(10, int(30.0 / est))
    for _ in range(n // 2): f()
    T.sync(); T.timer_start()
    for _ in range(n): f()
    return T.timer_stop() / n


def time_torch(f):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    est = max(e0.elapsed_time(e1) / 5, 1e-3)
    n = max(10, int(30.0 / est))
    for _ in range(n // 2): f()
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


cases = [  # (ms, os, ns)
    ((512, 512), (64,), (512,)), ((1024,), (64, 64), (1024,)), ((256, 64), (32, 32), (128,)), ((512,), (128,), (64, 64)),
    ((64, 64), (64,), (64, 64)), ((128,), (16, 16, 16), (128,)), ((32, 32, 32), (256,), (256,)), ((2048,), (8, 256), (2048,)),
    ((100, 30), (28, 28), (10,)), ((60000,), (28, 28), (300,)), ((16, 16), (16, 16), (16, 16)), ((4096,), (64, 64), (64,))]
for ms, os_, ns in cases:
    a = T.genRand(tuple(ms) + tuple(os_), "uniform", -1, 1, 1)
    b = T.genRand(tuple(reversed(os_)) + tuple(ns), "uniform", -1, 1, 2)
    ta = torch.rand(*ms, *os_, device="cuda"); tb = torch.rand(*reversed(os_), *ns, device="cuda")
    lo = len(os_)
    adims = list(range(len(ms), len(ms) + lo)); bdims = list(range(lo - 1, -1, -1))
    l0 = T.stats()["launches"]; T.gmul(len(ms), lo, len(ns), a, b); nl = T.stats()["launches"] - l0
    to = time_ours(lambda: T.gmul(len(ms), lo, len(ns), a, b))
    tv = time_torch(lambda: torch.tensordot(ta, tb, dims=(adims, bdims)))
    fl = 2.0 * np.prod(ms) * np.prod(os_) * np.prod(ns)
    print("ms %-14s os %-14s ns %-10s  ours %9.1f us (%d launches) %6.1f TF   torch %9.1f us   ratio %.2f" % (ms, os_, ns, to * 1e3, nl, fl / to / 1e9, tv * 1e3, tv / to), flush=True)
    del a, b, ta, tb
