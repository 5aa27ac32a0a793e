"""TOp vocabulary on batched data next to torch (protocol of ops_scan.py): softmax over B rows of n, zip3 (x y + z), a three-input
liftT, negate / scale, run through the host mirror's runTOp.  usage: tops_scan.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tensor_ops_amd import tops as H
from tensor_ops_amd.hipt import HipT
T = HipT(0); H.hlib()


def time_ours(f):
    for _ in range(5): f()
    T.sync(); T.timer_start()
    for _ in range(5): f()
    est = max(T.timer_stop() / 5, 1e-3)
    n = max(10, int(20.0 / est))
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
    n = max(10, int(20.0 / est))
    for _ in range(n // 2): f()
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


sm = H.softmax()
z3 = H.zip3(lambda x, y, z: x * y + z)
for B, n in [(1024, 10), (60000, 10), (60000, 100), (60000, 1000), (1000000, 10), (8192, 4096), (1024, 784)]:
    x = T.genRand((n,), "uniform", -1, 1, 1, batch=B); y = T.genRand((n,), "uniform", -1, 1, 2, batch=B); z = T.genRand((n,), "uniform", -1, 1, 3, batch=B)
    tx = torch.rand(B, n, device="cuda"); ty = torch.rand(B, n, device="cuda"); tz = torch.rand(B, n, device="cuda"); to_ = torch.empty(B, n, device="cuda")
    l0 = T.stats()["launches"]; sm.run([x]); nl = T.stats()["launches"] - l0
    a, b = time_ours(lambda: sm.run([x])), time_torch(lambda: torch.softmax(tx, dim=1, out=to_))
    print("softmax   %8d x %-5d ours %9.1f us (%d launches)  torch %9.1f us  ratio %.2f" % (B, n, a * 1e3, nl, b * 1e3, b / a), flush=True)
    l0 = T.stats()["launches"]; z3.run([x, y, z]); nl = T.stats()["launches"] - l0
    a, b = time_ours(lambda: z3.run([x, y, z])), time_torch(lambda: torch.addcmul(tz, tx, ty, out=to_))
    print("zip3 xy+z %8d x %-5d ours %9.1f us (%d launches)  torch %9.1f us  ratio %.2f" % (B, n, a * 1e3, nl, b * 1e3, b / a), flush=True)
    del x, y, z, tx, ty, tz, to_
    torch.cuda.empty_cache()
