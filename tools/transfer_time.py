"""Host <-> device transfers through the C ABI (pageable numpy memory, the library's pinned staging): GB/s.  usage: transfer_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tensor_ops_amd.hipt import HipT
T = HipT(0)
for mb in (1, 16, 188, 1024):
    n = mb * (1 << 20) // 4
    a = np.random.default_rng(0).random(n, dtype=np.float32)
    for _ in range(2): d = T.put(a)
    T.sync(); t0 = time.perf_counter()
    for _ in range(3): d = T.put(a)
    T.sync(); up = (time.perf_counter() - t0) / 3
    for _ in range(2): h = d.numpy()
    t0 = time.perf_counter()
    for _ in range(3): h = d.numpy()
    dn = (time.perf_counter() - t0) / 3
    ta = torch.from_numpy(a)
    for _ in range(2): td = ta.cuda()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): td = ta.cuda()
    torch.cuda.synchronize(); tup = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for _ in range(3): th = td.cpu()
    tdn = (time.perf_counter() - t0) / 3
    print("%5d MiB  put %6.2f GB/s  numpy() %6.2f GB/s   torch .cuda() %6.2f GB/s  .cpu() %6.2f GB/s" % (mb, n * 4 / up / 1e9, n * 4 / dn / 1e9, n * 4 / tup / 1e9, n * 4 / tdn / 1e9), flush=True)
