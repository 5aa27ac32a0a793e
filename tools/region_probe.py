"""What a timed region of K replayed steps costs beyond K times the steady step (the driver times 20-step regions between
two synchronisations): wall time for K = 1 .. 100 with and without the two device-timer events bench.py records inside the
region.  usage: region_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from tensor_ops_amd import tops
from tensor_ops_amd.hipt import HipT
T = HipT(0)
ws, X, Y = bench.synth(0, 1024)
net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
tr = tops.Trainer(net, "crossEntropy", bench.RATE / 1024, T.put(X, batched=True), T.put(Y, batched=True), use_memo=True, use_graph=True)
for _ in range(50): tr.step()
T.sync()
for events in (True, False):
    for K in (1, 2, 5, 10, 20, 50, 100):
        ts = []
        for _ in range(15):
            T.sync()
            t0 = time.perf_counter()
            if events: T.timer_start()
            for _ in range(K): tr.step()
            if events: T.timer_stop()
            T.sync()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print("events %s K %3d: median %.1f us  per step %.2f us" % (events, K, ts[len(ts) // 2] * 1e6, ts[len(ts) // 2] * 1e6 / K))
# what the contract's device-wide wait costs when the device is already idle, and as the call that waits for the steps
ts = []
for _ in range(200):
    T.sync()
    t0 = time.perf_counter(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort(); print("torch.cuda.synchronize() on an idle device: median %.1f us" % (ts[100] * 1e6))
for mode in ("stream wait, then device-wide", "device-wide only"):
    ts = []
    for _ in range(15):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): tr.step()
        if mode.startswith("stream"): T.sync()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort(); print("20 steps, %s: median %.1f us  per step %.2f us" % (mode, ts[7] * 1e6, ts[7] * 1e6 / 20))
# the same 20-step region with the library on a torch-owned stream and the parameters in torch-owned flat buffers (what bench.py does)
import ctypes as C
from tensor_ops_amd import capi
stream = torch.cuda.Stream()
capi.check(capi.lib().to_set_stream(C.c_void_p(stream.cuda_stream)))
with torch.cuda.stream(stream):
    net2 = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
    nflat = tops.Trainer.flat_size(net2)
    fp = torch.zeros(nflat, dtype=torch.float32, device="cuda"); fg = torch.zeros(nflat, dtype=torch.float32, device="cuda")
    stream.synchronize()
    tr2 = tops.Trainer(net2, "crossEntropy", bench.RATE / 1024, T.put(X, batched=True), T.put(Y, batched=True), use_memo=True, use_graph=True,
                       ext_params=fp.data_ptr(), ext_grads=fg.data_ptr())
    for _ in range(50): tr2.step()
    ts = []
    for _ in range(15):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): tr2.step()
        T.sync(); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort(); print("20 steps on a torch-owned stream + torch-owned flat buffers: median %.1f us  per step %.2f us" % (ts[7] * 1e6, ts[7] * 1e6 / 20))
    # which end of the bracket costs what: the region between every combination of the two waits
    def dev(): torch.cuda.synchronize()
    def lib(): T.sync()
    def both(): T.sync(); torch.cuda.synchronize()
    def both_r(): torch.cuda.synchronize(); T.sync()
    for sname, start in (("device-wide", dev), ("to_sync", lib), ("device-wide then to_sync", both_r)):
        for ename, end in (("to_sync", lib), ("device-wide", dev), ("to_sync then device-wide", both)):
            ts = []
            for _ in range(25):
                start()
                t0 = time.perf_counter()
                for _ in range(20): tr2.step()
                end()
                ts.append(time.perf_counter() - t0)
            ts.sort(); print("20 steps, start after %-26s end with %-26s median %.1f us (min %.1f)" % (sname + ",", ename + ":", ts[12] * 1e6, ts[0] * 1e6))
    ts = []
    for _ in range(25):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); tr2.step(); T.sync(); ts.append(time.perf_counter() - t0)
    ts.sort(); print("1 step after a device-wide wait, to_sync: median %.1f us (min %.1f)" % (ts[12] * 1e6, ts[0] * 1e6))
    ts = []
    for _ in range(25):
        T.sync()
        t0 = time.perf_counter(); tr2.step(); T.sync(); ts.append(time.perf_counter() - t0)
    ts.sort(); print("1 step after to_sync, to_sync: median %.1f us (min %.1f)" % (ts[12] * 1e6, ts[0] * 1e6))
