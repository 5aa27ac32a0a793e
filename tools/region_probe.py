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
