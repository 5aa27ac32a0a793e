"""How long the device takes to reach the step's steady rate after idling, and how fast it falls back: 20-step regions
(between two device-wide waits, as bench.py times them) after (a) a long busy stretch followed by an idle gap of 0 .. 100 ms,
(b) a second of idling, back to back, region by region.  usage: ramp_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


def region(k=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k): tr.step()
    T.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


for gap_ms in (0, 0.2, 1, 5, 20, 100, 1000):
    rs = []
    for _ in range(5):
        for _ in range(4000): tr.step()
        T.sync()
        if gap_ms: time.sleep(gap_ms / 1e3)
        rs.append(region())
    rs.sort()
    print("4000 steps, idle %6.1f ms, then a 20-step region: median %.1f us = %.2f us a step (min %.1f)" % (gap_ms, rs[2], rs[2] / 20, rs[0]))
time.sleep(1.0)
rs = [region() for _ in range(200)]
print("after 1 s of idling, regions back to back, us a step:", " ".join("%.2f" % (r / 20) for r in rs[:12]), "... #50 %.2f #100 %.2f #150 %.2f #199 %.2f" % (rs[50] / 20, rs[100] / 20, rs[150] / 20, rs[199] / 20))
time.sleep(1.0)
# the same with nothing between the regions' steps and the next region but the waits: how many steps until the rate settles
t0 = time.perf_counter(); n = 0; marks = []
while n < 6000:
    r = region(); n += 20
    if n in (20, 40, 100, 200, 400, 1000, 2000, 4000, 6000): marks.append((n, (time.perf_counter() - t0) * 1e3, r / 20))
print("after 1 s of idling:", "; ".join("step %d (%.1f ms in): %.2f us" % m for m in marks))
