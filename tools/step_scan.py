"""The batched gradTOp + SGD step of an ffLayer stack (logistic hidden layers, softmax, crossEntropy) over batch sizes and widths,
next to the same arithmetic written out in torch (forward, loss-head cotangent, backward, p -= r g; no autograd): us per step.
   usage: step_scan.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tensor_ops_amd import tops as H
from tensor_ops_amd.hipt import HipT
F64 = os.environ.get("SCAN_DTYPE") == "f64"   # SCAN_DTYPE=f64: the fp64 instance against torch in double
DT = np.float64 if F64 else np.float32
TD = torch.float64 if F64 else torch.float32
T = HipT(0, dtype=np.float64) if F64 else HipT(0); H.hlib()
if F64: H.set_elem_dtype(np.float64)
rng = np.random.default_rng(3)


def torch_step(ws, bs, X, Y, rate):
    acts = [X]
    for li, (W, b) in enumerate(zip(ws, bs)):
        z = torch.addmm(b, acts[-1], W.t())
        acts.append(torch.sigmoid(z) if li < len(ws) - 1 else torch.softmax(z, dim=1))
    dz = acts[-1] * Y.sum(dim=1, keepdim=True) - Y      # softmax >>> crossEntropy
    for li in range(len(ws) - 1, -1, -1):
        gW = dz.t() @ acts[li]; gb = dz.sum(dim=0)
        if li > 0:
            dz = (dz @ ws[li]) * acts[li] * (1 - acts[li])
        ws[li].sub_(gW, alpha=rate); bs[li].sub_(gb, alpha=rate)


def time_fn(f, sync, est_iters=5):
    import time
    for _ in range(3): f()
    sync(); t0 = time.perf_counter()
    for _ in range(est_iters): f()
    sync(); est = (time.perf_counter() - t0) / est_iters
    n = max(10, int(0.05 / max(est, 1e-6)))
    for _ in range(n // 2): f()
    sync(); t0 = time.perf_counter()
    for _ in range(n): f()
    sync()
    return (time.perf_counter() - t0) / n * 1e6


for dims in ([784, 256, 10], [784, 300, 100, 10], [1024, 1024, 1024, 10], [4096, 4096, 10]):
    for B in (32, 256, 1024, 8192, 60000):
        if B * max(dims) > (1.5e8 if F64 else 3e8): continue
        ws = [(0.5 * rng.standard_normal((o, i)) / np.sqrt(i), 0.5 * rng.standard_normal(o)) for i, o in zip(dims[:-1], dims[1:])]
        X = rng.uniform(0, 1, (B, dims[0])); Y = np.zeros((B, dims[-1])); Y[np.arange(B), rng.integers(0, dims[-1], B)] = 1
        net = H.genNet([(T.put(w.astype(DT)), T.put(b.astype(DT))) for w, b in ws], "actMapLogistic", "actSoftmax")
        tr = H.Trainer(net, "crossEntropy", 0.01 / B, T.put(X.astype(DT), batched=True), T.put(Y.astype(DT), batched=True))
        ours = time_fn(tr.step, T.sync)
        nl = tr.launches_per_step
        tw = [torch.tensor(w, dtype=TD, device="cuda") for w, _ in ws]; tb = [torch.tensor(b, dtype=TD, device="cuda") for _, b in ws]
        tX = torch.tensor(X, dtype=TD, device="cuda"); tY = torch.tensor(Y, dtype=TD, device="cuda")
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3): torch_step(tw, tb, tX, tY, 0.01 / B)
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                torch_step(tw, tb, tX, tY, 0.01 / B)
        tt = time_fn(g.replay, torch.cuda.synchronize)
        print("%-24s batch %6d   ours %9.1f us (%d launches, graph %s)   torch (graph replay) %9.1f us   ratio %.2f" % ("-".join(map(str, dims)), B, ours, nl, tr.graph, tt, tt / ours), flush=True)
        del tr, net, tw, tb, tX, tY, g
        torch.cuda.empty_cache()
