"""More forms next to torch (protocol of ops_scan.py): a map over a transposed view (the pack), the sum over the hidden batch,
arg_max over a batch of rows (includes the copy of B indices to the host, as torch's .cpu() does), scaleT of a transposed view,
batch_gather.  usage: ops_scan2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tensor_ops_amd.hipt import HipT, logistic_closure
T = HipT(0)
WARM, TIMED = 15.0, 20.0


def counts(est):
    est = max(est, 1e-3)
    return max(10, int(WARM / est)), max(10, int(TIMED / est))


def time_ours(f):
    def run(iters, warm):
        for _ in range(warm): f()
        T.sync(); T.timer_start()
        for _ in range(iters): f()
        return T.timer_stop() / iters
    w, i = counts(run(5, 2))
    return run(i, w)


def time_torch(f):
    def run(iters, warm):
        for _ in range(warm): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    w, i = counts(run(5, 2))
    return run(i, w)


rows = []


def report(name, shape, to, tv, by):
    rows.append((tv / to, name, shape, to, tv))
    print("%-12s %-18s ours %9.4f ms %7.1f GB/s   torch %9.4f ms %7.1f GB/s   ratio %.2f" % (name, shape, to, by / to / 1e6, tv, by / tv / 1e6, tv / to), flush=True)


e = T.expr(logistic_closure, 1, key="ops2-logi")
V = [100, 300, 1024, 4096, 10000, 60000]
for m in V:
    for k in V:
        if m * k > 6e8: continue
        a = T.genRand((m, k), "uniform", -1, 1, 1)
        ta = torch.rand(m, k, device="cuda"); tout = torch.empty(k, m, device="cuda")
        report("map_transp", "%dx%d" % (m, k), time_ours(lambda: T.liftT(e, [T.transp(a)])), time_torch(lambda: torch.sigmoid(ta.t(), out=tout)), 8.0 * m * k)
        report("scale_transp", "%dx%d" % (m, k), time_ours(lambda: T.scaleT(0.5, T.transp(a))), time_torch(lambda: torch.mul(ta.t(), 0.5, out=tout)), 8.0 * m * k)
        del a, ta, tout
        torch.cuda.empty_cache()
for B in [1000, 10000, 60000, 1000000]:
    for n in [10, 100, 784, 4096]:
        if B * n > 6e8: continue
        x = T.genRand((n,), "uniform", -1, 1, 3, batch=B)
        tx = torch.rand(B, n, device="cuda"); tn = torch.empty(n, device="cuda")
        report("batch_sum", "%dx%d" % (B, n), time_ours(lambda: T.batch_sum(x)), time_torch(lambda: torch.sum(tx, dim=0, out=tn)), 4.0 * B * n)
        report("arg_max", "%dx%d" % (B, n), time_ours(lambda: T.arg_max(x)), time_torch(lambda: torch.argmax(tx, dim=1).cpu()), 4.0 * B * n)
        idx = np.random.default_rng(1).integers(0, B, size=min(B, 4096))
        tidx = torch.from_numpy(idx).cuda()
        report("batch_gather", "%dx%d" % (B, n), time_ours(lambda: T.batch_gather(x, idx)), time_torch(lambda: tx[tidx]), 8.0 * len(idx) * n)
        del x, tx
        torch.cuda.empty_cache()
print("== below 0.90 of torch, worst first (%d of %d rows)" % (sum(r[0] < 0.90 for r in rows), len(rows)))
for r, name, shape, to, tv in sorted(rows):
    if r < 0.90:
        print("%-12s %-18s ours %9.4f ms   torch %9.4f ms   ratio %.2f" % (name, shape, to, tv, r))
