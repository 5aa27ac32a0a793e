"""The non-GEMM hot-path ops on a grid of sizes next to their torch equivalents (same protocol as gemm_scan.py): matVec / vecMat
(gemv), outerV (ger), map logistic, sumRows, sumT of two tensors, a packed transpose; prints every row and the rows below 0.9.
   usage: ops_scan.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensor_ops_amd.hipt import HipT, logistic_closure
import numpy as np
F64 = os.environ.get("SCAN_DTYPE") == "f64"   # SCAN_DTYPE=f64: the fp64 instance against torch in double
T = HipT(0, dtype=np.float64) if F64 else HipT(0)
TD = torch.float64 if F64 else torch.float32
ES = 8.0 if F64 else 4.0
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
    print("%-10s %-18s ours %9.4f ms %7.1f GB/s   torch %9.4f ms %7.1f GB/s   ratio %.2f" % (name, shape, to, by / to / 1e6, tv, by / tv / 1e6, tv / to), flush=True)


V = [100, 256, 300, 784, 1024, 4096, 10000, 60000]
for m in V:
    for k in V:
        if m * k > (0.6e9 if F64 else 1.2e9): continue
        a = T.genRand((m, k), "uniform", -1, 1, 1); x = T.genRand((k,), "uniform", -1, 1, 2); y = T.genRand((m,), "uniform", -1, 1, 3)
        ta = torch.rand(m, k, device="cuda", dtype=TD); tx = torch.rand(k, device="cuda", dtype=TD); ty = torch.rand(m, device="cuda", dtype=TD); to_ = torch.empty(m, device="cuda", dtype=TD); tk = torch.empty(k, device="cuda", dtype=TD); tmk = torch.empty(m, k, device="cuda", dtype=TD)
        by = ES * (m * k + m + k)
        report("matVec", "%dx%d" % (m, k), time_ours(lambda: T.matVec(a, x)), time_torch(lambda: torch.mv(ta, tx, out=to_)), by)
        report("vecMat", "%dx%d" % (m, k), time_ours(lambda: T.vecMat(y, a)), time_torch(lambda: torch.mv(ta.t(), ty, out=tk)), by)
        report("outerV", "%dx%d" % (m, k), time_ours(lambda: T.outerV(y, x)), time_torch(lambda: torch.outer(ty, tx, out=tmk)), by)
        report("sumRows", "%dx%d" % (m, k), time_ours(lambda: T.sumRows(a)), time_torch(lambda: torch.sum(ta, dim=0, out=tk)), by)
        report("transp", "%dx%d" % (m, k), time_ours(lambda: T.force(T.sumT([T.transp(a)], (k, m)))), time_torch(lambda: ta.t().contiguous()), 2 * ES * m * k)
        del a, x, y, ta, tx, ty, to_, tk, tmk
        torch.cuda.empty_cache()
e = T.expr(logistic_closure, 1, key="ops-scan-logi")
for n in [1000, 10000, 100000, 10 ** 6, 10 ** 7, 10 ** 8, (25 if F64 else 50) * 10 ** 7]:
    a = T.genRand((n,), "uniform", -1, 1, 1); b = T.genRand((n,), "uniform", -1, 1, 2)
    ta = torch.rand(n, device="cuda", dtype=TD); tb = torch.rand(n, device="cuda", dtype=TD); tc = torch.empty(n, device="cuda", dtype=TD)
    report("logistic", "%d" % n, time_ours(lambda: T.liftT(e, [a])), time_torch(lambda: torch.sigmoid(ta, out=tc)), 2 * ES * n)
    report("add", "%d" % n, time_ours(lambda: T.sumT([a, b], (n,))), time_torch(lambda: torch.add(ta, tb, out=tc)), 3 * ES * n)
    report("scale", "%d" % n, time_ours(lambda: T.scaleT(0.5, a)), time_torch(lambda: torch.mul(ta, 0.5, out=tc)), 2 * ES * n)
    del a, b, ta, tb, tc
    torch.cuda.empty_cache()
print("== below 0.90 of torch, worst first (%d of %d rows)" % (sum(r[0] < 0.90 for r in rows), len(rows)))
for r, name, shape, to, tv in sorted(rows):
    if r < 0.90:
        print("%-10s %-18s ours %9.4f ms   torch %9.4f ms   ratio %.2f" % (name, shape, to, tv, r))
