"""Where does the routing lose?  A coarse grid of fp32 GEMM extents (powers of two next to the ragged sizes of the reference's
networks), ours (gmul through the C ABI) next to the vendor GEMM (torch.mm), short protocol (15 ms warm-up, 20 ms timed);
prints every row and, at the end, the rows below 0.92 of the vendor GEMM, worst first.
   usage: gemm_scan.py [values ...]      (default grid below; SCAN_DTYPE=f64: the fp64 instance against torch.mm in double)"""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensor_ops_amd.hipt import HipT
import numpy as np
F64 = os.environ.get("SCAN_DTYPE") == "f64"
T = HipT(0, dtype=np.float64) if F64 else HipT(0)
TD = torch.float64 if F64 else torch.float32
V = [int(x) for x in sys.argv[1:]] or [8, 32, 100, 256, 300, 784, 1024, 2048, 4096, 10000, 60000]
WARM, TIMED = 15.0, 20.0


def counts(est):
    est = max(est, 1e-3)
    return max(10, int(WARM / est)), max(10, int(TIMED / est))


def ours(m, k, n):
    a = T.genRand((m, k), "uniform", -1, 1, 1); b = T.genRand((k, n), "uniform", -1, 1, 2)

    def run(iters, warm):
        for _ in range(warm): T.gmul(1, 1, 1, a, b)
        T.sync(); T.timer_start()
        for _ in range(iters): T.gmul(1, 1, 1, a, b)
        return T.timer_stop() / iters
    w, i = counts(run(5, 2))
    return run(i, w)


def vendor(m, k, n):
    a = torch.rand(m, k, device="cuda", dtype=TD) * 2 - 1; b = torch.rand(k, n, device="cuda", dtype=TD) * 2 - 1; c = torch.empty(m, n, device="cuda", dtype=TD)

    def run(iters, warm):
        for _ in range(warm): torch.mm(a, b, out=c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): torch.mm(a, b, out=c)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    w, i = counts(run(5, 2))
    return run(i, w)


rows = []
for m, k, n in itertools.product(V, V, V):
    fl = 2.0 * m * k * n
    if fl < 2e7 or fl > (1.5e12 if F64 else 3e12) or max(m * k, k * n, m * n) > (0.6e9 if F64 else 1.2e9):
        continue
    to, tv = ours(m, k, n), vendor(m, k, n)
    rows.append((tv / to, m, k, n, to, tv))
    print("%6d x %6d x %6d   ours %9.4f ms %7.2f TF   vendor %9.4f ms %7.2f TF   ratio %.2f" % (m, k, n, to, fl / to / 1e9, tv, fl / tv / 1e9, tv / to), flush=True)
    torch.cuda.empty_cache()
print("== below 0.92 of the vendor GEMM, worst first (%d of %d rows)" % (sum(r[0] < 0.92 for r in rows), len(rows)))
for r, m, k, n, to, tv in sorted(rows):
    if r < 0.92:
        print("%6d x %6d x %6d   ours %9.4f ms   vendor %9.4f ms   ratio %.2f" % (m, k, n, to, tv, r))
