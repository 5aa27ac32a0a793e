"""Square and skinny fp32 GEMM sweep: ours (gmul through the C ABI) next to the vendor GEMM (torch.mm), same protocol:
60 ms of warm-up launches, 40 ms timed (a short kernel timed over a few dozen launches runs at ~2.0 GHz, not 2.39)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensor_ops_amd.hipt import HipT
T = HipT(0)


WARM_MS, TIMED_MS = 60.0, 40.0   # steady state: the clock follows the load with a lag of tens of milliseconds


def counts(est_ms):
    est_ms = max(est_ms, 1e-3)
    return max(20, int(WARM_MS / est_ms)), max(20, int(TIMED_MS / est_ms))


def ours(m, k, n):
    a = T.genRand((m, k), "uniform", -1, 1, 1); b = T.genRand((k, n), "uniform", -1, 1, 2)

    def run(iters, warm):
        for _ in range(warm): T.gmul(1, 1, 1, a, b)
        T.sync(); T.timer_start()
        for _ in range(iters): T.gmul(1, 1, 1, a, b)
        return T.timer_stop() / iters
    warm, iters = counts(run(20, 5))
    return run(iters, warm)


def vendor(m, k, n):
    a = torch.rand(m, k, device="cuda") * 2 - 1; b = torch.rand(k, n, device="cuda") * 2 - 1; c = torch.empty(m, n, device="cuda")

    def run(iters, warm):
        for _ in range(warm): torch.mm(a, b, out=c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): torch.mm(a, b, out=c)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    warm, iters = counts(run(20, 5))
    return run(iters, warm)


shapes = [(s, s, s) for s in (512, 768, 1024, 1280, 1536, 2048, 2560, 3072, 3584, 4096, 5120, 6144, 8192)]
shapes += [(8192, 512, 8192), (16384, 256, 4096), (16384, 512, 4096), (4096, 16384, 4096), (1024, 8192, 1024), (4096, 784, 256), (4000, 4000, 4000), (1000, 1000, 1000)]
if len(sys.argv) > 1:   # gemm_sweep.py M K N [M K N ...]: these shapes instead
    v = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(v[i:i + 3]) for i in range(0, len(v), 3)]
for m, k, n in shapes:
    fl = 2.0 * m * k * n
    to, tv = ours(m, k, n), vendor(m, k, n)
    print("%6d x %6d x %6d   ours %8.4f ms %7.2f TF   vendor %8.4f ms %7.2f TF   ratio %.2f" % (m, k, n, to, fl / to / 1e9, tv, fl / tv / 1e9, tv / to))
