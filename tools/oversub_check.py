"""Is the eight-loops failure a matter of how many processes hold queues on the GPU?  Starts H idle holders
(a context, a stream, one tiny launch, then sleep), then runs P copies of a bit-exact tool at once -- ours
(kw_epilogue_fuzz.py, FUZZ_DTYPE from the environment) or the CONTROL: torch.mm (the vendor GEMM) on the same kind of
integer operands, compared with numpy's BLAS-free int64 product.  usage: oversub_check.py holders procs ours|torch [passes]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOLDER = r"""
import sys, time
sys.path.insert(0, %r)
from tensor_ops_amd.hipt import HipT
T = HipT(0)
x = T.genRand((256, 256), "uniform", -1, 1, 1); y = T.gmul(1, 1, 1, x, x); T.sync()
time.sleep(float(sys.argv[1]))
""" % ROOT
TORCH = r"""
import sys, numpy as np, torch
rng = np.random.default_rng(int(sys.argv[1]))
bad = 0
for case in range(int(sys.argv[2])):
    M = int(rng.integers(640, 1400)); N = int(rng.integers(130, 700)); K = int(rng.integers(128, 900))
    for dt, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
        X = rng.integers(-2, 3, (M, K)); W = rng.integers(-2, 3, (N, K))
        want = (X.astype(np.int64) @ W.T.astype(np.int64)).astype(dt)
        got = torch.mm(torch.from_numpy(X.astype(dt)).cuda(), torch.from_numpy(W.astype(dt)).cuda().t()).cpu().numpy()
        if not np.array_equal(got, want):
            bad += 1
            r = np.unique(np.nonzero(got != want)[0])
            print("TORCH MISMATCH case", case, (M, K, N), dt.__name__, "rows", r[:10].tolist(), "n", len(r), flush=True)
print("torch cases", int(sys.argv[2]), "mismatches", bad)
"""


def main():
    holders, procs, which = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    passes = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    hs = [subprocess.Popen([sys.executable, "-c", HOLDER, "600"], cwd=ROOT) for _ in range(holders)]
    time.sleep(25 if holders else 0)
    for p in range(passes):
        if which == "ours":
            cmd = [sys.executable, os.path.join(ROOT, "tools", "kw_epilogue_fuzz.py"), "25", "41"]
            ps = [subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, FUZZ_DIAG="1")) for _ in range(procs)]
        else:
            ps = [subprocess.Popen([sys.executable, "-c", TORCH, str(100 + i), "60"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for i in range(procs)]
        for q in ps:
            out = q.communicate()[0]
            lines = [l for l in out.splitlines() if "DIAG" in l or "MISMATCH" in l or "mismatches" in l]
            print("[%s holders %d pass %d]" % (which, holders, p), " | ".join(l[:700] for l in lines[-6:]), flush=True)
    for h in hs:
        h.kill()


if __name__ == "__main__":
    main()
