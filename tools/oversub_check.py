"""Is the eight-loops failure (DESIGN_HISTORY 10.1) a matter of how many processes hold queues on the GPU?  Starts H idle holders
(a context, a stream, one tiny launch, then sleep), then runs P copies of a bit-exact tool at once -- ours
(kw_epilogue_fuzz.py, FUZZ_DTYPE from the environment) or the CONTROL: torch.mm (the vendor GEMM) on the same kind of
integer operands, compared with numpy's BLAS-free int64 product.  `busy` more processes stream `map logistic` / 4096^3
products the whole time.  usage: oversub_check.py holders procs ours|torch [passes] [busy]"""
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
BUSY = r"""
import sys, time
sys.path.insert(0, %r)
from tensor_ops_amd import hipt
from tensor_ops_amd.hipt import HipT
T = HipT(0)
t_end = time.time() + float(sys.argv[1])
if int(sys.argv[2]) %% 2 == 0:     # a stream of `map logistic` over 512^3 (the harness's `corun`)
    x = T.genRand((512, 512, 512), "uniform", -1, 1, 3)
    while time.time() < t_end:
        for _ in range(50):
            y = T.liftT(hipt.logistic_closure, [x], key="busy-logistic"); del y
        T.sync()
else:                              # back-to-back 4096^3 products (the harness's `hot`)
    a = T.genRand((4096, 4096), "uniform", -1, 1, 1); b = T.genRand((4096, 4096), "uniform", -1, 1, 2)
    while time.time() < t_end:
        for _ in range(20):
            c = T.gmul(1, 1, 1, a, b); del c
        T.sync()
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
    busy = int(sys.argv[5]) if len(sys.argv) > 5 else 0     # working competitors (never on the library under test's override)
    clean_env = {k: v for k, v in os.environ.items() if k not in ("TOPS_HIP_LIB", "LD_LIBRARY_PATH")}
    hs = [subprocess.Popen([sys.executable, "-c", HOLDER, "600"], cwd=ROOT, env=clean_env) for _ in range(holders)]
    hs += [subprocess.Popen([sys.executable, "-c", BUSY, "600", str(i)], cwd=ROOT, env=clean_env) for i in range(busy)]
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
