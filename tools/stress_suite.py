"""Loop the bit-exact subset of the GPU suite (or all of it) under conditions that move timing, and write down WHAT fails.

  python tools/stress_suite.py --loops 100                      # the bit-exact subset, conditions rotated
  python tools/stress_suite.py --loops 5 --subset full          # every -m gpu test
  python tools/stress_suite.py --loops 20 --conditions hot,corun

Conditions (rotated loop by loop): `cold` (2 s of idle first: the chip drops to its idle clock), `hot` (the loop starts behind
150 ms of back-to-back 4096^3 products: 2.4 GHz, warm caches), `corun` (a second process streams `map logistic` over 512^3 on
the same GPU for the whole loop: the kernels under test share HBM, the fabric and the CUs with it), `kw2` (TOPS_GEMM_KW=2
TOPS_GEMM64_KW=2: the wave-split kernels wherever they can run -- A/B knobs, so the loop runs on the development library
of tools/build_ab_lib.py when it has been built; on the product library it is one more plain loop).
`--parallel P` keeps P loops in flight at once on the one GPU (each loop is host-bound: numpy references, process start-up).
Every loop is one pytest process (`-p no:cacheprovider --tb=short -rf`, no -x).  Per loop one line goes to
<out>/summary.jsonl; a failing loop keeps its whole log (<out>/loop_NNN_<condition>.log), tests/conftest.py appends each
failing node id with its assertion text to <out>/failures.jsonl and tools/mismatch_report.py writes the mismatching tiles /
waves / k-ranges of a failed bit-exact comparison to <out>/mismatch/.  The last line printed is the verdict:
`stress: L loops, G green, consecutive green C, failures F`.  Test infrastructure."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV_LIB = os.path.join(ROOT, "tensor-ops_amd", "build_ab", "libtensorops_hip.so")


def _sha(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:12]
    except OSError:
        return "missing"


LIB_SHA = _sha(os.path.join(ROOT, "tensor-ops_amd", "libtensorops_hip.so"))
BITEXACT = ["tests/test_gpu_full_size.py", "tests/test_gpu_fuzz_gemm.py", "tests/test_gpu_f64.py", "tests/test_golden.py"]
CORUN = r"""
import os, sys, time
sys.path.insert(0, %r)
from tensor_ops_amd import hipt
from tensor_ops_amd.hipt import HipT
T = HipT(0)
x = T.genRand((512, 512, 512), "uniform", -1, 1, 3)
stop = sys.argv[1]
while not os.path.exists(stop):
    for _ in range(200):
        y = T.liftT(hipt.logistic_closure, [x], key="corun-logistic")
        del y
    T.sync()
""" % ROOT
HOT = r"""
import sys
sys.path.insert(0, %r)
from tensor_ops_amd.hipt import HipT
T = HipT(0)
a = T.genRand((4096, 4096), "uniform", -1, 1, 1); b = T.genRand((4096, 4096), "uniform", -1, 1, 2)
for _ in range(160):
    c = T.gmul(1, 1, 1, a, b); del c
T.sync()
""" % ROOT


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--loops", type=int, default=10)
    ap.add_argument("--subset", default="bitexact", choices=["bitexact", "full"])
    ap.add_argument("--conditions", default="hot,cold,corun,kw2")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "stress"))
    ap.add_argument("--budget-s", type=float, default=0, help="stop starting new loops after this many seconds")
    ap.add_argument("-k", default=None, help="pytest -k expression")
    ap.add_argument("--parallel", type=int, default=1, help="loops in flight at once (they share the GPU: more contention, less wall time)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    conds = [c for c in args.conditions.split(",") if c]
    files = BITEXACT if args.subset == "bitexact" else ["tests"]
    summary = os.path.join(args.out, "summary.jsonl")
    t_start = time.time()
    green = consecutive = failures = done = 0
    import concurrent.futures
    import threading
    lock = threading.Lock()

    def one(loop):
        if args.budget_s and time.time() - t_start > args.budget_s:
            return None
        cond = conds[loop % len(conds)]
        env = dict(os.environ, TOPS_FAILURE_LOG=os.path.join(args.out, "failures.jsonl"),
                   TOPS_MISMATCH_DIR=os.path.join(args.out, "mismatch"), TOPS_STRESS_LOOP=str(loop), TOPS_STRESS_CONDITION=cond, FUZZ_DIAG="1")
        co = None
        stop = os.path.join(args.out, ".corun_stop_%d_%d" % (os.getpid(), loop))
        if cond == "cold":
            time.sleep(2.0)
        elif cond == "hot":
            subprocess.run([sys.executable, "-c", HOT], env=env, cwd=ROOT, timeout=300)
        elif cond == "corun":
            if os.path.exists(stop):
                os.remove(stop)
            co = subprocess.Popen([sys.executable, "-c", CORUN, stop], env=env, cwd=ROOT)
            time.sleep(3.0)
        elif cond == "kw2":
            # A/B knobs: read by a development library only (tools/build_ab_lib.py gemm_kwave.hip gemm_kwave_f64.hip); the
            # ctypes layer loads it through TOPS_HIP_LIB and the host mirror's dependency resolves to the same file
            env.update(TOPS_GEMM_KW="2", TOPS_GEMM64_KW="2")
            if os.path.exists(DEV_LIB):
                env.update(TOPS_HIP_LIB=DEV_LIB, LD_LIBRARY_PATH=os.path.dirname(DEV_LIB) + ":" + env.get("LD_LIBRARY_PATH", ""))
        cmd = [sys.executable, "-m", "pytest"] + files + ["-m", "gpu", "-q", "-p", "no:cacheprovider", "--tb=short", "-rf"]
        if args.k:
            cmd += ["-k", args.k]
        t0 = time.time()
        try:
            r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=3600)
            out, rc = r.stdout + r.stderr, r.returncode
        except subprocess.TimeoutExpired as e:
            out, rc = (e.stdout or b"").decode(errors="replace") + "\nTIMEOUT", -9
        # fresh extents every loop: the suite's own fuzz tests run fixed seeds
        for tool, n, dtype in (() if args.k else (("pinned_fuzz.py", 6, "f32"), ("pinned_fuzz.py", 4, "f64"), ("gemm_fuzz.py", 40, "f32"))):
            try:
                r2 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(n), str(7000 + loop)],
                                    env=dict(env, FUZZ_DTYPE=dtype), cwd=ROOT, capture_output=True, text=True, timeout=1200)
                o2, rc2 = r2.stdout + r2.stderr, r2.returncode
            except subprocess.TimeoutExpired:
                o2, rc2 = "TIMEOUT", -9
            if rc2 != 0 or "mismatches 0" not in o2:
                rc = rc or 1
                out += "\nFAILED tools/%s seed %d %s\n%s" % (tool, 7000 + loop, dtype, o2[-4000:])
        dt = time.time() - t0
        if co is not None:
            open(stop, "w").close()
            try:
                co.wait(timeout=60)
            except subprocess.TimeoutExpired:
                co.kill()
            os.remove(stop)
        tail = [ln for ln in out.splitlines() if " passed" in ln or " failed" in ln or " error" in ln]
        failed = [ln.split(" ", 1)[1] for ln in out.splitlines() if ln.startswith("FAILED ")]
        rec = {"loop": loop, "condition": cond, "lib": LIB_SHA + ("+dev" if env.get("TOPS_HIP_LIB") else ""), "subset": args.subset, "parallel": args.parallel, "rc": rc, "seconds": round(dt, 1),
               "result": tail[-1].strip("= ") if tail else None, "failed": failed}
        with lock:
            with open(summary, "a") as f:
                f.write(json.dumps(rec) + "\n")
            if rc != 0:
                with open(os.path.join(args.out, "loop_%03d_%s.log" % (loop, cond)), "w") as f:
                    f.write(out)
            print("loop %d [%s] rc %d %.0fs %s" % (loop, cond, rc, dt, rec["result"] or ""), flush=True)
        return rec

    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, args.parallel)) as ex:
        for rec in ex.map(one, range(args.loops)):     # (results in loop order: "consecutive" means what it says)
            if rec is None:
                continue
            ok = rec["rc"] == 0
            done += 1
            green += ok
            consecutive = consecutive + 1 if ok else 0
            failures += len(rec["failed"]) + (0 if ok or rec["failed"] else 1)
    print("stress: %d loops, %d green, consecutive green %d, failures %d" % (done, green, consecutive, failures))
    return 0 if green == done else 1


if __name__ == "__main__":
    sys.exit(main())
