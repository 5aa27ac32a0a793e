"""One bounded experiment on the pageable-transfer loss (DESIGN_HISTORY.md 11.1): the suite's transfer-heavy file
(tests/test_gpu_fuzz_gemm.py: every test is a fresh tool process that uploads / downloads hundreds of numpy arrays and never
forks) looped eight in flight with the library's fence OFF (TOPS_PINNED_STAGING=0) and the download sentinel on, in two arms
that alternate loop by loop so that both see the same box at the same time:

  as_is      glibc's defaults: numpy's large arrays come from mmap and go back with munmap (fresh physical pages at recycled
             virtual addresses), the heap top is trimmed
  no_return  MALLOC_MMAP_MAX_=0 MALLOC_TRIM_THRESHOLD_=2^62 MALLOC_TOP_PAD_=256 MiB: malloc never uses mmap and never gives
             heap pages back -- a virtual address keeps its physical pages for the life of the process

(The third arm the verdict names -- tools in-process instead of subprocess, i.e. no fork beside in-flight user-pointer mappings --
is settled by reading: the processes that saw the loss in round 5, tools/t32_check.py and tools/kw_epilogue_fuzz.py, never fork.)
usage: pageable_arms.py --budget-s 900 --parallel 8 --out gpurun_out/pageable_arms       Test infrastructure."""
import argparse
import concurrent.futures
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARMS = {
    "as_is": {},
    "no_return": {"MALLOC_MMAP_MAX_": "0", "MALLOC_TRIM_THRESHOLD_": str(1 << 62), "MALLOC_TOP_PAD_": str(256 << 20)},
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--budget-s", type=float, default=900)
    ap.add_argument("--parallel", type=int, default=8)
    ap.add_argument("--max-loops", type=int, default=400)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pageable_arms"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    t_start = time.time()
    lock = threading.Lock()
    names = sorted(ARMS)
    stats = {a: {"loops": 0, "red": 0, "failed_tests": 0, "seconds": 0.0} for a in names}

    def one(loop):
        if time.time() - t_start > args.budget_s:
            return
        arm = names[loop % len(names)]
        env = dict(os.environ, PYTHONPATH=ROOT, TOPS_PINNED_STAGING="0", TOPS_DL_SENTINEL="1", FUZZ_DIAG="1",
                   TOPS_FAILURE_LOG=os.path.join(args.out, "failures_%s.jsonl" % arm),
                   TOPS_MISMATCH_DIR=os.path.join(args.out, "mismatch_%s" % arm), TOPS_STRESS_LOOP=str(loop), TOPS_STRESS_CONDITION=arm, **ARMS[arm])
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_fuzz_gemm.py", "-m", "gpu", "-q", "-p", "no:cacheprovider", "--tb=short", "-rf"],
                               env=env, cwd=ROOT, capture_output=True, text=True, timeout=1800)
            out, rc = r.stdout + r.stderr, r.returncode
        except subprocess.TimeoutExpired as e:
            out, rc = (e.stdout or b"").decode(errors="replace") + "\nTIMEOUT", -9
        dt = time.time() - t0
        failed = [ln.split(" ", 1)[1][:200] for ln in out.splitlines() if ln.startswith("FAILED ")]
        with lock:
            s = stats[arm]
            s["loops"] += 1
            s["red"] += rc != 0
            s["failed_tests"] += len(failed)
            s["seconds"] += dt
            with open(os.path.join(args.out, "summary.jsonl"), "a") as f:
                f.write(json.dumps({"loop": loop, "arm": arm, "rc": rc, "seconds": round(dt, 1), "failed": failed}) + "\n")
            if rc != 0:
                with open(os.path.join(args.out, "loop_%03d_%s.log" % (loop, arm)), "w") as f:
                    f.write(out[-20000:])
            print("loop %d [%s] rc %d %.0fs %s" % (loop, arm, rc, dt, "; ".join(failed)), flush=True)

    with concurrent.futures.ThreadPoolExecutor(max_workers=args.parallel) as ex:
        list(ex.map(one, range(args.max_loops)))
    verdict = {"budget_s": args.budget_s, "parallel": args.parallel, "wall_s": round(time.time() - t_start, 1), "arms": stats,
               "env": {a: ARMS[a] for a in names}}
    with open(os.path.join(args.out, "verdict.json"), "w") as f:
        json.dump(verdict, f, indent=1)
    print(json.dumps(verdict))


if __name__ == "__main__":
    main()
