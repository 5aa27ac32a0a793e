"""Stand-alone attempt at the pageable-transfer loss of DESIGN_HISTORY.md 11.1 in the environment it was seen in: CPython + numpy,
raw HIP through ctypes -- libtensorops_hip is NOT loaded, no kernel of this repository runs.

Box 9 (round 5) lost pieces of `hipMemcpyAsync(pageable, device)` transfers in both directions inside the test suite's
processes (eight sharing the device) with the library's pinned staging switched off; the C++ probe beside this file
(dma_pageable.hip), which recycles malloc / mmap blocks, has moved terabytes without a failure.  This script does what the
suite's processes do around their transfers and the C++ probe does not:
  * the host arrays are numpy's (its allocator, its huge-page advice from 4 MiB, glibc's dynamic mmap threshold at work),
    of the shapes and dtypes the failing cases had (hundreds to ~1300 rows and columns, fp32 and fp64);
  * between transfers a float64 reference product is computed with numpy (BLAS threads, large temporaries freed just before
    the download's destination is allocated);
  * now and then a subprocess is spawned (fork/exec next to pages a transfer may have pinned);
  * downloads go into `np.empty` blocks pre-filled with a NaN sentinel, uploads are read back through PINNED memory.
A worker prints one JSON line; the driver (no arguments = driver) starts N workers and four compute co-runners
(preempt_lds_dma, if built) and prints a summary.   usage: pageable_repro.py [--workers 8] [--seconds 120] [--no-fork]
First run (round 5's last GPU minute, box 11): eight workers + four co-runners for 30 s -- 200 iterations, 1.6 GB, no
failure, no crash: it works mechanically and has proved nothing yet (the suite lost six transfers in two process-hours)."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

H2D, D2H = 1, 2
SENT32 = np.uint32(0x7FC0DEAD)


def hip():
    lib = C.CDLL(os.environ.get("HIP_LIB", "/opt/rocm/lib/libamdhip64.so"))
    lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    lib.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    lib.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    lib.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    lib.hipStreamSynchronize.argtypes = [C.c_void_p]
    lib.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    return lib


def ck(e, what):
    if e != 0:
        raise RuntimeError("%s -> hip error %d" % (what, e))


def runs_of(mask, limit=24):
    """(byte offset, bytes) of the runs of True in a boolean word mask"""
    idx = np.flatnonzero(mask)
    if idx.size == 0:
        return []
    cut = np.flatnonzero(np.diff(idx) > 1)
    starts = np.concatenate([[idx[0]], idx[cut + 1]])
    ends = np.concatenate([idx[cut], [idx[-1]]])
    return [(int(a) * 4, int(b - a + 1) * 4) for a, b in list(zip(starts, ends))[:limit]]


def worker(args):
    lib = hip()
    ck(lib.hipSetDevice(0), "hipSetDevice")
    s = C.c_void_p()
    ck(lib.hipStreamCreateWithFlags(C.byref(s), 1), "hipStreamCreateWithFlags")      # hipStreamNonBlocking
    cap = 16 << 20
    dbuf, pin = C.c_void_p(), C.c_void_p()
    ck(lib.hipMalloc(C.byref(dbuf), cap), "hipMalloc")
    ck(lib.hipHostMalloc(C.byref(pin), cap, 0), "hipHostMalloc")
    pinned = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint32)), shape=(cap // 4,))
    rng = np.random.default_rng(0x7E5 + args.worker * 7919 + os.getpid())
    t_end = time.time() + args.seconds
    rec = {"probe": "pageable_repro", "worker": args.worker, "pid": os.getpid(), "iters": 0, "bytes": 0, "h2d_fail": 0, "d2h_fail": 0,
           "forks": 0, "details": []}
    keep = []                                                 # a few live arrays, so that frees leave holes of every size
    while time.time() < t_end:
        rec["iters"] += 1
        B, K, N = (int(rng.integers(64, 1400)) for _ in range(3))
        dt = np.float64 if rng.integers(0, 2) else np.float32
        X = rng.integers(-3, 4, (B, K)).astype(dt)
        W = rng.integers(-3, 4, (N, K)).astype(dt)
        # ---- upload of W from numpy memory, read back through pinned memory
        ck(lib.hipMemcpyAsync(dbuf, W.ctypes.data_as(C.c_void_p), W.nbytes, H2D, s), "H2D")
        ck(lib.hipStreamSynchronize(s), "sync")
        ck(lib.hipMemcpyAsync(pin, dbuf, W.nbytes, D2H, s), "D2H pinned")
        ck(lib.hipStreamSynchronize(s), "sync")
        got = pinned[: W.nbytes // 4]
        want = W.reshape(-1).view(np.uint32)
        bad = got != want
        if bad.any():
            rec["h2d_fail"] += 1
            if len(rec["details"]) < 8:
                rec["details"].append({"dir": "h2d", "iter": rec["iters"], "shape": [N, K], "dtype": np.dtype(dt).name, "wrong_words": int(bad.sum()),
                                       "src_mod_4096": int(W.ctypes.data % 4096), "runs": runs_of(bad)})
        rec["bytes"] += 2 * W.nbytes
        # ---- the numpy reference: BLAS threads, large float64 temporaries, freed right before the destination is made
        ref = X.astype(np.float64) @ W.T.astype(np.float64)
        chk = float(ref[0, 0])
        del ref
        # ---- a device result of the product's shape (the bytes of W repeated: what matters is that the DEVICE bytes are known)
        n_out = B * N * np.dtype(dt).itemsize
        n_out -= n_out % 4
        n_out = min(n_out, cap)
        ck(lib.hipMemsetAsync(dbuf, 0x5A, n_out, s), "memset")
        dst = np.empty(n_out // 4, dtype=np.uint32)
        dst[:] = SENT32
        ck(lib.hipMemcpyAsync(dst.ctypes.data_as(C.c_void_p), dbuf, n_out, D2H, s), "D2H pageable")
        ck(lib.hipStreamSynchronize(s), "sync")
        bad = dst != np.uint32(0x5A5A5A5A)
        if bad.any():
            rec["d2h_fail"] += 1
            if len(rec["details"]) < 8:
                rec["details"].append({"dir": "d2h", "iter": rec["iters"], "bytes": int(n_out), "wrong_words": int(bad.sum()),
                                       "sentinel_words_left": int((dst == SENT32).sum()), "dst_mod_4096": int(dst.ctypes.data % 4096),
                                       "runs": runs_of(bad)})
        rec["bytes"] += n_out
        if rng.integers(0, 4) == 0:
            keep.append(dst)
            if len(keep) > 6:
                keep.pop(int(rng.integers(0, len(keep))))
        del dst, X, W
        if not args.no_fork and rec["iters"] % 10 == 0:
            subprocess.run(["true"], check=False)              # fork + exec beside whatever the runtime keeps pinned
            rec["forks"] += 1
        if chk != chk:
            print("unreachable", chk)
    rec["GB_moved"] = round(rec["bytes"] / 1e9, 2)
    print(json.dumps(rec), flush=True)
    return 1 if rec["h2d_fail"] or rec["d2h_fail"] else 0


def driver(args):
    here = os.path.dirname(os.path.abspath(__file__))
    co = []
    lds = os.path.join(here, "preempt_lds_dma")
    if os.path.exists(lds) and not args.no_corun:
        co = [subprocess.Popen([lds, str(args.seconds), str(200 + i), "dma", "0"], stdout=subprocess.DEVNULL) for i in range(4)]
    ws = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(i), "--seconds", str(args.seconds)] +
                           (["--no-fork"] if args.no_fork else []), stdout=subprocess.PIPE, text=True) for i in range(args.workers)]
    outs = []
    for w in ws:
        out, _ = w.communicate(timeout=args.seconds * 3 + 300)
        line = [l for l in out.splitlines() if l.startswith("{")]
        outs.append(json.loads(line[-1]) if line else {"crashed": w.returncode})
    for c in co:
        c.wait(timeout=args.seconds * 3 + 300)
    summ = {k: sum(o.get(k, 0) for o in outs) for k in ("iters", "h2d_fail", "d2h_fail", "forks")}
    summ["GB_moved"] = round(sum(o.get("GB_moved", 0) for o in outs), 1)
    summ["crashed"] = sum(1 for o in outs if "crashed" in o)
    summ["co_runners"] = len(co)
    print(json.dumps({"summary": summ, "workers": outs}))
    return 1 if summ["h2d_fail"] or summ["d2h_fail"] else 0


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker", type=int, default=-1)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--no-fork", action="store_true")
    ap.add_argument("--no-corun", action="store_true")
    a = ap.parse_args()
    sys.exit(worker(a) if a.worker >= 0 else driver(a))
