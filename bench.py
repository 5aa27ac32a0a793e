#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X:
    "gradTOp steps/sec (ffLayer MNIST 784->256->10) + gmul TFLOP/s vs roofline"

  value      batched-gradTOp steps per second, whole job.  A "step" is one pass of the hot
             path over one 1024-sample batch (BASELINE config 3): batched gradTOp of the
             784->256->10 ffLayer stack (hidden `actMap logistic`, output softmax, loss
             crossEntropy, app/MNIST.hs:264-265,396) + the SGD update p <- p - r*G.
             The network is the reference's `Network` record -- an op and its parameters, no
             activation tags -- driven through gradTOp; the launches it collapses into are found
             by the library in the class-method stream (csrc/lazy.cpp).  --steps K steps are
             timed in each of 5 back-to-back regions (barrier + synchronize on both sides of
             each, max over ranks); value comes from the MEDIAN region, all five are listed.
             On N GPUs every rank steps its own 1024-row shard of a 1024*N global batch
             (config 4) with one RCCL all-reduce of the flat weight gradients per step, so
             the job completes N shard-steps per synchronised iteration ("weak" scaling).
  roofline   the `gmul '[4096,4096] x '[4096,4096]` fp32 GEMM kernel (config 2), timed with
             HIP events on the stream it is launched on, against the dense fp32 MFMA peak.
  extra      config 5 (rank>2 gmul + mapped logistic, HBM roofline) and the step's own
             flop rate, in the same JSON line.
  cpu_baseline  the oracle's plain-C restatement of the reference's per-sample HMat path
             (single thread) on a bounded sample of the same workload.

Inputs are synthetic (seed 0x7e500001) and resident in HBM before the timed region.
Usage: python bench.py [--gpus N --steps K --warmup W]; for N>1 launch through
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 0x7E500001
SIZES = (784, 256, 10)
RATE = 0.02                       # app/MNIST.hs:93
PEAK_MFMA_F32_TF = 157.3          # dense fp32 MFMA, MI355X (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0             # HBM3E spec
PEAK_MFMA_F64_TF = 78.6           # MI355X fp64 matrix spec (half the fp32 MFMA rate)
STEP_FLOPS = 837_812_224          # GEMM flops of one B=1024 step (dX1 excluded)


def synth(rank, batch):
    """Parameters from the global seed (replicated); the shard's rows from seed + 1 + rank."""
    i, h, o = SIZES
    rp = np.random.default_rng(SEED)
    ws = [(0.5 * rp.standard_normal((h, i)), 0.5 * rp.standard_normal(h)),     # normalDistr 0 0.5
          (0.5 * rp.standard_normal((o, h)), 0.5 * rp.standard_normal(o))]     # (FeedForward.hs:206-207)
    rd = np.random.default_rng(SEED + 1 + rank)
    X = rd.uniform(0, 1, size=(batch, i))                                      # pixel/255 (MNIST.hs:207)
    Y = np.zeros((batch, o))
    Y[np.arange(batch), rd.integers(0, o, size=batch)] = 1.0                   # one-hot (MNIST.hs:211)
    return ws, X, Y


def time_launches(T, fn, iters, warm=3):
    for _ in range(warm):
        fn()
    T.sync()
    T.timer_start()
    for _ in range(iters):
        fn()
    return T.timer_stop() / iters  # ms per launch, HIP events on the launch stream


def time_steady(T, fn, warm_ms=60.0, timed_ms=40.0):
    """ms per launch in steady state for a SHORT kernel: the chip's clock follows the load with a lag of tens of
    milliseconds (a 50-launch loop of a 23 us kernel runs at ~2.0 GHz, the same loop after 60 ms of them at 2.37-2.39 GHz),
    so warm up and time by duration, not by count."""
    est = time_launches(T, fn, 20, warm=5)
    est = max(est, 1e-3)
    return time_launches(T, fn, max(20, int(timed_ms / est)), warm=max(20, int(warm_ms / est)))


def sample_power_state(T, fn, seconds=1.5, batch=50):
    """Run `fn` back to back for `seconds` while rocm-smi is sampled every ~50 ms: median ms per launch (HIP events),
    mean shader clock and socket power over the last two thirds of the samples.  Config 5 runs at the socket power cap
    (~1.39 kW) with sclk throttled to ~2.0 GHz while config 2 runs at 2.39 GHz / 1.24 kW: that, not the schedule, is the
    box-to-box spread of the config-5 time, and the dense-MFMA bound at the clock the chip actually holds is
    109 us * 2400 / sclk."""
    import subprocess
    import threading
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True,
                                     text=True, timeout=10).stdout
                c = json.loads(out)
                c = c.get("card%d" % T_DEVICE[0], c)
                sclk = [v for k, v in c.items() if "sclk clock speed" in k][0]
                pw = [v for k, v in c.items() if "Power" in k][0]
                samples.append((float(sclk.strip("()Mhz")), float(pw)))
            except Exception:  # noqa: BLE001
                return
            time.sleep(0.05)
    for _ in range(20):
        fn()
    T.sync()
    th = threading.Thread(target=sampler)
    th.start()
    ms = []
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        T.timer_start()
        for _ in range(batch):
            fn()
        ms.append(T.timer_stop() / batch)
    stop[0] = True
    th.join()
    out = {"ms_per_launch_sustained": round(sorted(ms)[len(ms) // 2], 4)}
    s = samples[len(samples) // 3:]
    if s:
        out["sclk_mhz"] = round(sum(x[0] for x in s) / len(s))
        out["socket_power_w"] = round(sum(x[1] for x in s) / len(s))
    else:
        out["note"] = "rocm-smi not available"
    return out


T_DEVICE = [0]


def pmc_traffic(kernel_key):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), if any."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(path)).get(kernel_key)
    except Exception:
        return None


def pmc_traffic_source():
    """where `roofline.traffic` comes from: NOT measured in this run (counters need rocprofv3 around the process) but read
    from the committed summary of the PMC passes; the file names its round"""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(path))
        return "profiles/pmc_traffic.json (%s; rocprofv3 --pmc passes of tools/collect_profiles.sh, not this run)" % d.get("_round", "round unknown")
    except Exception:
        return None


def exchange_roofline(world, payload, collective, collective_us):
    """N > 1: the exchange as a roofline object.  One all-reduce(sum) of the flat gradient (814,128 B) per step; the
    one-shot peer-to-peer form moves 2 (N-1)/N of the payload out of (and into) every rank, spread over its N-1 links.
    `achieved` = those bytes over the transport's measured stand-alone latency (config.collective_us_alone); `peak` = one
    xGMI link (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU).  The payload is latency-bound: the fractions say how far."""
    key = {"direct": "rccl_to_comm_allreduce_sum", "p2p": "p2p_one_shot_to_p2p_allreduce_sum"}.get(collective)
    us = (collective_us or {}).get(key) if key else None
    out = {"bound": "xgmi-latency", "achieved": None, "peak": 153.0, "unit": "GB/s", "frac": None, "traffic": None,
           "kernel": {"direct": "RCCL all-reduce (to_comm_allreduce_sum)", "p2p": "p2p_allreduce_kernel (csrc/p2p.hip)",
                      "torch": "torch.distributed nccl all-reduce"}.get(collective, collective),
           "payload_bytes": payload, "us_alone": us}
    if us:
        moved = 2.0 * (world - 1) / world * payload          # bytes out of (= into) one rank
        out["bytes_out_per_rank"] = int(moved)
        out["achieved"] = round(moved / (us * 1e-6) / 1e9, 2)
        out["frac"] = round(out["achieved"] / 153.0, 4)                    # of ONE link
        out["frac_of_7_links"] = round(out["achieved"] / (7 * 153.0), 4)   # of everything a rank has
        out["wire_time_us_at_peak"] = {"one_link": round(moved / 153e9 * 1e6, 2),
                                       "n_minus_1_links": round(moved / (max(world - 1, 1) * 153e9) * 1e6, 2)}   # (world 1, forced: nothing leaves the rank)
    return out


def step_variants(T):
    """the step with and without the SGD update, graph replay against direct issue, the
    logistic + squaredError variant of the same stack (the Dots-style head), and the same class-method stream
    with the library's fusion switched off (one launch per call: what round 1's generic path was)."""
    from tensor_ops_amd import tops
    ws, X, Y = synth(0, 1024)
    dX, dY = T.put(X, batched=True), T.put(Y, batched=True)
    out = {}

    def timed(tr, with_update):
        def f():
            tr.grad()
            if with_update:
                tr.apply()
        return round(time_launches(T, f, 300, warm=30), 5)

    def whole_step(tr):
        return round(time_launches(T, tr.step, 300, warm=30), 5)

    for name, hid, head, loss in (("softmax_crossEntropy", "actMapLogistic", "actSoftmax", "crossEntropy"),
                                  ("logistic_squaredError", "actLogistic", "actLogistic", "squaredError")):
        net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], hid, head)
        tr = tops.Trainer(net, loss, RATE, dX, dY, use_graph=True)
        v = {"ms_grad_only": timed(tr, False), "ms_grad_then_sgd_launch": timed(tr, True),
             "ms_step": whole_step(tr), "launches_grad": tr.launches_per_step, "launches_step": tr.step_launches,
             "graph_replay": True}
        del tr
        tr = tops.Trainer(net, loss, RATE, dX, dY, use_graph=False)
        # no capture: the 54 class-method calls of a step cross the C ABI every step, planned and launched each time
        # (the mirror evaluates gradTOp's thunk graph, built once; *_fresh_thunks builds it per step like the reference)
        v["ms_step_issued_directly"] = whole_step(tr)
        del tr
        tr = tops.Trainer(net, loss, RATE, dX, dY, use_graph=False, fresh_thunks=True)
        v["ms_step_issued_directly_fresh_thunks"] = whole_step(tr)
        del tr
        out[name] = v
    net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
    tr = tops.Trainer(net, "crossEntropy", RATE, dX, dY, use_fused=False)
    out["fusion_off_one_launch_per_method_call"] = {"ms_grad_and_sgd": timed(tr, True),
                                                     "launches": tr.launches_per_step + 1, "graph_replay": tr.graph}
    return out


def online_sgd_leg(T):
    """The reference's own training loop -- per-sample online SGD, `foldl' trainNetwork` (app/MNIST.hs:390-396) -- on the
    app's default stack 784 -> 300 -> 100 -> 10 over resident synthetic samples: the persistent one-XCD kernel the
    library substitutes when it recognises the captured step (csrc/online_sgd.hip), and the same loop as one replayed
    step (6 launches) per sample."""
    from tensor_ops_amd import tops
    rng = np.random.default_rng(SEED + 21)
    sizes = [784, 300, 100, 10]
    n = 20000
    ws = [(rng.normal(0, 0.5, size=(o, i)) / np.sqrt(i), rng.normal(0, 0.5, size=o)) for i, o in zip(sizes, sizes[1:])]
    X = rng.uniform(0, 1, size=(n, 784))
    Y = np.zeros((n, 10))
    Y[np.arange(n), rng.integers(0, 10, size=n)] = 1.0
    net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
    dX, dY = T.put(X, batched=True), T.put(Y, batched=True)
    res = {"stack": "784->300->100->10, actMap logistic, softmax, crossEntropy, rate 0.02 (app/MNIST.hs defaults)"}
    for key, flag, m in (("persistent_kernel", "1", n), ("replayed_step_per_sample", "0", 4000)):
        os.environ["TOPS_ONLINE_KERNEL"] = flag
        best = None
        for _ in range(2):
            T.sync()
            t0 = time.perf_counter()
            trained = tops.trainAll(net, "crossEntropy", RATE, dX, dY, n=m)
            T.sync()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        finite = all(bool(np.isfinite(p.numpy()).all()) for p in trained.params)
        res[key] = {"samples": m, "us_per_sample": round(best / m * 1e6, 2), "samples_per_s": round(m / best, 1),
                    "params_finite": finite}
    os.environ.pop("TOPS_ONLINE_KERNEL", None)
    # what the parameters were drawn from -- and that it is NOT the reference's draw
    res["init"] = ("W ~ N(0, 0.5) / sqrt(fan_in), b ~ N(0, 0.5).  NOT the reference's W ~ N(0, 0.5) (FeedForward.hs:206-207): under "
                   "that draw 784 inputs in [0, 1) put every hidden pre-activation at |z| ~ 8 and this leg's 20,000 fp32 steps of rate 0.02 "
                   "leave the finite range; the CPU legs below start from the SAME parameters and samples.  (The app-level leg runs the "
                   "app as it is, reference draw included.)")
    res["cpu_baseline"] = online_cpu_baseline(ws, X, Y)
    pk = res.get("persistent_kernel", {}).get("samples_per_s")
    cb = res["cpu_baseline"]
    if pk and isinstance(cb, dict):
        res["gpu_over_cpu_same_precision"] = {
            "f32_kernel_over_f32_port_with_recompute": round(pk / cb["f32"]["with_reference_recompute"]["samples_per_s"], 1),
            "f32_kernel_over_f32_port_without_recompute": round(pk / cb["f32"]["each_primitive_once"]["samples_per_s"], 1)}
    res["app"] = app_level_leg()
    return res


def online_cpu_baseline(ws, X, Y, seconds=3.0):
    """The reference's OWN loop on the host: `foldl' trainNetwork` per sample on 784 -> 300 -> 100 -> 10 (oracle/hmat_path.c
    hmat_train_online_stack, one thread), fp64 and the f32 build of the same text, with the reference's forward recomputation
    (Types.hs:155: a hidden layer's `W a + b` three times a sample) and with every primitive once.  ~`seconds` per leg."""
    try:
        from oracle import hmat
        out = {"cores": 1, "kind": "port", "unit": "samples/s",
               "what": "oracle/hmat_path.c hmat_train_online_stack: per-sample gemv / axpy / liftB / ger + `p - r g` over every parameter, "
                       "the same samples and initial parameters as the GPU legs; CPU restatement of the hmatrix path, not GHC-compiled tensor-ops"}
        for prec, f32 in (("f64", False), ("f32", True)):
            legs = {}
            for label, rec in (("with_reference_recompute", True), ("each_primitive_once", False)):
                t = time.perf_counter()
                hmat.train_online_stack(X[:64], Y[:64], ws, RATE, recompute=rec, f32=f32)
                per = (time.perf_counter() - t) / 64
                n = int(max(128, min(len(X), seconds / max(per, 1e-6))))
                t = time.perf_counter()
                p, loss = hmat.train_online_stack(X[:n], Y[:n], ws, RATE, recompute=rec, f32=f32)
                dt = time.perf_counter() - t
                legs[label] = {"samples": n, "us_per_sample": round(dt / n * 1e6, 1), "samples_per_s": round(n / dt, 1),
                               "params_finite": bool(all(np.isfinite(w).all() and np.isfinite(b).all() for w, b in p))}
            out[prec] = legs
        return out
    except Exception as e:  # noqa: BLE001
        return "unavailable: %s" % e


def app_level_leg():
    """`tensor-ops-mnist-hip` (host/apps/mnist.cpp = app/MNIST.hs on the backend) on a synthetic IDX-format set of MNIST's size,
    its own defaults (layers [300, 100], rate 0.02, batch 5000, the reference's N(0, 0.5) draw): what it prints for two batches --
    "Trained on N samples in ..." (the reference's own `time` of `trainAll`, app/MNIST.hs:390-398) and the two validation folds
    (runNetwork + argMax over the batch and the 10,000 validation samples, :335-389) -- beside the same two phases of the C port
    on one host thread."""
    import re
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(here, "tensor-ops_amd", "tensor-ops-mnist-hip")
    out = {"command": "tensor-ops-mnist-hip --synthetic 60000,10000 --batch 5000 --epochs 1 --max-batches 2 --noconfusion"}
    try:
        r = subprocess.run([exe, "--synthetic", "60000,10000", "--batch", "5000", "--epochs", "1", "--max-batches", "2", "--noconfusion"],
                           capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            return dict(out, error=(r.stderr or r.stdout)[-400:])
        tr = [(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"Trained on (\d+) samples in ([0-9.]+)s", r.stdout)]
        va = [(int(m.group(1)), int(m.group(2)), float(m.group(3))) for m in re.finditer(r"Validated on (\d+) \+ (\d+) samples in ([0-9.]+)s", r.stdout)]
        err = [float(m.group(1)) for m in re.finditer(r"Validation: ([0-9.]+)% error", r.stdout)]
        n_tr, s_tr = tr[-1]                       # (the second batch: the first carries the capture and the kernel's first launch)
        n_b, n_v, s_va = va[-1]
        out["gpu"] = {"trained_samples": n_tr, "train_seconds": s_tr, "train_samples_per_s": round(n_tr / s_tr, 1),
                      "first_batch_train_seconds": tr[0][1],
                      "validated_samples": n_b + n_v, "validate_seconds": s_va, "validate_samples_per_s": round((n_b + n_v) / s_va, 1),
                      "validation_error_percent_after_each_batch": err}
    except Exception as e:  # noqa: BLE001
        return dict(out, error=str(e))
    try:
        from oracle import hmat
        rng = np.random.default_rng(SEED + 31)
        sizes = [784, 300, 100, 10]
        ws = [(rng.normal(0, 0.5, size=(o, i)), rng.normal(0, 0.5, size=o)) for i, o in zip(sizes, sizes[1:])]
        n_t, n_val = 1500, 6000                    # a bounded sample of the same two phases (~3 s each), scaled to the app's counts
        X = rng.uniform(0, 1, size=(max(n_t, n_val), 784))
        Y = np.zeros((n_t, 10))
        Y[np.arange(n_t), rng.integers(0, 10, size=n_t)] = 1.0
        cpu = {}
        for prec, f32 in (("f64", False), ("f32", True)):
            t = time.perf_counter()
            hmat.train_online_stack(X[:n_t], Y, ws, RATE, recompute=True, f32=f32)
            dt_t = time.perf_counter() - t
            t = time.perf_counter()
            hmat.classify_stack(X[:n_val], ws, f32=f32)
            dt_v = time.perf_counter() - t
            cpu[prec] = {"train_samples_per_s": round(n_t / dt_t, 1), "validate_samples_per_s": round(n_val / dt_v, 1),
                         "train_seconds_for_the_apps_batch": round(out["gpu"]["trained_samples"] / (n_t / dt_t), 2),
                         "validate_seconds_for_the_apps_folds": round(out["gpu"]["validated_samples"] / (n_val / dt_v), 2),
                         "sample": "%d samples trained (%.1f s), %d classified (%.1f s), one thread" % (n_t, dt_t, n_val, dt_v)}
        out["cpu_port"] = cpu
        out["gpu_over_cpu_f32"] = {"train": round(out["gpu"]["train_samples_per_s"] / cpu["f32"]["train_samples_per_s"], 1),
                                   "validate": round(out["gpu"]["validate_samples_per_s"] / cpu["f32"]["validate_samples_per_s"], 1)}
    except Exception as e:  # noqa: BLE001
        out["cpu_port"] = "unavailable: %s" % e
    return out


def aux_benchmarks(T):
    from tensor_ops_amd.hipt import logistic_closure
    out = {}
    out["step_variants"] = step_variants(T)
    out["online_sgd"] = online_sgd_leg(T)
    # ---- config 2: gmul '[4096,4096] x '[4096,4096] fp32 ----
    n = 4096
    a = T.genRand((n, n), "uniform", -1.0, 1.0, SEED + 11)
    b = T.genRand((n, n), "uniform", -1.0, 1.0, SEED + 12)
    # steady state: the chip needs ~30 back-to-back launches of this kernel before its duration settles
    # (1.18 ms for the 4th launch, 1.05 ms from the 30th on)
    ms = time_launches(T, lambda: T.gmul(1, 1, 1, a, b), 50, warm=30)
    flops = 2.0 * n * n * n
    tf = flops / ms / 1e9
    out["roofline"] = {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_MFMA_F32_TF,
                       "unit": "TFLOP/s", "frac": round(tf / PEAK_MFMA_F32_TF, 4),
                       "traffic": pmc_traffic("gmul_4096"), "traffic_source": pmc_traffic_source(),
                       "kernel": "gemm_mfma_kernel<256,256,16,2,2,0,0,5> (gmul '[4096,4096]x'[4096,4096], "
                                 "137,438,953,472 flop/launch)",
                       "ms_per_launch": round(ms, 4)}
    out["roofline"]["power_state"] = sample_power_state(T, lambda: T.gmul(1, 1, 1, a, b), batch=10)
    del a, b
    # ---- config 5a: gmul '[512,512,64] x '[64,512]  (rank > 2: ONE flat GEMM) ----
    a = T.genRand((512, 512, 64), "uniform", -1.0, 1.0, SEED + 13)
    b = T.genRand((64, 512), "uniform", -1.0, 1.0, SEED + 14)
    ms5 = time_launches(T, lambda: T.gmul(2, 1, 1, a, b), 200, warm=100)   # (steady state: the clock settles at the power cap)
    bytes5 = 604_110_848
    flops5 = 17_179_869_184
    out["gmul_c5a"] = {"ms_per_launch": round(ms5, 4), "tflops": round(flops5 / ms5 / 1e9, 2),
                       "frac_mfma": round(flops5 / ms5 / 1e9 / PEAK_MFMA_F32_TF, 4),
                       "gbps": round(bytes5 / ms5 / 1e6, 1),
                       "frac_hbm": round(bytes5 / ms5 / 1e6 / PEAK_HBM_GBS, 4),
                       "traffic": pmc_traffic("gmul_c5a"),
                       "kernel": "gemm_skinnyk3_kernel<8,0,1,64,true> (short-K streaming GEMM, csrc/gemm_skinnyk.hip)",
                       "bound": "near the ridge: t_mfma 109 us vs t_hbm 76 us at spec peaks"}
    ps = sample_power_state(T, lambda: T.gmul(2, 1, 1, a, b))
    if "sclk_mhz" in ps:
        ps["mfma_bound_us_at_this_sclk"] = round(109.2 * 2400.0 / ps["sclk_mhz"], 1)
        ps["frac_of_that_bound"] = round(ps["mfma_bound_us_at_this_sclk"] / (ps["ms_per_launch_sustained"] * 1e3), 4)
    out["gmul_c5a"]["power_state"] = ps
    c = T.gmul(2, 1, 1, a, b)
    # ---- config 5 as BASELINE states it: the contraction + mapped logistic.  Recorded in a fusion scope the
    # map is applied in the GEMM's epilogue: C is stored once, the 1.07 GB round trip of 5b disappears ----
    e = T.expr(logistic_closure, 1, key="bench_logistic")

    def c5_fused_keep():   # (forced inside the scope: closing a scope demands nothing)
        with T.memo():
            r = T.force(T.liftT(e, [T.gmul(2, 1, 1, a, b)]))
        return r
    l0 = T.stats()["launches"]
    c5_fused_keep()
    fused_launches = T.stats()["launches"] - l0
    ms5f = time_launches(T, c5_fused_keep, 200, warm=100)
    out["gmul_map_c5_fused"] = {"ms_per_launch": round(ms5f, 4), "launches": fused_launches,
                                "tflops": round(flops5 / ms5f / 1e9, 2),
                                "frac_mfma": round(flops5 / ms5f / 1e9 / PEAK_MFMA_F32_TF, 4),
                                "gbps": round(bytes5 / ms5f / 1e6, 1),
                                "frac_hbm": round(bytes5 / ms5f / 1e6 / PEAK_HBM_GBS, 4),
                                "algorithmic_bytes": bytes5, "traffic": pmc_traffic("gmul_map_c5_fused"),
                                "kernel": "gemm_skinnyk3_kernel<8,1,1,64,false>",
                                "note": "liftT logistic (gmul ...) recorded in one scope: logistic in the GEMM epilogue"}
    out["gmul_map_c5_fused"]["power_state"] = sample_power_state(T, c5_fused_keep)
    del a, b
    # ---- config 5b: map logistic over the 512^3 result (8 B/element), as a launch of its own ----
    msm = time_launches(T, lambda: T.liftT(e, [c]), 100, warm=50)
    gbs = 8.0 * 512 ** 3 / msm / 1e6
    out["map_logistic_c5b"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS,
                               "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                               "traffic": pmc_traffic("map_logistic_512cubed"),
                               "kernel": "ew_stream_kernel<float,1,FLogistic> (1,073,741,824 B/launch)",
                               "ms_per_launch": round(msm, 4)}
    out["map_logistic_c5b"]["power_state"] = sample_power_state(T, lambda: T.liftT(e, [c]))
    del c
    # ---- between the latency-bound and the full-chip regime: the wave-split kernel (csrc/gemm_kwave.hip) ----
    mid = {}
    # (round 4: 640^3, 1024x1024x512, 512x2048x512, 384x4096x384 -- fewer tiles than CUs, two to six workgroups per tile;
    #  round 6: 768^3 and 1280^3 on the tile menu of csrc/gemm_kw16.hip -- 256 tiles of 48x48 / 80x80 --, 1088^3 and
    #  1152x2048x1152 as stream-K over 512 workgroups, 768x1024x1024 on 48x64 tiles)
    for m_, k_, n_ in ((640, 640, 640), (768, 768, 768), (1000, 1000, 1000), (1024, 1024, 1024), (1088, 1088, 1088), (1280, 1280, 1280),
                       (1536, 1536, 1536), (2048, 2048, 2048), (3072, 3072, 3072), (4096, 784, 256), (1024, 1024, 512), (512, 2048, 512),
                       (384, 4096, 384), (768, 1024, 1024), (1152, 2048, 1152)):
        am = T.genRand((m_, k_), "uniform", -1.0, 1.0, SEED + 31)
        bm = T.genRand((k_, n_), "uniform", -1.0, 1.0, SEED + 32)
        msm = time_steady(T, lambda: T.gmul(1, 1, 1, am, bm))
        mid["%dx%dx%d" % (m_, k_, n_)] = {"ms": round(msm, 4), "tflops": round(2.0 * m_ * k_ * n_ / msm / 1e9, 1),
                                          "frac_mfma": round(2.0 * m_ * k_ * n_ / msm / 1e9 / PEAK_MFMA_F32_TF, 3)}
        del am, bm
    out["gmul_mid_sizes"] = mid
    # ---- the reference's own network (784 -> 300 -> 100 -> 10, app/MNIST.hs) under a whole data set / a big batch: a tall batch
    #      through a narrow layer (a tile per wave / per workgroup of csrc/gemm_kwave.hip), the cotangent coming back (K = 100),
    #      a narrow layer's weight gradient (ten tiles under K = 8192 / 60000: stream-K over 256 workgroups) ----
    learn = {}
    for m_, k_, n_ in ((60000, 784, 300), (60000, 300, 100), (8192, 300, 100), (8192, 100, 300), (100, 8192, 300), (100, 60000, 300)):
        am = T.genRand((m_, k_), "uniform", -1.0, 1.0, SEED + 33)
        bm = T.genRand((k_, n_), "uniform", -1.0, 1.0, SEED + 34)
        msm = time_steady(T, lambda: T.gmul(1, 1, 1, am, bm))
        by = 4.0 * (m_ * k_ + k_ * n_ + m_ * n_)
        learn["%dx%dx%d" % (m_, k_, n_)] = {"ms": round(msm, 4), "tflops": round(2.0 * m_ * k_ * n_ / msm / 1e9, 1),
                                            "frac_mfma": round(2.0 * m_ * k_ * n_ / msm / 1e9 / PEAK_MFMA_F32_TF, 3),
                                            "gbps": round(by / msm / 1e6, 1), "frac_hbm": round(by / msm / 1e6 / PEAK_HBM_GBS, 3)}
        del am, bm
    out["gmul_learn_shapes"] = learn
    # ---- the per-sample forms at size: matVec / vecMat / outer / sumRows move every matrix element once (csrc/gemv.hip): HBM-bound ----
    hb = {}
    n_ = 16384
    Am = T.genRand((n_, n_), "uniform", -1.0, 1.0, SEED + 35)
    xv = T.genRand((n_,), "uniform", -1.0, 1.0, SEED + 36)
    for name, fn, by in (("matVec_16384x16384", lambda: T.matVec(Am, xv), 4.0 * n_ * n_), ("vecMat_16384x16384", lambda: T.vecMat(xv, Am), 4.0 * n_ * n_),
                         ("outerV_16384x16384", lambda: T.outerV(xv, xv), 4.0 * n_ * n_), ("sumRows_16384x16384", lambda: T.sumRows(Am), 4.0 * n_ * n_)):
        msm = time_steady(T, fn)
        hb[name] = {"ms": round(msm, 4), "gbps": round(by / msm / 1e6, 1), "frac_hbm": round(by / msm / 1e6 / PEAK_HBM_GBS, 3),
                    "algorithmic_bytes": int(by), "traffic": (pmc_traffic("hbm_bound_forms") or {}).get(name) if isinstance(pmc_traffic("hbm_bound_forms"), dict) else None}
    del Am, xv
    out["hbm_bound_forms"] = hb
    # ---- fp64 instance (the reference's apps run `HMat Double`) ----
    from tensor_ops_amd.hipt import HipT
    T64 = HipT(0, dtype=np.float64)
    a = T64.genRand((n, n), "uniform", -1.0, 1.0, SEED + 15)
    b = T64.genRand((n, n), "uniform", -1.0, 1.0, SEED + 16)
    ms64 = time_launches(T64, lambda: T64.gmul(1, 1, 1, a, b), 30, warm=30)   # (steady state, as for the fp32 kernel)
    del a, b
    mid64 = {}
    for m_, k_, n_ in ((1000, 1000, 1000), (1024, 1024, 1024), (4096, 784, 256), (2048, 2048, 2048)):
        am = T64.genRand((m_, k_), "uniform", -1.0, 1.0, SEED + 33)
        bm = T64.genRand((k_, n_), "uniform", -1.0, 1.0, SEED + 34)
        msm = time_steady(T64, lambda: T64.gmul(1, 1, 1, am, bm))
        mid64["%dx%dx%d" % (m_, k_, n_)] = {"ms": round(msm, 4), "tflops": round(2.0 * m_ * k_ * n_ / msm / 1e9, 1),
                                            "frac_mfma": round(2.0 * m_ * k_ * n_ / msm / 1e9 / PEAK_MFMA_F64_TF, 3)}
        del am, bm
    a5 = T64.genRand((512, 512, 64), "uniform", -1.0, 1.0, SEED + 35)
    b5 = T64.genRand((64, 512), "uniform", -1.0, 1.0, SEED + 36)
    ms5_64 = time_launches(T64, lambda: T64.gmul(2, 1, 1, a5, b5), 50, warm=20)
    c5_64 = {"ms_per_launch": round(ms5_64, 4), "tflops": round(17_179_869_184 / ms5_64 / 1e9, 2),
             "frac_mfma": round(17_179_869_184 / ms5_64 / 1e9 / PEAK_MFMA_F64_TF, 4),
             "gbps": round(2 * 604_110_848 / ms5_64 / 1e6, 1),
             "frac_hbm": round(2 * 604_110_848 / ms5_64 / 1e6 / PEAK_HBM_GBS, 4),
             "kernel": "gemm_skinnyk64_kernel<16,0,false,true> (csrc/gemm_skinnyk_f64.hip: B panel resident in LDS, wave streams of "
                       "16-row blocks, the previous block leaving under the MFMAs)",
             "bound": "t_mfma 218 us vs t_hbm 151 us at spec peaks"}
    e64 = T64.expr(logistic_closure, 1, key="bench_logistic64")

    def c5_fused64():
        with T64.memo():
            T64.force(T64.liftT(e64, [T64.gmul(2, 1, 1, a5, b5)]))
    l0 = T64.stats()["launches"]
    c5_fused64()
    nl64 = T64.stats()["launches"] - l0
    ms5f_64 = time_launches(T64, c5_fused64, 50, warm=20)
    c5_64["with_map_logistic_fused"] = {"ms_per_launch": round(ms5f_64, 4), "launches": nl64,
                                        "tflops": round(17_179_869_184 / ms5f_64 / 1e9, 2)}
    del a5, b5
    x = T64.genRand((512, 512, 256), "uniform", -4.0, 4.0, SEED + 17)
    msm64 = time_launches(T64, lambda: T64.liftT(e, [x]), 30, warm=10)
    gb64 = 16.0 * 512 * 512 * 256 / msm64 / 1e6
    # the config-3 step in the reference's own precision (ElemT = Double): same networks, fp64 kernels
    from tensor_ops_amd import tops
    tops.set_elem_dtype(np.float64)
    try:
        ws64, X64, Y64 = synth(0, 1024)
        net64 = tops.genNet([(T64.put(w), T64.put(b)) for w, b in ws64], "actMapLogistic", "actSoftmax")
        tr64 = tops.Trainer(net64, "crossEntropy", RATE, T64.put(X64, batched=True), T64.put(Y64, batched=True))

        def step64():
            tr64.step()
        ms_step64 = time_launches(T64, step64, 300, warm=30)
        step64_info = {"ms_per_step": round(ms_step64, 5), "steps_per_s": round(1e3 / ms_step64, 1),
                       "library_fusion": tr64.fused, "kernel_launches": tr64.step_launches}
        del tr64, net64
    finally:
        tops.set_elem_dtype(np.float32)
    out["fp64"] = {"step_c3": step64_info, "gmul_4096": {"tflops": round(flops / ms64 / 1e9, 2), "peak": PEAK_MFMA_F64_TF,
                                 "frac": round(flops / ms64 / 1e9 / PEAK_MFMA_F64_TF, 4),
                                 "kernel": "gemm_f64_w4_kernel<0,0,4,2> (v_mfma_f64_16x16x4_f64, 8 waves of 64x64 on a pinned schedule)"},
                   "gmul_mid_sizes": mid64, "gmul_c5a": c5_64,
                   "map_logistic": {"gbps": round(gb64, 1), "frac_hbm": round(gb64 / PEAK_HBM_GBS, 4),
                                    "elements": 512 * 512 * 256, "bytes_per_element": 16}}
    return out


def cpu_baseline(ws, X, Y, seconds):
    """Single-thread port: per-sample BLAS-2 sequence of the reference incl. its forward
    recomputation (Types.hs:155), gradients summed at fixed params, one SGD update."""
    from oracle import hmat
    hmat.lib()
    W1, b1, W2, b2 = ws[0][0], ws[0][1], ws[1][0], ws[1][1]
    t = time.perf_counter()
    hmat.batched_grads(X[:32], Y[:32], W1, b1, W2, b2, True)
    per = (time.perf_counter() - t) / 32
    n = int(max(64, min(len(X) * 64, seconds / per)))
    reps, rem = divmod(n, len(X))
    t = time.perf_counter()
    done = 0
    for _ in range(reps):
        g, _l = hmat.batched_grads(X, Y, W1, b1, W2, b2, True)
        done += len(X)
    if rem:
        hmat.batched_grads(X[:rem], Y[:rem], W1, b1, W2, b2, True)
        done += rem
    dt = time.perf_counter() - t
    sps = done / dt
    # the same text compiled in single precision (oracle/hmat_path.c, -DHMAT_F32): the like-for-like figure beside the fp32 GPU step
    f32_build = None
    try:
        hmat.batched_grads_f32(X[:32], Y[:32], W1, b1, W2, b2, True)
        t32 = time.perf_counter()
        done32 = 0
        while done32 < len(X) or time.perf_counter() - t32 < seconds / 2.0:
            hmat.batched_grads_f32(X, Y, W1, b1, W2, b2, True)
            done32 += len(X)
        dt32 = time.perf_counter() - t32
        f32_build = {"value": round(done32 / dt32 / len(X), 4), "unit": "steps/s", "cores": 1, "dtype": "f32",
                     "samples_per_s": round(done32 / dt32, 1),
                     "sample": "%d samples (%.1f s), liboracle_hmat_f32.so: the same C text with float for double" % (done32, dt32)}
    except Exception as e:  # noqa: BLE001
        f32_build = "unavailable: %s" % e
    # host context: core count, CPU model, and what the host's own BLAS (numpy's bundled
    # OpenBLAS, all threads) does on the config-2 contraction at 2048^3 -- reported, not a target
    host = {"nproc": os.cpu_count()}
    try:
        host["cpu_model"] = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo")
                             if l.startswith("model name")][0]
    except Exception:
        host["cpu_model"] = "unknown"
    try:
        n2 = 2048
        a = np.random.default_rng(SEED).uniform(-1, 1, (n2, n2)).astype(np.float32)
        a @ a
        t = time.perf_counter()
        reps2 = 0
        while time.perf_counter() - t < 1.0:
            a @ a
            reps2 += 1
        host["numpy_sgemm_2048_gflops"] = round(2.0 * n2 ** 3 * reps2 / (time.perf_counter() - t) / 1e9, 1)
    except Exception:
        pass
    # the config-2 and config-5a contractions through the host's BLAS (numpy's bundled
    # OpenBLAS: a stand-in for hmatrix's sgemm, NOT GHC-compiled tensor-ops), one thread and all threads
    try:
        from threadpoolctl import threadpool_limits
        r2 = np.random.default_rng(SEED + 11)
        a2 = r2.uniform(-1, 1, (4096, 4096)).astype(np.float32)
        b2 = r2.uniform(-1, 1, (4096, 4096)).astype(np.float32)
        a5 = r2.uniform(-1, 1, (512 * 512, 64)).astype(np.float32)
        b5 = r2.uniform(-1, 1, (64, 512)).astype(np.float32)

        def rate(fn, flop, budget):
            fn()
            t = time.perf_counter()
            k = 0
            while k < 1 or time.perf_counter() - t < budget:
                fn()
                k += 1
            return round(flop * k / (time.perf_counter() - t) / 1e9, 1)
        blas = {}
        for label, lim in (("1_thread", 1), ("all_threads", None)):
            with threadpool_limits(limits=lim):
                blas[label] = {"gmul_c2_4096_gflops": rate(lambda: a2 @ b2, 2.0 * 4096 ** 3, 2.0),
                               "gmul_c5a_gflops": rate(lambda: a5 @ b5, 17_179_869_184.0, 1.0)}
        host["openblas_sgemm_stand_in"] = blas
    except Exception as e:  # noqa: BLE001
        host["openblas_sgemm_stand_in"] = "unavailable: %s" % e
    # BASELINE.md section 3, CPU-B: the same per-sample loop with the samples split over all host cores (pthreads, one
    # gradient buffer per thread, added up at the end) -- and CPU-D: `cmap logistic` over 2^27 fp32 as the scalar loop it
    # is, one thread and all threads.  Restatements like the rest of this block, bounded to a few seconds each.
    nproc = os.cpu_count() or 1
    try:
        # a persistent pool, several batches per call (round 5): a thread per batch of 1024 samples spent its time in
        # pthread_create and in zeroing / adding up one private 1.6 MB gradient sum per thread (256 threads: 416 MB each
        # way per batch for 4 samples of work per thread).  Every thread keeps >= 8 samples.
        legs = {}
        for th in sorted({min(nproc, max(1, len(X) // 32)), min(nproc, max(1, len(X) // 8))}):
            hmat.batched_grads_pool(X, Y, W1, b1, W2, b2, th, 1, True)
            t = time.perf_counter()
            hmat.batched_grads_pool(X, Y, W1, b1, W2, b2, th, 2, True)
            per_batch = (time.perf_counter() - t) / 2
            reps_b = int(max(2, min(512, seconds / 4.0 / max(per_batch, 1e-4))))
            t = time.perf_counter()
            hmat.batched_grads_pool(X, Y, W1, b1, W2, b2, th, reps_b, True)
            dtb = time.perf_counter() - t
            legs[th] = {"steps_per_s": round(reps_b / dtb, 3), "samples_per_s": round(reps_b * len(X) / dtb, 1), "threads": th,
                        "samples_per_thread_per_batch": round(len(X) / th, 1),
                        "speedup_over_1_thread": round(reps_b * len(X) / dtb / sps, 2),
                        "sample": "%d batches of %d samples in one call on a persistent pool of %d pthreads (%.1f s); "
                                  "oracle/hmat_path.c hmat_batched_grads_pool" % (reps_b, len(X), th, dtb)}
        best = max(legs.values(), key=lambda v: v["steps_per_s"])
        host["cpu_b_all_cores"] = dict(best, thread_counts_tried={str(k): v["steps_per_s"] for k, v in legs.items()},
                                       why_not_more="the reference's per-sample path ends in `ger`: every sample adds an outer product into the WHOLE "
                                                    "200,704-element gradient of layer 1, so every thread streams its private 1.6 MB sum (read + "
                                                    "write) once per sample -- %d threads x %d samples x 3.2 MB = %.1f GB of memory traffic a batch. "
                                                    "One thread keeps its sum in its own L2; all cores share the DRAM.  More threads than ~32 lose."
                                                    % (best["threads"], int(best["samples_per_thread_per_batch"]), best["threads"] * best["samples_per_thread_per_batch"] * 3.2e-3))
    except Exception as e:  # noqa: BLE001
        host["cpu_b_all_cores"] = "unavailable: %s" % e
    try:
        n_map = 1 << 27
        xm = np.random.default_rng(SEED + 5).uniform(-4, 4, n_map).astype(np.float32)
        legs = {}
        for label, th, n_el in (("1_thread", 1, n_map // 8), ("all_threads", nproc, n_map)):   # (1 thread: an eighth, ~1 s)
            hmat.map_logistic_f32(xm[:1 << 20], th)
            t = time.perf_counter()
            hmat.map_logistic_f32(xm[:n_el], th)
            dtm = time.perf_counter() - t
            legs[label] = {"gelem_per_s": round(n_el / dtm / 1e9, 3), "gbytes_per_s": round(8.0 * n_el / dtm / 1e9, 2),
                           "elements": n_el, "threads": th}
        host["cpu_d_map_logistic_f32"] = legs
        del xm
    except Exception as e:  # noqa: BLE001
        host["cpu_d_map_logistic_f32"] = "unavailable: %s" % e
    return {"value": round(sps / len(X), 4), "unit": "steps/s", "cores": 1, "kind": "port", "dtype": "f64",
            "f32_build": f32_build,
            "samples_per_s": round(sps, 1), "host": host,
            "sample": "%d samples of the same batch (%.1f s), oracle/hmat_path.c: per-sample gemv/ger/"
                      "axpy/liftB sequence in fp64 with the reference's 3x layer-1 forward recompute; "
                      "CPU restatement of the hmatrix path, not GHC-compiled tensor-ops" % (done, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1024, help="rows per GPU (BASELINE config 3)")
    ap.add_argument("--no-aux", action="store_true", help="skip gmul / map / cpu_baseline legs")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--regions", type=int, default=5, help="timed regions of --steps steps each (value = median)")
    ap.add_argument("--collective", choices=["auto", "torch", "direct", "p2p"], default="auto",
                    help="all-reduce transport.  auto (default): both C-ABI transports are set up, probed and timed, the "
                         "one that measured faster here (slowest rank's latency) carries the step.  "
                         "torch.distributed (nccl = RCCL) or the library's own C-ABI "
                         "collective (to_comm_*, RCCL loaded by the library; torch only carries the 128-byte id), or "
                         "p2p: the one-shot peer-to-peer all-reduce over hipIpc-mapped buffers with the SGD update in "
                         "the same launch (to_p2p_*)")
    ap.add_argument("--two-call", action="store_true",
                    help="single GPU: run the step as grad() + apply() (what a data-parallel rank runs around "
                         "its all-reduce) instead of Trainer.step() with the update fused into the gradient launches")
    ap.add_argument("--same-gpu", action="store_true",
                    help="every rank uses GPU 0 (a one-GPU box): the N>1 code path end to end -- sharding, set-up, the "
                         "peer-to-peer exchange over hipIpc, timing, the JSON line -- with the ranks time-sharing the "
                         "device.  RCCL refuses two ranks on one device; the step's all-reduce is the p2p exchange")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and all-reduce even at world size 1 (self-test)")
    args = ap.parse_args()

    # stdout carries ONE JSON line.  gloo and RCCL print banners on the C-level stdout (RCCL's arrive when its
    # buffers flush at exit), so file descriptor 1 is pointed at stderr for the whole run and the line is written
    # to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N>1 through torch.distributed.run"
                         % (args.gpus, world))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # before the HIP runtime starts (dmabuf IPC for RCCL)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP backend has no CPU fallback")
    if args.same_gpu:
        local_dev = 0
        os.environ.setdefault("TOPS_P2P_TIMEOUT_S", "30")   # ranks time-share one device: a peer may be scheduled late
    else:
        local_dev = local_rank
    torch.cuda.set_device(local_dev)
    dist = None
    if world > 1 or args.force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.collective in ("auto", "direct", "p2p"):
            dist.init_process_group(backend="gloo")   # bootstrap only: carries the RCCL unique id / IPC handles
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_dev))

    if local_rank == 0:   # build artefacts are git-ignored: (re)compile when missing or stale, once per node
        import __graft_entry__
        __graft_entry__.build()
    if dist is not None:
        dist.barrier()
    from tensor_ops_amd import capi, tops
    from tensor_ops_amd.dist import DataParallel
    from tensor_ops_amd.hipt import HipT

    T = HipT(local_dev)
    T_DEVICE[0] = local_dev
    stream = torch.cuda.Stream()                       # torch owns the stream; ours = the same one
    capi.check(capi.lib().to_set_stream(C.c_void_p(stream.cuda_stream)))

    ws, X, Y = synth(rank, args.batch)
    with torch.cuda.stream(stream):
        net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
        dX, dY = T.put(X, batched=True), T.put(Y, batched=True)
        nflat = tops.Trainer.flat_size(net)
        flat_p = torch.zeros(nflat, dtype=torch.float32, device="cuda")   # torch owns the buffers
        flat_g = torch.zeros(nflat, dtype=torch.float32, device="cuda")   # the all-reduce works on
        stream.synchronize()
        # the reference's rate is per sample (online SGD, app/MNIST.hs:93,396); a batched step sums B * world gradients,
        # so it steps by rate / (B * world): the same expected step length, and the parameters stay finite for as long
        # as the measurement runs (checked after the timed regions)
        rate = RATE / (args.batch * world)
        # N > 1: the whole step -- local gradients, exchange, update -- is captured as ONE launch list below
        # (DataParallel.capture), so the trainer issues directly; at N = 1 the trainer replays its own capture
        whole_step_capture = dist is not None and world > 1 and args.collective != "torch" and not args.no_graph
        n1 = None
        if dist is not None and world > 1:
            # this rank's N = 1 rate, measured in this run (same protocol: regions of --steps steps, median), on a net of its
            # own: what `weak_scaling_efficiency` in the JSON line is relative to
            net1 = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
            tr1 = tops.Trainer(net1, "crossEntropy", rate, dX, dY, use_memo=True, use_graph=not args.no_graph)
            for _ in range(max(args.warmup, 5)):
                tr1.step()
            r1 = []
            for _ in range(max(1, args.regions)):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    tr1.step()
                torch.cuda.synchronize()
                r1.append(time.perf_counter() - t0)
            n1 = args.steps / sorted(r1)[len(r1) // 2]
            del tr1, net1
        tr = tops.Trainer(net, "crossEntropy", rate, dX, dY, use_memo=True,
                          use_graph=(not args.no_graph) and not whole_step_capture, ext_params=flat_p.data_ptr(),
                          ext_grads=flat_g.data_ptr())
        direct = p2p_params = torch_group = None
        collective_us = None
        if dist is not None and args.collective in ("auto", "direct", "p2p"):
            from tensor_ops_amd.dist import HipCollectives, setup_collectives
            got = setup_collectives(HipCollectives(T, one_device_per_rank=not args.same_gpu), dist, rank, world, flat_g,
                                    flat_p, nflat, args.collective)
            direct, p2p_params, torch_group = got["direct"], got["p2p_params"], got["torch_group"]
            args.collective, collective_us = got["collective"], got["collective_us"]
        dp = DataParallel(flat_g, tr.grad, tr.apply, world, force=args.force_dist, direct_handle=direct,
                          step_fn=None if args.two_call else tr.step, p2p_params=p2p_params, p2p_rate=rate,
                          group=torch_group)

        step_captured = False
        if whole_step_capture and dp.world > 1:
            dp.step()   # (once directly: pool and plan cache warm, the exchange's epoch started)
            try:
                step_captured = dp.capture()
            except Exception as e:  # noqa: BLE001 -- every rank must take the same way out
                sys.stderr.write("bench.py: the step could not be captured on rank %d: %r\n" % (rank, e))
            flag = torch.tensor([1 if step_captured else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if step_captured and not flag.item():
                dp.release()
            step_captured = bool(flag.item())
        for _ in range(args.warmup):
            dp.step()
        l0 = T.stats()["launches"]
        dp.step()
        launches = T.stats()["launches"] - l0   # kernels per step (the collective, if any, not counted)
        # Each region: barrier + synchronize, EXACTLY --steps steps, synchronize + barrier; max over ranks.
        # A region of 20 steps is ~0.6 ms: one region swings by 10 % from run to run, the median of five does not.
        # Round 5: nothing but the steps sits between the two synchronisations.  The two device-timer events bench.py used
        # to record INSIDE the region (for step.device_ms_per_step) cost 24 us per region -- 1.2 us a step at 20 steps
        # (tools/region_probe.py: 536 vs 513 us for 20 steps) -- and torch.cuda.synchronize(), a device-wide wait, another
        # ~40 us when it is the call that has to notice the stream going idle; the library's own stream wait (to_sync) goes
        # first, so the device-wide one that brackets the region finds the device idle.  The device time of a region is
        # measured on an untimed twin region right after each timed one.
        regions, dev_regions = [], []
        for _ in range(max(1, args.regions)):
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                dp.step()
            T.sync()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            el = time.perf_counter() - t0
            T.timer_start()                      # (the untimed twin: device time of --steps steps, HIP events on the launch stream)
            for _ in range(args.steps):
                dp.step()
            dev_regions.append(T.timer_stop())
            if dist is not None:
                tmax = torch.tensor([el], dtype=torch.float64,
                                    device="cpu" if dist.get_backend() == "gloo" else "cuda")
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                el = float(tmax.item())
            regions.append(el)
        # the same step in regions long enough for the fixed cost of a region's two ends to vanish (200 steps): a separately
        # named figure, never `value` -- `value` is what the driver's protocol (--steps K, regions of exactly K steps) gives
        steady = None
        if world == 1 and not args.no_aux:
            rs = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(200):
                    dp.step()
                torch.cuda.synchronize()
                rs.append(time.perf_counter() - t0)
            el200 = sorted(rs)[1]
            steady = {"steps_per_region": 200, "regions": 3, "ms_per_step": round(el200 / 200 * 1e3, 5),
                      "steps_per_s": round(200 / el200, 2),
                      "note": "not `value`: the fixed cost of a timed region's two ends (first launch onto an idle queue, the "
                              "host noticing the last one has finished) is ~40-60 us, 2-3 us per step of a 20-step region"}
            # ... and with a DIFFERENT batch every step: eight resident 1024-row batches (25.7 MB of X) taken
            # round robin by eight trainers that share the one parameter buffer, each replaying its own captured step -- the
            # same three launches bound to another X, Y.  (`value` replays one batch, as SURVEY 8(d) defines the step.)
            try:
                others = []
                for i in range(1, 8):
                    _w, Xi, Yi = synth(100 + i, args.batch)
                    others.append(tops.Trainer(net, "crossEntropy", rate, T.put(Xi, batched=True), T.put(Yi, batched=True), use_memo=True,
                                               use_graph=not args.no_graph, ext_params=flat_p.data_ptr(), ext_grads=flat_g.data_ptr()))
                ring = [tr.step] + [o.step for o in others]
                for f in ring * 3:
                    f()
                rs = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for k in range(200):
                        ring[k & 7]()
                    torch.cuda.synchronize()
                    rs.append(time.perf_counter() - t0)
                eld = sorted(rs)[1]
                steady["distinct_batches"] = {"batches": 8, "ms_per_step": round(eld / 200 * 1e3, 5), "steps_per_s": round(200 / eld, 2),
                                              "bytes_of_input_cycled": int(8 * args.batch * (SIZES[0] + SIZES[2]) * 4)}
                del others, ring
            except Exception as e:  # noqa: BLE001
                steady["distinct_batches"] = "unavailable: %r" % (e,)
        stream.synchronize()
        params_finite = bool(torch.isfinite(flat_p).all().item())
        if not params_finite:
            raise SystemExit("bench.py: the parameters are not finite after the timed region -- the measured steps did "
                             "not compute a training step's arithmetic on meaningful numbers")
        order = sorted(range(len(regions)), key=lambda k: regions[k])
        mid = order[len(order) // 2]
        elapsed, dev_ms = regions[mid], dev_regions[mid]

        # N > 1: the first line a multi-GPU box produces is also the first config-4 parity evidence.
        # After the timed regions the replicated parameters are reset to the initial ones, THREE data-parallel steps run
        # through the very step object that was timed (captured launch list, transport and all), and rank 0 compares its
        # parameters with a single-GPU replay of the same three steps on the full batch (world x rows) at 1e-5 --
        # the check of tests/test_gpu_multi.py:72-80; replicas must be bit-identical.
        c4 = None
        if dist is not None and world > 1:
            def flat_of(params):
                out, off = np.zeros(nflat, dtype=np.float32), 0
                for p in params:
                    out[off:off + p.size] = np.asarray(p, dtype=np.float32).ravel()
                    off += (p.size + 3) // 4 * 4
                return out
            init = flat_of([ws[0][0], ws[0][1], ws[1][0], ws[1][1]])
            flat_p.copy_(torch.from_numpy(init).to(flat_p.device))
            stream.synchronize()
            dist.barrier()
            for _ in range(3):
                dp.step()
            stream.synchronize()
            mine = flat_p.detach().cpu()
            gathered = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine)
            identical = all(bool(torch.equal(g.view(torch.int32), gathered[0].view(torch.int32))) for g in gathered)
            v = C.c_int(0)
            capi.check(capi.lib().to_comm_world(C.byref(v)))   # ncclCommCount of the library's communicator, 0 if none was made
            nranks = v.value or None
            if rank == 0:
                shards = [synth(r, args.batch) for r in range(world)]
                Xf, Yf = np.concatenate([sh[1] for sh in shards]), np.concatenate([sh[2] for sh in shards])
                netf = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
                trf = tops.Trainer(netf, "crossEntropy", rate, T.put(Xf, batched=True), T.put(Yf, batched=True), use_memo=True, use_graph=False)
                for _ in range(3):
                    trf.grad()
                    trf.apply()
                want = flat_of([p.numpy() for p in trf.net.params])
                got = gathered[0].numpy()
                errs, off = [], 0
                for w in (ws[0][0], ws[0][1], ws[1][0], ws[1][1]):
                    a, b = got[off:off + w.size].astype(np.float64), want[off:off + w.size].astype(np.float64)
                    errs.append(float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)))
                    off += (w.size + 3) // 4 * 4
                c4 = {"rel_err": max(errs), "rel_err_by_tensor": [float("%.3g" % e) for e in errs], "steps": 3, "tolerance": 1e-5,
                      "ok": bool(max(errs) < 1e-5 and identical), "replicas_bit_identical": identical,
                      "reference": "single-GPU replay of the same 3 steps on the full batch of %d rows (this process, same library)" % (args.batch * world),
                      "rccl_nranks": nranks}
                if not c4["ok"]:
                    sys.stderr.write("bench.py: CONFIG-4 PARITY FAILED: %r\n" % (c4,))
                del trf, netf
            dist.barrier()

        # what RCCL itself says the library's communicator spans (ncclCommCount; None: no communicator was made, e.g. ranks
        # sharing one GPU) -- asked for whenever a process group exists, a forced world of 1 included
        rccl_nranks = None
        if dist is not None:
            v = C.c_int(0)
            capi.check(capi.lib().to_comm_world(C.byref(v)))
            rccl_nranks = v.value or None
        result = None
        if rank == 0:
            steps_total = args.steps * world
            result = {
                "metric": "gradTOp steps/sec (ffLayer MNIST 784->256->10) + gmul TFLOP/s vs roofline",
                "value": round(steps_total / elapsed, 2),
                "unit": "steps/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 5),
                "timing": {"regions": len(regions), "steps_per_region": args.steps, "value_from": "median region",
                           "device_state": "coming out of idle: with --warmup 5 the regions are the first ~2 ms of work after "
                                           "seconds of set-up on the host.  tools/ramp_probe.py (profiles/r05_ramp_probe.txt): the same "
                                           "20-step region costs 28.6 us a step after 1 s of idling, 27.2 after 5 ms, 26.3 right "
                                           "behind 4,000 steps; from idle the rate settles after ~400 steps (10 ms) -- `steady_state` "
                                           "below is the figure for a device that stays busy",
                           "ms_per_step_by_region": [round(r / args.steps * 1e3, 5) for r in regions]},
                "steady_state": steady,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": "C3/C4 batched gradTOp + SGD step, ffLayer 784->256->10 "
                                       "(actMap logistic, softmax, crossEntropy), %d rows per GPU; "
                                       "a step = one 1024-row shard pass" % args.batch,
                           "global_batch": args.batch * world, "rows_per_gpu": args.batch,
                           "parallelism": "dp%d" % world,
                           "collective": ("1 all-reduce(sum) of %d fp32 per step via %s" % (
                               nflat, {"direct": "to_comm_allreduce_sum (RCCL, C ABI)",
                                       "p2p": "to_p2p_allreduce_sgd (one-shot peer-to-peer over hipIpc, update fused)",
                                       "torch": "torch.distributed nccl (RCCL)"}[args.collective]))
                           if (world > 1 or args.force_dist) else "none",
                           "collective_us_alone": collective_us},
                "samples_per_s": round(steps_total * args.batch / elapsed, 1),
                "step": {"kernel_launches": tr.step_launches if (dp.world == 1 and not args.two_call) else launches,
                         "path": "Network{op, params} (no activation tags) -> the reference's gradTOp, unchanged: seed "
                                 "through generateA, sumRows' gradient through the general mapRows, per-sample "
                                 "cotangents summed over the batch where gradTOp returns -> class-method stream -> "
                                 "deferred + fused by the library (csrc/lazy.cpp); pinned to the oracle's stream by "
                                 "tests/test_gpu_call_trace.py",
                         "sgd_update": "p - r*g recorded like any other method call; lands in the epilogue of the "
                                       "weight-gradient launches" if (dp.world == 1 and not args.two_call)
                         else "separate launch after the all-reduce",
                         "graph_replay": not args.no_graph, "library_fusion": tr.fused,
                         "rate": rate, "params_finite_after_timed_region": params_finite,
                         "device_ms_per_step": round(dev_ms / args.steps, 5),
                         "algorithmic_flops": STEP_FLOPS * args.batch // 1024,
                         "tflops": round(STEP_FLOPS * args.batch / 1024 / (dev_ms / args.steps) / 1e9, 3),
                         "frac_mfma": round(STEP_FLOPS * args.batch / 1024 / (dev_ms / args.steps) / 1e9
                                            / PEAK_MFMA_F32_TF, 4),
                         "note": "latency-bound: ~5 MB working set, 0.84 GFLOP = 5.3 us at MFMA peak"},
            }
            if world == 1 and not args.no_aux:
                result.update(aux_benchmarks(T))
                result["cpu_baseline"] = cpu_baseline(ws, X, Y, args.cpu_seconds)
                cb = result["cpu_baseline"]
                # the headline GPU value is fp32, CPU-A is fp64 (the reference apps' element type): both ratios, labelled
                cb["gpu_over_cpu"] = round(result["value"] / max(cb["value"], 1e-12), 1)
                cb["gpu_over_cpu_note"] = "fp32 GPU step over the fp64 CPU port (mixed precision); like for like below"
                same = {}
                s64 = ((result.get("fp64") or {}).get("step_c3") or {}).get("steps_per_s")
                if s64:
                    same["f64"] = {"gpu_steps_per_s": s64, "cpu_steps_per_s": cb["value"], "ratio": round(s64 / max(cb["value"], 1e-12), 1)}
                if isinstance(cb.get("f32_build"), dict):
                    same["f32"] = {"gpu_steps_per_s": result["value"], "cpu_steps_per_s": cb["f32_build"]["value"],
                                   "ratio": round(result["value"] / max(cb["f32_build"]["value"], 1e-12), 1)}
                result["gpu_over_cpu_same_precision"] = same
            else:
                result["cpu_baseline"] = None
                result["roofline"] = exchange_roofline(world, nflat * 4, args.collective, collective_us)
                result["c4_parity"] = c4
                result["c4_parity_rel_err"] = c4["rel_err"] if c4 else None
                result["rccl_nranks"] = rccl_nranks
                result["step"]["whole_step_captured_as_one_launch_list"] = step_captured
                if n1 is not None:
                    result["n1_steps_per_s_this_run"] = round(n1, 2)
                    result["weak_scaling_efficiency"] = round(result["value"] / (world * n1), 4)
        del tr, dp
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if result is not None:
        os.write(real_stdout, (json.dumps(result) + "\n").encode())


if __name__ == "__main__":
    main()
