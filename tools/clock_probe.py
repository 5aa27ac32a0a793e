"""sclk / socket power (rocm-smi, sampled every ~50 ms) while one kernel runs back to back: config 2's GEMM, config 5a
plain, config 5 fused, config 5b's map.  Explains the box-to-box spread of config 5 (it runs at the socket power cap)."""
import json
import os
import subprocess
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd.hipt import HipT, logistic_closure  # noqa: E402

T = HipT(0)


def sample_while(fn, seconds=3.0, batch=100):
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True,
                                     timeout=10).stdout
                c = json.loads(out)
                c = c.get("card0", c)
                sclk = [v for k, v in c.items() if "sclk clock speed" in k][0]
                pw = [v for k, v in c.items() if "Power" in k][0]
                samples.append((float(sclk.strip("()Mhz")), float(pw)))
            except Exception:  # noqa: BLE001
                pass
            time.sleep(0.05)
    for _ in range(30):
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
    s = samples[len(samples) // 3:] or samples or [(0.0, 0.0)]
    return sorted(ms)[len(ms) // 2], sum(x[0] for x in s) / len(s), sum(x[1] for x in s) / len(s)


a = T.genRand((4096, 4096), "uniform", -1, 1, 1)
b = T.genRand((4096, 4096), "uniform", -1, 1, 2)
ms, sclk, pw = sample_while(lambda: T.gmul(1, 1, 1, a, b), batch=20)
print("gmul 4096^3        %.4f ms  %6.1f TF   sclk %.0f MHz  %.0f W" % (ms, 137.439 / ms, sclk, pw))
del a, b
a = T.genRand((512, 512, 64), "uniform", -1, 1, 1)
b = T.genRand((64, 512), "uniform", -1, 1, 2)
ms, sclk, pw = sample_while(lambda: T.gmul(2, 1, 1, a, b))
print("gmul c5a           %.4f ms  %6.1f TF   sclk %.0f MHz  %.0f W" % (ms, 17.18 / ms, sclk, pw))
e = T.expr(logistic_closure, 1, key="c5b")


def fused():
    with T.memo():
        return T.force(T.liftT(e, [T.gmul(2, 1, 1, a, b)]))


ms, sclk, pw = sample_while(fused)
print("gmul+logistic c5   %.4f ms  %6.1f TF   sclk %.0f MHz  %.0f W" % (ms, 17.18 / ms, sclk, pw))
c = T.gmul(2, 1, 1, a, b)
ms, sclk, pw = sample_while(lambda: T.liftT(e, [c]))
print("map logistic c5b   %.4f ms  %6.0f GB/s sclk %.0f MHz  %.0f W" % (ms, 1073.74 / ms, sclk, pw))
for env in ("TOPS_SKINNYK_NT",):
    pass
