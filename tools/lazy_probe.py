"""Launch counts and step time of the tag-less Trainer on config 3 (lazy fusion on / off)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd import tops, capi
from tensor_ops_amd.hipt import HipT
import ctypes as C

T = HipT(0)
rng = np.random.default_rng(1)
i, h, o, B = 784, 256, 10, 1024
ws = [(0.5 * rng.standard_normal((h, i)), 0.5 * rng.standard_normal(h)),
      (0.5 * rng.standard_normal((o, h)), 0.5 * rng.standard_normal(o))]
X = rng.uniform(0, 1, size=(B, i)); Y = np.zeros((B, o)); Y[np.arange(B), rng.integers(0, o, size=B)] = 1.0
L = capi.lib()

def stats():
    a = [C.c_int64() for _ in range(4)]
    L.to_lazy_stats(*[C.byref(v) for v in a]); return [v.value for v in a]

def times():
    a = [C.c_int64() for _ in range(2)]
    L.to_lazy_time(*[C.byref(v) for v in a]); return [v.value for v in a]

for head, loss in (("actSoftmax", "crossEntropy"), ("actLogistic", "squaredError")):
  for fused in (True, False):
    for graph in (False, True):
        net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", head)
        tr = tops.Trainer(net, loss, 0.02, T.put(X, batched=True), T.put(Y, batched=True), use_graph=graph, use_fused=fused)
        s0 = stats()
        tr.step(); T.sync()
        s1 = stats()
        for _ in range(20): tr.step()
        T.sync(); t0 = time.perf_counter(); tm0 = times()
        n = 300
        for _ in range(n): tr.step()
        tm1 = times(); t_issue = (time.perf_counter() - t0) / n
        T.sync(); dt = (time.perf_counter() - t0) / n
        print(f"   host issue {t_issue*1e6:.1f} us/step, of which planning {(tm1[0]-tm0[0])/n/1e3:.1f} us, plan+launch {(tm1[1]-tm0[1])/n/1e3:.1f} us")
        print(f"{head:11s} fused={fused} graph={graph}: grad launches {tr.launches_per_step}, step launches {tr.step_launches}, "
              f"{dt*1e3:.4f} ms/step ; one step: recorded {s1[0]-s0[0]} fused-launches {s1[1]-s0[1]} elided {s1[2]-s0[2]} flushes {s1[3]-s0[3]}", flush=True)
