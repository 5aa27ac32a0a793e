"""Where the host time of an UN-replayed step goes: the mirror's closures + entry-point calls, the planner, the launches.
Two forms: Trainer.step() issued directly (in-place update through to_copy_into_many) and the pure step of the Haskell
shim's trainBatch (trainNetwork on a batch + force_many inside a scope, fresh parameter buffers every step)."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tensor_ops_amd import capi, tops  # noqa: E402
from tensor_ops_amd.hipt import HipT  # noqa: E402

T = HipT(0)
ws, X, Y = bench.synth(0, 1024)
dX, dY = T.put(X, batched=True), T.put(Y, batched=True)
net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")


def lazy_time():
    a, b = C.c_int64(), C.c_int64()
    capi.check(capi.lib().to_lazy_time(C.byref(a), C.byref(b)))
    return a.value, b.value


def api_time():
    a, b = C.c_int64(), C.c_int64()
    capi.check(capi.lib().to_api_time(C.byref(a), C.byref(b)))
    return a.value, b.value


def run(name, f, n=3000):
    for _ in range(100):
        f()
    T.sync()
    p0, f0 = lazy_time()
    a0, c0 = api_time()
    l0 = T.stats()["launches"]
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    t1 = time.perf_counter()
    T.sync()
    t2 = time.perf_counter()
    p1, f1 = lazy_time()
    a1, c1 = api_time()
    print("%-28s inside the library %.2f us/step in %.1f calls" % ("", (a1 - a0) / n / 1e3, (c1 - c0) / n))
    print("%-28s host %.2f us/step (drained %.2f), planning %.2f us, plan+launch %.2f us, %.1f launches/step"
          % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6, (p1 - p0) / n / 1e3, (f1 - f0) / n / 1e3,
             (T.stats()["launches"] - l0) / n))


tr = tops.Trainer(net, "crossEntropy", bench.RATE / 1024, dX, dY, use_graph=False)
run("Trainer.step() direct", tr.step)
trg = tops.Trainer(net, "crossEntropy", bench.RATE / 1024, dX, dY, use_graph=True)
run("Trainer.step() replayed", trg.step)

state = {"net": net}


def pure():
    with T.memo():
        new = tops.trainNetwork(state["net"], "crossEntropy", bench.RATE / 1024, dX, dY)
        T.force_many(new.params)
    state["net"] = new


run("pure trainBatch (shim form)", pure)
