"""Same three launches (to_fflayer_stack_sgd on config 3), issued directly vs replayed as a HIP graph."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tensor_ops_amd import capi
from tensor_ops_amd.hipt import HipT
T = HipT(0); L = capi.lib()
ws, X, Y = bench.synth(0, 1024)
W = [T.put(ws[0][0]), T.put(ws[1][0])]; Bv = [T.put(ws[0][1]), T.put(ws[1][1])]
dX, dY = T.put(X, batched=True), T.put(Y, batched=True)
w = (capi.c_tensor * 2)(W[0].h, W[1].h); b = (capi.c_tensor * 2)(Bv[0].h, Bv[1].h)
def step():
    capi.check(L.to_fflayer_stack_sgd(2, w, b, 0, 2, 1, dX.h, dY.h, C.c_double(1e-7), None))
def timeit(fn, n=2000, warm=200):
    for _ in range(warm): fn()
    T.sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    T.sync(); return (time.perf_counter() - t0) / n * 1e6
print("direct: %.2f us/step" % timeit(step))
capi.check(L.to_graph_begin()); step(); g = capi.c_graph(); capi.check(L.to_graph_end(C.byref(g)))
print("graph : %.2f us/step" % timeit(lambda: capi.check(L.to_graph_launch(g))))
print("direct: %.2f us/step" % timeit(step))
