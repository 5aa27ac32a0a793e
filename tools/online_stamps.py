"""Per-phase time of ONE sample of the persistent online-SGD kernel (csrc/online_sgd.hip, development build: TOPS_ONLINE_STAMPS):
workgroup 0, sample 64, wall_clock64 (10 ns).  Runs the stack in a subprocess with the stamps on and labels what it printed.
  D=$PWD/tensor-ops_amd/build_ab; TOPS_HIP_LIB=$D/libtensorops_hip.so LD_LIBRARY_PATH=$D python tools/online_stamps.py [784 300 100 10]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = [int(v) for v in sys.argv[1:]] or [784, 300, 100, 10]
L = len(sizes) - 1
code = r'''
import sys, time, ctypes as C, numpy as np
sys.path.insert(0, %r)
from tensor_ops_amd import capi
from tensor_ops_amd.hipt import HipT
T = HipT(0); Lb = capi.lib()
sizes = %r; n = 20000
rng = np.random.default_rng(7)
ws = [((0.5 * rng.standard_normal((o, i)) / np.sqrt(i)).astype(np.float32), (0.5 * rng.standard_normal(o)).astype(np.float32)) for i, o in zip(sizes[:-1], sizes[1:])]
X = rng.uniform(0, 1, (n, sizes[0])).astype(np.float32)
Y = np.zeros((n, sizes[-1]), np.float32); Y[np.arange(n), rng.integers(0, sizes[-1], n)] = 1
dw = [T.put(w) for w, _ in ws]; db = [T.put(b) for _, b in ws]
dX, dY = T.put(X, batched=True), T.put(Y, batched=True)
wa = (capi.c_tensor * len(ws))(*[t.h for t in dw]); ba = (capi.c_tensor * len(ws))(*[t.h for t in db])
for rep in range(3):
    T.sync(); t0 = time.perf_counter()
    capi.check(Lb.to_fflayer_stack_online_sgd(len(ws), wa, ba, 0, 2, 1, dX.h, dY.h, n, None, 0.02))
    dt = time.perf_counter() - t0
print("us_per_sample %%.3f" %% (dt / n * 1e6))
''' % (ROOT, sizes)
r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TOPS_ONLINE_STAMPS="1"), capture_output=True, text=True, timeout=600)
m = re.search(r"\[online\] phase stamps[^:]*:(.*)", r.stderr)
us = re.search(r"us_per_sample ([0-9.]+)", r.stdout)
if not m or not us:
    sys.exit("no stamps (a product build?):\n" + r.stderr[-2000:] + r.stdout[-500:])
st = [0.0] + [float(v) for v in m.group(1).split()]
names = ["x (registers) -> LDS, barrier", "layer 1: this workgroup's rows of W1 (LDS) . x, one wave a row, logistic",
         "partial W2[:, R_g] h1[R_g] -> the exchange (tagged 64-bit stores), poll the other workgroups' partials, next row requested"]
names += ["z2 = b2 + the G partials in workgroup order, logistic"]
names += ["replicated layer %d forward" % (l + 1) for l in range(2, L)]
names += ["loss head on z_L (wave 0: softmax, crossEntropy cotangent)"]
names += ["back through replicated layer %d (old weights)" % (l + 1) for l in range(L - 1, 1, -1)]
names += ["dz1 on this workgroup's rows (W2 columns in LDS)", "p <- p - rate g on everything this workgroup holds (W1 rows, W2 columns, replicated layers)"]
print("stack %s, workgroup 0, sample 64 of 20000; measured %s us per sample end to end (third run)" % ("x".join(map(str, sizes)), us.group(1)))
print("%8s %8s  phase" % ("ends us", "takes us"))
for i in range(1, len(st)):
    print("%8.2f %8.2f  %s" % (st[i], st[i] - st[i - 1], names[i - 1] if i - 1 < len(names) else "?"))
print("(the stamps themselves cost a few hundred ns a sample; the sum is the stamped sample's length, the end-to-end figure the average)")
