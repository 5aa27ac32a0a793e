"""The persistent online-SGD kernel (csrc/online_sgd.hip) against a numpy per-sample loop, and its time per sample."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tensor_ops_amd import capi
from tensor_ops_amd.hipt import HipT

T = HipT(0)
L = capi.lib()


def ref(ws, X, Y, order, rate, head):
    ws = [(w.astype(np.float64).copy(), b.astype(np.float64).copy()) for w, b in ws]
    for s in order:
        a = [X[s].astype(np.float64)]
        for l, (w, b) in enumerate(ws):
            z = w @ a[-1] + b
            a.append(z if l == len(ws) - 1 else 1 / (1 + np.exp(-z)))
        z, y = a[-1], Y[s].astype(np.float64)
        if head == 1:
            e = np.exp(z - z.max())
            dz = e / e.sum() * y.sum() - y
        else:
            sg = 1 / (1 + np.exp(-z))
            dz = -2 * (y - sg) * sg * (1 - sg)
        for l in range(len(ws) - 1, -1, -1):
            w, b = ws[l]
            dprev = (w.T @ dz) * a[l] * (1 - a[l]) if l > 0 else None
            ws[l] = (w - rate * np.outer(dz, a[l]), b - rate * dz)
            dz = dprev
    return ws


def run(sizes, n, rate, head, N=None, check=True):
    rng = np.random.default_rng(7)
    N = N or n
    ws = [(0.5 * rng.standard_normal((o, i)).astype(np.float32), 0.5 * rng.standard_normal(o).astype(np.float32))
          for i, o in zip(sizes[:-1], sizes[1:])]
    X = rng.uniform(0, 1, (N, sizes[0])).astype(np.float32)
    Y = np.zeros((N, sizes[-1]), np.float32)
    Y[np.arange(N), rng.integers(0, sizes[-1], N)] = 1
    order = rng.permutation(N)[:n]
    dw = [T.put(w) for w, _ in ws]
    db = [T.put(b) for _, b in ws]
    dX, dY = T.put(X, batched=True), T.put(Y, batched=True)
    wa = (capi.c_tensor * len(ws))(*[t.h for t in dw])
    ba = (capi.c_tensor * len(ws))(*[t.h for t in db])
    idx = (C.c_int64 * n)(*[int(v) for v in order])
    out_act, loss = (2, 1) if head == 1 else (0, 0)
    T.sync()
    t0 = time.perf_counter()
    capi.check(L.to_fflayer_stack_online_sgd(len(ws), wa, ba, 0, out_act, loss, dX.h, dY.h, n, idx, rate))
    dt = time.perf_counter() - t0
    msg = "%-22s n=%-6d %.2f us/sample" % ("x".join(map(str, sizes)), n, dt / n * 1e6)
    if check:
        want = ref(ws, X, Y, order, rate, head)
        errs = []
        for (w, b), gw, gb in zip(want, dw, db):
            errs.append(np.linalg.norm(gw.numpy() - w) / np.linalg.norm(w))
            errs.append(np.linalg.norm(gb.numpy() - b) / np.linalg.norm(b))
        msg += "  max rel err %.2e" % max(errs)
    print(msg)


run([20, 12, 5], 48, 0.1, 1, N=64)
run([2, 16, 1], 200, 1.0, 2)
run([784, 256, 10], 500, 0.02, 1)
run([784, 300, 100, 10], 500, 0.02, 1)
run([30, 20, 16, 12, 6], 300, 0.05, 2)
run([784, 300, 100, 10], 20000, 0.02, 1, check=False)
run([784, 300, 100, 10], 60000, 0.02, 1, check=False)
