"""Repro of the stress run's intermittent fp64 mismatch: (M, K, N) = (1346, 339, 258) and (1305, 341, 192), `W x + b` recorded
and forced, again and again on operands uploaded once; on a mismatch: which rows, is the operand on the device intact, does
an immediate second run of the same launch agree.  usage: repro_f64.py [iters] [tag]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd.hipt import HipT
DT = np.float32 if os.environ.get("REPRO_DTYPE") == "f32" else np.float64
T = HipT(0, dtype=DT) if DT is np.float64 else HipT(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tag = sys.argv[2] if len(sys.argv) > 2 else "?"
rng = np.random.default_rng(1234)
bad = 0
shapes = [(1346, 339, 258), (1305, 341, 192)]
if os.environ.get("REPRO_SHAPES"):
    shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["REPRO_SHAPES"].split(",")]
for (M, K, N) in shapes:
    W = rng.integers(-2, 3, (N, K)).astype(DT); X = rng.integers(-2, 3, (M, K)).astype(DT); b = rng.integers(-3, 4, N).astype(DT)
    want = (X.astype(np.float64) @ W.T.astype(np.float64) + b).astype(DT)
    dW, dX, db = T.put(W), T.put(X, batched=True), T.put(b)
    for it in range(iters):
        with T.memo():
            z = T.force(T.sumT([T.matVec(dW, dX), db], (N,)))
        got = z.numpy().reshape(M, N)
        if not np.array_equal(got, want):
            bad += 1
            rows = np.unique(np.nonzero(got != want)[0]); cols = np.unique(np.nonzero(got != want)[1])
            x_ok = np.array_equal(dX.numpy().reshape(M, K), X); w_ok = np.array_equal(dW.numpy(), W)
            got_again = z.numpy().reshape(M, N)     # the same device result downloaded again
            with T.memo():
                z2 = T.force(T.sumT([T.matVec(dW, dX), db], (N,)))
            again = np.array_equal(z2.numpy().reshape(M, N), want)
            print("[%s] MISMATCH %s it %d: rows %d..%d (%d rows) cols %d..%d (%d) | X intact %s W intact %s | same result on 2nd download %s | relaunch right %s | ints %s"
                  % (tag, (M, K, N), it, rows[0], rows[-1], len(rows), cols[0], cols[-1], len(cols), x_ok, w_ok,
                     np.array_equal(got_again, got), again, bool(np.all(got == np.round(got)))), flush=True)
            if bad > 12:
                break
        del z
print("[%s] done, mismatches %d" % (tag, bad))
