"""Which kernel runs `logistic (W x + b)` on 65536 one-sample rows (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd import hipt
T = hipt.HipT(0)
K, N = 64, 512
rng = np.random.default_rng(0)
x = T.put(rng.integers(-3, 4, (65536, K)).astype(np.float32), batched=True)
W = T.put(rng.integers(-3, 4, (N, K)).astype(np.float32))
b = T.put(rng.integers(-2, 3, N).astype(np.float32))
for _ in range(3):
    with T.memo():
        zb = T.sumT([T.matVec(W, x), b], (N,))
    with T.memo():
        hb = T.liftT(hipt.logistic_closure, [T.sumT([T.matVec(W, x), b], (N,))], key="p")
T.sync()
print("ok")
