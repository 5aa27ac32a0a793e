"""One GEMM shape in all four operand layouts (A k- or m-contiguous, B n- or k-contiguous): steady-state time of each.
   usage: layout_time.py M K N [M K N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tensor_ops_amd.hipt import HipT
T = HipT(0)
v = [int(x) for x in sys.argv[1:]]
for i in range(0, len(v), 3):
    m, k, n = v[i:i + 3]
    out = []
    for ta in (0, 1):
        for tb in (0, 1):
            a = T.transp(T.genRand((k, m), "uniform", -1, 1, 1)) if ta else T.genRand((m, k), "uniform", -1, 1, 1)
            b = T.transp(T.genRand((n, k), "uniform", -1, 1, 2)) if tb else T.genRand((k, n), "uniform", -1, 1, 2)

            def run(iters, warm):
                for _ in range(warm): T.gmul(1, 1, 1, a, b)
                T.sync(); T.timer_start()
                for _ in range(iters): T.gmul(1, 1, 1, a, b)
                return T.timer_stop() / iters
            est = max(run(20, 5), 1e-3)
            ms = run(max(20, int(40.0 / est)), max(20, int(60.0 / est)))
            out.append("A%s B%s %.4f ms %6.1f TF" % ("mk"[ta == 0] if False else ("k" if not ta else "m"), "n" if not tb else "k", ms, 2.0 * m * k * n / ms / 1e9))
    print("%d x %d x %d: " % (m, k, n) + " | ".join(out), flush=True)
