"""Per-workgroup timeline of the last GEMM launch (TOPS_GEMM_DBG): start / after prologue / loop end / end,
in units of the 100 MHz wall clock (10 ns)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.environ.setdefault("TOPS_GEMM_DBG", "/tmp/gemm_dbg.txt")
from tensor_ops_amd.hipt import HipT
m, k, n = (int(a) for a in sys.argv[1:4])
T = HipT(0)
a = T.genRand((m, k), "uniform", -1, 1, 1); b = T.genRand((k, n), "uniform", -1, 1, 2)
for _ in range(int(os.environ.get("REPS", "4"))):
    T.gmul(1, 1, 1, a, b)
T.sync()
d = np.loadtxt(path, dtype=np.int64)
t0 = d[:, 1].min()
st, pro, loop, end = (d[:, i] - t0 for i in (1, 2, 3, 4))
us = 0.01
print("blocks", len(d), "kernel span %.1f us" % ((end.max()) * us))
print("start   : min %.1f max %.1f us" % (st.min() * us, st.max() * us))
print("prologue: mean %.1f max %.1f us" % ((pro - st).mean() * us, (pro - st).max() * us))
print("k-loop  : mean %.1f min %.1f max %.1f us" % ((loop - pro).mean() * us, (loop - pro).min() * us, (loop - pro).max() * us))
print("epilogue: mean %.1f max %.1f us" % ((end - loop).mean() * us, (end - loop).max() * us))
print("end     : min %.1f mean %.1f max %.1f us" % (end.min() * us, end.mean() * us, end.max() * us))
q1, q2 = d[:, 9] - t0, d[:, 10] - t0
print("loop quarters (mean us): 0-25%%: %.1f  25-50%%: %.1f  50-100%%: %.1f (per quarter %.1f)" % (
    (q1 - pro).mean() * us, (q2 - q1).mean() * us, (loop - q2).mean() * us, (loop - q2).mean() * us / 2))
xcc = d[:, 5] & 0xf
for x in range(8):
    sel = xcc == x
    if sel.any():
        print(" xcc %d: n=%d loop mean %.1f max %.1f end max %.1f" % (x, sel.sum(), (loop - pro)[sel].mean() * us, (loop - pro)[sel].max() * us, end[sel].max() * us))
