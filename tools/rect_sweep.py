"""A/B of the mid-size tile shapes (TOPS_GEMM_W4_RECT = 0 square 128x128, 1 = 128x256, 2 = 256x128) x workgroup targets:
one process per setting (the switches are read once)."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from tensor_ops_amd.hipt import HipT
T = HipT(0)
shapes = [(768,768,768),(1024,1024,1024),(1536,1536,1536),(2048,2048,2048),(2304,2304,2304),(16384,256,4096),(1024,8192,1024),(2048,8192,2048)]
out = []
rng = np.random.default_rng(1)
for m,k,n in shapes:
    a = T.genRand((m,k),"uniform",-1,1,1); b = T.genRand((k,n),"uniform",-1,1,2)
    for _ in range(10): T.gmul(1,1,1,a,b)
    T.sync(); T.timer_start()
    for _ in range(30): T.gmul(1,1,1,a,b)
    ms = T.timer_stop()/30
    out.append("%%6.1f" %% (2.0*m*k*n/ms/1e9))
# exactness on integers, one ragged and one aligned shape
for m,k,n in [(1028,512,1540),(2048,256,2048)]:
    A = rng.integers(-3,4,(m,k)).astype(np.float32); B = rng.integers(-3,4,(k,n)).astype(np.float32)
    got = T.gmul(1,1,1,T.put(A),T.put(B)).numpy()
    out.append("ok" if np.array_equal(got, A@B) else "WRONG")
print(" ".join(out))
''' % ROOT
print("%-22s %s" % ("setting", "768 1024 1536 2048 2304 16384x256x4096 1024x8192x1024 2048x8192x2048 | exact"))
for rect in ("0", "1", "2"):
    for wg in ("1", "256", "512"):
        env = dict(os.environ, TOPS_GEMM_W4_RECT=rect, TOPS_GEMM_W4_128=wg)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print("rect=%s wg_target=%-4s   %s" % (rect, wg, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]))
