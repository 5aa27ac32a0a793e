"""Config-5a style (short K, many tiles) correctness under the persistent kernels: several shapes and layouts,
repeated, against numpy fp64 on sampled rows."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd.hipt import HipT
T = HipT(0)
rng = np.random.default_rng(5)
bad = 0
for (m, k, n) in ((8192, 64, 4096), (16384, 16, 2048), (4096, 48, 8192), (262144, 64, 512), (8192, 256, 8192), (6400, 32, 5120)):
    a = rng.uniform(-1, 1, (m, k)).astype(np.float32); b = rng.uniform(-1, 1, (k, n)).astype(np.float32)
    for ta in (0, 1):
        for tb in (0, 1):
            da = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
            db = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
            for rep in range(3):
                got = T.gmul(1, 1, 1, da, db).numpy()
                rows = np.concatenate([rng.integers(0, m, 24), [0, m - 1, 255, 256, m - 257]])
                ref = a[rows].astype(np.float64) @ b.astype(np.float64)
                err = np.abs(got[rows] - ref).max()
                # full-matrix checksum against a second run guards against rare races
                if rep == 0:
                    first = got
                elif not np.array_equal(first, got):
                    print("NONDETERMINISTIC", m, k, n, ta, tb); bad += 1
                if err > 1e-4:
                    print("FAIL m%d k%d n%d ta%d tb%d err %.3e" % (m, k, n, ta, tb, err)); bad += 1
    print("shape", m, k, n, "done")
print("bad =", bad)
