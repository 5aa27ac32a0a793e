"""gemm_kwave.hip: bit-exact integer check on every operand layout (forced route), then a timing A/B.
   usage: kw_check.py check | kw_check.py time   (TOPS_GEMM_KW / TOPS_GEMM_KW_WAVES are read once per process)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tensor_ops_amd.hipt import HipT
T = HipT(0, dtype=np.float64) if os.environ.get("KW_DTYPE") == "f64" else HipT(0)


def check():
    bad = 0
    shapes = [(1024, 1024, 1024), (1000, 1000, 1000), (768, 768, 768), (128, 128, 128), (128, 80, 192), (1100, 528, 900),
              (260, 1000, 388), (128, 1030, 136), (1001, 66, 1003), (512, 4096, 512), (196, 333, 200), (1536, 200, 1536)]
    for m, k, n in shapes:
        for ta in (0, 1):
            for tb in (0, 1):
                if (ta and m % 4) or (not tb and n % 4):
                    continue
                dt = np.float64 if os.environ.get("KW_DTYPE") == "f64" else np.float32
                rng = np.random.default_rng(m + 3 * k + 7 * n + ta * 2 + tb)
                a = rng.integers(-2, 3, size=(m, k)).astype(dt)
                b = rng.integers(-2, 3, size=(k, n)).astype(dt)
                da = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
                db = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
                l0 = T.stats()["launches"]
                got = T.gmul(1, 1, 1, da, db).numpy()
                nl = T.stats()["launches"] - l0
                want = (a.astype(np.float64) @ b.astype(np.float64)).astype(dt)
                ok = np.array_equal(got, want)
                bad += not ok
                print(m, k, n, "ta", ta, "tb", tb, "launches", nl, "OK" if ok else "MISMATCH %d of %d, max %g" % ((got != want).sum(), got.size, np.abs(got - want).max()))
    print("bad", bad)
    sys.exit(1 if bad else 0)


def timeit():
    def ours(m, k, n, iters=int(os.environ.get("KW_ITERS", "50")), warm=int(os.environ.get("KW_WARM", "20"))):
        a = T.genRand((m, k), "uniform", -1, 1, 1); b = T.genRand((k, n), "uniform", -1, 1, 2)
        for _ in range(warm): T.gmul(1, 1, 1, a, b)
        T.sync(); T.timer_start()
        for _ in range(iters): T.gmul(1, 1, 1, a, b)
        return T.timer_stop() / iters
    shapes = [(s, s, s) for s in (512, 640, 768, 896, 1000, 1024, 1152, 1280, 1408, 1536, 1792, 2048)]
    shapes += [(1024, 4096, 1024), (1024, 8192, 1024), (2048, 512, 2048), (1024, 784, 256), (4096, 784, 256), (16384, 256, 4096)]
    if os.environ.get("KW_SHAPES"):
        shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["KW_SHAPES"].split(",")]
    for m, k, n in shapes:
        t = ours(m, k, n)
        print("%6d x %6d x %6d   %8.4f ms %7.2f TF" % (m, k, n, t, 2.0 * m * k * n / t / 1e9), flush=True)


if __name__ == "__main__":
    check() if sys.argv[1] == "check" else timeit()
