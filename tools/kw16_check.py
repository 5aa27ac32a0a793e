"""gemm_kw16.hip: bit-exact integer check of one tile shape of the menu on every operand layout (forced route, development
build: TOPS_GEMM_KW16=2 TOPS_GEMM_KW16_TILE=0..7 = 48x48 / 48x64 / 64x48 / 80x80 / 32x64 / 64x32 / 64x80 / 80x64), or a timing run.
   usage: kw16_check.py check | kw16_check.py time [M K N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tensor_ops_amd.hipt import HipT
T = HipT(0)


def check():
    bad = 0
    shapes = [(768, 768, 768), (1280, 1280, 1280), (96, 64, 96), (100, 80, 104), (144, 1030, 240), (1000, 1000, 1000), (500, 264, 332),
              (1152, 96, 1152), (260, 333, 388), (1001, 66, 1003), (768, 16 * 9, 816), (480, 16 * 5 + 3, 496),
              (1088, 272, 1088), (1024, 300, 1280), (1276, 264, 1024), (1100, 256, 1100)]
    for m, k, n in shapes:
        for ta in (0, 1):
            for tb in (0, 1):
                if (ta and m % 4) or (not tb and n % 4):
                    continue
                rng = np.random.default_rng(m + 3 * k + 7 * n + ta * 2 + tb)
                a = rng.integers(-2, 3, size=(m, k)).astype(np.float32)
                b = rng.integers(-2, 3, size=(k, n)).astype(np.float32)
                da = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
                db = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
                l0 = T.stats()["launches"]
                got = T.gmul(1, 1, 1, da, db).numpy()
                nl = T.stats()["launches"] - l0
                want = (a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32)
                ok = np.array_equal(got, want)
                bad += not ok
                if not ok:
                    w = np.argwhere(got != want)
                    print(m, k, n, "ta", ta, "tb", tb, "launches", nl, "MISMATCH %d of %d, max %g; rows %s cols %s" % (
                        len(w), got.size, np.abs(got - want).max(), np.unique(w[:, 0])[:12], np.unique(w[:, 1])[:12]))
    print("kw16_check mismatches", bad)
    sys.exit(1 if bad else 0)


def timeit(v):
    shapes = [(768, 768, 768), (1280, 1280, 1280), (1152, 1152, 1152), (640, 640, 640), (1024, 512, 1024), (896, 896, 896)]
    if v:
        shapes = [tuple(v[i:i + 3]) for i in range(0, len(v), 3)]
    for m, k, n in shapes:
        a = T.genRand((m, k), "uniform", -1, 1, 1); b = T.genRand((k, n), "uniform", -1, 1, 2)

        def run(iters, warm):
            for _ in range(warm): T.gmul(1, 1, 1, a, b)
            T.sync(); T.timer_start()
            for _ in range(iters): T.gmul(1, 1, 1, a, b)
            return T.timer_stop() / iters
        est = max(run(20, 5), 1e-3)
        ms = run(max(20, int(40.0 / est)), max(20, int(60.0 / est)))
        print("%6d x %6d x %6d  %8.4f ms %7.2f TF" % (m, k, n, ms, 2.0 * m * k * n / ms / 1e9), flush=True)


if __name__ == "__main__":
    check() if sys.argv[1] == "check" else timeit([int(x) for x in sys.argv[2:]])
