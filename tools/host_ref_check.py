"""Is the HOST reference of the bit-exact tests itself reliable under load?  numpy's float matmul goes to a multi-threaded
BLAS; its int64 matmul does not.  On integer-valued operands the two must agree exactly.  N processes at once (the stress
harness runs 8 test loops side by side on a 256-core box, every one of them calling the BLAS with its default thread count).
usage: host_ref_check.py [processes] [iterations]      -- no GPU involved"""
import os, sys, time
import numpy as np


def work(tag, iters):
    rng = np.random.default_rng(17 + tag)
    shapes = [(1346, 339, 258), (1305, 341, 192), (2129, 157, 981), (1613, 896, 811), (802, 777, 1095), (1024, 1024, 1024)]
    bad = 0
    t0 = time.time()
    for it in range(iters):
        M, K, N = shapes[it % len(shapes)]
        for dt in (np.float64, np.float32):
            X = rng.integers(-2, 3, (M, K)).astype(dt); W = rng.integers(-2, 3, (N, K)).astype(dt)
            want = X.astype(np.float64) @ W.T.astype(np.float64)          # what the tests use
            exact = X.astype(np.int64) @ W.T.astype(np.int64)             # no BLAS
            if not np.array_equal(want, exact.astype(np.float64)):
                bad += 1
                rows = np.unique(np.nonzero(want != exact)[0]); cols = np.unique(np.nonzero(want != exact)[1])
                print("[p%d] HOST REFERENCE WRONG it %d %s %s: %d rows (%d..%d), %d cols (%d..%d)"
                      % (tag, it, (M, K, N), dt.__name__, len(rows), rows[0], rows[-1], len(cols), cols[0], cols[-1]), flush=True)
    print("[p%d] %d iterations, %d wrong references, %.0f s" % (tag, iters, bad, time.time() - t0), flush=True)


if __name__ == "__main__":
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    if len(sys.argv) > 3:
        work(int(sys.argv[3]), iters)
    else:
        import subprocess
        try:
            import threadpoolctl
            print(threadpoolctl.threadpool_info())
        except Exception as e:
            print("threadpoolctl:", e)
        ps = [subprocess.Popen([sys.executable, __file__, str(nproc), str(iters), str(i)]) for i in range(nproc)]
        for p in ps:
            p.wait()
