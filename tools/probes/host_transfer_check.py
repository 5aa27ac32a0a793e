"""Does THIS host lose pieces of pageable transfers on a shared GPU?  A stand-alone check (raw HIP through ctypes, no library):
upload a numpy array, download it into a buffer full of a sentinel, compare -- fresh arrays every round, so freed pages
come back at recycled addresses.  Run several copies at once beside your workload:  python host_transfer_check.py [seconds]"""
import ctypes as C, sys, time
import numpy as np
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
budget, t0, rounds, lost = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, time.time(), 0, 0
rng = np.random.default_rng()
dev = C.c_void_p()
assert hip.hipMalloc(C.byref(dev), 64 << 20) == 0
while time.time() - t0 < budget:
    n = int(rng.integers(1 << 10, 16 << 20))                      # 4 KiB .. 64 MiB of fp32
    src = rng.integers(1, 1 << 20, n).astype(np.float32)          # (a fresh pageable array)
    dst = np.full(n, -7.0, dtype=np.float32)                      # sentinel: no source value is negative
    assert hip.hipMemcpy(dev, src.ctypes.data_as(C.c_void_p), n * 4, 1) == 0        # host -> device
    assert hip.hipMemcpy(dst.ctypes.data_as(C.c_void_p), dev, n * 4, 2) == 0        # device -> host
    bad = int((dst != src).sum())
    if bad:
        lost += 1
        print("round %d: %d of %d words wrong, %d still the sentinel" % (rounds, bad, n, int((dst == -7.0).sum())), flush=True)
    rounds += 1
print("host_transfer_check: %d rounds, %d with lost pieces" % (rounds, lost))
sys.exit(1 if lost else 0)
