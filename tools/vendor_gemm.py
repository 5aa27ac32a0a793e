"""Yardstick, not a dependency: the vendor GEMM (torch.mm -> hipBLASLt / rocBLAS) on the shapes bench.py
reports, timed the same way (30 warm-up launches, 50 timed, HIP events)."""
import torch

assert torch.cuda.is_available()
dev = torch.device("cuda", 0)


def timeit(fn, iters=50, warm=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


torch.backends.cuda.matmul.allow_tf32 = False
for name, dt, m, k, n in (("fp32 4096^3", torch.float32, 4096, 4096, 4096),
                          ("fp32 8192^3", torch.float32, 8192, 8192, 8192),
                          ("fp32 262144x64x512 (config 5a)", torch.float32, 262144, 64, 512),
                          ("fp32 1024x784x256 (config 3 forward)", torch.float32, 1024, 784, 256),
                          ("fp64 4096^3", torch.float64, 4096, 4096, 4096)):
    a = torch.rand(m, k, device=dev, dtype=dt) * 2 - 1
    b = torch.rand(k, n, device=dev, dtype=dt) * 2 - 1
    c = torch.empty(m, n, device=dev, dtype=dt)
    ms = timeit(lambda: torch.mm(a, b, out=c))
    print("torch.mm %-40s %8.4f ms  %7.2f TFLOP/s" % (name, ms, 2.0 * m * k * n / ms / 1e9))
