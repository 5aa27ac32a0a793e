"""Oracle (test infrastructure) -- ctypes loader for the plain-C HMat-path restatement
(oracle/hmat_path.c).  Used by tests and by bench.py's `cpu_baseline` leg only."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle_hmat.so")
LIB32 = os.path.join(HERE, "liboracle_hmat_f32.so")   # the same text compiled in single precision (hmat_path.c, head)
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_lib = None


def build():
    src = os.path.join(HERE, "hmat_path.c")
    for lib in (LIB, LIB32):
        if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", HERE, "-s", os.path.basename(lib)])
    return LIB


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.hmat_batched_grads.restype = C.c_double
        L.hmat_batched_grads.argtypes = [C.c_int] * 4 + [_dp] * 10 + [C.c_int]
        L.hmat_train_online.restype = C.c_double
        L.hmat_train_online.argtypes = [C.c_int] * 4 + [_dp] * 6 + [C.c_double, C.c_int]
        L.hmat_gemm.restype = None
        L.hmat_gemm.argtypes = [C.c_int] * 3 + [_dp] * 3
        L.hmat_map_logistic.restype = None
        L.hmat_map_logistic.argtypes = [C.c_long, _dp, _dp]
        L.hmat_batched_grads_mt.restype = C.c_double
        L.hmat_batched_grads_mt.argtypes = [C.c_int] * 4 + [_dp] * 10 + [C.c_int, C.c_int]
        L.hmat_map_logistic_f32_mt.restype = None
        _fp = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
        L.hmat_map_logistic_f32_mt.argtypes = [C.c_long, _fp, _fp, C.c_int]
        L.hmat_call_counts.restype = None
        L.hmat_call_counts.argtypes = [C.POINTER(C.c_int)]
        L.hmat_batched_grads_pool.restype = C.c_double
        L.hmat_batched_grads_pool.argtypes = [C.c_int] * 5 + [_dp] * 10 + [C.c_int, C.c_int]
        _lib = L
    return _lib


_lib32 = None


def lib32():
    """the fp32 build: same entry points, float arrays, float results"""
    global _lib32
    if _lib32 is None:
        build()
        L = C.CDLL(LIB32)
        fp = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
        L.hmat_batched_grads.restype = C.c_float
        L.hmat_batched_grads.argtypes = [C.c_int] * 4 + [fp] * 10 + [C.c_int]
        _lib32 = L
    return _lib32


def batched_grads_f32(X, Y, W1, b1, W2, b2, recompute=True):
    """`batched_grads` through the fp32 build (what `HMat Float` would compute): the like-for-like CPU figure beside the
    fp32 GPU step.  Parity claims stay on the fp64 build."""
    X, Y, W1, b1, W2, b2 = (np.ascontiguousarray(a, dtype=np.float32) for a in (X, Y, W1, b1, W2, b2))
    B, i = X.shape
    h, o = W1.shape[0], W2.shape[0]
    g = [np.empty_like(W1), np.empty_like(b1), np.empty_like(W2), np.empty_like(b2)]
    loss = lib32().hmat_batched_grads(B, i, h, o, X, Y, W1, b1, W2, b2, *g, int(recompute))
    return g, loss


def batched_grads_pool(X, Y, W1, b1, W2, b2, threads, reps=1, recompute=True):
    """`reps` batches in one call on a persistent pool of `threads` pthreads (CPU-B)."""
    X, Y, W1, b1, W2, b2 = map(_c, (X, Y, W1, b1, W2, b2))
    B, i = X.shape
    h, o = W1.shape[0], W2.shape[0]
    g = [np.empty_like(W1), np.empty_like(b1), np.empty_like(W2), np.empty_like(b2)]
    loss = lib().hmat_batched_grads_pool(int(reps), B, i, h, o, X, Y, W1, b1, W2, b2, *g, int(recompute), int(threads))
    return g, loss


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def batched_grads(X, Y, W1, b1, W2, b2, recompute=True):
    """G = sum_b networkGradient(x_b, y_b) at fixed params; returns ([gW1,gb1,gW2,gb2], sum loss)."""
    X, Y, W1, b1, W2, b2 = map(_c, (X, Y, W1, b1, W2, b2))
    B, i = X.shape
    h, o = W1.shape[0], W2.shape[0]
    g = [np.empty_like(W1), np.empty_like(b1), np.empty_like(W2), np.empty_like(b2)]
    loss = lib().hmat_batched_grads(B, i, h, o, X, Y, W1, b1, W2, b2, *g, int(recompute))
    return g, loss


def batched_grads_mt(X, Y, W1, b1, W2, b2, threads, recompute=True):
    """batched_grads with the samples split over `threads` pthreads (CPU-B: a reported baseline)."""
    X, Y, W1, b1, W2, b2 = map(_c, (X, Y, W1, b1, W2, b2))
    B, i = X.shape
    h, o = W1.shape[0], W2.shape[0]
    g = [np.empty_like(W1), np.empty_like(b1), np.empty_like(W2), np.empty_like(b2)]
    loss = lib().hmat_batched_grads_mt(B, i, h, o, X, Y, W1, b1, W2, b2, *g, int(recompute), int(threads))
    return g, loss


def map_logistic_f32(x, threads=1):
    """`cmap logistic` over fp32, scalar loop, `threads` slices (CPU-D)."""
    x = np.ascontiguousarray(x, dtype=np.float32).ravel()
    y = np.empty_like(x)
    lib().hmat_map_logistic_f32_mt(x.size, x, y, int(threads))
    return y


def train_online(X, Y, W1, b1, W2, b2, rate, recompute=True):
    """Per-sample online SGD (app/MNIST.hs:390-396); returns updated params and sum loss."""
    X, Y = _c(X), _c(Y)
    p = [np.array(a, dtype=np.float64, order="C", copy=True) for a in (W1, b1, W2, b2)]
    B, i = X.shape
    loss = lib().hmat_train_online(B, i, p[0].shape[0], p[2].shape[0], X, Y, *p, float(rate), int(recompute))
    return p, loss


def _stack_lib(f32):
    L = lib32() if f32 else lib()
    if not getattr(L, "_stack_ready", False):
        fp = np.ctypeslib.ndpointer(dtype=np.float32 if f32 else np.float64, flags="C_CONTIGUOUS")
        ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
        L.hmat_train_online_stack.restype = C.c_float if f32 else C.c_double
        L.hmat_train_online_stack.argtypes = [C.c_int, C.c_int, ip, fp, fp, fp, C.c_float if f32 else C.c_double, C.c_int]
        L.hmat_classify_stack.restype = None
        L.hmat_classify_stack.argtypes = [C.c_int, C.c_int, ip, fp, fp, ip]
        L.hmat_stack_call_counts.restype = None
        L.hmat_stack_call_counts.argtypes = [C.POINTER(C.c_int)]
        L._stack_ready = True
    return L


def _flat(ws, dt):
    return np.ascontiguousarray(np.concatenate([np.concatenate([np.asarray(w, dtype=dt).ravel(), np.asarray(b, dtype=dt).ravel()])
                                                for w, b in ws]))


def _unflat(p, dims):
    out, o = [], 0
    for i, n in zip(dims, dims[1:]):
        w = p[o:o + n * i].reshape(n, i); o += n * i
        b = p[o:o + n]; o += n
        out.append((w.copy(), b.copy()))
    return out


def train_online_stack(X, Y, ws, rate, recompute=True, f32=False):
    """Per-sample online SGD (app/MNIST.hs:390-396) on an ffLayer stack of any depth (hidden `actMap logistic`, `actSoftmax`,
    crossEntropy): ws = [(W1, b1), ...]; returns the updated [(W, b)] and the summed loss.  f32: the single-precision build."""
    dt = np.float32 if f32 else np.float64
    X, Y = np.ascontiguousarray(X, dtype=dt), np.ascontiguousarray(Y, dtype=dt)
    dims = np.asarray([ws[0][0].shape[1]] + [w.shape[0] for w, _ in ws], dtype=np.int32)
    p = _flat(ws, dt)
    loss = _stack_lib(f32).hmat_train_online_stack(len(X), len(ws), dims, X, Y, p, float(rate), int(recompute))
    return _unflat(p, [int(v) for v in dims]), float(loss)


def classify_stack(X, ws, f32=False):
    """`runNetwork` + `argMax` per sample (the app's validation loop, app/MNIST.hs:366-389)."""
    dt = np.float32 if f32 else np.float64
    X = np.ascontiguousarray(X, dtype=dt)
    dims = np.asarray([ws[0][0].shape[1]] + [w.shape[0] for w, _ in ws], dtype=np.int32)
    out = np.empty(len(X), dtype=np.int32)
    _stack_lib(f32).hmat_classify_stack(len(X), len(ws), dims, X, _flat(ws, dt), out)
    return out


def stack_call_counts():
    a = (C.c_int * 5)()
    _stack_lib(False).hmat_stack_call_counts(a)
    return dict(zip(["hidden_gemv", "hidden_add", "hidden_logistic", "last_gemv", "last_add"], list(a)))


def gemm(A, B):
    A, B = _c(A), _c(B)
    out = np.empty((A.shape[0], B.shape[1]))
    lib().hmat_gemm(A.shape[0], A.shape[1], B.shape[1], A, B, out)
    return out


def map_logistic(x):
    x = _c(x)
    y = np.empty_like(x)
    lib().hmat_map_logistic(x.size, x.ravel(), y.ravel())
    return y


def call_counts():
    a = (C.c_int * 10)()
    lib().hmat_call_counts(a)
    names = ["gemv_l1", "add_b1", "logistic", "gemv_l2", "add_b2", "exp", "sum_rows", "recip",
             "scale_sv", "log"]
    return dict(zip(names, list(a)))
