"""`instance BLAS HipB` (src/TensorOps/BLAS.hs:90-173) over device handles, via the C ABI -- the Python twin of
hs/TensorOps/BLAS/HIP.hs, method for method: every class method is ONE `to_blas_*` / `to_lift` / `to_index` call;
the Applicative-effectful traversals (`iRowsB`, `iElemsB`, `bgenA`, `bgenRowsA`, BLAS.hs:140-159) are one download,
a host traversal and one upload, as in the Haskell shim (the effects are arbitrary, they cannot run on the device).

This is the reference's INNER boundary, the one its README prescribes ("make your type an instance of the `BLAS`
typeclass ... and you get it for free", README.md:150-154): with it `BTensor v HipB` is a `Tensor` through
src/TensorOps/Backend/BTensor.hs:775-879 unchanged.  Harness-side plumbing only (like hipt.py); the tests drive
BTensor's dispatcher (restated in oracle/btensor.py, test infrastructure) over this dictionary on the GPU.
"""
import ctypes as C
import itertools

import numpy as np

from . import capi
from .capi import check, lib
from .hipt import DT, HipT, Sym, _arr, _out


class HipB:
    def __init__(self, device=0, dtype=np.float32):
        self.T = HipT(device, dtype=dtype) if np.dtype(dtype) == np.float64 else HipT(device)
        self.dt = np.dtype(dtype)
        self._closures = {}

    # run-time shape of a handle (Haskell: `Sing s`)
    def dimsB(self, x):
        return tuple(x.shape)

    def fromArray(self, a):
        return self.T.put(np.asarray(a, dtype=self.dt))

    def toArray(self, x):
        return x.numpy()

    def _new(self, fn, *args):
        h = _out()
        check(fn(*args, C.byref(h)))
        return DT(h)

    # liftB (BLAS.hs:92-96): the closure is applied ONCE to symbolic elements and compiled (to_expr_compile); a closure
    # that ignores its arguments, or the empty vector of operands, is a constant of shape s (HMat.hs:115-119)
    def liftB(self, dims, f, xs):
        xs = list(xs)
        if not xs:
            return self.T.konst(dims, float(f([])))
        key = (f, len(xs))
        e = self._closures.get(key)
        if e is None:
            probe = self.T.expr(f, len(xs), key=("hipb", id(f), len(xs)))
            e = self._closures[key] = probe
        return self.T.liftT(e, xs)

    def axpy(self, a, x, y):                                                 # (:97-101)
        return self._new(lib().to_blas_axpy, float(a), x.h, y.h if y is not None else None)

    def dot(self, x, y):                                                     # (:102-104)
        v = C.c_double()
        check(lib().to_blas_dot(x.h, y.h, C.byref(v)))
        return v.value

    def ger(self, x, y):                                                     # (:108-110)
        return self._new(lib().to_blas_ger, x.h, y.h)

    def gemv(self, a, A, x, by):                                             # (:111-116)
        if by is None:
            return self._new(lib().to_blas_gemv, float(a), A.h, x.h, 0.0, None)
        return self._new(lib().to_blas_gemv, float(a), A.h, x.h, float(by[0]), by[1].h)

    def gemm(self, a, A, B, bc):                                             # (:117-123)
        if bc is None:
            return self._new(lib().to_blas_gemm, float(a), A.h, B.h, 0.0, None)
        return self._new(lib().to_blas_gemm, float(a), A.h, B.h, float(bc[0]), bc[1].h)

    def scaleB(self, a, x):                                                  # (:124-127)
        return self._new(lib().to_blas_scale, float(a), x.h)

    def addB(self, x, y):                                                    # (:128)
        return self._new(lib().to_blas_add, x.h, y.h)

    def indexB(self, idx, x):                                                # (:129-132)
        return self.T.index(x, tuple(idx))

    def indexRowB(self, i, A):                                               # (:133-136) a zero-copy view
        return self._new(lib().to_blas_index_row, int(i), A.h)

    def transpB(self, A):                                                    # (:137-139) a view, like `tr`
        return self._new(lib().to_blas_transp, A.h)

    def iRowsB(self, f, A):                                                  # (:140-143)
        host = A.numpy()
        rows = [f(i, self.T.put(host[i])) for i in range(host.shape[0])]
        return self.T.put(np.stack([r.numpy() for r in rows]))

    def iElemsB(self, f, x):                                                 # (:144-147)
        host = x.numpy()
        out = np.empty_like(host)
        for i in itertools.product(*[range(d) for d in host.shape]):
            out[i] = f(i, float(host[i]))
        return self.T.put(out)

    def bgen(self, dims, f):                                                 # (:149-153)
        return self.T.generate(tuple(dims), f)

    def bgenRows(self, n, f):                                                # (:154-159) to_stack of the rows
        return self.T.stack((n,), [f(i) for i in range(n)])

    def eye(self, n):                                                        # (:160-161)
        return self._new(lib().to_blas_eye, self.T.to_dtype, int(n))

    def traceB(self, A):                                                     # (:162-164)
        v = C.c_double()
        check(lib().to_blas_trace(A.h, C.byref(v)))
        return v.value

    def diagB(self, x):                                                      # (:165-167)
        return self._new(lib().to_blas_diag, x.h)

    def getDiagB(self, A):                                                   # (:168-170)
        return self._new(lib().to_blas_get_diag, A.h)

    def sumB(self, x):                                                       # (:171-173)
        v = C.c_double()
        check(lib().to_blas_sum(x.h, C.byref(v)))
        return v.value
