"""Oracle (TEST INFRASTRUCTURE): `BTensor v b`, the reference's generic tensor over ANY `class BLAS b`
(src/TensorOps/Backend/BTensor.hs), restated once and parametrised over a BLAS dictionary.

This is the route the reference's README prescribes for a new backend ("make your type an instance of the `BLAS`
typeclass ... and you get it for free", README.md:150-154): `instance Tensor (BTensor v b)` (BTensor.hs:775-879) is
written ONLY in terms of the class methods of `b` (src/TensorOps/BLAS.hs:90-173).  `BTensorOps(blas)` below is that
instance; `blas` is any object with the class's methods:

    liftB(dims, f, xs) axpy(a, x, y|None) dot(x, y) ger(x, y) gemv(a, A, x, None|(b, y)) gemm(a, A, B, None|(b, C))
    scaleB(a, x) addB(x, y) indexB(idx, x) indexRowB(i, A) transpB(A) iRowsB(f, A) iElemsB(f, x) bgen(dims, f)
    bgenRows(n, f) eye(n) traceB(A) diagB(x) getDiagB(A) sumB(x)        + dimsB(x) (Haskell has `Sing s` instead)

`HMatB` is the reference's own instance (src/TensorOps/BLAS/HMat.hs:103-231, hmatrix forms restated in numpy); the
GPU tests instantiate the same `BTensorOps` with `tensor_ops_amd.hipb.HipB` (= hs/TensorOps/BLAS/HIP.hs over the
`to_blas_*` entry points) and compare both with the authoritative definition `oracle.nested.gmul`
(src/Data/Nested.hs:451-473).  `Counting(blas)` counts the class-method calls a dispatch makes.

What a Python restatement cannot carry: the type-level lengths (`Length ms/os/ns`, `Sing ns`) are run-time ints /
tuples here, and Haskell's impossible cases (`case ss of {}`) are assertions.  Association of the monoidal folds
(`ifoldMapBTensor` through `Const`, BTensor.hs:301-308; `sum` of `BTN`, :773) is the list `traverse`'s: right-nested
with the unit last -- x1 + (x2 + (... + (xn + 0))).  It is unobservable on the integer-valued data the tests use.
"""
import itertools

import numpy as np


# ---- the value -------------------------------------------------------------------------------------------------------
class BT:
    """`data BTensor v b ns` (BTensor.hs:58-63): BTS scalar | BTV b('BV n) | BTM b('BM n m) | BTN (v n (BTensor ...))."""
    __slots__ = ("tag", "val", "dims")

    def __init__(self, tag, val, dims):
        self.tag, self.val, self.dims = tag, val, tuple(int(d) for d in dims)

    def __repr__(self):
        return "BT%s%s" % (self.tag, list(self.dims))


def _prod_indices(dims):
    return itertools.product(*[range(d) for d in dims])


class BTensorOps:
    def __init__(self, blas):
        self.b = blas

    # constructors --------------------------------------------------------------------------------------------------
    def BTS(self, x):
        return BT("S", x, ())

    def BTV(self, v):
        return BT("V", v, self.b.dimsB(v))

    def BTM(self, m):
        return BT("M", m, self.b.dimsB(m))

    def BTN(self, xs):
        xs = list(xs)
        assert xs and len(xs[0].dims) >= 2, "BTN nests tensors of rank >= 2 (BTensor.hs:62)"
        return BT("N", xs, (len(xs),) + xs[0].dims)

    # genBTensorA at the identity functor (BTensor.hs:498-519) -------------------------------------------------------
    def gen(self, dims, f):
        dims = tuple(dims)
        if len(dims) == 0:
            return self.BTS(f(()))
        if len(dims) <= 2:
            return BT("V" if len(dims) == 1 else "M", self.b.bgen(dims, f), dims)
        return self.BTN([self.gen(dims[1:], lambda is_, i=i: f((i,) + is_)) for i in range(dims[0])])

    # indexBTensor (:521-543) ------------------------------------------------------------------------------------------
    def index(self, idx, t):
        idx = tuple(idx)
        if t.tag == "S":
            assert idx == ()
            return t.val
        if t.tag in "VM":
            return self.b.indexB(idx, t.val)
        return self.index(idx[1:], t.val[idx[0]])

    # btn (:545-556): how a vector of sub-tensors becomes a tensor ------------------------------------------------------
    def btn(self, child_rank, xs):
        xs = list(xs)
        n = len(xs)
        if child_rank == 0:
            return self.BTV(self.b.bgen((n,), lambda i: xs[i[0]].val))
        if child_rank == 1:
            return self.BTM(self.b.bgenRows(n, lambda i: xs[i].val))
        return self.BTN(xs)

    # mapBase / zipBase (:453-489) -------------------------------------------------------------------------------------
    def mapBase(self, f, g, h, t):
        if t.tag == "S":
            return self.BTS(f(t.val))
        if t.tag == "V":
            return self.BTV(g(t.val))
        if t.tag == "M":
            return self.BTM(h(t.val))
        return self.BTN([self.mapBase(f, g, h, x) for x in t.val])

    def zipBase(self, f, g, h, t, u):
        assert t.tag == u.tag and t.dims == u.dims, (t, u)
        if t.tag == "S":
            return self.BTS(f(t.val, u.val))
        if t.tag == "V":
            return self.BTV(g(t.val, u.val))
        if t.tag == "M":
            return self.BTM(h(t.val, u.val))
        return self.BTN([self.zipBase(f, g, h, x, y) for x, y in zip(t.val, u.val)])

    # instance Num (BTensor v b ns) (:100-131): matrices add THROUGH GEMM with an identity -------------------------------
    def add(self, t, u):
        b = self.b
        return self.zipBase(lambda x, y: x + y, lambda xs, ys: b.axpy(1, xs, ys),
                            lambda xs, ys: b.gemm(1, xs, b.eye(b.dimsB(xs)[1]), (1, ys)), t, u)

    def sub(self, t, u):
        b = self.b
        return self.zipBase(lambda x, y: x - y, lambda xs, ys: b.axpy(-1, ys, xs),
                            lambda xs, ys: b.gemm(1, xs, b.eye(b.dimsB(xs)[1]), (-1, ys)), t, u)

    def negate(self, t):
        b = self.b
        # (:121-124; the matrix case is `gemm 1 xs (eye sM) Nothing` in the source -- x itself, not -x: kept as written
        #  would make `negate` the identity on matrices.  TOp.negate goes through scaleT, never through this method.)
        return self.mapBase(lambda x: -x, lambda xs: b.axpy(-1, xs, None), lambda xs: b.gemm(1, xs, b.eye(b.dimsB(xs)[1]), None), t)

    def zero(self, dims):
        return self.gen(dims, lambda _: 0.0)           # fromInteger (:130)

    def _msum(self, dims, parts):
        """`getSum . foldMap Sum` over a list traversal: x1 + (x2 + (... + (xn + 0)))."""
        acc = self.zero(dims)
        for p in reversed(parts):
            acc = self.add(p, acc)
        return acc

    # dispatchBLAS (:141-175) ------------------------------------------------------------------------------------------
    def dispatchBLAS(self, lM, lO, lN, v, r):
        b = self.b
        key = (lM, lO, lN)
        if key == (0, 0, 0):
            return self.BTS(v.val * r.val)                                   # scalar-scalar (:152)
        if key == (0, 0, 1):
            return self.BTV(b.axpy(v.val, r.val, None))                      # scalar-vector (:155)
        if key == (0, 1, 0):
            return self.BTS(b.dot(v.val, r.val))                             # dot (:158)
        if key == (0, 1, 1):
            return self.BTV(b.gemv(1, b.transpB(r.val), v.val, None))        # vector-matrix (:162)
        if key == (1, 0, 0):
            return self.BTV(b.axpy(r.val, v.val, None))                      # vector-scalar (:165)
        if key == (1, 0, 1):
            return self.BTM(b.ger(v.val, r.val))                             # outer (:168)
        if key == (1, 1, 0):
            return self.BTV(b.gemv(1, v.val, r.val, None))                   # matrix-vector (:171)
        if key == (1, 1, 1):
            return self.BTM(b.gemm(1, v.val, r.val, None))                   # matrix-matrix (:174)
        raise AssertionError(key)

    # bIxRows at the identity functor (:222-261) -------------------------------------------------------------------------
    def bIxRows(self, sN, lO, f, t):
        b = self.b
        sN = tuple(sN)
        if not sN:
            return f((), t)
        s, ss = sN[0], sN[1:]
        if t.tag == "V":
            assert not ss
            xs = t.val
            if lO == 0:
                return self.BTV(b.iElemsB(lambda i, x: f(i, self.BTS(x)).val, xs))
            if lO == 1:
                return self.BTM(b.bgenRows(s, lambda i: f((i,), self.BTS(b.indexB((i,), xs))).val))
            return self.BTN([f((i,), self.BTS(b.indexB((i,), xs))) for i in range(s)])
        if t.tag == "M":
            xs = t.val
            if not ss:                                                        # ns ~ '[n], ms ~ '[m]
                if lO == 0:
                    return self.BTV(b.bgen((s,), lambda i: f(i, self.BTV(b.indexRowB(i[0], xs))).val))
                if lO == 1:
                    return self.BTM(b.iRowsB(lambda i, row: f((i,), self.BTV(row)).val, xs))
                return self.BTN([f((i,), self.BTV(b.indexRowB(i, xs))) for i in range(s)])
            assert len(ss) == 1                                               # ns ~ '[n,m], ms ~ '[]
            if lO == 0:
                return self.BTM(b.iElemsB(lambda ij, x: f(ij, self.BTS(x)).val, xs))
            return self.BTN([self.btn(lO, [f((i, j), self.BTS(b.indexB((i, j), xs))) for j in range(ss[0])]) for i in range(s)])
        assert t.tag == "N"
        return self.btn(len(ss) + lO, [self.bIxRows(ss, lO, lambda is_, y, i=i: f((i,) + is_, y), x) for i, x in enumerate(t.val)])

    def mapRows(self, sN, lO, f, t):                                          # mapRowsBTensor (:177-186)
        return self.bIxRows(sN, lO, lambda _i, x: f(x), t)

    # indexRowBTensor (:263-281) ----------------------------------------------------------------------------------------
    def indexRow(self, idx, t):
        idx = tuple(idx)
        if not idx:
            return t
        if t.tag == "V":
            assert len(idx) == 1
            return self.BTS(self.b.indexB(idx, t.val))
        if t.tag == "M":
            if len(idx) == 1:
                return self.BTV(self.b.indexRowB(idx[0], t.val))
            assert len(idx) == 2
            return self.BTS(self.b.indexB(idx, t.val))
        return self.indexRow(idx[1:], t.val[idx[0]])

    # ifoldMapBTensor (:301-322): every element with its index, row-major ------------------------------------------------
    def ielems(self, t):
        out = []
        if t.tag == "S":
            return [((), t.val)]
        if t.tag in "VM":
            self.b.iElemsB(lambda i, x: out.append((i, x)) or x, t.val)
            return out
        for i, x in enumerate(t.val):
            out += [((i,) + is_, e) for is_, e in self.ielems(x)]
        return out

    # traverseBTM at the identity functor (:391-406) ---------------------------------------------------------------------
    def mapBTM(self, sN, lM, f, t):
        sN = tuple(sN)
        if not sN:
            assert t.tag == "M"
            return f(t.val)
        assert t.tag == "N", "a tensor of rank >= 3 is nested (BTensor.hs:402-403 are empty cases)"
        return self.btn(len(sN) - 1 + lM, [self.mapBTM(sN[1:], lM, f, x) for x in t.val])

    # naiveGMul (:619-646) ------------------------------------------------------------------------------------------------
    def naiveGMul(self, sM, lO, lN, v, r):
        b = self.b
        ns = r.dims[lO:]

        def row(x_os):                                                       # x_os : BTensor os
            parts = []
            for is_, x in self.ielems(x_os):
                y = self.indexRow(tuple(reversed(is_)), r)
                parts.append(self.mapBase(lambda e: x * e, lambda ys: b.scaleB(x, ys), lambda ys: b.scaleB(x, ys), y))
            return self._msum(ns, parts)
        return self.mapRows(sM, lN, row, v)

    # gmulBLAS (:648-716) ------------------------------------------------------------------------------------------------
    def gmulBLAS(self, sM, lO, lN, v, r):
        b = self.b
        sM = tuple(sM)
        if lO == 0:
            if len(sM) <= 1:
                return self.dispatchBLAS(len(sM), 0, lN, v, r)
            if len(sM) == 2:
                if lN == 0:
                    return self.BTM(b.scaleB(r.val, v.val))                 # (:669)
                return self.naiveGMul(sM, 0, lN, v, r)                       # (:674)
            if lN == 0:                                                       # ms ~ ms0 ++ '[m1,m2] (:676-682)
                return self.mapBTM(sM[:-2], 2, lambda xs: self.BTM(b.scaleB(r.val, xs)), v)
            return self.naiveGMul(sM, 0, lN, v, r)                           # (:687)
        assert lO == 1
        if len(sM) <= 1:
            return self.dispatchBLAS(len(sM), 1, lN, v, r)                   # (:689)
        sM0 = sM[:-1]                                                         # ms ~ ms0 ++ '[m1] (:690-713)
        if lN == 0:
            return self.mapBTM(sM0, 1, lambda xs: self.BTV(b.gemv(1, xs, r.val, None)), v)       # (:700-702)
        return self.mapBTM(sM0, 2, lambda xs: self.BTM(b.gemm(1, xs, r.val, None)), v)           # (:703-710)

    # gmulB (:583-617) = `gmul` of the instance (:802-811) ---------------------------------------------------------------
    def gmul(self, lM, lO, lN, v, r):
        sM = v.dims[:lM]
        assert len(v.dims) == lM + lO and len(r.dims) == lO + lN, (lM, lO, lN, v, r)
        assert tuple(reversed(v.dims[lM:])) == r.dims[:lO], "B : Reverse os ++ ns (Types.hs:60-66)"
        if lN <= 1:
            if lO <= 1:
                return self.gmulBLAS(sM, lO, lN, v, r)                       # (:609-610)
            if lO == 2:
                if lN == 0:                                                   # trace(gemm) (:611-613)
                    ys = r.val
                    return self.mapBTM(sM, 0, lambda xs: self.BTS(self.b.traceB(self.b.gemm(1, xs, ys, None))), v)
                return self.naiveGMul(sM, lO, lN, v, r)                      # (:614)
            return self.naiveGMul(sM, lO, lN, v, r)                          # (:615)
        return self.naiveGMul(sM, lO, lN, v, r)                              # (:616)

    # liftBTensor (:345-369) = liftT ----------------------------------------------------------------------------------------
    def liftT(self, f, xs):
        xs = list(xs)
        t0 = xs[0]
        if t0.tag == "S":
            return self.BTS(f([x.val for x in xs]))
        if t0.tag in "VM":
            return BT(t0.tag, self.b.liftB(t0.dims, f, [x.val for x in xs]), t0.dims)
        return self.BTN([self.liftT(f, [x.val[i] for x in xs]) for i in range(t0.dims[0])])     # liftVecD: distribute

    # sumT = sum' (:796; Data/List/Util.hs:7-10): 0 for the empty list, else a LEFT fold -----------------------------------
    def sumT(self, ts, dims=None):
        ts = list(ts)
        if not ts:
            return self.zero(dims)
        acc = ts[0]
        for t in ts[1:]:
            acc = self.add(acc, t)
        return acc

    def scaleT(self, a, t):                                                   # (:799)
        b = self.b
        return self.mapBase(lambda x: a * x, lambda xs: b.scaleB(a, xs), lambda xs: b.scaleB(a, xs), t)

    # transpBTensor (:740-752) ------------------------------------------------------------------------------------------
    def transp(self, t):
        if t.tag in "SV":
            return t
        if t.tag == "M":
            return self.BTM(self.b.transpB(t.val))
        return self.gen(tuple(reversed(t.dims)), lambda i: self.index(tuple(reversed(i)), t))

    # sumBTensor (:754-773) = sumRows ---------------------------------------------------------------------------------------
    def sumRows(self, t):
        b = self.b
        if t.tag == "V":
            return self.BTS(b.sumB(t.val))
        if t.tag == "M":
            n = t.dims[0]
            return self.BTV(b.gemv(1, b.transpB(t.val), b.bgen((n,), lambda _: 1.0), None))
        return self._msum(t.dims[1:], list(t.val))

    # diagBTensor / getDiag (:718-738, :816-824) ------------------------------------------------------------------------------
    def diag(self, rank, t):
        assert t.tag == "V"
        n = t.dims[0]
        if rank == 1:
            return t
        if rank == 2:
            return self.BTM(self.b.diagB(t.val))
        return self.gen((n,) * rank, lambda i: self.b.indexB((i[0],), t.val) if len(set(i)) == 1 else 0.0)

    def getDiag(self, t):
        if t.tag == "M":
            return self.BTV(self.b.getDiagB(t.val))
        n, rank = t.dims[0], len(t.dims)
        return self.gen((n,), lambda i: self.index((i[0],) * rank, t))

    # host <-> BTensor (generateA / toList of the harness) -------------------------------------------------------------------
    def from_array(self, a):
        a = np.asarray(a)
        if a.ndim == 0:
            return self.BTS(float(a))
        if a.ndim <= 2:
            return BT("V" if a.ndim == 1 else "M", self.b.fromArray(a), a.shape)
        return self.BTN([self.from_array(a[i]) for i in range(a.shape[0])])

    def to_array(self, t):
        if t.tag == "S":
            return np.asarray(float(t.val))
        if t.tag in "VM":
            return np.asarray(self.b.toArray(t.val))
        return np.stack([self.to_array(x) for x in t.val])


# ---- the reference's own instance: HMat (src/TensorOps/BLAS/HMat.hs:103-231) ---------------------------------------------
class HMatB:
    """`instance BLAS (HMat a)`: hmatrix calls restated in numpy, in the forms the source writes them
    (`scale a` applied to the operand BEFORE the product, :135-163)."""

    def __init__(self, dtype=np.float64):
        self.dt = np.dtype(dtype)

    def dimsB(self, x):
        return tuple(x.shape)

    def fromArray(self, a):
        return np.array(a, dtype=self.dt)

    def toArray(self, x):
        return x

    def liftB(self, dims, f, xs):                                            # (:103-131)
        xs = list(xs)
        if not xs:
            return np.full(dims, f([]), dtype=self.dt)                       # konst (f ØV)
        out = np.empty(dims, dtype=self.dt)
        for i in _prod_indices(dims):
            out[i] = f([x[i] for x in xs])                                   # cmap / zipWith / liftB' (:94-101)
        return out

    def axpy(self, a, x, y):                                                 # (:135-139)
        r = self.dt.type(a) * x
        return r if y is None else y + r

    def dot(self, x, y):                                                     # (:141-142)
        return float(x @ y)

    def ger(self, x, y):                                                     # (:144-145)
        return np.outer(x, y)

    def gemv(self, a, A, x, by):                                             # (:147-152)
        r = A @ (self.dt.type(a) * x)
        return r if by is None else self.dt.type(by[0]) * by[1] + r

    def gemm(self, a, A, B, bc):                                             # (:154-159)
        r = A @ (self.dt.type(a) * B)
        return r if bc is None else self.dt.type(bc[0]) * bc[1] + r

    def scaleB(self, a, x):                                                  # (:161)
        return self.dt.type(a) * x

    def addB(self, x, y):                                                    # (:163)
        return x + y

    def indexB(self, idx, x):                                                # (:165-171)
        return float(x[tuple(idx)])

    def indexRowB(self, i, A):                                               # (:173)
        return A[i].copy()

    def transpB(self, A):                                                    # (:175)
        return A.T

    def iRowsB(self, f, A):                                                  # (:177-182)
        return np.stack([f(i, A[i].copy()) for i in range(A.shape[0])])

    def iElemsB(self, f, x):                                                 # (:183-198)
        out = np.empty_like(x)
        for i in _prod_indices(x.shape):
            out[i] = f(i, float(x[i]))
        return out

    def bgen(self, dims, f):                                                 # (:200-214)
        out = np.empty(dims, dtype=self.dt)
        for i in _prod_indices(dims):
            out[i] = f(i)
        return out

    def bgenRows(self, n, f):                                                # (:215-222)
        return np.stack([f(i) for i in range(n)])

    def eye(self, n):                                                        # (:224)
        return np.eye(n, dtype=self.dt)

    def traceB(self, A):                                                     # (:230)
        return float(np.trace(A))

    def diagB(self, x):                                                      # (:226)
        return np.diag(x)

    def getDiagB(self, A):                                                   # (:228)
        return np.diag(A).copy()

    def sumB(self, x):                                                       # (:232-235)
        return float(x.sum())


class Counting:
    """counts the class-method calls a dispatch makes on a BLAS dictionary"""

    def __init__(self, blas):
        self._b = blas
        self.calls = {}

    def __getattr__(self, name):
        f = getattr(self._b, name)
        if not callable(f) or name in ("dimsB", "fromArray", "toArray"):
            return f

        def g(*a, **k):
            self.calls[name] = self.calls.get(name, 0) + 1
            return f(*a, **k)
        return g


# ---- `instance Tensor (BTensor v b)` (BTensor.hs:775-879) in the shape oracle.top / oracle.neuralnet take a backend ----------
class BTensorT:
    """The dictionary `oracle.top`'s polymorphic TOp closures are applied to (same method names as `oracle.tensor.OTensor`
    and `tensor_ops_amd.hipt.HipT`), every method going through BTensor's dispatcher over the BLAS dictionary `blas`."""

    def __init__(self, blas):
        self.ops = BTensorOps(blas)
        self.dtype = np.dtype(getattr(blas, "dt", np.float64))               # `ElemT (BTensor v b) = ElemB b` (:789)

    def put(self, a):
        return self.ops.from_array(a)

    def get(self, t):
        return self.ops.to_array(t)

    def from_list(self, shape, xs):
        xs = list(xs)
        n = int(np.prod(shape)) if len(shape) else 1
        return None if len(xs) < n else self.put(np.array(xs[:n], dtype=np.float64).reshape(tuple(shape)))

    def generate(self, shape, f):                                            # generateA (:828)
        return self.ops.gen(tuple(shape), f)

    def konst(self, shape, x):
        return self.ops.gen(tuple(shape), lambda _: x)

    def liftT(self, f, xs):                                                  # (:790-794)
        return self.ops.liftT(f, xs)

    def gmul(self, lM, lO, lN, x, y):                                        # (:802-811)
        return self.ops.gmul(lM, lO, lN, x, y)

    def sumT(self, xs, shape):                                               # (:796)
        return self.ops.sumT(xs, tuple(shape))

    def scaleT(self, a, x):                                                  # (:799)
        return self.ops.scaleT(a, x)

    def transp(self, x):                                                     # (:825)
        return self.ops.transp(x)

    def mapRows(self, len_n, f, x):                                          # (:868-879)
        return self.ops.mapRows(x.dims[:len_n], len(x.dims) - len_n, f, x)

    def ixRows(self, len_m, f, x):                                           # (:834-846); `Length os` from the first row
        lead = x.dims[:len_m]
        probe = f(tuple(0 for _ in lead), self.ops.indexRow(tuple(0 for _ in lead), x))
        return self.ops.bIxRows(lead, len(probe.dims), f, x)

    def sumRows(self, x):                                                    # (:851-858)
        return self.ops.sumRows(x)

    def diag(self, rank, x):                                                 # (:813-815)
        return self.ops.diag(rank, x)

    def getDiag(self, x):                                                    # (:816-824)
        return self.ops.getDiag(x)

    def index(self, x, i):                                                   # (!) (:848)
        return self.ops.index(tuple(i), x)
