"""Oracle (test infrastructure) -- `class Tensor` (src/TensorOps/Types.hs:52-109)
restated over numpy arrays, with the semantics of the nested-vector backend
(`instance Tensor (NTensor v a)`, src/TensorOps/Backend/NTensor.hs:132-250).

A tensor `t ns` is a numpy array of shape `ns`; values are immutable by
convention (every method returns a fresh array), like the pure Haskell values.
"""
import itertools

import numpy as np

from . import nested


class OTensor:
    """One backend instance = one element type (`ElemT t`)."""

    def __init__(self, dtype=np.float64):
        self.dtype = np.dtype(dtype)

    # -- construction ------------------------------------------------------
    def from_list(self, shape, xs):
        """`TT.fromList` (src/TensorOps/Tensor.hs:187-191): row-major fill;
        returns None when the list is too short (the `Maybe`)."""
        n = nested._prod(shape)
        xs = list(xs)
        if len(xs) < n:
            return None
        return np.array(xs[:n], dtype=self.dtype).reshape(tuple(shape))

    def generate(self, shape, f):
        """`generateA` at Identity (src/TensorOps/Types.hs:97-99): f gets the
        index tuple, first index slowest."""
        out = np.empty(tuple(shape), dtype=self.dtype)
        for i in itertools.product(*[range(d) for d in shape]):
            out[i] = f(i)
        return out

    def konst(self, shape, x):
        """`TT.konst` (src/TensorOps/Tensor.hs:49-54)."""
        return np.full(tuple(shape), x, dtype=self.dtype)

    # -- the class methods ----------------------------------------------------
    def liftT(self, f, xs):
        """`liftT` (Types.hs:56-59): apply `f :: Vec n e -> e` to every element
        position of n same-shaped tensors."""
        xs = [np.asarray(x, dtype=self.dtype) for x in xs]
        r = f(xs)
        shape = xs[0].shape if xs else ()
        return np.broadcast_to(np.asarray(r, dtype=self.dtype), shape).copy()

    def gmul(self, len_m, len_o, len_n, x, y):
        """`gmul` (Types.hs:60-66) = `Nested.gmul'` (Nested.hs:451-473)."""
        return nested.gmul(len_m, len_o, len_n,
                           np.asarray(x, dtype=self.dtype),
                           np.asarray(y, dtype=self.dtype)).astype(self.dtype, copy=False)

    def sumT(self, xs, shape):
        """`sumT` (Types.hs:69) = `sum'` left fold (Data/List/Util.hs:7-10);
        `shape` stands for the `SingI o` dictionary (needed for `[]`)."""
        return nested.sum_list([np.asarray(x, dtype=self.dtype) for x in xs],
                               tuple(shape), self.dtype)

    def scaleT(self, alpha, x):
        """`scaleT` (Types.hs:70)."""
        return (self.dtype.type(alpha) * np.asarray(x, dtype=self.dtype)).astype(self.dtype)

    def transp(self, x):
        """`transp` (Types.hs:71-73) = `transpose'` (Nested.hs:520-528)."""
        return np.array(nested.transpose(x), order="C", copy=True)

    def mapRows(self, len_n, f, x):
        """`mapRows` (Types.hs:77-81): apply f to every `ms`-slice under the
        leading `len_n` dims."""
        x = np.asarray(x, dtype=self.dtype)
        out = np.empty_like(x)
        lead = x.shape[:len_n]
        for i in itertools.product(*[range(d) for d in lead]):
            out[i] = f(x[i])
        return out

    def sumRows(self, x):
        """`sumRows` (Types.hs:82-84) = `sumRowsNested` (Nested.hs:550-560)."""
        return nested.sum_rows(np.asarray(x, dtype=self.dtype))

    def diag(self, rank, x):
        return nested.diag(rank, np.asarray(x, dtype=self.dtype))

    def getDiag(self, x):
        return nested.get_diag(np.asarray(x, dtype=self.dtype))

    def ixRows(self, len_m, f, x):
        """`ixRows` at Identity (Types.hs:100-106): f gets (index, ns-slice)."""
        x = np.asarray(x, dtype=self.dtype)
        lead = x.shape[:len_m]
        rows = {}
        for i in itertools.product(*[range(d) for d in lead]):
            rows[i] = np.asarray(f(i, x[i]), dtype=self.dtype)
        any_row = next(iter(rows.values())) if rows else np.zeros((), self.dtype)
        out = np.empty(lead + any_row.shape, dtype=self.dtype)
        for i, r in rows.items():
            out[i] = r
        return out

    def index(self, x, i):
        """`(!)` (Types.hs:107-109)."""
        return np.asarray(x, dtype=self.dtype)[tuple(i)]

    # -- Tensor.hs helpers ------------------------------------------------------
    def one_hot(self, n, hot, cold, i):
        """`TT.oneHot` (src/TensorOps/Tensor.hs:275-289)."""
        return self.generate((n,), lambda j: hot if j[0] == i else cold)

    def arg_max(self, x):
        """`TT.argMax` (src/TensorOps/Tensor.hs:291-305): a `Max (Arg x j)` semigroup fold.
        `Arg`'s `max` (base-4.9 `Data.Semigroup`, not vendored: `max x@(Arg a _) y@(Arg b _)
        | a >= b = x | otherwise = y`) keeps its LEFT argument on ties, so the EARLIEST index
        of the maximum wins."""
        x = np.asarray(x)
        best, bi = None, None
        for j in range(x.shape[0]):
            if best is None or not (best >= x[j]):
                best, bi = x[j], j
        return bi

    def arg_min(self, x):
        """`TT.argMin` (src/TensorOps/Tensor.hs:307-321): a `Min (Arg x j)` fold; `Arg`'s `min`
        (`min x@(Arg a _) y@(Arg b _) | a <= b = x | otherwise = y`) keeps its LEFT argument on ties."""
        x = np.asarray(x)
        best, bi = None, None
        for j in range(x.shape[0]):
            if best is None or not (best <= x[j]):
                best, bi = x[j], j
        return bi
