"""Oracle (test infrastructure) -- `Data.Nested` semantics over numpy arrays.

Two statements of every function:
  *_literal : pure-Python loops that follow the Haskell source term by term
              (small cases only), and
  the plain name : a vectorised numpy formulation used at larger sizes.
`tests/test_oracle_nested.py` checks the two agree exactly on integer data.

Index order is row-major, first dim slowest (`genNestedA`,
src/Data/Nested.hs:362-369; `genBTensorA`, src/TensorOps/Backend/BTensor.hs:503-511).
"""
import itertools

import numpy as np


def _prod(xs):
    p = 1
    for x in xs:
        p *= int(x)
    return p


def gmul_literal(len_m, len_o, len_n, x, y):
    """`gmul'` (src/Data/Nested.hs:451-473), term by term.

    x : ms ++ os,  y : Reverse os ++ ns,  result : ms ++ ns
    For every m-slice of x (`mapNVecSlices f lM x`, :465) sum over every index
    `i` of that slice, in row-major order of `i` (`itraverseNested`, :472),
    `x'[i] * indexNested' (reverse i) y` (:472).
    """
    x = np.asarray(x)
    y = np.asarray(y)
    ms = x.shape[:len_m]
    os_ = x.shape[len_m:]
    assert len(os_) == len_o, "x must have shape ms ++ os"
    assert y.shape[:len_o] == tuple(reversed(os_)), "y must have shape Reverse os ++ ns"
    ns = y.shape[len_o:]
    assert len(ns) == len_n
    out = np.zeros(ms + ns, dtype=np.result_type(x, y))
    for m in itertools.product(*[range(d) for d in ms]):
        acc = np.zeros(ns, dtype=out.dtype)
        for i in itertools.product(*[range(d) for d in os_]):
            acc = acc + x[m + i] * y[tuple(reversed(i))]
        out[m] = acc
    return out


def gmul(len_m, len_o, len_n, x, y):
    """Vectorised `gmul'`: contract x's trailing `len_o` axes with y's leading
    `len_o` axes *in reverse order* (src/Data/Nested.hs:472,
    src/TensorOps/Types.hs:60-66)."""
    x = np.asarray(x)
    y = np.asarray(y)
    assert x.ndim >= len_o and y.ndim >= len_o
    assert x.ndim - len_o == len_m and y.ndim - len_o == len_n
    os_ = x.shape[len_m:]
    assert y.shape[:len_o] == tuple(reversed(os_)), (x.shape, y.shape, len_o)
    ax_x = list(range(len_m, len_m + len_o))
    ax_y = list(reversed(range(len_o)))
    return np.tensordot(x, y, axes=(ax_x, ax_y))


def transpose_literal(x):
    """`transpose'` (src/Data/Nested.hs:520-528):
    `genNested sR $ \\i -> indexNested (reverse i) x`."""
    x = np.asarray(x)
    rdims = tuple(reversed(x.shape))
    out = np.empty(rdims, dtype=x.dtype)
    for i in itertools.product(*[range(d) for d in rdims]):
        out[i] = x[tuple(reversed(i))]
    return out


def transpose(x):
    """Full axis reversal `ns -> Reverse ns` (src/TensorOps/Types.hs:71-73)."""
    return np.transpose(np.asarray(x))


def sum_rows(x):
    """`sumRowsNested` (src/Data/Nested.hs:550-560): left fold `sum'` over the
    leading dim (src/Data/List/Util.hs:7-10)."""
    x = np.asarray(x)
    assert x.ndim >= 1
    if x.shape[0] == 0:
        return np.zeros(x.shape[1:], dtype=x.dtype)
    acc = x[0].copy()
    for r in range(1, x.shape[0]):
        acc = acc + x[r]
    return acc


def sum_list(xs, shape, dtype):
    """`sum'` (src/Data/List/Util.hs:7-10): `[] -> 0`, else `foldl1' (+)`."""
    if len(xs) == 0:
        return np.zeros(shape, dtype=dtype)
    acc = np.asarray(xs[0])
    for x in xs[1:]:
        acc = acc + np.asarray(x)
    return acc


def diag(rank, x):
    """`diag` (src/TensorOps/Types.hs:82-85; BTensor.hs:718-738): vector [n] ->
    rank-`rank` tensor, x[i] on the i,i,..,i diagonal, 0 elsewhere."""
    x = np.asarray(x)
    n = x.shape[0]
    if rank == 1:
        return x.copy()
    out = np.zeros((n,) * rank, dtype=x.dtype)
    for i in range(n):
        out[(i,) * rank] = x[i]
    return out


def get_diag(x):
    """`getDiag` (src/TensorOps/Types.hs:86-89; BTensor.hs:822-833)."""
    x = np.asarray(x)
    n = x.shape[0]
    return np.array([x[(i,) * x.ndim] for i in range(n)], dtype=x.dtype)


# ---- flat-GEMM formulation (what the HIP backend computes); kept here so the
# ---- tests can assert it equals the nested definition on integer data.
def digit_reversal_perm(os_):
    """Row gather `P` such that (P B_f)[rowmajor(o1..oq)] = B_f[rowmajor_revdims(oq..o1)]."""
    k = _prod(os_)
    if len(os_) <= 1:
        return np.arange(k)
    return np.arange(k).reshape(tuple(reversed(os_))).T.reshape(-1)


def gmul_flat(len_m, len_o, len_n, x, y):
    """C_f[M,N] = A_f[M,K] . (P B_f)[K,N] on the flat row-major buffers."""
    x = np.asarray(x)
    y = np.asarray(y)
    ms, os_ = x.shape[:len_m], x.shape[len_m:]
    ns = y.shape[len_o:]
    m, k, n = _prod(ms), _prod(os_), _prod(ns)
    a = x.reshape(m, k)
    b = y.reshape(k, n)[digit_reversal_perm(os_)]
    return (a @ b).reshape(ms + ns)
