"""Pin the oracle's contraction/transpose semantics with hand-derived exact
known-answer tests (small integers, so every product and sum is exact) and the
identity  literal-loop definition == vectorised == flat-GEMM formulation."""
import numpy as np
import pytest

from oracle import nested


def iota(*shape):
    return np.arange(1, int(np.prod(shape)) + 1, dtype=np.float64).reshape(shape)


def test_kat_matmat_2x3_3x2():
    # A = [[1,2,3],[4,5,6]], B = [[1,2],[3,4],[5,6]]: by hand.
    c = nested.gmul_literal(1, 1, 1, iota(2, 3), iota(3, 2))
    assert c.tolist() == [[22.0, 28.0], [49.0, 64.0]]


def test_kat_matvec_and_dot_and_outer():
    a = iota(2, 3)
    x = np.array([1.0, 0.0, -1.0])
    assert nested.gmul_literal(1, 1, 0, a, x).tolist() == [-2.0, -2.0]
    assert float(nested.gmul_literal(0, 1, 0, x, np.array([2.0, 5.0, 7.0]))) == -5.0
    o = nested.gmul_literal(1, 0, 1, np.array([1.0, 2.0]), np.array([3.0, 4.0, 5.0]))
    assert o.tolist() == [[3.0, 4.0, 5.0], [6.0, 8.0, 10.0]]


def test_kat_reversed_contraction_order():
    # |os| = 2: C[m,n] = sum_{a,b} A[m,a,b] * B[b,a,n]   (Nested.hs:472 reverses i)
    # A[0] = [[1,2,3],[4,5,6]] (a in 0..1, b in 0..2); B has dims [3,2,1],
    # B[b,a,0] = 10*b + a  ->  sum = 1*0+2*10+3*20 + 4*1+5*11+6*21 = 80 + 185 = 265.
    a = iota(1, 2, 3)
    b = np.array([[[0.0], [1.0]], [[10.0], [11.0]], [[20.0], [21.0]]])
    assert nested.gmul_literal(1, 2, 1, a, b).tolist() == [[265.0]]
    # the NON-reversed reading would need B dims [2,3,1] and is a different number
    wrong = np.einsum("mab,abn->mn", a, b.reshape(2, 3, 1))
    assert wrong.tolist() != [[265.0]]


def test_kat_transpose_rank3():
    x = iota(2, 3, 4)
    t = nested.transpose_literal(x)
    assert t.shape == (4, 3, 2)
    assert t[3, 1, 0] == x[0, 1, 3] == 8.0
    assert t[0, 2, 1] == x[1, 2, 0] == 21.0
    assert np.array_equal(t, nested.transpose(x))


CASES = [
    ((3,), (4,), (2,)), ((2, 3), (4,), (5,)), ((2,), (3, 4), (2,)),
    ((2, 3), (2, 3), ()), ((2,), (2, 3, 2), (3, 2)), ((2, 3), (), (4,)),
    ((), (5,), ()), ((), (), (3,)), ((4,), (), ()), ((), (), ()),
    ((2, 2, 2), (3,), (2, 2)),
]


@pytest.mark.parametrize("ms,os_,ns", CASES)
def test_literal_equals_vectorised_equals_flat(ms, os_, ns):
    rng = np.random.default_rng(0x7e500001)
    a = rng.integers(-4, 5, size=ms + os_).astype(np.float64)
    b = rng.integers(-4, 5, size=tuple(reversed(os_)) + ns).astype(np.float64)
    lit = nested.gmul_literal(len(ms), len(os_), len(ns), a, b)
    vec = nested.gmul(len(ms), len(os_), len(ns), a, b)
    flat = nested.gmul_flat(len(ms), len(os_), len(ns), a, b)
    assert lit.shape == ms + ns
    assert np.array_equal(lit, vec)
    assert np.array_equal(lit, flat)


def test_blas_dispatch_equals_definition():
    """BTensor's dispatch (BTensor.hs:149-174) picks dot/gemv/ger/gemm/axpy;
    each must equal the nested definition."""
    rng = np.random.default_rng(1)
    ri = lambda *s: rng.integers(-3, 4, size=s).astype(np.float64)
    x, y, A, B, s = ri(4), ri(4), ri(3, 4), ri(4, 5), ri()
    assert nested.gmul(0, 0, 0, s, s) == s * s                                 # SS  (:152)
    assert np.array_equal(nested.gmul(0, 0, 1, s, x), s * x)                   # SV axpy (:155)
    assert nested.gmul(0, 1, 0, x, y) == x @ y                                 # dot (:158)
    assert np.array_equal(nested.gmul(0, 1, 1, x, B), B.T @ x)                 # VM gemv A^T (:162)
    assert np.array_equal(nested.gmul(1, 0, 0, x, s), s * x)                   # VS  (:165)
    assert np.array_equal(nested.gmul(1, 0, 1, x, y), np.outer(x, y))          # ger (:168)
    assert np.array_equal(nested.gmul(1, 1, 0, A, x), A @ x)                   # MV gemv (:171)
    assert np.array_equal(nested.gmul(1, 1, 1, A, B), A @ B)                   # MM gemm (:174)
    # trace(gemm) for |os| = 2, |ns| = 0 (BTensor.hs:613)
    P, Q = ri(3, 4), ri(4, 3)
    assert nested.gmul(0, 2, 0, P, Q) == np.trace(P @ Q)
    # batched-over-leading-dims gemm (BTensor.hs:707-710)
    T3 = ri(2, 3, 4)
    assert np.array_equal(nested.gmul(2, 1, 1, T3, B), np.stack([T3[i] @ B for i in range(2)]))


def test_sum_rows_and_sum_list_and_diag():
    x = iota(3, 2)
    assert nested.sum_rows(x).tolist() == [9.0, 12.0]
    assert nested.sum_list([], (2,), np.float64).tolist() == [0.0, 0.0]
    assert nested.sum_list([x, x, x], x.shape, x.dtype).tolist() == (3 * x).tolist()
    d = nested.diag(3, np.array([1.0, 2.0]))
    assert d.shape == (2, 2, 2) and d[1, 1, 1] == 2.0 and d.sum() == 3.0
    assert nested.get_diag(d).tolist() == [1.0, 2.0]
