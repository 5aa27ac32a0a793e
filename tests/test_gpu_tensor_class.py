"""GPU parity: every `class Tensor` / `class BLAS` entry point of the C ABI against
the numpy oracle on the same seeded inputs.

Bar: bit-exact on small-integer data (every product and sum is exact in fp32) and
on shape/index work; <= 1e-5 relative (Frobenius) on random fp32 data, the
tolerance BASELINE.json's north_star states for fp32.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import nested  # noqa: E402
from oracle.tensor import OTensor  # noqa: E402

RTOL = 1e-5
SEED = 0x7e500001


@pytest.fixture(scope="module")
def T():
    from tensor_ops_amd.hipt import HipT
    return HipT(0)


@pytest.fixture(scope="module")
def O():
    return OTensor(np.float64)


def rel_err(got, want):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    den = np.linalg.norm(want.ravel())
    return np.linalg.norm((got - want).ravel()) / (den if den > 0 else 1.0)


def ints(rng, *shape):
    return rng.integers(-4, 5, size=shape).astype(np.float32)


GMUL_CASES = [
    ((3,), (4,), (2,)), ((2, 3), (4,), (5,)), ((2,), (3, 4), (2,)), ((2, 3), (2, 3), ()),
    ((2,), (2, 3, 2), (3, 2)), ((2, 3), (), (4,)), ((), (5,), ()), ((), (), (3,)), ((4,), (), ()),
    ((), (), ()), ((2, 2, 2), (3,), (2, 2)), ((70,), (33,), (65,)), ((130, 3), (17,), (50,)),
    ((5,), (4, 3, 2), ()), ((), (6, 2), (7,)),
]


@pytest.mark.parametrize("ms,os_,ns", GMUL_CASES)
def test_gmul_exact_on_integers(T, ms, os_, ns):
    rng = np.random.default_rng(SEED)
    a = ints(rng, *(ms + os_))
    b = ints(rng, *(tuple(reversed(os_)) + ns))
    want = nested.gmul(len(ms), len(os_), len(ns), a.astype(np.float64), b.astype(np.float64))
    got = T.gmul(len(ms), len(os_), len(ns), T.put(a), T.put(b)).numpy()
    assert got.shape == ms + ns
    assert np.array_equal(got, want.astype(np.float32))


@pytest.mark.parametrize("m,k,n", [(128, 128, 128), (300, 513, 129), (1024, 784, 256), (64, 8, 64),
                                   (1000, 10, 256), (33, 1000, 47), (256, 1024, 784), (2048, 64, 512)])
def test_gemm_random_1e5(T, m, k, n):
    rng = np.random.default_rng(SEED + m + k + n)
    a = rng.uniform(-1, 1, size=(m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, size=(k, n)).astype(np.float32)
    want = a.astype(np.float64) @ b.astype(np.float64)
    got = T.gmul(1, 1, 1, T.put(a), T.put(b)).numpy()
    assert rel_err(got, want) < RTOL


def test_gemm_all_transpose_combinations(T):
    """`transp` is a zero-copy view; gmul must consume it through trans flags."""
    rng = np.random.default_rng(SEED)
    m, k, n = 200, 136, 168
    a = rng.uniform(-1, 1, size=(m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, size=(k, n)).astype(np.float32)
    want = a.astype(np.float64) @ b.astype(np.float64)
    da, db = T.put(a), T.put(b)
    dat, dbt = T.transp(T.put(a.T.copy())), T.transp(T.put(b.T.copy()))
    for x in (da, dat):
        for y in (db, dbt):
            assert rel_err(T.gmul(1, 1, 1, x, y).numpy(), want) < RTOL
    st = T.stats()
    T.gmul(1, 1, 1, dat, dbt)
    assert T.stats()["launches"] - st["launches"] <= 2  # GEMM (+ split-K sum): no materialisation pass


def test_transp_rank3_bit_exact(T):
    x = np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4)
    t = T.transp(T.put(x))
    assert t.shape == (4, 3, 2)
    assert np.array_equal(t.numpy(), nested.transpose_literal(x))
    # gmul gradient form on rank>2: dA = gmul dC (transp B)   (TOp.hs:81)
    rng = np.random.default_rng(SEED)
    a, b = ints(rng, 2, 3, 4), ints(rng, 4, 5)
    dc = ints(rng, 2, 3, 5)
    got = T.gmul(2, 1, 1, T.put(dc), T.transp(T.put(b))).numpy()
    assert np.array_equal(got, np.einsum("xyn,kn->xyk", dc, b))
    got = T.gmul(1, 2, 1, T.transp(T.put(a)), T.put(dc)).numpy()
    assert np.array_equal(got, np.einsum("xyk,xyn->kn", a, dc))


def test_c5_shape_rank3_times_matrix(T):
    """BASELINE config 5 shape family: '[M1,M2,K] x '[K,N] is ONE flat GEMM."""
    rng = np.random.default_rng(SEED)
    a = rng.uniform(-1, 1, size=(24, 32, 64)).astype(np.float32)
    b = rng.uniform(-1, 1, size=(64, 96)).astype(np.float32)
    st = T.stats()
    got = T.gmul(2, 1, 1, T.put(a), T.put(b))
    assert T.stats()["launches"] - st["launches"] <= 2  # one GEMM (+ split-K sum), not 24 per-slice GEMMs
    want = np.tensordot(a.astype(np.float64), b.astype(np.float64), axes=1)
    assert got.shape == (24, 32, 96)
    assert rel_err(got.numpy(), want) < RTOL


def test_gmul_shape_errors_are_loud(T):
    from tensor_ops_amd.capi import TensorOpsError
    a, b = T.put(np.zeros((3, 4), np.float32)), T.put(np.zeros((5, 2), np.float32))
    with pytest.raises(TensorOpsError) as ei:
        T.gmul(1, 1, 1, a, b)
    assert ei.value.code == 2 and "contracted dims differ" in str(ei.value)
    with pytest.raises(TensorOpsError):
        T.gmul(2, 1, 1, a, b)


# ---- liftT -------------------------------------------------------------------------------------
def _lift_cases():
    from oracle import ad, neuralnet as NN
    return [
        ("logistic", 1, lambda v: NN.logistic(v[0]), 6, False),
        ("exp", 1, lambda v: ad.exp(v[0]), 3, False),
        ("log", 1, lambda v: ad.log(v[0]), 4, True),
        ("recip", 1, lambda v: ad.recip(v[0]), 5, True),
        ("tanh", 1, lambda v: ad.tanh(v[0]), 8, False),
        ("sqrt", 1, lambda v: ad.sqrt(v[0]), 9, True),
        ("affine_sgd", 2, lambda v: v[0] - 0.02 * v[1], 1, False),
        ("affine3", 3, lambda v: 2.0 * v[0] - v[1] + 0.5 * v[2] + 1.0, 1, False),
        ("mul", 2, lambda v: v[0] * v[1], 2, False),
        ("div", 2, lambda v: v[0] / v[1], 10, True),
        ("d_logistic", 2, lambda v: v[0] * NN.logistic_prime(v[1]), 7, False),
        ("d_logistic_ad", 2, lambda v: v[0] * ad.diff(NN.logistic)(v[1]), 7, False),
        # no pre-fused functor: specialised at run time with hiprtc (kind 100); the bytecode VM
        # (kind 0) runs them when TOPS_EXPR_JIT=0 -- see test_bytecode_vm_fallback
        ("jit_mixed", 2, lambda v: ad.sin(v[0]) * v[1] + ad.exp(-v[0] * v[0]), 100, False),
        ("jit_poly5", 4, lambda v: v[0] * v[1] - v[2] * ad.tanh(v[3]) + abs(v[0]), 100, False),
        ("jit_8ary", 8, lambda v: (v[0] + v[1] * v[2]) / (2.0 + v[3] * v[3]) - v[4] * v[5] + ad.cos(v[6]) * v[7], 100, False),
    ]


@pytest.mark.parametrize("case", _lift_cases(), ids=lambda c: c[0])
@pytest.mark.parametrize("shape", [(7,), (16, 12), (3, 5, 2)])
def test_lift_matches_oracle(T, O, case, shape):
    name, n, f, kind, positive = case
    rng = np.random.default_rng(SEED)
    lo, hi = (0.25, 2.0) if positive else (-2.0, 2.0)
    xs = [rng.uniform(lo, hi, size=shape).astype(np.float32) for _ in range(n)]
    e = T.expr(f, n, key=(name, n))
    assert e.kind == kind, "classifier picked kernel %d for %s" % (e.kind, name)
    got = T.liftT(e, [T.put(x) for x in xs]).numpy()
    want = O.liftT(f, [x.astype(np.float64) for x in xs])
    assert rel_err(got, want) < RTOL


def test_lift_large_vectorised_and_ragged(T, O):
    from oracle import neuralnet as NN
    rng = np.random.default_rng(SEED)
    for n in (1 << 20, (1 << 20) + 3):
        x = rng.uniform(-6, 6, size=n).astype(np.float32)
        got = T.liftT(lambda v: NN.logistic(v[0]), [T.put(x)], key="logi").numpy()
        want = 1 / (1 + np.exp(-x.astype(np.float64)))
        assert np.max(np.abs(got - want)) < 2e-7


def test_sumT_scaleT(T, O):
    rng = np.random.default_rng(SEED)
    xs = [ints(rng, 5, 6) for _ in range(6)]
    d = [T.put(x) for x in xs]
    for n in (0, 1, 2, 3, 4, 5, 6):
        got = T.sumT(d[:n], (5, 6)).numpy()
        assert np.array_equal(got, O.sumT(xs[:n], (5, 6)).astype(np.float32))
    assert np.array_equal(T.scaleT(-3.0, d[0]).numpy(), -3.0 * xs[0])


def test_sumRows_mapRows_diag_index(T, O):
    rng = np.random.default_rng(SEED)
    x = ints(rng, 9, 4, 3)
    dx = T.put(x)
    assert np.array_equal(T.sumRows(dx).numpy(), x.sum(axis=0))
    v = ints(rng, 300)
    assert T.sumRows(T.put(v)).numpy() == v.sum()
    wide = ints(rng, 50, 130)
    assert np.array_equal(T.sumRows(T.put(wide)).numpy(), wide.sum(axis=0))
    row = ints(rng, 4, 3)
    got = T.mapRows_const(1, T.put(row), dx).numpy()
    assert np.array_equal(got, np.broadcast_to(row, x.shape))
    # general mapRows = host traversal over row views
    got = T.mapRows(1, lambda r: T.scaleT(2.0, r), dx).numpy()
    assert np.array_equal(got, 2 * x)
    got = T.ixRows(2, lambda i, r: T.scaleT(float(i[0] * 10 + i[1]), r), dx).numpy()
    want = np.stack([np.stack([(i * 10 + j) * x[i, j] for j in range(4)]) for i in range(9)])
    assert np.array_equal(got, want)
    d = ints(rng, 5)
    for rank in (1, 2, 3):
        dd = T.diag(rank, T.put(d))
        assert np.array_equal(dd.numpy(), nested.diag(rank, d))
        if rank >= 2:
            assert np.array_equal(T.getDiag(dd).numpy(), d)
    assert T.index(dx, (8, 3, 2)) == x[8, 3, 2]
    assert T.index(T.transp(dx), (2, 3, 8)) == x[8, 3, 2]


def test_big_reductions(T):
    rng = np.random.default_rng(SEED)
    v = rng.uniform(-1, 1, size=(1 << 21) + 17).astype(np.float32)
    got = float(T.sumRows(T.put(v)).numpy())
    assert abs(got - v.astype(np.float64).sum()) < 1e-5 * np.abs(v).sum()
    w = rng.uniform(-1, 1, size=v.shape).astype(np.float32)
    got = float(T.gmul(0, 1, 0, T.put(v), T.put(w)).numpy())
    want = float(v.astype(np.float64) @ w.astype(np.float64))
    assert abs(got - want) < 1e-5 * float(np.abs(v * w).sum())


# ---- hidden batch dimension -------------------------------------------------------------------------
def test_batched_gmul_equals_per_sample(T):
    rng = np.random.default_rng(SEED)
    B = 37
    W = rng.uniform(-1, 1, size=(48, 70)).astype(np.float32)
    X = rng.uniform(-1, 1, size=(B, 70)).astype(np.float32)
    dW, dX = T.put(W), T.put(X, batched=True)
    # matVec W x_b for all b = one GEMM
    st = T.stats()
    z = T.gmul(1, 1, 0, dW, dX)
    assert T.stats()["launches"] - st["launches"] <= 2
    assert z.batch == B and z.shape == (48,)
    assert rel_err(z.numpy(), X.astype(np.float64) @ W.T.astype(np.float64)) < RTOL
    # per-sample outer product
    D = rng.uniform(-1, 1, size=(B, 48)).astype(np.float32)
    o = T.gmul(1, 0, 1, T.put(D, batched=True), dX).numpy()
    assert rel_err(o, np.einsum("bm,bn->bmn", D, X)) < RTOL
    # ... and its sum over the batch, fused: dW = D^T X   (one launch)
    dD = T.put(D, batched=True)
    st = T.stats()
    g = T.gmul_batch_sum(1, 0, 1, dD, dX)
    assert T.stats()["launches"] - st["launches"] <= 2
    assert g.batch == 0
    assert rel_err(g.numpy(), D.T.astype(np.float64) @ X.astype(np.float64)) < RTOL
    # W^T d_b  (vecMat through a transposed view)
    dh = T.gmul(1, 1, 0, T.transp(dW), dD).numpy()
    assert rel_err(dh, D.astype(np.float64) @ W.astype(np.float64)) < RTOL
    # per-sample dot and scalar*vector (softmax / crossEntropy pieces)
    Y = rng.uniform(-1, 1, size=(B, 48)).astype(np.float32)
    dot = T.gmul(0, 1, 0, dD, T.put(Y, batched=True)).numpy()
    assert rel_err(dot, np.einsum("bi,bi->b", D, Y)) < RTOL
    s = rng.uniform(0.5, 2, size=(B,)).astype(np.float32)
    sv = T.gmul(0, 0, 1, T.put(s, batched=True), dD).numpy()
    assert rel_err(sv, s[:, None] * D) < RTOL
    # batched x batched matrices
    P = rng.uniform(-1, 1, size=(B, 20, 33)).astype(np.float32)
    Q = rng.uniform(-1, 1, size=(B, 33, 40)).astype(np.float32)
    pq = T.gmul(1, 1, 1, T.put(P, batched=True), T.put(Q, batched=True)).numpy()
    assert rel_err(pq, np.einsum("bik,bkj->bij", P, Q)) < RTOL
    red = T.gmul_batch_sum(1, 1, 1, T.put(P, batched=True), T.put(Q, batched=True)).numpy()
    assert rel_err(red, np.einsum("bik,bkj->ij", P.astype(np.float64), Q.astype(np.float64))) < RTOL


def test_batched_lift_sum_rows_and_batch_sum(T):
    rng = np.random.default_rng(SEED)
    B = 19
    X = rng.uniform(-1, 1, size=(B, 12)).astype(np.float32)
    b = rng.uniform(-1, 1, size=(12,)).astype(np.float32)
    dX, db = T.put(X, batched=True), T.put(b)
    z = T.sumT([dX, db], (12,))
    assert z.batch == B
    assert np.allclose(z.numpy(), X + b, rtol=0, atol=1e-7)
    z2 = T.sumT([db, dX], (12,))
    assert np.allclose(z2.numpy(), X + b, rtol=0, atol=1e-7)
    assert np.allclose(T.sumRows(dX).numpy(), X.sum(axis=1), atol=1e-5)
    assert np.allclose(T.batch_sum(dX).numpy(), X.sum(axis=0), atol=1e-5)
    s = T.put(rng.uniform(-1, 1, size=(B,)).astype(np.float32), batched=True)
    bc = T.mapRows_const(1, s, dX)
    assert np.array_equal(bc.numpy(), np.broadcast_to(s.numpy()[:, None], (B, 12)))
    assert np.array_equal(T.batch_select(dX, 5).numpy(), X[5])
    assert np.array_equal(T.batch_bcast(db, 4).numpy(), np.broadcast_to(b, (4, 12)))


# ---- class BLAS ---------------------------------------------------------------------------------------
def test_blas_class_entry_points(T):
    from tensor_ops_amd import capi
    from tensor_ops_amd.hipt import DT
    L = capi.lib()
    rng = np.random.default_rng(SEED)
    A, Bm, Cm = ints(rng, 6, 5), ints(rng, 5, 7), ints(rng, 6, 7)
    x, y, y6 = ints(rng, 5), ints(rng, 5), ints(rng, 6)
    dA, dB, dC, dx, dy, dy6 = (T.put(v) for v in (A, Bm, Cm, x, y, y6))

    def call(fn, *args):
        h = capi.c_tensor()
        capi.check(fn(*args, C.byref(h)))
        return DT(h).numpy()

    assert np.array_equal(call(L.to_blas_axpy, 2.0, dx.h, dy.h), 2 * x + y)          # BLAS.hs:97-101
    assert np.array_equal(call(L.to_blas_axpy, -1.0, dx.h, None), -x)
    v = C.c_double()
    capi.check(L.to_blas_dot(dx.h, dy.h, C.byref(v)))
    assert v.value == float(x @ y)                                                   # :102-104
    assert np.array_equal(call(L.to_blas_ger, dy6.h, dx.h), np.outer(y6, x))        # :108-110
    assert np.array_equal(call(L.to_blas_gemv, 2.0, dA.h, dx.h, 0.0, None), 2 * (A @ x))
    assert np.array_equal(call(L.to_blas_gemv, 2.0, dA.h, dx.h, 3.0, dy6.h), 2 * (A @ x) + 3 * y6)
    assert np.array_equal(call(L.to_blas_gemm, 1.0, dA.h, dB.h, 0.0, None), A @ Bm)
    assert np.array_equal(call(L.to_blas_gemm, -1.0, dA.h, dB.h, 2.0, dC.h), -(A @ Bm) + 2 * Cm)
    assert np.array_equal(call(L.to_blas_scale, 4.0, dA.h), 4 * A)
    assert np.array_equal(call(L.to_blas_add, dA.h, dA.h), 2 * A)
    assert np.array_equal(call(L.to_blas_index_row, 3, dA.h), A[3])
    assert np.array_equal(call(L.to_blas_transp, dA.h), A.T)
    assert np.array_equal(call(L.to_blas_eye, 0, 4), np.eye(4, dtype=np.float32))
    sq = ints(rng, 5, 5)
    dsq = T.put(sq)  # keep the handle alive across the raw C calls
    capi.check(L.to_blas_trace(dsq.h, C.byref(v)))
    assert v.value == float(np.trace(sq))
    assert np.array_equal(call(L.to_blas_diag, dx.h), np.diag(x))
    assert np.array_equal(call(L.to_blas_get_diag, dsq.h), np.diag(sq))
    capi.check(L.to_blas_sum(dA.h, C.byref(v)))
    assert v.value == float(A.sum())
    # BTensor's "matrix add through gemm with eye" (BTensor.hs:113) gives the same values
    eye = capi.c_tensor()
    capi.check(L.to_blas_eye(0, 7, C.byref(eye)))
    got = call(L.to_blas_gemm, 1.0, dC.h, eye, 1.0, dC.h)
    L.to_release(eye)
    assert np.array_equal(got, 2 * Cm)


def test_genRand_is_counter_based_and_matches_host_restatement(T):
    def splitmix64(z):
        z = (z + 0x9e3779b97f4a7c15) & (2**64 - 1)
        z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & (2**64 - 1)
        return z ^ (z >> 31)
    seed = SEED
    u = T.genRand((1000,), "uniform", -1.0, 1.0, seed).numpy()
    want = np.array([np.float32(-1.0) + np.float32(2.0) * (np.float32(
        splitmix64((seed + 0x9e3779b97f4a7c15 * i) & (2**64 - 1)) >> 40) * np.float32(1 / 16777216.0))
        for i in range(1000)], dtype=np.float32)
    assert np.array_equal(u, want)
    n = T.genRand((200000,), "normal", 0.0, 0.5, seed + 1).numpy()   # FeedForward.hs:206
    assert abs(n.mean()) < 5e-3 and abs(n.std() - 0.5) < 5e-3


def test_memo_scope_is_cse(T):
    rng = np.random.default_rng(SEED)
    W, x = T.put(ints(rng, 8, 6)), T.put(ints(rng, 6))
    st = T.stats()
    with T.memo():
        a = T.gmul(1, 1, 0, W, x)
        b = T.gmul(1, 1, 0, W, x)
        c = T.scaleT(2.0, a)
        d = T.force(T.scaleT(2.0, b))
    # CSE: b IS a and d IS c.  The scope is also a fusion scope: what was launched is c alone, with the
    # scale folded into the GEMM's alpha; a (consumed by c, never asked for so far) gets no storage of its own.
    assert T.stats()["launches"] - st["launches"] == 1
    assert a.h.value == b.h.value and c.h.value == d.h.value
    assert np.array_equal(c.numpy(), d.numpy())
    assert T.stats()["launches"] - st["launches"] == 1
    assert np.array_equal(a.numpy(), b.numpy())          # asked for now: computed on demand
    assert T.stats()["launches"] - st["launches"] == 2
    assert np.array_equal(c.numpy(), 2.0 * a.numpy())
    e = T.gmul(1, 1, 0, W, x)  # outside the scope: computed again
    assert np.array_equal(e.numpy(), a.numpy())


def test_graph_capture_replay(T):
    from tensor_ops_amd.hipt import Graph
    rng = np.random.default_rng(SEED)
    Wn = rng.uniform(-1, 1, size=(64, 32)).astype(np.float32)
    X = T.put(np.zeros((16, 32), np.float32), batched=True)
    W = T.put(Wn)
    T.gmul(1, 1, 0, W, X)  # warm the pool
    with Graph() as g:
        z = T.gmul(1, 1, 0, W, X)
        h = T.liftT(lambda v: v[0] * 2.0 + 1.0, [z], key="aff")
    from tensor_ops_amd import capi
    for trial in range(3):
        xn = rng.uniform(-1, 1, size=(16, 32)).astype(np.float32)
        capi.check(capi.lib().to_upload(X.h, xn.ctypes.data_as(C.c_void_p), xn.nbytes))
        g.launch()
        assert rel_err(h.numpy(), 2 * (xn.astype(np.float64) @ Wn.T) + 1) < RTOL


def test_no_handle_leaks(T):
    import gc
    gc.collect()
    before = T.stats()["live_handles"]
    rng = np.random.default_rng(SEED)
    for _ in range(3):
        a, b = T.put(ints(rng, 10, 10)), T.put(ints(rng, 10, 10))
        c = T.gmul(1, 1, 1, a, T.transp(b))
        del a, b, c
    gc.collect()
    assert T.stats()["live_handles"] == before


def test_argMax_oneHot_and_batched_inference(T, O):
    """validation = runNetwork + argMax per sample (app/MNIST.hs:368-389)."""
    rng = np.random.default_rng(SEED)
    v = rng.uniform(-1, 1, size=37).astype(np.float32)
    assert T.arg_max(T.put(v)) == O.arg_max(v) == int(np.argmax(v))
    ties = np.array([1, 3, 3, 2, 3], dtype=np.float32)      # earliest maximum wins (Max/Arg fold)
    assert T.arg_max(T.put(ties)) == O.arg_max(ties) == 1
    X = rng.uniform(-1, 1, size=(300, 10)).astype(np.float32)
    X[5, 2] = X[5, 7] = 9.0
    X[6, :] = 0.0
    got = T.arg_max(T.put(X, batched=True))
    assert np.array_equal(got, [O.arg_max(r) for r in X])
    wide = rng.uniform(-1, 1, size=(7, 1000)).astype(np.float32)
    wide[3, 100] = wide[3, 900] = 5.0
    assert np.array_equal(T.arg_max(T.put(wide, batched=True)), [O.arg_max(r) for r in wide])
    assert np.array_equal(T.arg_max(T.transp(T.put(v))), O.arg_max(v))
    assert np.array_equal(T.one_hot(10, 1.0, 0.0, 3).numpy(), O.one_hot(10, 1.0, 0.0, 3))
    idx = rng.integers(0, 10, size=50)
    oh = T.one_hot(10, 1.0, 0.0, list(idx))
    assert oh.batch == 50 and np.array_equal(T.arg_max(oh), idx)
    from tensor_ops_amd.capi import TensorOpsError
    with pytest.raises(TensorOpsError):
        T.one_hot(10, 1.0, 0.0, 10)


def test_bytecode_vm_fallback(repo_root):
    """With the run-time specialisation disabled the same closures run on the LDS-slot VM."""
    import subprocess
    import sys
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from oracle import ad\n"
        "from tensor_ops_amd.hipt import HipT\n"
        "T = HipT(0)\n"
        "f = lambda v: ad.sin(v[0]) * v[1] + ad.exp(-v[0] * v[0])\n"
        "e = T.expr(f, 2, key='vm')\n"
        "assert e.kind == 0, e.kind\n"
        "rng = np.random.default_rng(1)\n"
        "for shape, b in (((1000,), 0), ((16, 12), 0), ((33,), 7)):\n"
        "    full = ((b,) if b else ()) + shape\n"
        "    x, y = (rng.uniform(-2, 2, size=full).astype(np.float32) for _ in range(2))\n"
        "    got = T.liftT(e, [T.put(x, batched=bool(b)), T.put(y, batched=bool(b))]).numpy()\n"
        "    want = np.sin(x.astype(np.float64)) * y + np.exp(-x.astype(np.float64) ** 2)\n"
        "    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-5\n"
        "print('vm ok')\n") % repo_root
    import os
    env = dict(os.environ, TOPS_EXPR_JIT="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "vm ok" in out.stdout, out.stdout + out.stderr


def test_calls_from_many_threads_are_serialised_correctly(T, O):
    """The class methods may be called from any OS thread in demand order (a `-threaded -N` Haskell RTS,
    tensor-ops.cabal:72; SURVEY.md 8(b) "Threading"): concurrent callers must each get their own correct
    results (one global lock, one stream, pure stream-ordered ops)."""
    import threading
    rng = np.random.default_rng(SEED)
    errs = []

    def worker(k):
        try:
            r = np.random.default_rng(SEED + k)
            for it in range(40):
                a = r.integers(-4, 5, size=(17 + k, 9)).astype(np.float32)
                b = r.integers(-4, 5, size=(9, 5 + it % 3)).astype(np.float32)
                da, db = T.put(a), T.put(b)
                got = T.gmul(1, 1, 1, da, db).numpy()
                if not np.array_equal(got, a @ b):
                    errs.append(("gmul", k, it))
                s = T.sumT([da, da, da], a.shape).numpy()
                if not np.array_equal(s, 3 * a):
                    errs.append(("sumT", k, it))
                t = T.transp(T.scaleT(2.0, da)).numpy()
                if not np.array_equal(t, 2 * a.T):
                    errs.append(("transp", k, it))
                if float(T.sumRows(T.put(a.ravel())).numpy()) != float(a.sum()):
                    errs.append(("sumRows", k, it))
        except Exception as e:  # noqa: BLE001
            errs.append(("exception", k, repr(e)))

    before = T.stats()["live_handles"]
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs[:5]
    del rng
    import gc
    gc.collect()
    assert T.stats()["live_handles"] == before


def test_c_abi_collective_world_of_one(T):
    """to_comm_* (the exchange step of SURVEY.md 8(e) on the C ABI): unique id -> communicator -> in-place
    all-reduce(sum) on the library stream.  A world of one rank is what a 1-GPU box can exercise: the sum is
    the identity, for both element types, and the error paths are loud."""
    from tensor_ops_amd import capi
    from tensor_ops_amd.hipt import HipT
    L = capi.lib()
    w = C.c_int(-1)
    capi.check(L.to_comm_world(C.byref(w)))
    assert w.value == 0
    x = T.put(np.arange(1000, dtype=np.float32))
    assert L.to_comm_allreduce_sum(x.h) != 0 and b"to_comm_init" in L.to_last_error()
    uid = (C.c_char * 128)()
    capi.check(L.to_comm_unique_id(uid))
    assert any(b != b"\x00" for b in uid.raw)
    assert L.to_comm_init(1, 1, uid) != 0                      # rank out of range
    capi.check(L.to_comm_init(0, 1, uid))
    capi.check(L.to_comm_world(C.byref(w)))
    assert w.value == 1
    assert L.to_comm_init(0, 1, uid) != 0                      # second communicator refused
    for dt in (np.float32, np.float64):
        Td = HipT(0, dtype=dt)
        v = np.random.default_rng(SEED).integers(-9, 10, 203532).astype(dt)
        d = Td.put(v)
        capi.check(L.to_comm_allreduce_sum(d.h))
        assert np.array_equal(d.numpy(), v)
    capi.check(L.to_comm_shutdown())
    capi.check(L.to_comm_world(C.byref(w)))
    assert w.value == 0


def test_argMin(T, O):
    """`TT.argMin` (Tensor.hs:307-321): per sample, ties -> earliest index, strided views."""
    rng = np.random.default_rng(SEED + 5)
    X = rng.uniform(-1, 1, (41, 13)).astype(np.float32)
    assert list(T.arg_min(T.put(X, batched=True))) == [O.arg_min(r) for r in X] == list(np.argmin(X, axis=1))
    assert list(T.arg_max(T.put(X, batched=True))) == [O.arg_max(r) for r in X]
    tie = np.array([[3.0, 1.0, 1.0, 2.0], [5.0, 5.0, 5.0, 5.0], [0.0, -1.0, 7.0, -1.0]], dtype=np.float32)
    assert list(T.arg_min(T.put(tie, batched=True))) == [1, 0, 1] == [O.arg_min(r) for r in tie]
    v = rng.uniform(-1, 1, 1000).astype(np.float32)
    assert T.arg_min(T.put(v)) == int(np.argmin(v)) == O.arg_min(v)
    col = T.slice(T.transp(T.put(X)), (4,))          # row 4 of the transpose: a strided vector view
    assert T.arg_min(col) == int(np.argmin(X[:, 4]))


def test_tensor_hs_helpers(T):
    """TT.inner/outer/outerV/dot/matVec/vecMat/matMat, toList/unScalar/toRows/rows (Tensor.hs:132-273)."""
    rng = np.random.default_rng(SEED + 6)
    A, Bm, x, y = ints(rng, 6, 5), ints(rng, 5, 7), ints(rng, 5), ints(rng, 6)
    dA, dB, dx, dy = (T.put(v) for v in (A, Bm, x, y))
    assert np.array_equal(T.matVec(dA, dx).numpy(), A @ x)
    assert np.array_equal(T.vecMat(dy, dA).numpy(), y @ A)
    assert np.array_equal(T.matMat(dA, dB).numpy(), A @ Bm)
    assert T.unScalar(T.dot(dx, dx)) == float(x @ x)
    assert np.array_equal(T.outerV(dy, dx).numpy(), np.outer(y, x))
    assert np.array_equal(T.outer(2, 1, dA, dx).numpy(), np.einsum("ij,k->ijk", A, x))
    assert np.array_equal(T.inner(1, 1, dA, dB).numpy(), A @ Bm)
    assert T.toList(dA) == [float(v) for v in A.ravel()]
    rows = T.toRows(dA)
    assert len(rows) == 6 and np.array_equal(rows[4].numpy(), A[4])
    assert np.array_equal(T.rows(list(reversed(rows))).numpy(), A[::-1])
