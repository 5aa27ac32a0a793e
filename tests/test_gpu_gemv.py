"""csrc/gemv.hip (round 6, last): matVec / vecMat / outer products / tall column sums beyond the small-GEMM kernel's range, found
2 x ... 200 x off torch.mv / torch.outer / torch.sum by tools/ops_scan.py.  The reference's forms: `matVec`, `vecMat`, `outer`
(Types.hs:52-109 via `gmul`; TOp.hs:56-94), `sumRows`.  Exact on small integers in both element types: a row per output and a
column per output, 16 lanes and 64 lanes a row, 16-byte and scalar loads (odd extents), a reduction split over workgroups
(few outputs under a long K), the ragged end of a row, outer products with and without 16-byte stores."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[np.float32, np.float64], ids=["f32", "f64"])
def T(request):
    from tensor_ops_amd.hipt import HipT
    return HipT(0, dtype=request.param), request.param


SHAPES = [(1536, 4096), (100, 60000), (4099, 515), (10000, 300), (257, 8191), (30000, 100), (5000, 2049), (64, 40000), (3, 70000),
          (2048, 2048)]


@pytest.mark.parametrize("m,k", SHAPES)
def test_matvec_and_vecmat_every_layout_exact_on_integers(T, m, k):
    T, dt = T
    rng = np.random.default_rng(m * 7 + k)
    A = rng.integers(-2, 3, (m, k)).astype(dt)
    xk = rng.integers(-2, 3, k).astype(dt)
    ym = rng.integers(-2, 3, m).astype(dt)
    dA, dAt = T.put(A), T.transp(T.put(np.ascontiguousarray(A.T)))   # (the same matrix, k-contiguous and m-contiguous)
    dx, dy = T.put(xk), T.put(ym)
    want_mv, want_vm = A.astype(np.float64) @ xk, ym @ A.astype(np.float64)
    for mat in (dA, dAt):
        l0 = T.stats()["launches"]
        got = T.matVec(mat, dx).numpy()
        assert T.stats()["launches"] - l0 <= 2
        assert np.array_equal(got.astype(np.float64), want_mv)
        assert np.array_equal(T.vecMat(dy, mat).numpy().astype(np.float64), want_vm)
    # through the transposed view as the matrix itself
    assert np.array_equal(T.matVec(T.transp(dA), dy).numpy().astype(np.float64), want_vm)
    assert np.array_equal(T.vecMat(dx, T.transp(dA)).numpy().astype(np.float64), want_mv)


@pytest.mark.parametrize("m,n", [(4096, 784), (1031, 2050), (10000, 300), (300, 10000), (2048, 2048), (60000, 100)])
def test_outer_products_exact(T, m, n):
    T, dt = T
    rng = np.random.default_rng(m + 3 * n)
    a = rng.integers(-3, 4, m).astype(dt); b = rng.integers(-3, 4, n).astype(dt)
    l0 = T.stats()["launches"]
    got = T.outerV(T.put(a), T.put(b)).numpy()
    assert T.stats()["launches"] - l0 == 1
    assert np.array_equal(got, np.outer(a, b))


@pytest.mark.parametrize("m,k", [(60000, 1024), (10000, 10000), (4096, 4100), (20000, 257), (1000000, 256)])
def test_tall_column_sums_exact(T, m, k):
    T, dt = T
    if m * k * np.dtype(dt).itemsize > (1 << 31):
        pytest.skip("kept small")
    rng = np.random.default_rng(m + k)
    A = rng.integers(-2, 3, (m, k)).astype(dt)
    got = T.sumRows(T.put(A)).numpy()
    assert np.array_equal(got.astype(np.float64), A.astype(np.float64).sum(axis=0))


@pytest.mark.parametrize("i,o", [(4096, 4096), (60000, 300), (300, 60000)])
def test_a_wide_layer_on_one_sample_keeps_its_epilogue(T, i, o):
    """`logistic (W x + b)` for ONE sample through the planner: the bias and the activation in gemv.hip's epilogue, at most two
    launches (a split reduction has a finishing pass); values at 2e-6 / 1e-12 of fp64."""
    T, dt = T
    from tensor_ops_amd.hipt import logistic_closure
    rng = np.random.default_rng(i + o)
    W = (rng.integers(-2, 3, (o, i)) / 64).astype(dt); x = rng.integers(-2, 3, i).astype(dt); b = (rng.integers(-3, 4, o) / 8).astype(dt)
    dW, dx, db = T.put(W), T.put(x), T.put(b)
    z = W.astype(np.float64) @ x + b
    l0 = T.stats()["launches"]
    with T.memo():
        out = T.force(T.liftT(logistic_closure, [T.sumT([T.matVec(dW, dx), db], (o,))], key="gemv-logistic"))
    assert T.stats()["launches"] - l0 <= 3
    tol = 2e-6 if dt == np.float32 else 1e-12
    assert np.max(np.abs(out.numpy().astype(np.float64) - 1 / (1 + np.exp(-z)))) < tol


@pytest.mark.parametrize("dt", [np.float32, np.float64], ids=["f32", "f64"])
def test_blas_class_gemv_and_ger_at_size(dt):
    """`class BLAS` (BLAS.hs:108-116): `gemv alpha a x (Just (beta, y))` and `ger x y` beyond the small-GEMM kernel's range --
    alpha and beta * y in gemv.hip's epilogue; exact on small integers."""
    from tensor_ops_amd.hipb import HipB
    B = HipB(0, dtype=dt)
    rng = np.random.default_rng(5)
    for m, k in ((3000, 4100), (100, 50000), (20000, 120)):
        A = rng.integers(-2, 3, (m, k)).astype(dt); x = rng.integers(-2, 3, k).astype(dt); y = rng.integers(-4, 5, m).astype(dt)
        dA, dx, dy = B.T.put(A), B.T.put(x), B.T.put(y)
        got = B.gemv(2.0, dA, dx, (-3.0, dy)).numpy()
        assert np.array_equal(got.astype(np.float64), 2.0 * (A.astype(np.float64) @ x) - 3.0 * y)
        assert np.array_equal(B.gemv(1.0, dA, dx, None).numpy().astype(np.float64), A.astype(np.float64) @ x)
        assert abs(B.dot(dx, dx) - float(x.astype(np.float64) @ x)) == 0.0
    a = rng.integers(-3, 4, 5000).astype(dt); b = rng.integers(-3, 4, 3000).astype(dt)
    assert np.array_equal(B.ger(B.T.put(a), B.T.put(b)).numpy(), np.outer(a, b))


@pytest.mark.parametrize("dt", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("m,k,n", [(4096, 8, 4096), (5000, 3, 3000), (2048, 15, 4100), (1000, 7, 4098), (60000, 8, 256), (300, 12, 20000)])
def test_rank_k_updates_exact(dt, m, k, n):
    """A rank-2 .. 15 update of a large matrix (the weight gradient of a minibatch of a few samples, `gemm alpha a b (Just (beta, c))`
    with K below the MFMA kernels' 16): gemv.hip's outer-product kernel with the rank as a template parameter (8 or 16 rows of b in
    registers, zero beyond K); all four layouts, with beta * C; exact on small integers, one launch."""
    from tensor_ops_amd.hipb import HipB
    B = HipB(0, dtype=dt)
    rng = np.random.default_rng(m + 5 * k + n)
    a = rng.integers(-2, 3, (m, k)).astype(dt); b = rng.integers(-2, 3, (k, n)).astype(dt); c = rng.integers(-5, 6, (m, n)).astype(dt)
    want = a.astype(np.float64) @ b.astype(np.float64)
    for ta in (0, 1):
        for tb in (0, 1):
            da = B.T.transp(B.T.put(np.ascontiguousarray(a.T))) if ta else B.T.put(a)
            db = B.T.transp(B.T.put(np.ascontiguousarray(b.T))) if tb else B.T.put(b)
            l0 = B.T.stats()["launches"]
            got = B.T.gmul(1, 1, 1, da, db).numpy()
            assert B.T.stats()["launches"] - l0 == 1
            assert np.array_equal(got.astype(np.float64), want), (ta, tb)
    got = B.gemm(2.0, B.T.put(a), B.T.put(b), (-3.0, B.T.put(c))).numpy()
    assert np.array_equal(got.astype(np.float64), 2.0 * want - 3.0 * c)
