"""GPU parity on the edges of the index space: zero extents (a type-level `Nat` dim may be 0), extents of
one, a contraction over nothing, ragged sizes that are no multiple of any tile, and views at odd offsets.
Oracle: the nested-vector restatement (`oracle/nested.py`, Nested.hs:451-473) / numpy on the same inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import nested  # noqa: E402

RTOL = 1e-5
RNG = np.random.default_rng(0x7e500011)


@pytest.fixture(scope="module", params=["f32", "f64"])
def T(request):
    from tensor_ops_amd.hipt import HipT
    from tensor_ops_amd import tops
    dt = np.float32 if request.param == "f32" else np.float64
    return HipT(0, dtype=dt)


def rel_err(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    den = np.linalg.norm(want.ravel())
    return np.linalg.norm((got - want).ravel()) / (den if den > 0 else 1.0)


ZERO_CASES = [
    ((0,), (3,), (4,)), ((3,), (4,), (0,)), ((3,), (0,), (4,)), ((0,), (0,), (0,)), ((2, 0), (3,), (2,)),
    ((2,), (3, 0), (2,)), ((), (0,), ()), ((0,), (), (5,)), ((4,), (), (0,)), ((300,), (0,), (70,)),
]


@pytest.mark.parametrize("ms,os_,ns", ZERO_CASES)
def test_gmul_zero_extents(T, ms, os_, ns):
    """An empty output stays empty; contracting over an empty index set gives zeros (the empty sum)."""
    a = RNG.integers(-3, 4, size=ms + os_).astype(T.dtype)
    b = RNG.integers(-3, 4, size=tuple(reversed(os_)) + ns).astype(T.dtype)
    got = T.gmul(len(ms), len(os_), len(ns), T.put(a), T.put(b)).numpy()
    assert got.shape == ms + ns
    assert np.array_equal(got, np.zeros(ms + ns, dtype=T.dtype))


@pytest.mark.parametrize("shape", [(0,), (0, 5), (3, 0, 2), ()])
def test_elementwise_and_sums_on_empty_and_scalar(T, shape):
    from tensor_ops_amd.hipt import logistic_closure
    x = RNG.uniform(-1, 1, size=shape).astype(T.dtype)
    dx = T.put(x)
    got = T.liftT(logistic_closure, [dx]).numpy()
    assert got.shape == shape and rel_err(got, 1 / (1 + np.exp(-x.astype(np.float64)))) < RTOL
    assert np.array_equal(T.scaleT(2.0, dx).numpy(), 2 * x)
    assert np.array_equal(T.sumT([dx, dx, dx], shape).numpy(), (x + x + x).astype(T.dtype))
    assert np.array_equal(T.transp(dx).numpy(), np.transpose(x))
    if len(shape) >= 1:
        assert rel_err(T.sumRows(dx).numpy(), x.astype(np.float64).sum(axis=0)) < RTOL


@pytest.mark.parametrize("ms,os_,ns", [
    ((1,), (1,), (1,)), ((1, 1), (1,), (1, 1)), ((1,), (777,), (1,)), ((1,), (3,), (1025,)), ((1023,), (1,), (1,)),
    ((257,), (129,), (33,)), ((17,), (1031,), (19,)), ((2, 3, 5), (7,), (11, 1)), ((31, 1), (1, 9), (1, 31)),
    ((513,), (255,), (511,)), ((1,), (4097,), (3,)),
])
def test_gmul_ragged_sizes(T, ms, os_, ns):
    """No extent is a multiple of a tile, a chunk or a wave."""
    a = RNG.uniform(-1, 1, size=ms + os_).astype(T.dtype)
    b = RNG.uniform(-1, 1, size=tuple(reversed(os_)) + ns).astype(T.dtype)
    K = int(np.prod(os_, dtype=np.int64))
    want = (a.astype(np.float64).reshape(-1, K) @
            np.transpose(b.astype(np.float64), tuple(reversed(range(len(os_)))) +
                         tuple(range(len(os_), len(os_) + len(ns)))).reshape(K, -1)).reshape(ms + ns)
    got = T.gmul(len(ms), len(os_), len(ns), T.put(a), T.put(b)).numpy()
    assert rel_err(got, want) < (RTOL if T.dtype == np.float32 else 1e-12)


def test_gmul_ragged_matches_nested_definition(T):
    """the same through the authoritative nested definition (reversed `os` order) on small-integer data"""
    for ms, os_, ns in [((3,), (2, 5), (4,)), ((2, 2), (3, 2), (2,)), ((1,), (2, 3, 2), (1,))]:
        a = RNG.integers(-4, 5, size=ms + os_).astype(T.dtype)
        b = RNG.integers(-4, 5, size=tuple(reversed(os_)) + ns).astype(T.dtype)
        want = nested.gmul(len(ms), len(os_), len(ns), a.astype(np.float64), b.astype(np.float64))
        got = T.gmul(len(ms), len(os_), len(ns), T.put(a), T.put(b)).numpy()
        assert np.array_equal(got, want.astype(T.dtype))


@pytest.mark.parametrize("off", [1, 2, 3, 5])
def test_rows_at_odd_offsets(T, off):
    """Row views of a batched tensor start at addresses that are not 16-byte aligned: the vector-load paths
    must fall back (batch_slice -> matVec / map on the view)."""
    n, k = 37, 10   # 10 elements per row: 40-byte (f32) row pitch
    X = RNG.uniform(-1, 1, size=(n, k)).astype(T.dtype)
    W = RNG.uniform(-1, 1, size=(6, k)).astype(T.dtype)
    dX = T.put(X, batched=True)
    v = T.batch_slice(dX, off, 9)
    got = T.gmul(1, 1, 0, T.put(W), v).numpy()          # batched matVec on the view
    want = X[off:off + 9].astype(np.float64) @ W.astype(np.float64).T
    assert rel_err(got, want) < (RTOL if T.dtype == np.float32 else 1e-12)
    assert np.array_equal(T.scaleT(-1.0, v).numpy(), -X[off:off + 9])


def test_prefused_step_equals_generic_step_on_random_stacks():
    """Property sweep: on random `genNet` stacks (depth 1..4, widths across every kernel-selection boundary:
    16/17 outputs for the fused loss head, 256/257 for the fused tail, batch 1 = the outer-product path),
    two `Trainer.step()`s on the pre-fused kernels (update in the weight-gradient epilogues, paired launches,
    in-place parameters) leave the same parameters as two steps of the generic TOp composition, whose every
    primitive is tested against the oracle.  Both element types."""
    from tensor_ops_amd import tops
    from tensor_ops_amd.hipt import HipT
    rng = np.random.default_rng(0x7e500012)
    widths = [1, 2, 3, 10, 15, 16, 17, 31, 33, 64, 100, 255, 256, 257, 300, 784]
    for dt, tol in ((np.float32, 1e-5), (np.float64, 1e-11)):
        tops.hlib()
        tops.set_elem_dtype(dt)
        try:
            T = HipT(0, dtype=dt)
            for case in range(28):
                depth = int(rng.integers(1, 5))
                sizes = [int(rng.choice(widths)) for _ in range(depth + 1)]
                B = int(rng.choice([1, 2, 7, 64, 200, 1024, 1100]))
                head = ("actSoftmax", "crossEntropy") if rng.random() < 0.6 else ("actLogistic", "squaredError")
                ws = [(rng.normal(0, 0.5, size=(o, i)) / np.sqrt(i), rng.normal(0, 0.5, size=o))
                      for i, o in zip(sizes, sizes[1:])]
                X = rng.uniform(0, 1, size=(B, sizes[0]))
                Y = np.zeros((B, sizes[-1]))
                Y[np.arange(B), rng.integers(0, sizes[-1], size=B)] = 1.0
                dX, dY = T.put(X, batched=True), T.put(Y, batched=True)
                res = []
                for fused in (True, False):
                    net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", head[0])
                    tr = tops.Trainer(net, head[1], 0.3 / B, dX, dY, use_graph=False, use_fused=fused)
                    assert tr.fused == fused
                    tr.step()
                    tr.step()
                    res.append([p.numpy().astype(np.float64) for p in tr.net.params])
                    del tr
                for a, b in zip(*res):
                    assert np.all(np.isfinite(a)) and rel_err(a, b) < tol, (case, sizes, B, head)
        finally:
            tops.set_elem_dtype(np.float32)
