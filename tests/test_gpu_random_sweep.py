"""Seeded random sweep of the contraction / layout entry points against the oracle: random index groups
(|ms|, |os|, |ns| in 0..3), sizes including 1 and non-multiples of the tile sizes, hidden batches on either or
both operands, operands given as strided `transp` views, both element types.  Integer-valued data, so every
result must be BIT-EXACT (products and sums stay exactly representable)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import nested  # noqa: E402

SEED = 0x7e5000aa


@pytest.fixture(scope="module", params=["f32", "f64"])
def T(request):
    from tensor_ops_amd.hipt import HipT
    return HipT(0, dtype=np.float32 if request.param == "f32" else np.float64)


def _dims(rng, n, big):
    pool = [1, 2, 3, 4, 5, 6] + ([7, 17, 33, 65] if big else [])
    return tuple(int(rng.choice(pool)) for _ in range(n))


def _strided(T, x, rng, batched):
    """x as a device value, half of the time through a transposed view of its transposed copy"""
    if batched or x.ndim < 2 or rng.random() < 0.5:
        return T.put(x, batched=batched)
    return T.transp(T.put(np.array(nested.transpose(x), order="C")))


def test_gmul_random_index_groups(T):
    rng = np.random.default_rng(SEED)
    dt = T.dtype
    for case in range(160):
        lm, lo, ln = (int(v) for v in rng.integers(0, 4, 3))
        big = rng.random() < 0.25 and lm + lo + ln <= 4
        ms, os_, ns = _dims(rng, lm, big), _dims(rng, lo, big), _dims(rng, ln, big)
        a = rng.integers(-3, 4, size=ms + os_).astype(dt)
        b = rng.integers(-3, 4, size=tuple(reversed(os_)) + ns).astype(dt)
        want = nested.gmul(lm, lo, ln, a.astype(np.float64), b.astype(np.float64))
        got = T.gmul(lm, lo, ln, _strided(T, a, rng, False), _strided(T, b, rng, False)).numpy()
        assert got.shape == ms + ns and np.array_equal(got, want), (case, ms, os_, ns)


def test_gmul_random_hidden_batches(T):
    rng = np.random.default_rng(SEED + 1)
    dt = T.dtype
    for case in range(80):
        lm, lo, ln = (int(v) for v in rng.integers(0, 3, 3))
        ms, os_, ns = _dims(rng, lm, False), _dims(rng, lo, False), _dims(rng, ln, False)
        B = int(rng.choice([1, 2, 5, 37]))
        ba, bb = rng.random() < 0.7, rng.random() < 0.7
        if not (ba or bb):
            ba = True
        a = rng.integers(-3, 4, size=((B,) if ba else ()) + ms + os_).astype(dt)
        b = rng.integers(-3, 4, size=((B,) if bb else ()) + tuple(reversed(os_)) + ns).astype(dt)
        want = np.stack([nested.gmul(lm, lo, ln, (a[i] if ba else a).astype(np.float64),
                                     (b[i] if bb else b).astype(np.float64)) for i in range(B)])
        da, db = T.put(a, batched=ba), T.put(b, batched=bb)
        got = T.gmul(lm, lo, ln, da, db)
        assert got.batch == B and np.array_equal(got.numpy(), want), (case, ms, os_, ns, B, ba, bb)
        # the cotangent-of-an-unbatched-operand form: the same contraction summed over the samples
        gs = T.gmul_batch_sum(lm, lo, ln, da, db)
        assert gs.batch == 0 and np.array_equal(gs.numpy(), want.sum(axis=0)), (case, "batch_sum")


def test_layout_and_reduction_random(T):
    rng = np.random.default_rng(SEED + 2)
    dt = T.dtype
    for case in range(60):
        rank = int(rng.integers(1, 5))
        shape = _dims(rng, rank, rng.random() < 0.3 and rank <= 2)
        x = rng.integers(-5, 6, size=shape).astype(dt)
        dx = T.put(x)
        assert np.array_equal(T.transp(dx).numpy(), nested.transpose(x)), (case, shape)
        assert np.array_equal(T.transp(T.transp(dx)).numpy(), x)
        assert np.array_equal(T.sumRows(dx).numpy(), x.sum(axis=0)), (case, shape)
        assert np.array_equal(T.sumRows(T.transp(dx)).numpy(), nested.transpose(x).sum(axis=0))
        k = int(rng.integers(0, 5))
        parts = [rng.integers(-5, 6, size=shape).astype(dt) for _ in range(k)]
        assert np.array_equal(T.sumT([T.put(p) for p in parts], shape).numpy(),
                              sum(parts) if parts else np.zeros(shape, dt))
        idx = tuple(int(rng.integers(0, d)) for d in shape)
        assert T.index(dx, idx) == x[idx]
        assert T.index(T.transp(dx), tuple(reversed(idx))) == x[idx]
        alpha = float(rng.integers(-3, 4))
        assert np.array_equal(T.scaleT(alpha, T.transp(dx)).numpy(), alpha * nested.transpose(x))
        # liftT over mixed views (a strided transp view and a contiguous value of the same logical shape)
        y = rng.integers(-5, 6, size=tuple(reversed(shape))).astype(dt)
        got = T.liftT(lambda v: 2.0 * v[0] - v[1] * v[1], [T.transp(dx), T.put(y)], key="sweep_mix").numpy()
        assert np.array_equal(got, 2.0 * nested.transpose(x) - y * y), (case, shape)
