"""Pin the oracle's `gradTOp'` closures: every op's gradient is checked against
central finite differences of its own `runTOp` (fp64), and the ffLayer-stack
gradient against the closed-form backprop of SURVEY.md section 3.2."""
import numpy as np
import pytest

from oracle import ad, neuralnet as NN, top as TO
from oracle.tensor import OTensor

T = OTensor(np.float64)
RNG = np.random.default_rng(0x7e500001)


def scalar_of(op, xs, w):
    """sum_k <w_k, out_k> -- makes any TOp scalar-valued for finite differences."""
    ys = TO.runTOp(op, T, xs)
    return sum(float(np.sum(np.asarray(y) * wk)) for y, wk in zip(ys, w))


def fd_check(op, xs, rtol=1e-6, eps=1e-6):
    ys = TO.runTOp(op, T, xs)
    w = [RNG.standard_normal(np.shape(y)) for y in ys]
    g = op.grad(T, xs, [np.asarray(wk, dtype=np.float64) for wk in w])
    assert len(g) == len(xs)
    for k, x in enumerate(xs):
        x = np.asarray(x, dtype=np.float64)
        num = np.zeros_like(x)
        it = np.nditer(x, flags=["multi_index"])
        for _ in it:
            i = it.multi_index
            xp, xm = x.copy(), x.copy()
            xp[i] += eps
            xm[i] -= eps
            num[i] = (scalar_of(op, xs[:k] + [xp] + xs[k + 1:], w) -
                      scalar_of(op, xs[:k] + [xm] + xs[k + 1:], w)) / (2 * eps)
        assert np.shape(g[k]) == x.shape
        np.testing.assert_allclose(np.asarray(g[k]), num, rtol=rtol, atol=1e-7)


def rnd(*shape):
    return RNG.uniform(-1, 1, size=shape)


@pytest.mark.parametrize("ms,os_,ns", [
    ((3,), (4,), (2,)), ((2, 3), (4,), (5,)), ((2,), (3, 4), (2,)),
    ((2, 3), (2, 3), ()), ((2,), (), (3,)), ((), (4,), ()), ((), (), (3,)),
])
def test_gmul_grad(ms, os_, ns):
    op = TO.gmul(len(ms), len(os_), len(ns))
    fd_check(op, [rnd(*(ms + os_)), rnd(*(tuple(reversed(os_)) + ns))])


def test_named_contractions_grad():
    fd_check(TO.dot(), [rnd(5), rnd(5)])
    fd_check(TO.matVec(), [rnd(3, 4), rnd(4)])
    fd_check(TO.vecMat(), [rnd(3), rnd(3, 4)])
    fd_check(TO.matMat(), [rnd(3, 4), rnd(4, 2)])
    fd_check(TO.outer(1, 1), [rnd(3), rnd(4)])


def test_lift_ops_grad():
    fd_check(TO.map_(NN.logistic), [rnd(2, 3)])
    fd_check(TO.map_(NN.logistic, NN.logistic_prime), [rnd(5)])
    fd_check(TO.map_(ad.exp), [rnd(4)])
    fd_check(TO.map_(ad.log), [RNG.uniform(0.5, 2, size=4)])
    fd_check(TO.map_(ad.recip), [RNG.uniform(0.5, 2, size=4)])
    fd_check(TO.zip_(lambda x, y: x * y + ad.sin(x)), [rnd(3), rnd(3)])
    fd_check(TO.zip3(lambda x, y, z: x * y / (2 + z)), [rnd(3), rnd(3), rnd(3)])
    fd_check(TO.zipN(4, lambda v: v[0] * v[1] - v[2] * ad.tanh(v[3])), [rnd(2, 2) for _ in range(4)])


def test_structural_ops_grad():
    fd_check(TO.add(), [rnd(3), rnd(3)])
    fd_check(TO.add3(), [rnd(3), rnd(3), rnd(3)])
    fd_check(TO.duplicate(), [rnd(3)])
    fd_check(TO.replicate(3), [rnd(2, 2)])
    fd_check(TO.swap(), [rnd(3), rnd(2)])
    fd_check(TO.scale(2.5), [rnd(3)])
    fd_check(TO.negate(), [rnd(3)])
    fd_check(TO.transpOp(), [rnd(2, 3, 4)])
    fd_check(TO.sumRows(), [rnd(4, 3)])
    fd_check(TO.sumRows(), [rnd(4)])
    fd_check(TO.sumOp(3, (2,)), [rnd(2), rnd(2), rnd(2)])
    fd_check(TO.shuffle([1, 1, 0], [(2,), (3,)]), [rnd(2), rnd(3)])
    fd_check(TO.drop(1, [(2,), (3,)]), [rnd(2), rnd(3)])
    fd_check(TO.take(1, [(2,), (3,)]), [rnd(2), rnd(3)])


def test_combinators_grad():
    fd_check(TO.first(TO.matVec(), 1), [rnd(3, 4), rnd(4), rnd(2)])
    fd_check(TO.secondOp(1, TO.matVec()), [rnd(2), rnd(3, 4), rnd(4)])
    fd_check(TO.par(TO.dot(), TO.map_(ad.exp)), [rnd(3), rnd(3), rnd(2)])
    fd_check(TO.fanout(TO.map_(ad.exp), TO.scale(3.0), [(3,)]), [rnd(3)])
    fd_check(TO.then_first(TO.matVec(), TO.add()), [rnd(3, 4), rnd(4), rnd(3)])
    fd_check(TO.idOp(2), [rnd(2), rnd(3)])


def test_losses_and_softmax_grad():
    fd_check(NN.softmax(), [rnd(5)])
    fd_check(NN.squaredError(), [rnd(4), rnd(4)])
    fd_check(NN.crossEntropy(), [RNG.uniform(0.1, 0.9, size=4), rnd(4)])
    sm = TO.runTOp(NN.softmax(), T, [np.array([1.0, 2.0, 3.0])])[0]
    e = np.exp([1.0, 2.0, 3.0])
    np.testing.assert_allclose(sm, e / e.sum(), rtol=1e-15)


def _closed_form_logistic_se(x, w1, b1, w2, b2, y):
    """SURVEY.md section 3.2 backward sequence (config C1: logistic/logistic/squaredError)."""
    sig = lambda z: 1 / (1 + np.exp(-z))
    z1 = w1 @ x + b1
    h = sig(z1)
    z2 = w2 @ h + b2
    yh = sig(z2)
    e = y - yh
    dyh = -2 * e
    dz2 = dyh * yh * (1 - yh)
    dh = w2.T @ dz2
    dz1 = dh * h * (1 - h)
    return float(e @ e), [w1.T @ dz1, np.outer(dz1, x), dz1, np.outer(dz2, h), dz2]


def test_c1_dots_network_matches_closed_form():
    """BASELINE config 1: 2 -> 16 -> 1, actLogistic, squaredError, rate 1 (app/Dots.hs:60-92)."""
    x, y = rnd(2), np.array([1.0])
    w1, b1, w2, b2 = 0.5 * RNG.standard_normal((16, 2)), 0.5 * RNG.standard_normal(16), \
        0.5 * RNG.standard_normal((1, 16)), 0.5 * RNG.standard_normal(1)
    net = NN.genNet([(w1, b1), (w2, b2)], NN.actLogistic, NN.actLogistic)
    g = NN.netGrad(T, NN.squaredError(), x, y, net)
    loss, ref = _closed_form_logistic_se(x, w1, b1, w2, b2, y)
    assert len(g) == 5
    for a, b in zip(g, ref):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-15)
    op = TO.then_first(net.op, NN.squaredError())
    assert abs(float(TO.runTOp(op, T, [x] + net.params + [y])[0]) - loss) < 1e-14
    # one SGD step (FeedForward.hs:131-148)
    net2 = NN.trainNetwork(T, NN.squaredError(), 1.0, x, y, net)
    for p, p2, gr in zip(net.params, net2.params, ref[1:]):
        np.testing.assert_allclose(p2, p - 1.0 * gr, rtol=1e-12, atol=1e-15)
    # whole-network finite differences too
    fd_check(op, [x] + net.params + [y], rtol=1e-5)


def test_c3_style_network_softmax_crossentropy():
    """MNIST-style stack (app/MNIST.hs:264-265,396): actMap logistic (AD-derived
    derivative), actSoftmax, crossEntropy -- small dims."""
    i, h, o = 7, 5, 3
    x = RNG.uniform(0, 1, size=i)
    y = np.zeros(o)
    y[1] = 1.0
    ws = [(0.5 * RNG.standard_normal((h, i)), 0.5 * RNG.standard_normal(h)),
          (0.5 * RNG.standard_normal((o, h)), 0.5 * RNG.standard_normal(o))]
    net = NN.genNet(ws, lambda: NN.actMap(NN.logistic), NN.actSoftmax)
    op = TO.then_first(net.op, NN.crossEntropy())
    fd_check(op, [x] + net.params + [y], rtol=1e-5)
    # closed form: dz2 = softmax(z2) - y  (for one-hot y)
    g = NN.networkGradient(T, NN.crossEntropy(), x, y, net)
    sig = lambda z: 1 / (1 + np.exp(-z))
    hcur = sig(ws[0][0] @ x + ws[0][1])
    z2 = ws[1][0] @ hcur + ws[1][1]
    p = np.exp(z2) / np.exp(z2).sum()
    np.testing.assert_allclose(g[3], p - y, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(g[2], np.outer(p - y, hcur), rtol=1e-10, atol=1e-13)


def test_recompute_count_of_layer1():
    """Types.hs:155 recomputes f1 xs in every composition node: the layer-1
    matVec runs 3x per gradTOp."""
    class Counting(OTensor):
        calls = 0

        def gmul(self, lm, lo, ln, x, y):
            if (lm, lo, ln) == (1, 1, 0) and np.shape(x) == (4, 3):
                Counting.calls += 1
            return super().gmul(lm, lo, ln, x, y)
    Tc = Counting(np.float64)
    ws = [(rnd(4, 3), rnd(4)), (rnd(2, 4), rnd(2))]
    net = NN.genNet(ws, NN.actLogistic, NN.actLogistic)
    NN.netGrad(Tc, NN.squaredError(), rnd(3), rnd(2), net)
    assert Counting.calls == 3
