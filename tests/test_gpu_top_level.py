"""GPU parity at the `TOp` level: the oracle's polymorphic TOp closures
(`forall t. Tensor t => ...`, Types.hs:122-125) are run once with the numpy
backend and once with the HIP backend -- same closures, same inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ad, neuralnet as NN, top as TO  # noqa: E402
from oracle.tensor import OTensor  # noqa: E402

RTOL = 1e-5
RNG = np.random.default_rng(0x7e500001)


@pytest.fixture(scope="module")
def T():
    from tensor_ops_amd.hipt import HipT
    return HipT(0)


O = OTensor(np.float64)


def rel_err(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape
    den = np.linalg.norm(want.ravel())
    return np.linalg.norm((got - want).ravel()) / (den if den > 0 else 1.0)


def both(T, op, xs, grad_seed=None):
    ys_o = TO.runTOp(op, O, xs)
    dxs = [T.put(x) for x in xs]
    ys_t = TO.runTOp(op, T, dxs)
    for a, b in zip(ys_t, ys_o):
        assert rel_err(a.numpy(), b) < RTOL
    ds = [RNG.uniform(-1, 1, size=np.shape(y)) for y in ys_o]
    g_o = op.grad(O, list(xs), ds)
    g_t = op.grad(T, dxs, [T.put(d) for d in ds])
    for a, b in zip(g_t, g_o):
        assert rel_err(a.numpy(), b) < RTOL


def rnd(*s):
    return RNG.uniform(-1, 1, size=s)


def test_op_vocabulary(T):
    both(T, TO.gmul(2, 1, 1), [rnd(2, 3, 4), rnd(4, 5)])
    both(T, TO.gmul(1, 2, 1), [rnd(2, 3, 4), rnd(4, 3, 2)])
    both(T, TO.matVec(), [rnd(9, 7), rnd(7)])
    both(T, TO.vecMat(), [rnd(9), rnd(9, 7)])
    both(T, TO.matMat(), [rnd(9, 7), rnd(7, 5)])
    both(T, TO.dot(), [rnd(11), rnd(11)])
    both(T, TO.outer(1, 1), [rnd(4), rnd(6)])
    both(T, TO.map_(NN.logistic), [rnd(3, 4)])
    both(T, TO.zip_(lambda x, y: x * y + ad.sin(x)), [rnd(6), rnd(6)])
    both(T, TO.zip3(lambda x, y, z: x * y / (2 + z)), [rnd(5), rnd(5), rnd(5)])
    both(T, TO.add(), [rnd(5), rnd(5)])
    both(T, TO.duplicate(), [rnd(5)])
    both(T, TO.replicate(3), [rnd(2, 2)])
    both(T, TO.scale(2.5), [rnd(4)])
    both(T, TO.transpOp(), [rnd(2, 3, 4)])
    both(T, TO.sumRows(), [rnd(6, 3)])
    both(T, TO.sumOp(3, (4,)), [rnd(4), rnd(4), rnd(4)])
    both(T, TO.shuffle([1, 1, 0], [(2,), (3,)]), [rnd(2), rnd(3)])
    both(T, TO.fanout(TO.map_(ad.exp), TO.scale(3.0), [(3,)]), [rnd(3)])
    both(T, NN.softmax(), [rnd(10)])
    both(T, NN.squaredError(), [rnd(4), rnd(4)])
    both(T, NN.crossEntropy(), [RNG.uniform(0.1, 0.9, size=6), rnd(6)])


def _net(sizes, hidden, out):
    ws = [(0.5 * RNG.standard_normal((o, i)), 0.5 * RNG.standard_normal(o))
          for i, o in zip(sizes[:-1], sizes[1:])]
    return ws, NN.genNet(ws, hidden, out)


def test_c1_dots_step_on_gpu_in_fp32(T):
    """BASELINE config 1 network (2 -> 16 -> 1, logistic, squaredError, rate 1),
    one runTOp + gradTOp + SGD step; the reference config is fp64/CPU, the GPU
    runs it in fp32 against the fp64 oracle at 1e-5."""
    ws, net_o = _net([2, 16, 1], NN.actLogistic, NN.actLogistic)
    net_t = NN.Network(net_o.op, [T.put(p) for p in net_o.params])
    x, y = rnd(2), np.array([1.0])
    assert rel_err(NN.runNetwork(T, net_t, T.put(x)).numpy(), NN.runNetwork(O, net_o, x)) < RTOL
    g_o = NN.netGrad(O, NN.squaredError(), x, y, net_o)
    with T.memo():
        g_t = NN.netGrad(T, NN.squaredError(), T.put(x), T.put(y), net_t)
    for a, b in zip(g_t, g_o):
        assert rel_err(a.numpy(), b) < RTOL
    n_o = NN.trainNetwork(O, NN.squaredError(), 1.0, x, y, net_o)
    n_t = NN.trainNetwork(T, NN.squaredError(), 1.0, T.put(x), T.put(y), net_t)
    for a, b in zip(n_t.params, n_o.params):
        assert rel_err(a.numpy(), b) < RTOL


def test_mnist_style_net_unbatched(T):
    ws, net_o = _net([20, 12, 5], lambda: NN.actMap(NN.logistic), NN.actSoftmax)
    net_t = NN.Network(net_o.op, [T.put(p) for p in net_o.params])
    x = RNG.uniform(0, 1, size=20)
    y = np.zeros(5)
    y[3] = 1
    g_o = NN.netGrad(O, NN.crossEntropy(), x, y, net_o)
    g_t = NN.netGrad(T, NN.crossEntropy(), T.put(x), T.put(y), net_t)
    for a, b in zip(g_t, g_o):
        assert rel_err(a.numpy(), b) < RTOL
