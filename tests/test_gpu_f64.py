"""GPU parity of the fp64 instance (`ElemT = Double`, the reference HMat backend's
element type; BASELINE config 1 is quoted on it) against the fp64 oracle.

Bar: bit-exact on integer data and index work; <= 1e-12 relative on random data
(the only differences are summation order and the device libm's last-ulp
rounding of exp/log/tanh).  Same C ABI, handles created with TO_F64.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ad, nested, neuralnet as NN, top as TO  # noqa: E402
from oracle.tensor import OTensor  # noqa: E402

RTOL = 1e-12
SEED = 0x7e500064
RNG = np.random.default_rng(SEED)
O = OTensor(np.float64)


@pytest.fixture(scope="module")
def T():
    from tensor_ops_amd.hipt import HipT
    return HipT(0, dtype=np.float64)


@pytest.fixture(scope="module")
def T32():
    from tensor_ops_amd.hipt import HipT
    return HipT(0)


@pytest.fixture()
def H():
    """host mirror with ElemT = Double for the duration of one test"""
    from tensor_ops_amd import tops
    tops.hlib()
    tops.set_elem_dtype(np.float64)
    yield tops
    tops.set_elem_dtype(np.float32)


def rel_err(got, want):
    got, want = np.asarray(got), np.asarray(want, np.float64)
    assert got.dtype == np.float64, got.dtype
    assert got.shape == want.shape, (got.shape, want.shape)
    den = np.linalg.norm(want.ravel())
    return np.linalg.norm((got - want).ravel()) / (den if den > 0 else 1.0)


def ints(*shape):
    # beyond 2^24: exact in fp64, would round in fp32
    return RNG.integers(-5000, 5001, size=shape).astype(np.float64)


def rnd(*s):
    return RNG.uniform(-1, 1, size=s)


GMUL_CASES = [
    ((3,), (4,), (2,)), ((2, 3), (4,), (5,)), ((2,), (3, 4), (2,)), ((2, 3), (2, 3), ()),
    ((2,), (2, 3, 2), (3, 2)), ((2, 3), (), (4,)), ((), (5,), ()), ((), (), (3,)), ((4,), (), ()),
    ((), (), ()), ((70,), (33,), (65,)), ((130, 3), (17,), (50,)), ((5,), (4, 3, 2), ()),
    ((200,), (300,), (136,)),
]


@pytest.mark.parametrize("ms,os_,ns", GMUL_CASES)
def test_gmul_exact_on_integers(T, ms, os_, ns):
    a = ints(*(ms + os_))
    b = ints(*(tuple(reversed(os_)) + ns))
    want = nested.gmul(len(ms), len(os_), len(ns), a, b)
    got = T.gmul(len(ms), len(os_), len(ns), T.put(a), T.put(b)).numpy()
    assert got.dtype == np.float64 and got.shape == ms + ns
    assert np.array_equal(got, want)


@pytest.mark.parametrize("m,k,n", [(128, 128, 128), (300, 513, 129), (1024, 784, 256), (64, 8, 64),
                                   (1000, 10, 256), (33, 1000, 47), (256, 1024, 784), (2048, 64, 512),
                                   (1, 700, 300), (700, 300, 1), (65, 17, 63)])
def test_gemm_random(T, m, k, n):
    a, b = rnd(m, k), rnd(k, n)
    got = T.gmul(1, 1, 1, T.put(a), T.put(b)).numpy()
    assert rel_err(got, a @ b) < RTOL


def test_gemm_transposes_and_batches(T):
    m, k, n = 200, 136, 168
    a, b = rnd(m, k), rnd(k, n)
    for ta in (False, True):
        for tb in (False, True):
            da = T.transp(T.put(np.array(a.T, order="C"))) if ta else T.put(a)
            db = T.transp(T.put(np.array(b.T, order="C"))) if tb else T.put(b)
            assert rel_err(T.gmul(1, 1, 1, da, db).numpy(), a @ b) < RTOL
    B = 37
    W, X = rnd(70, 50), rnd(B, 50)
    dW, dX = T.put(W), T.put(X, batched=True)
    Z = T.gmul(1, 1, 0, dW, dX).numpy()
    assert rel_err(Z, X @ W.T) < RTOL
    D = rnd(B, 70)
    dD = T.put(D, batched=True)
    gW = T.gmul_batch_sum(1, 0, 1, dD, dX).numpy()          # sum_b outer(d_b, x_b)
    assert rel_err(gW, D.T @ X) < RTOL
    per = T.gmul(1, 0, 1, dD, dX).numpy()
    assert rel_err(per, np.einsum("bi,bj->bij", D, X)) < RTOL
    assert rel_err(T.batch_sum(T.put(per, batched=True)).numpy(), D.T @ X) < RTOL


def _lift_cases():
    return [
        ("logistic", 1, lambda v: NN.logistic(v[0]), False),
        ("exp", 1, lambda v: ad.exp(v[0]), False),
        ("log", 1, lambda v: ad.log(v[0]), True),
        ("recip", 1, lambda v: ad.recip(v[0]), True),
        ("tanh", 1, lambda v: ad.tanh(v[0]), False),
        ("sqrt", 1, lambda v: ad.sqrt(v[0]), True),
        ("affine_sgd", 2, lambda v: v[0] - 0.02 * v[1], False),
        ("affine3", 3, lambda v: 2.0 * v[0] - v[1] + 0.5 * v[2] + 1.0, False),
        ("mul", 2, lambda v: v[0] * v[1], False),
        ("div", 2, lambda v: v[0] / v[1], True),
        ("d_logistic", 2, lambda v: v[0] * NN.logistic_prime(v[1]), False),
        ("d_logistic_ad", 2, lambda v: v[0] * ad.diff(NN.logistic)(v[1]), False),
        ("jit_mixed", 2, lambda v: ad.sin(v[0]) * v[1] + ad.exp(-v[0] * v[0]), False),
        ("jit_poly5", 4, lambda v: v[0] * v[1] - v[2] * ad.tanh(v[3]) + abs(v[0]), False),
        ("jit_8ary", 8, lambda v: (v[0] + v[1] * v[2]) / (2.0 + v[3] * v[3]) - v[4] * v[5] + ad.cos(v[6]) * v[7], False),
    ]


@pytest.mark.parametrize("case", _lift_cases(), ids=lambda c: c[0])
@pytest.mark.parametrize("shape", [(7,), (16, 12), (3, 5, 2), (70001,)])
def test_lift_matches_oracle(T, case, shape):
    name, n, f, positive = case
    lo, hi = (0.25, 2.0) if positive else (-2.0, 2.0)
    xs = [RNG.uniform(lo, hi, size=shape) for _ in range(n)]
    got = T.liftT(T.expr(f, n, key=(name, n)), [T.put(x) for x in xs]).numpy()
    assert rel_err(got, O.liftT(f, xs)) < RTOL


def test_same_expression_serves_both_dtypes(T, T32):
    """one compiled closure, two element types (`forall a. RealFloat a`, Types.hs:56-59)"""
    f = lambda v: ad.sin(v[0]) * v[1] + ad.exp(-v[0] * v[0])  # noqa: E731
    e = T.expr(f, 2, key="both_dtypes")
    x, y = rnd(1000), rnd(1000)
    got64 = T.liftT(e, [T.put(x), T.put(y)]).numpy()
    got32 = T32.liftT(e, [T32.put(x), T32.put(y)]).numpy()
    want = np.sin(x) * y + np.exp(-x * x)
    assert got64.dtype == np.float64 and got32.dtype == np.float32
    assert rel_err(got64, want) < RTOL
    assert np.linalg.norm(got32 - want) / np.linalg.norm(want) < 1e-5


def test_mixing_dtypes_is_a_loud_error(T, T32):
    from tensor_ops_amd.capi import TensorOpsError
    a64, a32 = T.put(rnd(4, 4)), T32.put(rnd(4, 4))
    with pytest.raises(TensorOpsError):
        T.gmul(1, 1, 1, a64, a32)
    with pytest.raises(TensorOpsError):
        T.liftT(lambda v: v[0] + v[1], [a64, a32], key="mix")
    with pytest.raises(TensorOpsError):
        T.sumT([a64, a32], (4, 4))


def test_reductions_layout_and_index(T):
    x = ints(9, 4, 3)
    dx = T.put(x)
    assert np.array_equal(T.sumRows(dx).numpy(), x.sum(axis=0))
    v = ints(300)
    assert T.sumRows(T.put(v)).numpy() == v.sum()
    wide = ints(50, 130)
    assert np.array_equal(T.sumRows(T.put(wide)).numpy(), wide.sum(axis=0))
    tall = ints(5000, 6)
    assert np.array_equal(T.sumRows(T.put(tall)).numpy(), tall.sum(axis=0))
    row = ints(4, 3)
    assert np.array_equal(T.mapRows_const(1, T.put(row), dx).numpy(), np.broadcast_to(row, x.shape))
    assert np.array_equal(T.mapRows(1, lambda r: T.scaleT(2.0, r), dx).numpy(), 2 * x)
    assert np.array_equal(T.transp(dx).numpy(), nested.transpose(x))
    d = ints(5)
    for rank in (1, 2, 3):
        dd = T.diag(rank, T.put(d))
        assert np.array_equal(dd.numpy(), nested.diag(rank, d))
        if rank >= 2:
            assert np.array_equal(T.getDiag(dd).numpy(), d)
    assert T.index(dx, (8, 3, 2)) == x[8, 3, 2]
    assert T.index(T.transp(dx), (2, 3, 8)) == x[8, 3, 2]
    xs = [ints(5, 6) for _ in range(6)]
    ds = [T.put(v) for v in xs]
    for n in range(7):
        got = T.sumT(ds[:n], (5, 6)).numpy()
        assert got.dtype == np.float64 and np.array_equal(got, O.sumT(xs[:n], (5, 6)))
    big = rnd((1 << 21) + 17)
    got = float(T.sumRows(T.put(big)).numpy())
    assert abs(got - big.sum()) < 1e-13 * np.abs(big).sum()
    # precision that fp32 cannot hold
    tiny = np.array([1.0, 1e-12, -1.0])
    assert float(T.sumRows(T.put(tiny)).numpy()) == pytest.approx(1e-12, rel=1e-3)


def test_batched_ops_argmax_onehot(T):
    B = 29
    X = rnd(B, 12)
    dX = T.put(X, batched=True)
    assert np.array_equal(T.batch_select(dX, 5).numpy(), X[5])
    assert rel_err(T.batch_sum(dX).numpy(), X.sum(axis=0)) < RTOL
    assert rel_err(T.sumRows(dX).numpy(), X.sum(axis=1)) < RTOL
    b = rnd(12)
    assert np.array_equal(T.batch_bcast(T.put(b), 4).numpy(), np.broadcast_to(b, (4, 12)))
    got = T.liftT(lambda v: v[0] + v[1], [dX, T.put(b)], key="bias64").numpy()
    assert np.array_equal(got, X + b)
    idx = T.arg_max(dX)
    assert list(idx) == list(np.argmax(X, axis=1))
    # ties: earliest index (Tensor.hs:291-305 via base's `Arg` maximum)
    tie = np.array([[1.0, 3.0, 3.0, 2.0], [5.0, 5.0, 5.0, 5.0]])
    assert list(T.arg_max(T.put(tie, batched=True))) == [1, 0]
    # a gap only fp64 sees
    close = np.array([1.0, 1.0 + 2e-16 * 2, 1.0])
    assert T.arg_max(T.put(close)) == 1
    oh = T.one_hot(5, 1.0, 0.0, [3, 0, 4]).numpy()
    assert oh.dtype == np.float64 and np.array_equal(oh, np.eye(5)[[3, 0, 4]])


def test_genRand_fp64_is_counter_based(T):
    def splitmix64(z):
        z = (z + 0x9e3779b97f4a7c15) & (2**64 - 1)
        z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & (2**64 - 1)
        return z ^ (z >> 31)
    u = T.genRand((1000,), "uniform", -1.0, 1.0, SEED).numpy()
    want = np.array([-1.0 + 2.0 * ((splitmix64((SEED + 0x9e3779b97f4a7c15 * i) & (2**64 - 1)) >> 11)
                                   * (1.0 / 9007199254740992.0)) for i in range(1000)])
    assert u.dtype == np.float64 and np.array_equal(u, want)
    n = T.genRand((200000,), "normal", 0.0, 0.5, SEED + 1).numpy()
    assert abs(n.mean()) < 5e-3 and abs(n.std() - 0.5) < 5e-3


def test_blas_class_entry_points_fp64(T):
    from tensor_ops_amd import capi
    from tensor_ops_amd.hipt import DT
    L = capi.lib()
    A, Bm, Cm = ints(6, 5), ints(5, 7), ints(6, 7)
    x, y, y6 = ints(5), ints(5), ints(6)
    dA, dB, dC, dx, dy, dy6 = (T.put(v) for v in (A, Bm, Cm, x, y, y6))

    def call(fn, *args):
        h = capi.c_tensor()
        capi.check(fn(*args, C.byref(h)))
        return DT(h).numpy()

    assert np.array_equal(call(L.to_blas_axpy, 2.0, dx.h, dy.h), 2 * x + y)
    v = C.c_double()
    capi.check(L.to_blas_dot(dx.h, dy.h, C.byref(v)))
    assert v.value == float(x @ y)
    assert np.array_equal(call(L.to_blas_ger, dy6.h, dx.h), np.outer(y6, x))
    assert np.array_equal(call(L.to_blas_gemv, 2.0, dA.h, dx.h, 3.0, dy6.h), 2 * (A @ x) + 3 * y6)
    assert np.array_equal(call(L.to_blas_gemm, -1.0, dA.h, dB.h, 2.0, dC.h), -(A @ Bm) + 2 * Cm)
    assert np.array_equal(call(L.to_blas_eye, capi.TO_F64, 4), np.eye(4))
    capi.check(L.to_blas_sum(dA.h, C.byref(v)))
    assert v.value == float(A.sum())
    sq = ints(5, 5)
    dsq = T.put(sq)
    capi.check(L.to_blas_trace(dsq.h, C.byref(v)))
    assert v.value == float(np.trace(sq))


# ---- TOp level: the same oracle closures over the fp64 HIP instance -----------------------------------
def both(T, op, xs):
    ys_o = TO.runTOp(op, O, xs)
    dxs = [T.put(x) for x in xs]
    for a, b in zip(TO.runTOp(op, T, dxs), ys_o):
        assert rel_err(a.numpy(), b) < RTOL
    ds = [RNG.uniform(-1, 1, size=np.shape(y)) for y in ys_o]
    for a, b in zip(op.grad(T, dxs, [T.put(d) for d in ds]), op.grad(O, list(xs), ds)):
        assert rel_err(a.numpy(), b) < RTOL


def test_op_vocabulary_fp64(T):
    both(T, TO.gmul(2, 1, 1), [rnd(2, 3, 4), rnd(4, 5)])
    both(T, TO.matVec(), [rnd(9, 7), rnd(7)])
    both(T, TO.matMat(), [rnd(9, 7), rnd(7, 5)])
    both(T, TO.dot(), [rnd(11), rnd(11)])
    both(T, TO.map_(NN.logistic), [rnd(3, 4)])
    both(T, TO.zip_(lambda x, y: x * y + ad.sin(x)), [rnd(6), rnd(6)])
    both(T, TO.sumRows(), [rnd(6, 3)])
    both(T, TO.sumOp(3, (4,)), [rnd(4), rnd(4), rnd(4)])
    both(T, NN.softmax(), [rnd(10)])
    both(T, NN.squaredError(), [rnd(4), rnd(4)])
    both(T, NN.crossEntropy(), [RNG.uniform(0.1, 0.9, size=6), rnd(6)])


def _weights(sizes):
    return [(0.5 * RNG.standard_normal((o, i)), 0.5 * RNG.standard_normal(o))
            for i, o in zip(sizes[:-1], sizes[1:])]


def test_c1_dots_config_in_the_reference_precision(T):
    """BASELINE config 1 exactly as the reference runs it: 2 -> 16 -> 1 logistic net, squaredError,
    rate 1, Double elements (app/Dots.hs:60-92, HMat Double); 20 online steps."""
    ws = _weights([2, 16, 1])
    net_o = NN.genNet(ws, NN.actLogistic, NN.actLogistic)
    net_t = NN.Network(net_o.op, [T.put(p) for p in net_o.params])
    for _ in range(20):
        x, y = rnd(2), np.array([float(RNG.integers(0, 2))])
        net_o = NN.trainNetwork(O, NN.squaredError(), 1.0, x, y, net_o)
        net_t = NN.trainNetwork(T, NN.squaredError(), 1.0, T.put(x), T.put(y), net_t)
    for a, b in zip(net_t.params, net_o.params):
        assert rel_err(a.numpy(), b) < 1e-11


def test_against_the_plain_c_hmat_restatement(T, H):
    """the oracle's plain-C HMat BLAS sequence (oracle/hmat_path.c) is fp64: the fp64 HIP instance
    must agree with it to rounding on the summed per-sample gradients and on online SGD."""
    from oracle import hmat
    ws = _weights([20, 12, 5])
    B = 16
    X = RNG.uniform(0, 1, size=(B, 20))
    Y = np.zeros((B, 5))
    Y[np.arange(B), RNG.integers(0, 5, size=B)] = 1.0
    want, _ = hmat.batched_grads(X, Y, ws[0][0], ws[0][1], ws[1][0], ws[1][1])
    net_h = H.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
    tr = H.Trainer(net_h, "crossEntropy", 0.5, T.put(X, batched=True), T.put(Y, batched=True),
                   use_memo=True, use_graph=False)
    tr.grad()
    before = [p.numpy() for p in tr.net.params]
    tr.apply()
    for b, a, w in zip(before, tr.net.params, want):
        assert rel_err(a.numpy(), b - 0.5 * w) < 1e-11
    # per-sample online SGD (app/MNIST.hs:390-396) through trainNetwork, 8 samples
    p_c, _ = hmat.train_online(X[:8], Y[:8], ws[0][0], ws[0][1], ws[1][0], ws[1][1], 0.1)
    net = H.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
    for i in range(8):
        net = H.trainNetwork(net, "crossEntropy", 0.1, T.put(X[i]), T.put(Y[i]))
    for a, b in zip(net.params, p_c):
        assert rel_err(a.numpy(), b) < 1e-11


@pytest.mark.parametrize("sizes,hid,out,loss,B", [
    ([20, 12, 5], "actMapLogistic", "actSoftmax", "crossEntropy", 33),
    ([784, 256, 10], "actMapLogistic", "actSoftmax", "crossEntropy", 48),
    ([12, 9, 7, 4], "actLogistic", "actLogistic", "squaredError", 21),
])
@pytest.mark.parametrize("graph", [False, True])
def test_host_mirror_batched_gradTOp_fp64(T, H, sizes, hid, out, loss, B, graph):
    oact = {"actLogistic": NN.actLogistic, "actMapLogistic": lambda: NN.actMap(NN.logistic),
            "actSoftmax": NN.actSoftmax}
    ws = _weights(sizes)
    net_o = NN.genNet(ws, oact[hid], oact[out])
    net_h = H.genNet([(T.put(w), T.put(b)) for w, b in ws], hid, out)
    X = RNG.uniform(0, 1, size=(B, sizes[0]))
    Y = np.zeros((B, sizes[-1]))
    Y[np.arange(B), RNG.integers(0, sizes[-1], size=B)] = 1.0
    oloss = {"crossEntropy": NN.crossEntropy, "squaredError": NN.squaredError}[loss]()
    want = NN.batched_param_grads(O, oloss, list(X), list(Y), net_o)
    tr = H.Trainer(net_h, loss, 0.02, T.put(X, batched=True), T.put(Y, batched=True),
                   use_memo=True, use_graph=graph, use_fused=True)
    # the pre-fused kernels serve fp64 too (every contraction of these latency-bound shapes runs on the
    # small-GEMM kernel; a shape beyond it makes the library answer TO_ERR_UNSUPPORTED and the trainer
    # falls back to the generic composition)
    assert tr.fused
    tr.grad()
    before = [p.numpy() for p in tr.net.params]
    assert all(p.dtype == np.float64 for p in before)
    tr.apply()
    for b, a, w in zip(before, tr.net.params, want):
        assert rel_err(a.numpy(), b - 0.02 * w) < 1e-11
    tr.grad()                    # graph replay (or direct re-run) on the updated parameters
    tr.apply()
    net_o2 = NN.Network(net_o.op, [b - 0.02 * w for b, w in zip(before, want)])
    want2 = NN.batched_param_grads(O, oloss, list(X), list(Y), net_o2)
    for p2, a, w in zip(net_o2.params, tr.net.params, want2):
        assert rel_err(a.numpy(), p2 - 0.02 * w) < 1e-11


def test_bytecode_vm_fallback_fp64(repo_root):
    """TOPS_EXPR_JIT=0: the LDS-slot bytecode VM evaluates unclassified closures, in fp64 too"""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np
from oracle import ad
from tensor_ops_amd.hipt import HipT
T = HipT(0, dtype=np.float64)
f = lambda v: ad.sin(v[0]) * v[1] + ad.exp(-v[0] * v[0])
e = T.expr(f, 2, key="vm64")
assert e.kind == 0, e.kind
rng = np.random.default_rng(7)
for n in (5, 4099):
    x, y = rng.uniform(-2, 2, n), rng.uniform(-2, 2, n)
    got = T.liftT(e, [T.put(x), T.put(y)]).numpy()
    want = np.sin(x) * y + np.exp(-x * x)
    assert got.dtype == np.float64
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-12
print("ok")
'''
    env = dict(os.environ, TOPS_EXPR_JIT="0", PYTHONPATH=repo_root)
    out = subprocess.run([sys.executable, "-c", code], cwd=repo_root, env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,k,n", [(4096, 288, 4096), (2048, 160, 8192), (4352, 48, 4096), (3072, 208, 3072),
                                   (2048, 800, 2048), (4352, 160, 4096), (4352, 1024, 4096)])
def test_full_tile_f64_kernel_every_layout_bit_exact_on_integers(ta, tb, m, k, n):
    """`gemm_f64_w4_kernel` (whole rounds of full 256x128 tiles: DMA-fed LDS images, 16-byte fragments -- two k of
    a k-contiguous operand or two OWNED rows/columns of an m-/n-contiguous one, mapped back in the epilogue): the
    whole output on all four operand layouts, integer data so that any summation order is exact.
    (3072^2 = 288 tiles, 2048^2 = 128 tiles and 17 x 32 = 544 tiles do not fill whole rounds of 256: stream-K, every
    workgroup an equal share of the k-tile stream, partial tiles added up in workgroup order by the fix-up pass.)"""
    from tensor_ops_amd.hipt import HipT
    T = HipT(0, dtype=np.float64)
    rng = np.random.default_rng(900 + 2 * ta + tb)
    a = rng.integers(-3, 4, size=(m, k)).astype(np.float64)
    b = rng.integers(-3, 4, size=(k, n)).astype(np.float64)
    da = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
    db = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
    assert np.array_equal(T.gmul(1, 1, 1, da, db).numpy(), a @ b)


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,k,n", [(1024, 1024, 1024), (1000, 1000, 1000), (768, 200, 896), (1100, 531, 900), (640, 4096, 640)])
def test_wave_split_f64_kernel_every_layout_bit_exact_on_integers(ta, tb, m, k, n):
    """`gemm_kw64_kernel` (gemm_kwave_f64.hip: 100 .. 320 tiles of 64x64, the K loop split over the four waves of each tile's
    workgroup, partial tiles summed in LDS in wave order, ragged M / N by clamped loads, K tails of 8, 3 and 0 inside the
    kernel) on all four operand layouts: the whole output, integer data."""
    from tensor_ops_amd.hipt import HipT
    T = HipT(0, dtype=np.float64)
    rng = np.random.default_rng(940 + 2 * ta + tb)
    a = rng.integers(-3, 4, size=(m, k)).astype(np.float64)
    b = rng.integers(-3, 4, size=(k, n)).astype(np.float64)
    da = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
    db = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
    assert np.array_equal(T.gmul(1, 1, 1, da, db).numpy(), a @ b)


def test_mid_size_layers_keep_their_epilogues_in_fp64(T, H):
    """A 1024-row batch through 272 -> 640 -> 200 -> 10 in the reference's precision: the 640-wide layer's forward
    `W x + b` + logistic and backward `dZ W (.) h (1 - h)` are 160-tile contractions and run on the wave-split fp64 kernel
    with the epilogue in its final reduction (the tiled fp64 kernel has none and used to force them unfused).  Checked by
    linearity over the batch against sixteen 64-row gradients (other kernels) and against the oracle on the first 64."""
    rng = np.random.default_rng(955)
    B, n, dims = 1024, 64, (272, 640, 200, 10)
    ws = [(0.3 * rng.standard_normal((o, i)), 0.3 * rng.standard_normal(o)) for i, o in zip(dims[:-1], dims[1:])]
    X = rng.uniform(0, 1, size=(B, dims[0]))
    Y = np.zeros((B, dims[-1]))
    Y[np.arange(B), rng.integers(0, dims[-1], size=B)] = 1.0

    def grads(lo, hi):
        tr = H.Trainer(H.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax"), "crossEntropy", 1.0,
                       T.put(X[lo:hi], batched=True), T.put(Y[lo:hi], batched=True), use_graph=False)
        before = [p.numpy() for p in tr.net.params]
        tr.grad()
        tr.apply()
        return [b - a.numpy() for b, a in zip(before, tr.net.params)]

    big = grads(0, B)
    acc = None
    for c in range(B // n):
        g = grads(c * n, (c + 1) * n)
        acc = g if acc is None else [x + y for x, y in zip(acc, g)]
        if c == 0:
            net_o = NN.genNet(ws, lambda: NN.actMap(NN.logistic), NN.actSoftmax)
            want = NN.batched_param_grads(O, NN.crossEntropy(), list(X[:n]), list(Y[:n]), net_o)
            for a, w in zip(g, want):
                assert rel_err(a, w) < 1e-11
    for a, w in zip(big, acc):
        assert rel_err(a, w) < 1e-11


@pytest.mark.parametrize("rows,K,N,b_transposed", [(65536, 64, 512, False), (16 * 8193, 64, 256, True), (16 * 4099 * 4, 64, 128, False),
                                                   (16 * 4500, 64, 384, True), (65536 + 5, 64, 256, False), (65536, 32, 256, False)])
def test_short_k_streaming_gemm_fp64(T, rows, K, N, b_transposed):
    """gemm_skinnyk_f64.hip (config 5's shape class in Double: B resident in LDS, barrier-free wave streams of 16-row blocks,
    whole-row stores through wave-private strips): exact on small integers, both B layouts, one to four column panels, even and odd numbers of blocks
    per wave stream (the last two shapes -- ragged rows, K = 32 -- stay on the tiled kernel); then with bias + logistic recorded behind it -- ONE launch."""
    from tensor_ops_amd import hipt
    rng = np.random.default_rng(970 + K + N)
    a = rng.integers(-3, 4, (rows, K)).astype(np.float64)
    bn = rng.integers(-3, 4, (K, N)).astype(np.float64)
    bias = rng.integers(-2, 3, N).astype(np.float64)
    A = T.put(a)
    B = T.transp(T.put(np.ascontiguousarray(bn.T))) if b_transposed else T.put(bn)
    want = a @ bn
    streaming = K == 64 and rows % 16 == 0
    st = T.stats()["launches"]
    got = T.gmul(1, 1, 1, A, B)
    if streaming:
        assert T.stats()["launches"] - st == 1
    assert np.array_equal(got.numpy(), want)
    x = T.put(a, batched=True)
    W = T.put(np.ascontiguousarray(bn.T))
    bt = T.put(bias)
    st = T.stats()["launches"]
    with T.memo():
        hb = T.force(T.liftT(hipt.logistic_closure, [T.sumT([T.matVec(W, x), bt], (N,))], key="skinny-logistic64"))
    if streaming and rows * N >= 1 << 24:   # (large enough for the streaming kernel to be chosen for the recorded form as well)
        assert T.stats()["launches"] - st == 1
    assert np.max(np.abs(hb.numpy().reshape(-1, N) - 1 / (1 + np.exp(-(want + bias))))) < 1e-14


def test_fused_fp64_logistic_keeps_nan_and_the_infinities(T):
    """the table-driven fp64 logistic of the fused short-K epilogue (gemm_skinnyk_f64.hip logistic64_tab) clamped
    its denominator with fmin, which swallows a NaN and turned inf - inf into 1e-38: a diverged run came back as finite
    activations.  logistic(NaN) is NaN, logistic(+inf) = 1, logistic(-inf) = 0, |z| beyond exp's range saturates -- as
    1 / (1 + exp(-z)) does -- and every finite row next to them is untouched (1e-14)."""
    from tensor_ops_amd import hipt
    rows, K, N = 65536, 64, 256
    rng = np.random.default_rng(4242)
    a = rng.integers(-3, 4, (rows, K)).astype(np.float64)
    bn = rng.integers(-3, 4, (K, N)).astype(np.float64)
    bias = rng.integers(-2, 3, N).astype(np.float64)
    a[5, 0] = np.nan          # row 5: every z is NaN
    a[9, :] = 0.0
    a[9, 1] = np.inf          # row 9: z = +-inf where B[1, n] != 0, NaN where it is 0 (inf * 0)
    a[12, :] = 0.0
    a[12, 2] = 1e300          # row 12: |z| ~ 1e300: saturates to 1 / 0 (0 where B[2, n] == 0: logistic(bias))
    x = T.put(a, batched=True)
    W = T.put(np.ascontiguousarray(bn.T))
    bt = T.put(bias)
    st = T.stats()["launches"]
    with T.memo():
        hb = T.force(T.liftT(hipt.logistic_closure, [T.sumT([T.matVec(W, x), bt], (N,))], key="skinny-logistic64-wild"))
    assert T.stats()["launches"] - st == 1          # the fused streaming launch is what is under test
    got = hb.numpy().reshape(-1, N)
    with np.errstate(over="ignore", invalid="ignore"):
        z = a @ bn + bias
        want = 1 / (1 + np.exp(-z))
    assert np.isnan(got[5]).all()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    fin = ~np.isnan(want)
    assert np.max(np.abs(got[fin] - want[fin])) < 1e-14
    assert set(np.unique(got[9][~np.isnan(got[9])])) <= {0.0, 1.0}
    assert (got[12][bn[2] > 0] == 1.0).all() and (got[12][bn[2] < 0] == 0.0).all()
