"""GPU parity of the C++ host mirror (TOp DSL + Learn layer over the C ABI) against
the oracle: same op trees, same inputs; 1e-5 relative for fp32 (north_star)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ad, neuralnet as NN, top as TO  # noqa: E402
from oracle.tensor import OTensor  # noqa: E402

RTOL = 1e-5
RNG = np.random.default_rng(0x7e500001)
O = OTensor(np.float64)


@pytest.fixture(scope="module")
def T():
    from tensor_ops_amd.hipt import HipT
    return HipT(0)


@pytest.fixture(scope="module")
def H():
    from tensor_ops_amd import tops
    tops.hlib()
    return tops


def rel_err(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    den = np.linalg.norm(want.ravel())
    return np.linalg.norm((got - want).ravel()) / (den if den > 0 else 1.0)


def rnd(*s):
    return RNG.uniform(-1, 1, size=s)


def both(T, hop, oop, xs):
    """run + grad of a host-mirror op against the same oracle op."""
    dxs = [T.put(x) for x in xs]
    ys_o = TO.runTOp(oop, O, xs)
    ys_h = hop.run(dxs)
    assert len(ys_o) == len(ys_h)
    for a, b in zip(ys_h, ys_o):
        assert rel_err(a.numpy(), b) < RTOL
    ds = [RNG.uniform(-1, 1, size=np.shape(y)) for y in ys_o]
    g_o = oop.grad(O, list(xs), ds)
    g_h = hop.grad(dxs, [T.put(d) for d in ds])
    for a, b in zip(g_h, g_o):
        assert rel_err(a.numpy(), b) < RTOL


def test_vocabulary(T, H):
    both(T, H.gmul(2, 1, 1), TO.gmul(2, 1, 1), [rnd(2, 3, 4), rnd(4, 5)])
    both(T, H.gmul(1, 2, 1), TO.gmul(1, 2, 1), [rnd(2, 3, 4), rnd(4, 3, 2)])
    both(T, H.gmul(0, 2, 0), TO.gmul(0, 2, 0), [rnd(3, 4), rnd(4, 3)])
    both(T, H.matVec(), TO.matVec(), [rnd(9, 7), rnd(7)])
    both(T, H.vecMat(), TO.vecMat(), [rnd(9), rnd(9, 7)])
    both(T, H.matMat(), TO.matMat(), [rnd(9, 7), rnd(7, 5)])
    both(T, H.dot(), TO.dot(), [rnd(11), rnd(11)])
    both(T, H.outer(1, 1), TO.outer(1, 1), [rnd(4), rnd(6)])
    both(T, H.outer(0, 1), TO.outer(0, 1), [rnd(), rnd(6)])
    both(T, H.map_(NN.logistic), TO.map_(NN.logistic), [rnd(3, 4)])
    both(T, H.map_(NN.logistic, NN.logistic_prime), TO.map_(NN.logistic, NN.logistic_prime), [rnd(5)])
    both(T, H.map_(lambda x: ad.sin(x) * x), TO.map_(lambda x: ad.sin(x) * x), [rnd(5)])
    f2 = lambda x, y: x * y + ad.sin(x)  # noqa: E731
    both(T, H.zip_(f2), TO.zip_(f2), [rnd(6), rnd(6)])
    f3 = lambda x, y, z: x * y / (2 + z)  # noqa: E731
    both(T, H.zip3(f3), TO.zip3(f3), [rnd(5), rnd(5), rnd(5)])
    f4 = lambda v: v[0] * v[1] - v[2] * ad.tanh(v[3])  # noqa: E731
    both(T, H.zipN(4, f4), TO.zipN(4, f4), [rnd(2, 2) for _ in range(4)])
    both(T, H.add(), TO.add(), [rnd(5), rnd(5)])
    both(T, H.add3(), TO.add3(), [rnd(5), rnd(5), rnd(5)])
    both(T, H.duplicate(), TO.duplicate(), [rnd(5)])
    both(T, H.replicate(3), TO.replicate(3), [rnd(2, 2)])
    both(T, H.swap(), TO.swap(), [rnd(3), rnd(2)])
    both(T, H.scale(2.5), TO.scale(2.5), [rnd(4)])
    both(T, H.negate(), TO.negate(), [rnd(4)])
    both(T, H.transpOp(), TO.transpOp(), [rnd(2, 3, 4)])
    both(T, H.sumRows(), TO.sumRows(), [rnd(6, 3)])
    both(T, H.sumRows(), TO.sumRows(), [rnd(6)])
    both(T, H.sumOp(3, (4,)), TO.sumOp(3, (4,)), [rnd(4), rnd(4), rnd(4)])
    both(T, H.shuffle([1, 1, 0], 2), TO.shuffle([1, 1, 0], [(2,), (3,)]), [rnd(2), rnd(3)])
    both(T, H.drop(1, 2), TO.drop(1, [(2,), (3,)]), [rnd(2), rnd(3)])
    both(T, H.take(1, 2), TO.take(1, [(2,), (3,)]), [rnd(2), rnd(3)])
    both(T, H.idOp(2), TO.idOp(2), [rnd(2), rnd(3)])


def test_combinators(T, H):
    both(T, H.firstOp(H.matVec(), 1), TO.first(TO.matVec(), 1), [rnd(3, 4), rnd(4), rnd(2)])
    both(T, H.secondOp(1, H.matVec()), TO.secondOp(1, TO.matVec()), [rnd(2), rnd(3, 4), rnd(4)])
    both(T, H.par(H.dot(), H.map_(ad.exp)), TO.par(TO.dot(), TO.map_(ad.exp)), [rnd(3), rnd(3), rnd(2)])
    both(T, H.fanout(H.map_(ad.exp), H.scale(3.0)), TO.fanout(TO.map_(ad.exp), TO.scale(3.0), [(3,)]),
         [rnd(3)])
    both(T, H.then_first(H.matVec(), H.add()), TO.then_first(TO.matVec(), TO.add()),
         [rnd(3, 4), rnd(4), rnd(3)])
    both(T, H.matVec() >> H.map_(NN.logistic), TO.matVec() >> TO.map_(NN.logistic), [rnd(3, 4), rnd(4)])
    both(T, H.softmax(), NN.softmax(), [rnd(10)])
    both(T, H.squaredError(), NN.squaredError(), [rnd(4), rnd(4)])
    both(T, H.crossEntropy(), NN.crossEntropy(), [RNG.uniform(0.1, 0.9, size=6), rnd(6)])
    both(T, H.named("ffLayer"), NN.ffLayer_op(), [rnd(4), rnd(3, 4), rnd(3)])


def test_arity_and_shape_errors(T, H):
    from tensor_ops_amd.capi import TensorOpsError
    with pytest.raises(TensorOpsError):
        H.matVec() >> H.dot()  # 1 output into 2 inputs
    with pytest.raises(TensorOpsError) as ei:
        H.matVec().run([T.put(rnd(3, 4)), T.put(rnd(5))])
    assert ei.value.code == 2


def test_laziness_unwanted_cotangents_are_not_computed(T, H):
    op = H.matVec()
    W, x = T.put(rnd(8, 6)), T.put(rnd(6))
    d = T.put(rnd(8))
    st = T.stats()["launches"]
    g = op.grad([W, x], [d], want=[True, False])
    assert g[1] is None and g[0] is not None
    assert T.stats()["launches"] - st == 1


def _weights(sizes):
    return [(0.5 * RNG.standard_normal((o, i)), 0.5 * RNG.standard_normal(o))
            for i, o in zip(sizes[:-1], sizes[1:])]


def _nets(T, H, sizes, hid, out):
    oact = {"actLogistic": NN.actLogistic, "actMapLogistic": lambda: NN.actMap(NN.logistic),
            "actSoftmax": NN.actSoftmax}
    ws = _weights(sizes)
    net_o = NN.genNet(ws, oact[hid], oact[out])
    net_h = H.genNet([(T.put(w), T.put(b)) for w, b in ws], hid, out)
    return ws, net_o, net_h


def test_c1_dots_network_one_step(T, H):
    """BASELINE config 1: 2->16->1, actLogistic, squaredError, rate 1 (app/Dots.hs:60-92),
    runTOp + gradTOp + one trainNetwork step, through the C++ host mirror."""
    ws, net_o, net_h = _nets(T, H, [2, 16, 1], "actLogistic", "actLogistic")
    x, y = rnd(2), np.array([1.0])
    dx, dy = T.put(x), T.put(y)
    assert rel_err(H.runNetwork(net_h, dx).numpy(), NN.runNetwork(O, net_o, x)) < RTOL
    g_o = NN.netGrad(O, NN.squaredError(), x, y, net_o)
    g_h = H.netGrad(net_h, "squaredError", dx, dy)
    for a, b in zip(g_h, g_o):
        assert rel_err(a.numpy(), b) < RTOL
    n_o = NN.trainNetwork(O, NN.squaredError(), 1.0, x, y, net_o)
    n_h = H.trainNetwork(net_h, "squaredError", 1.0, dx, dy)
    for a, b in zip(n_h.params, n_o.params):
        assert rel_err(a.numpy(), b) < RTOL
    # a few online-SGD steps, like `foldl' trainEach` (Dots.hs:74-80)
    for _ in range(5):
        x, y = rnd(2), np.array([float(RNG.integers(0, 2))])
        n_o = NN.trainNetwork(O, NN.squaredError(), 1.0, x, y, n_o)
        n_h = H.trainNetwork(n_h, "squaredError", 1.0, T.put(x), T.put(y))
    for a, b in zip(n_h.params, n_o.params):
        assert rel_err(a.numpy(), b) < RTOL


def test_mnist_style_network_unbatched(T, H):
    ws, net_o, net_h = _nets(T, H, [20, 12, 5], "actMapLogistic", "actSoftmax")
    x = RNG.uniform(0, 1, size=20)
    y = np.zeros(5)
    y[3] = 1
    g_o = NN.netGrad(O, NN.crossEntropy(), x, y, net_o)
    g_h = H.netGrad(net_h, "crossEntropy", T.put(x), T.put(y))
    for a, b in zip(g_h, g_o):
        assert rel_err(a.numpy(), b) < RTOL
    g_h = H.netGrad(net_h, "crossEntropy", T.put(x), T.put(y), want_x=False)
    assert g_h[0] is None


def _batch(B, i, o):
    X = RNG.uniform(0, 1, size=(B, i))
    Y = np.zeros((B, o))
    Y[np.arange(B), RNG.integers(0, o, size=B)] = 1.0
    return X, Y


@pytest.mark.parametrize("sizes,hid,out,loss,B", [
    ([20, 12, 5], "actMapLogistic", "actSoftmax", "crossEntropy", 33),
    ([6, 16, 3], "actLogistic", "actLogistic", "squaredError", 64),
    ([784, 256, 10], "actMapLogistic", "actSoftmax", "crossEntropy", 48),
    ([12, 9, 7, 4], "actLogistic", "actSoftmax", "crossEntropy", 21),
    ([30, 70, 40, 5], "actMapLogistic", "actLogistic", "squaredError", 130),
])
@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("fused", [False, True])
def test_batched_gradTOp_equals_sum_of_per_sample(T, H, sizes, hid, out, loss, B, graph, fused):
    """batched gradTOp = sum_b gradTOp(x_b, p, y_b) at fixed params."""
    ws, net_o, net_h = _nets(T, H, sizes, hid, out)
    X, Y = _batch(B, sizes[0], sizes[-1])
    oloss = {"crossEntropy": NN.crossEntropy, "squaredError": NN.squaredError}[loss]()
    want = NN.batched_param_grads(O, oloss, list(X), list(Y), net_o)
    dX, dY = T.put(X, batched=True), T.put(Y, batched=True)
    tr = H.Trainer(net_h, loss, 0.02, dX, dY, use_memo=True, use_graph=graph, use_fused=fused)
    assert tr.fused == fused
    tr.grad()
    got = tr.net  # parameters now live in the flat buffer; grads next to them
    p_ptr, g_ptr, n = tr.flat()
    import ctypes as C
    from tensor_ops_amd import capi
    flat = np.empty(n, dtype=np.float32)
    h = capi.c_tensor()
    d = (C.c_int64 * 1)(n)
    capi.check(capi.lib().to_wrap(C.c_void_p(g_ptr), 0, 1, d, 0, C.byref(h)))
    capi.check(capi.lib().to_download(h, flat.ctypes.data_as(C.c_void_p), flat.nbytes))
    capi.lib().to_release(h)
    off = 0
    for w in want:
        sz = w.size
        assert rel_err(flat[off:off + sz].reshape(w.shape), w) < RTOL
        off += (sz + 3) // 4 * 4
    # the SGD step on the flat buffer = zip (\p g -> p - r*g)   (FeedForward.hs:141-147)
    before = [p.numpy() for p in got.params]
    tr.apply()
    for b, a, w in zip(before, tr.net.params, want):
        assert rel_err(a.numpy(), b.astype(np.float64) - 0.02 * w) < RTOL
    # replay with new data in the same buffers
    X2, Y2 = _batch(B, sizes[0], sizes[-1])
    for t, v in ((dX, X2), (dY, Y2)):
        v32 = np.ascontiguousarray(v, dtype=np.float32)
        capi.check(capi.lib().to_upload(t.h, v32.ctypes.data_as(C.c_void_p), v32.nbytes))
    net_o2 = NN.Network(net_o.op, [p.numpy().astype(np.float64) for p in tr.net.params])
    want2 = NN.batched_param_grads(O, oloss, list(X2), list(Y2), net_o2)
    tr.grad()
    capi.check(capi.lib().to_wrap(C.c_void_p(g_ptr), 0, 1, d, 0, C.byref(h)))
    capi.check(capi.lib().to_download(h, flat.ctypes.data_as(C.c_void_p), flat.nbytes))
    capi.lib().to_release(h)
    off = 0
    for w in want2:
        assert rel_err(flat[off:off + w.size].reshape(w.shape), w) < RTOL
        off += (w.size + 3) // 4 * 4


@pytest.mark.parametrize("sizes,hid,out,loss,B", [
    ([20, 12, 5], "actMapLogistic", "actSoftmax", "crossEntropy", 33),
    ([6, 16, 3], "actLogistic", "actLogistic", "squaredError", 64),
    ([784, 256, 10], "actMapLogistic", "actSoftmax", "crossEntropy", 1024),
    ([12, 9, 7, 4], "actLogistic", "actSoftmax", "crossEntropy", 21),
    ([30, 70, 40, 5], "actMapLogistic", "actLogistic", "squaredError", 130),
    ([64, 300, 260, 10], "actMapLogistic", "actSoftmax", "crossEntropy", 1024),
    ([5, 3], "actLogistic", "actSoftmax", "crossEntropy", 1),
])
@pytest.mark.parametrize("fused", [False, True])
def test_trainer_step_is_trainNetwork_on_the_batch(T, H, sizes, hid, out, loss, B, fused):
    """`step()` = p <- p - rate * (batched gradTOp) (FeedForward.hs:247-260), three times in a row.  On the
    pre-fused path the update happens in the epilogue of the weight-gradient launches, in place: later layers'
    parameters must not be overwritten before the propagation has read them."""
    ws, net_o, net_h = _nets(T, H, sizes, hid, out)
    X, Y = _batch(B, sizes[0], sizes[-1])
    oloss = {"crossEntropy": NN.crossEntropy, "squaredError": NN.squaredError}[loss]()
    scale = (B // 64) if B > 64 else 1
    rate = 0.5 / B  # (gradients are batch SUMS: keep the step small enough that fp32 softmax does not underflow)
    tr = H.Trainer(net_h, loss, rate, T.put(X, batched=True), T.put(Y, batched=True), use_graph=False,
                   use_fused=fused)
    assert tr.fused == fused
    params = [np.asarray(p, dtype=np.float64) for p in net_o.params]
    Xs, Ys = list(X[:64]), list(Y[:64])  # the oracle is per-sample python: cap its work ...
    if B > 64:  # ... by giving the big batches 64 distinct rows, repeated
        reps = B // 64
        Xr, Yr = np.tile(X[:64], (reps, 1)), np.tile(Y[:64], (reps, 1))
        del tr
        tr = H.Trainer(net_h, loss, rate, T.put(Xr, batched=True), T.put(Yr, batched=True), use_graph=False,
                       use_fused=fused)
    for _ in range(3):
        g = NN.batched_param_grads(O, oloss, Xs, Ys, NN.Network(net_o.op, params))
        params = [p - rate * scale * gi for p, gi in zip(params, g)]
        tr.step()
    for a, w in zip(tr.net.params, params):
        assert rel_err(a.numpy(), w) < RTOL


@pytest.mark.parametrize("hid,out,loss", [("actMapLogistic", "actSoftmax", "crossEntropy"), ("actLogistic", "actLogistic", "squaredError")])
@pytest.mark.parametrize("fused", [True, False])
def test_a_kept_thunk_graph_issues_the_same_calls_as_fresh_thunks(T, H, hid, out, loss, fused):
    """A directly-issued step evaluates gradTOp's thunk graph, built once and kept (LT::Graph, host/tensorops/tensor.hpp),
    instead of building it on every step as the reference's evaluator does (TOH_TRAINER_FRESH_THUNKS): the same
    class-method calls (counted at the C ABI), the same launches, bit-identical parameters after five steps."""
    import ctypes as C
    from tensor_ops_amd import capi

    def calls():
        n = C.c_int64()
        capi.check(capi.lib().to_api_time(None, C.byref(n)))
        return n.value

    ws, net_o, net_h = _nets(T, H, [20, 12, 5], hid, out)
    X, Y = _batch(16, 20, 5)
    dX, dY = T.put(X, batched=True), T.put(Y, batched=True)
    res = {}
    for fresh in (False, True):
        tr = H.Trainer(net_h, loss, 0.02, dX, dY, use_graph=False, use_fused=fused, fresh_thunks=fresh)
        tr.step()
        T.sync()
        c0, l0 = calls(), T.stats()["launches"]
        for _ in range(4):
            tr.step()
        T.sync()
        res[fresh] = (calls() - c0, T.stats()["launches"] - l0, [p.numpy() for p in tr.net.params])
        del tr
    assert res[False][0] == res[True][0] and res[False][1] == res[True][1], (res[False][:2], res[True][:2])
    for a, b in zip(res[False][2], res[True][2]):
        assert np.array_equal(a, b)


def test_memo_removes_the_forward_recomputation(T, H):
    """Types.hs:155 recomputes f1 xs per composition node; inside a memo scope the
    repeated pure calls are cache hits, so the step launches fewer kernels."""
    ws, net_o, net_h = _nets(T, H, [20, 12, 5], "actMapLogistic", "actSoftmax")
    X, Y = _batch(16, 20, 5)
    dX, dY = T.put(X, batched=True), T.put(Y, batched=True)
    with_memo = H.Trainer(net_h, "crossEntropy", 0.02, dX, dY, use_memo=True, use_graph=False, use_fused=False)
    without = H.Trainer(net_h, "crossEntropy", 0.02, dX, dY, use_memo=False, use_graph=False, use_fused=False)
    assert with_memo.launches_per_step < without.launches_per_step


def test_batched_inference_matches_per_sample_runNetwork(T, H):
    """runNetwork on a batch = per-sample runNetwork (FeedForward.hs:123-129); argMax of the
    batch = the reference's validation loop (app/MNIST.hs:368-389)."""
    ws, net_o, net_h = _nets(T, H, [20, 12, 5], "actMapLogistic", "actSoftmax")
    X, _ = _batch(40, 20, 5)
    out = H.runNetwork(net_h, T.put(X, batched=True))
    want = np.stack([NN.runNetwork(O, net_o, x) for x in X])
    assert out.batch == 40 and rel_err(out.numpy(), want) < RTOL
    assert np.array_equal(T.arg_max(out), [O.arg_max(r) for r in want])


def test_dots_app_runs_on_the_hip_backend(repo_root):
    """tensor-ops-dots (app/Dots.hs:60-92) against the C++ host mirror: per-sample online SGD,
    then the 51x21 ASCII map from one batched runNetwork."""
    import os
    import subprocess
    exe = os.path.join(repo_root, "tensor-ops_amd", "tensor-ops-dots-hip")
    assert os.path.exists(exe), "build.py builds the app"
    accs = {}
    for prec in ("--f64", "--f32"):   # ElemT t ~ Double as in the reference (Dots.hs:49), and the fp32 instance
        out = subprocess.run([exe, "--samps", "4000", "--layers", "8", prec], capture_output=True, text=True,
                             timeout=300)
        assert out.returncode == 0, out.stderr
        lines = out.stdout.splitlines()
        rows = [l for l in lines if len(l) == 51 and set(l) <= set(" .-=#")]
        assert len(rows) == 21
        accs[prec] = float([l for l in lines if l.startswith("grid accuracy")][0].split(":")[1])
        assert 0.6 < accs[prec] <= 1.0
    # same samples, same initial weights, same update order: the two precisions learn the same map
    assert abs(accs["--f64"] - accs["--f32"]) < 0.05


def test_shared_weights_accumulate_like_a_tape(T, H):
    """Weights used at several time steps (an unrolled recurrent net built only from the op
    vocabulary): the cotangents of the shared W are accumulated with sumT, the reverse-mode
    'tape accumulation' of the north star (TOp.hs:106-131 shuffle, :287-302 replicate)."""
    n, steps = 6, 3
    W, h0 = rnd(n, n) * 0.5, rnd(n)
    xs = [rnd(n) for _ in range(steps)]

    def build(V, first, tanh_op, shuffle):
        # inputs: [h, W, x1..xT]; every step: h' = tanh(W h + x_t) with the SAME W
        op = None
        for t in range(steps):
            n_rest = steps - t - 1
            # [h, W, x_t, rest...] -> shuffle to [W, h, x_t, W, rest...]
            idx = [1, 0, 2, 1] + [3 + i for i in range(n_rest)]
            step = shuffle(idx, 3 + n_rest)
            step = step >> first(V.matVec(), 2 + n_rest)          # [W h, x_t, W, rest]
            step = step >> first(V.add(), 1 + n_rest)             # [W h + x_t, W, rest]
            step = step >> first(tanh_op, 1 + n_rest)
            op = step if op is None else op >> step
        return op                                                  # outputs [h_T, W]
    def o_shuffle(idx, k):  # the oracle's shuffle takes the `SingI ns` shapes explicitly
        shapes = [(n,), (n, n)] + [(n,)] * (k - 2)
        return TO.shuffle(idx, shapes)
    o_op = build(TO, TO.first, TO.map_(ad.tanh), o_shuffle)
    h_op = build(H, H.firstOp, H.map_(ad.tanh), lambda idx, k: H.shuffle(idx, k))
    inputs = [h0, W] + xs
    both(T, h_op, o_op, inputs)


def test_explicit_gradient_forms(T, H):
    """`zipN'`, `zip'`, `zip3'`, `map'` (TOp.hs:198-285): the caller supplies the derivative instead of `ad`;
    same values as the automatically differentiated forms and as the oracle."""
    f2 = lambda x, y: x * y + ad.sin(x)                                    # noqa: E731
    g2 = lambda x, y: (y + ad.cos(x), x)                                   # noqa: E731
    both(T, H.zip_with(f2, g2), TO.zip_(f2), [rnd(7), rnd(7)])
    both(T, H.zip_with(f2, g2), TO.zip_(f2, g2), [rnd(3, 4), rnd(3, 4)])
    f3 = lambda x, y, z: x * y / (2.0 + z * z)                             # noqa: E731
    g3 = lambda x, y, z: (y / (2.0 + z * z), x / (2.0 + z * z),            # noqa: E731
                          -2.0 * x * y * z / ((2.0 + z * z) * (2.0 + z * z)))
    both(T, H.zip3_with(f3, g3), TO.zip3(f3), [rnd(6), rnd(6), rnd(6)])
    fn = lambda v: v[0] * v[1] - v[2] * v[3]                               # noqa: E731
    gn = lambda v: [v[1], v[0], -v[3], -v[2]]                              # noqa: E731
    both(T, H.zipN_with(4, fn, gn), TO.zipN(4, fn), [rnd(5), rnd(5), rnd(5), rnd(5)])
    both(T, H.map_(NN.logistic, NN.logistic_prime), TO.map_(NN.logistic), [rnd(9)])
    # a deliberately WRONG gradient is used as given (the reference trusts `zipN'`'s second argument too)
    wrong = H.zip_with(lambda x, y: x * y, lambda x, y: (x, y))
    xs = [T.put(rnd(4)), T.put(rnd(4))]
    d = T.put(np.ones(4))
    gx, gy = wrong.grad(xs, [d])
    assert rel_err(gx.numpy(), xs[0].numpy()) < RTOL and rel_err(gy.numpy(), xs[1].numpy()) < RTOL


def test_feedforward_combinators(T, H):
    """buildNet / liftNet / ~*~ / ~* / *~ / nmap / networkGradient (FeedForward.hs:68-176)."""
    ws = _weights([6, 5, 4])
    x, y = rnd(6), RNG.uniform(0.1, 0.9, 4)
    # (scale 2 ~* ffLayer) ~*~ liftNet (map tanh) ~*~ nmap logistic ffLayer
    o1 = NN.then_net(TO.scale(2.0), NN.ffLayer(*ws[0]))
    net_o = NN.seq_net(NN.seq_net(o1, NN.liftNet(TO.map_(ad.tanh))), NN.nmap(NN.logistic, NN.ffLayer(*ws[1])))
    l1 = H.buildNet(H.firstOp(H.named("swap") >> H.matVec(), 1) >> H.add(), [T.put(ws[0][0]), T.put(ws[0][1])])
    h1 = H.net_after(H.scale(2.0), l1)
    l2 = H.genNet([(T.put(ws[1][0]), T.put(ws[1][1]))], "actLogistic", "actLogistic")   # ffLayer *~ logistic
    net_h = H.net_seq(H.net_seq(h1, H.liftNet(H.map_(ad.tanh))), l2)
    assert len(net_h.params) == 4
    assert rel_err(H.runNetwork(net_h, T.put(x)).numpy(), NN.runNetwork(O, net_o, x)) < RTOL
    want = NN.networkGradient(O, NN.squaredError(), x, y, net_o)
    got = H.networkGradient(net_h, "squaredError", T.put(x), T.put(y))
    assert len(got) == len(want) == 4
    for a, b in zip(got, want):
        assert rel_err(a.numpy(), b) < RTOL
    # *~ with an arbitrary op, then the batched trainer on the composed (non-genNet) network: generic path
    net_h2 = H.net_then(net_h, H.scale(0.5))
    net_o2 = NN.net_then(net_o, TO.scale(0.5))
    X, Y = RNG.uniform(-1, 1, (17, 6)), RNG.uniform(0.1, 0.9, (17, 4))
    tr = H.Trainer(net_h2, "squaredError", 0.1, T.put(X, batched=True), T.put(Y, batched=True))
    assert tr.fused   # the library fuses whatever it recognises in ANY network; correctness is what is checked below
    tr.grad()
    before = [p.numpy() for p in tr.net.params]
    tr.apply()
    wantb = NN.batched_param_grads(O, NN.squaredError(), list(X), list(Y), net_o2)
    for b, a, w in zip(before, tr.net.params, wantb):
        assert rel_err(a.numpy(), b.astype(np.float64) - 0.1 * w) < RTOL
