"""The plain-C HMat-path restatement (oracle/hmat_path.c) against the independent numpy
restatement (oracle/neuralnet.py): values, and the per-sample primitive counts the C
file hard-codes for the reference's recompute behaviour (Types.hs:155 + laziness)."""
import collections

import numpy as np

from oracle import hmat, neuralnet as NN
from oracle.tensor import OTensor

RNG = np.random.default_rng(0x7e500001)


def _setup(i, h, o, B):
    ws = [(0.5 * RNG.standard_normal((h, i)), 0.5 * RNG.standard_normal(h)),
          (0.5 * RNG.standard_normal((o, h)), 0.5 * RNG.standard_normal(o))]
    X = RNG.uniform(0, 1, size=(B, i))
    Y = np.zeros((B, o))
    Y[np.arange(B), RNG.integers(0, o, size=B)] = 1.0
    net = NN.genNet(ws, lambda: NN.actMap(NN.logistic), NN.actSoftmax)
    return ws, X, Y, net


def test_c_batched_grads_equal_numpy_oracle():
    ws, X, Y, net = _setup(23, 11, 5, 9)
    T = OTensor(np.float64)
    want = NN.batched_param_grads(T, NN.crossEntropy(), list(X), list(Y), net)
    for rec in (True, False):
        got, loss = hmat.batched_grads(X, Y, ws[0][0], ws[0][1], ws[1][0], ws[1][1], recompute=rec)
        for a, b in zip(got, want):
            np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-13)
        assert abs(loss - NN.batched_losses(T, NN.crossEntropy(), list(X), list(Y), net).sum()) < 1e-10


def test_c_online_sgd_equals_numpy_oracle():
    ws, X, Y, net = _setup(13, 7, 4, 6)
    T = OTensor(np.float64)
    for x, y in zip(X, Y):
        net = NN.trainNetwork(T, NN.crossEntropy(), 0.02, x, y, net)
    got, _ = hmat.train_online(X, Y, ws[0][0], ws[0][1], ws[1][0], ws[1][1], 0.02)
    for a, b in zip(got, net.params):
        np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-13)


def test_c_recompute_counts_match_traced_oracle():
    i, h, o = 7, 5, 3

    class Counting(OTensor):
        def __init__(self):
            super().__init__(np.float64)
            self.c = collections.Counter()

        def gmul(self, lm, lo, ln, x, y):
            self.c[("gmul", lm, lo, ln, np.shape(x), np.shape(y))] += 1
            return super().gmul(lm, lo, ln, x, y)

        def liftT(self, f, xs):
            self.c[("lift", len(xs), np.shape(xs[0]))] += 1
            return super().liftT(f, xs)

        def sumT(self, xs, sh):
            self.c[("sumT", len(xs), tuple(sh))] += 1
            return super().sumT(xs, sh)

        def sumRows(self, x):
            self.c[("sumRows", np.shape(x))] += 1
            return super().sumRows(x)
    ws, X, Y, net = _setup(i, h, o, 1)
    T = Counting()
    NN.netGrad(T, NN.crossEntropy(), X[0], Y[0], net)
    c, k = T.c, hmat.call_counts()
    assert c[("gmul", 1, 1, 0, (h, i), (i,))] == k["gemv_l1"] == 3
    assert c[("sumT", 2, (h,))] == k["add_b1"]
    assert c[("lift", 1, (h,))] == k["logistic"]
    assert c[("gmul", 1, 1, 0, (o, h), (h,))] == k["gemv_l2"]
    assert c[("sumT", 2, (o,))] == k["add_b2"] + 1          # + duplicate's backward sumT
    assert c[("lift", 1, (o,))] == k["exp"] + k["log"]
    assert c[("sumRows", (o,))] == k["sum_rows"]
    assert c[("lift", 1, ())] == k["recip"]
    assert c[("gmul", 0, 0, 1, (), (o,))] == k["scale_sv"] + 2  # + two backward scalar*vector


def test_c_gemm_and_map():
    A, B = RNG.integers(-3, 4, size=(5, 7)).astype(float), RNG.integers(-3, 4, size=(7, 4)).astype(float)
    assert np.array_equal(hmat.gemm(A, B), A @ B)
    x = RNG.uniform(-3, 3, size=100)
    np.testing.assert_allclose(hmat.map_logistic(x), 1 / (1 + np.exp(-x)), rtol=1e-15)


def test_two_independent_restatements_agree_on_config3():
    """oracle/hmat_path.c (the reference's per-sample BLAS sequence, restated) against tests/closed_form.py
    (textbook batched backprop, written separately): gradients and loss of the config-3 network agree to fp64
    round-off, with one-hot AND with general targets; so do the op-by-op Python oracle and the closed form on
    the logistic/squaredError head."""
    import numpy as np
    from oracle import hmat, neuralnet as NN
    from oracle.tensor import OTensor
    from tests.closed_form import logistic_se_grads, softmax_ce_grads
    rng = np.random.default_rng(0x7e500021)
    i, h, o, B = 784, 256, 10, 96
    W1, b1 = 0.5 * rng.standard_normal((h, i)), 0.5 * rng.standard_normal(h)
    W2, b2 = 0.5 * rng.standard_normal((o, h)), 0.5 * rng.standard_normal(o)
    X = rng.uniform(0, 1, (B, i))
    for Y in (np.eye(o)[rng.integers(0, o, B)], rng.uniform(0.05, 1.0, (B, o))):
        g_c, loss_c = hmat.batched_grads(X, Y, W1, b1, W2, b2, recompute=False)
        g_f, loss_f = softmax_ce_grads(X, Y, W1, b1, W2, b2)
        assert abs(loss_c - loss_f) <= 1e-11 * abs(loss_f)
        for a, b in zip(g_c, g_f):
            assert np.linalg.norm((a - b).ravel()) <= 1e-11 * np.linalg.norm(b.ravel())
    O = OTensor(np.float64)
    ws = [(W1[:12, :20], b1[:12]), (W2[:5, :12], b2[:5])]
    Xs, Ys = X[:9, :20], rng.uniform(0.1, 0.9, (9, 5))
    net = NN.genNet(ws, NN.actLogistic, NN.actLogistic)
    want = NN.batched_param_grads(O, NN.squaredError(), list(Xs), list(Ys), net)
    got, _ = logistic_se_grads(Xs, Ys, ws[0][0], ws[0][1], ws[1][0], ws[1][1])
    for a, b in zip(got, want):
        assert np.linalg.norm((a - np.asarray(b)).ravel()) <= 1e-11 * np.linalg.norm(np.asarray(b).ravel())


def test_threaded_baseline_legs_agree_with_the_single_thread_port():
    """BASELINE.md section 3, CPU-B / CPU-D: the all-cores legs bench.py reports are the same arithmetic -- gradients to
    fp64 round-off of the summation order, the fp32 map to one ulp of libm's expf."""
    from oracle import hmat
    rng = np.random.default_rng(3)
    i, h, o, B = 50, 20, 7, 37
    W1, b1, W2, b2 = rng.standard_normal((h, i)), rng.standard_normal(h), rng.standard_normal((o, h)), rng.standard_normal(o)
    X = rng.uniform(0, 1, (B, i))
    Y = np.zeros((B, o))
    Y[np.arange(B), rng.integers(0, o, B)] = 1
    g1, l1 = hmat.batched_grads(X, Y, W1, b1, W2, b2)
    for threads in (1, 3, 8, 64):
        gt, lt = hmat.batched_grads_mt(X, Y, W1, b1, W2, b2, threads)
        assert abs(lt - l1) <= 1e-12 * abs(l1)
        for a, b in zip(g1, gt):
            assert np.abs(a - b).max() <= 1e-13 * np.abs(a).max()
    # the persistent pool (several batches per call) gives the same sums
    for threads, reps in ((1, 1), (4, 3), (16, 2)):
        gp, lp = hmat.batched_grads_pool(X, Y, W1, b1, W2, b2, threads, reps)
        assert abs(lp - l1) <= 1e-12 * abs(l1)
        for a, b in zip(g1, gp):
            assert np.abs(a - b).max() <= 1e-13 * np.abs(a).max()
    # the fp32 build of the same text (the like-for-like CPU figure beside the fp32 GPU step): fp32 round-off of the fp64 one
    g32, l32 = hmat.batched_grads_f32(X, Y, W1, b1, W2, b2)
    assert all(g.dtype == np.float32 for g in g32) and abs(l32 - l1) <= 1e-4 * abs(l1)
    for a, b in zip(g1, g32):
        assert np.abs(a - b).max() <= 1e-4 * np.abs(a).max()
    x = rng.uniform(-6, 6, 100003).astype(np.float32)
    for threads in (1, 5):
        y = hmat.map_logistic_f32(x, threads)
        assert np.abs(y - 1 / (1 + np.exp(-x.astype(np.float64)))).max() < 2e-7


def _stack(sizes, B, seed):
    rng = np.random.default_rng(seed)
    ws = [(0.5 * rng.standard_normal((o, i)), 0.5 * rng.standard_normal(o)) for i, o in zip(sizes, sizes[1:])]
    X = rng.uniform(0, 1, size=(B, sizes[0]))
    Y = np.zeros((B, sizes[-1]))
    Y[np.arange(B), rng.integers(0, sizes[-1], size=B)] = 1.0
    return ws, X, Y


def test_c_stack_trainer_equals_numpy_oracle_at_any_depth():
    """hmat_train_online_stack (the app's 784 -> 300 -> 100 -> 10 shape class, bench.py's `online_sgd.cpu_baseline`):
    per-sample trainNetwork on 2-, 3- and 4-layer stacks against the op-by-op numpy oracle, with and without the
    reference's forward recomputation (same values: the recomputed passes write the same outputs); and the 2-layer
    case against the older two-layer entry point."""
    for sizes in ([13, 7, 4], [11, 9, 6, 4], [10, 9, 7, 5, 3]):
        ws, X, Y = _stack(sizes, 6, 40 + len(sizes))
        T = OTensor(np.float64)
        net = NN.genNet(ws, lambda: NN.actMap(NN.logistic), NN.actSoftmax)
        for x, y in zip(X, Y):
            net = NN.trainNetwork(T, NN.crossEntropy(), 0.05, x, y, net)
        for rec in (True, False):
            got, loss = hmat.train_online_stack(X, Y, ws, 0.05, recompute=rec)
            flat = [a for wb in got for a in wb]
            assert len(flat) == len(net.params)
            for a, b in zip(flat, net.params):
                np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-13)
            assert np.isfinite(loss)
    ws, X, Y = _stack([13, 7, 4], 6, 77)
    two, _ = hmat.train_online(X, Y, ws[0][0], ws[0][1], ws[1][0], ws[1][1], 0.05)
    got, _ = hmat.train_online_stack(X, Y, ws, 0.05)
    for a, b in zip([a for wb in got for a in wb], two):
        assert np.array_equal(a, b)
    # the single-precision build of the same text stays within fp32 round-off of it
    got32, _ = hmat.train_online_stack(X, Y, ws, 0.05, f32=True)
    for a, b in zip([a for wb in got32 for a in wb], two):
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-5)


def test_c_stack_recompute_counts_match_traced_oracle():
    """every hidden layer's `W a` / `+ b` three times and `map logistic` twice, the last layer's twice -- at every depth"""
    k = hmat.stack_call_counts()
    for sizes in ([7, 5, 3], [9, 7, 5, 3], [11, 9, 7, 5, 3]):
        ws, X, Y = _stack(sizes, 1, 5)

        class Counting(OTensor):
            def __init__(self):
                super().__init__(np.float64)
                self.c = collections.Counter()

            def gmul(self, lm, lo, ln, x, y):
                self.c[("gmul", lm, lo, ln, np.shape(x), np.shape(y))] += 1
                return super().gmul(lm, lo, ln, x, y)

            def liftT(self, f, xs):
                self.c[("lift", len(xs), np.shape(xs[0]))] += 1
                return super().liftT(f, xs)

            def sumT(self, xs, sh):
                self.c[("sumT", len(xs), tuple(sh))] += 1
                return super().sumT(xs, sh)
        T = Counting()
        NN.netGrad(T, NN.crossEntropy(), X[0], Y[0], NN.genNet(ws, lambda: NN.actMap(NN.logistic), NN.actSoftmax))
        L = len(sizes) - 1
        for l in range(1, L + 1):
            i, o = sizes[l - 1], sizes[l]
            last = l == L
            assert T.c[("gmul", 1, 1, 0, (o, i), (i,))] == (k["last_gemv"] if last else k["hidden_gemv"])
            assert T.c[("sumT", 2, (o,))] == (k["last_add"] + 1 if last else k["hidden_add"])   # (+ duplicate's backward sumT)
            if not last:
                assert T.c[("lift", 1, (o,))] == k["hidden_logistic"]
            assert T.c[("gmul", 1, 0, 1, (o,), (i,))] == 1                                       # one `ger` per layer
            # W^T dz once per layer; layer 1's is the INPUT's cotangent, which `netGrad` keeps (FeedForward.hs:187-198) and this
            # strict oracle therefore computes -- `trainNetwork` drops it unforced (`tail'`, :142), and so does the C file
            assert T.c[("gmul", 1, 1, 0, (i, o), (o,))] == 1


def test_c_classify_is_argmax_of_runNetwork():
    ws, X, _ = _stack([12, 9, 7, 5], 40, 9)
    T = OTensor(np.float64)
    net = NN.genNet(ws, lambda: NN.actMap(NN.logistic), NN.actSoftmax)
    want = [int(np.argmax(NN.runNetwork(T, net, x))) for x in X]
    assert list(hmat.classify_stack(X, ws)) == want
    assert list(hmat.classify_stack(X, ws, f32=True)) == want
