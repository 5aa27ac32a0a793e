"""Per-sample online SGD -- `foldl' (\\nt (i,o) -> trainNetwork loss rate i o nt)`, the reference's actual training loop
(app/MNIST.hs:390-396, app/Dots.hs:74-80) -- as one persistent launch (csrc/online_sgd.hip).

Checked against a per-sample loop written out in numpy/fp64 (forward, the two loss heads, backward, `p - r*g` after EVERY
sample; the C oracle's `hmat_train_online` covers the two-layer case in tests/test_mnist_app.py): the program-level entry
point on stacks of 2..5 layers, and `trainAll` of the host mirror, where nothing tells the library what the network is --
it recognises the captured one-sample step from the launches its own planner made of it."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(scope="module")
def T():
    from tensor_ops_amd.hipt import HipT
    return HipT(0)


def rel_err(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.linalg.norm((got - want).ravel()) / max(np.linalg.norm(want.ravel()), 1e-300)


def online_ref(ws, X, Y, order, rate, head):
    """the reference's loop, fp64: one trainNetwork step per sample"""
    ws = [(w.astype(np.float64).copy(), b.astype(np.float64).copy()) for w, b in ws]
    for s in order:
        a = [X[s].astype(np.float64)]
        for l, (w, b) in enumerate(ws):
            z = w @ a[-1] + b
            a.append(z if l == len(ws) - 1 else 1 / (1 + np.exp(-z)))
        z, y = a[-1], Y[s].astype(np.float64)
        if head == "softmax":
            e = np.exp(z - z.max())
            dz = e / e.sum() * y.sum() - y          # softmax >>> crossEntropy
        else:
            sg = 1 / (1 + np.exp(-z))
            dz = -2 * (y - sg) * sg * (1 - sg)      # logistic >>> squaredError
        for l in range(len(ws) - 1, -1, -1):
            w, b = ws[l]
            dprev = (w.T @ dz) * a[l] * (1 - a[l]) if l > 0 else None
            ws[l] = (w - rate * np.outer(dz, a[l]), b - rate * dz)
            dz = dprev
    return ws


def problem(sizes, N, seed, onehot=True):
    rng = np.random.default_rng(seed)
    ws = [(0.5 * rng.standard_normal((o, i)).astype(np.float32), 0.5 * rng.standard_normal(o).astype(np.float32))
          for i, o in zip(sizes[:-1], sizes[1:])]
    X = rng.uniform(0, 1, (N, sizes[0])).astype(np.float32)
    if onehot:
        Y = np.zeros((N, sizes[-1]), np.float32)
        Y[np.arange(N), rng.integers(0, sizes[-1], N)] = 1
    else:
        Y = rng.uniform(0.05, 0.95, (N, sizes[-1])).astype(np.float32)
    return ws, X, Y, rng


def run_entry(T, ws, X, Y, order, rate, head):
    from tensor_ops_amd import capi
    dw, db = [T.put(w) for w, _ in ws], [T.put(b) for _, b in ws]
    wa = (capi.c_tensor * len(ws))(*[t.h for t in dw])
    ba = (capi.c_tensor * len(ws))(*[t.h for t in db])
    idx = (C.c_int64 * max(len(order), 1))(*[int(v) for v in order]) if order is not None else None
    n = len(order) if order is not None else len(X)
    out_act, loss = (2, 1) if head == "softmax" else (0, 0)
    dX, dY = T.put(X, batched=True), T.put(Y, batched=True)
    st = capi.lib().to_fflayer_stack_online_sgd(len(ws), wa, ba, 0, out_act, loss, dX.h, dY.h, n, idx, rate)
    return st, dw, db


CASES = [
    ([784, 300, 100, 10], "softmax", 0.02, 400),     # the app's default stack (app/MNIST.hs:89-133)
    ([784, 256, 10], "softmax", 0.02, 300),          # BASELINE config 3's network, per sample
    ([2, 16, 1], "logistic", 1.0, 300),              # BASELINE config 1 (tensor-ops-dots)
    ([2, 12, 8, 1], "logistic", 1.0, 300),           # Dots' default hidden layers (app/Dots.hs:121-123)
    ([30, 20, 16, 12, 6], "logistic", 0.05, 200),
    ([40, 33, 7], "softmax", 0.1, 100),              # rows that do not divide over the workgroups
    ([64, 3, 5, 4], "softmax", 0.1, 60),             # fewer rows than workgroups
]


@pytest.mark.parametrize("sizes,head,rate,n", CASES, ids=lambda v: "x".join(map(str, v)) if isinstance(v, list) else None)
def test_entry_point_is_the_per_sample_loop(T, sizes, head, rate, n):
    ws, X, Y, rng = problem(sizes, n + 50, 11 + len(sizes), onehot=head == "softmax")
    order = rng.permutation(len(X))[:n]
    want = online_ref(ws, X, Y, order, rate, head)
    st, dw, db = run_entry(T, ws, X, Y, order, rate, head)
    assert st == 0
    for (w, b), gw, gb in zip(want, dw, db):
        assert rel_err(gw.numpy(), w) < RTOL and rel_err(gb.numpy(), b) < RTOL
    # the rows in their own order (idx = NULL), and a second run gives the same bits (workgroup-ordered sums)
    st1, dw1, _ = run_entry(T, ws, X, Y, None, rate, head)
    st2, dw2, _ = run_entry(T, ws, X, Y, None, rate, head)
    assert st1 == 0 and st2 == 0 and all(np.array_equal(a.numpy(), b.numpy()) for a, b in zip(dw1, dw2))
    want = online_ref(ws, X, Y, range(len(X)), rate, head)
    assert rel_err(dw1[0].numpy(), want[0][0]) < RTOL


def test_entry_point_refuses_what_the_kernel_cannot_hold(T):
    from tensor_ops_amd import capi
    for sizes, head in (([3000, 20, 5], "softmax"),      # input beyond the prefetch registers
                        ([20, 30, 100], "softmax"),      # head wider than one wave
                        ([20, 5], "softmax"),            # a single layer
                        ([600, 2000, 600, 5], "softmax")):   # replicated layers beyond the LDS
        ws, X, Y, _ = problem(sizes, 8, 3)
        st, dw, db = run_entry(T, ws, X, Y, None, 0.1, head)
        assert st != 0 and b"online SGD kernel" in capi.lib().to_last_error()
        for (w, b), gw, gb in zip(ws, dw, db):            # parameters untouched
            assert np.array_equal(gw.numpy(), w) and np.array_equal(gb.numpy(), b)


def _stats():
    from tensor_ops_amd import capi
    a, b = C.c_int64(), C.c_int64()
    capi.check(capi.lib().to_online_sgd_stats(C.byref(a), C.byref(b)))
    return a.value, b.value


@pytest.mark.parametrize("sizes,hidden,out,loss,head,recognised", [
    ([784, 300, 100, 10], "actMapLogistic", "actSoftmax", "crossEntropy", "softmax", True),
    ([2, 12, 8, 1], "actLogistic", "actLogistic", "squaredError", "logistic", True),
    ([30, 14, 6], "actMapLogistic", "actLogistic", "squaredError", "logistic", True),
    ([30, 14, 6], "actMapTanh", "actSoftmax", "crossEntropy", None, False),         # tanh: not this kernel's stack
    ([30, 14, 20], "actMapLogistic", "actSoftmax", "crossEntropy", "softmax", False),  # 20 outputs: no fused loss head
])
def test_trainAll_finds_the_stack_in_its_own_plan(T, sizes, hidden, out, loss, head, recognised):
    """`trainAll` of the mirror captures ONE step of the reference's `trainNetwork` on a `Network` without tags; the
    library decides from the launches it planned whether that step is an ffLayer stack's.  Recognised or not, the result
    is the per-sample loop's."""
    from oracle import ad, neuralnet as NN
    from oracle.tensor import OTensor
    from tensor_ops_amd import tops as H
    H.hlib()
    n = 120
    ws, X, Y, rng = problem(sizes, n + 20, 23, onehot=loss == "crossEntropy")
    order = rng.permutation(len(X))[:n]
    rate = 0.05
    net = H.genNet([(T.put(w), T.put(b)) for w, b in ws], hidden, out)
    s0 = _stats()
    got = H.trainAll(net, loss, rate, T.put(X, batched=True), T.put(Y, batched=True), order=list(order))
    s1 = _stats()
    assert (s1[0] - s0[0], s1[1] - s0[1]) == ((1, n) if recognised else (0, 0))
    if head is not None:
        want = [a for wb in online_ref(ws, X, Y, order, rate, head) for a in wb]
    else:
        O = OTensor(np.float64)
        net_o = NN.genNet([(w.astype(np.float64), b.astype(np.float64)) for w, b in ws], lambda: NN.actMap(ad.tanh), NN.actSoftmax)
        for s in order:
            net_o = NN.trainNetwork(O, NN.crossEntropy(), rate, X[s].astype(np.float64), Y[s].astype(np.float64), net_o)
        want = net_o.params
    for a, w in zip(got.params, want):
        assert rel_err(a.numpy(), w) < RTOL


@pytest.mark.parametrize("sizes,head,rate,n", [([784, 300, 100, 10], "softmax", 0.02, 200), ([2, 12, 8, 1], "logistic", 1.0, 300)],
                         ids=["mnist_stack", "dots_stack"])
def test_entry_point_in_the_references_precision(sizes, head, rate, n):
    """ElemT = Double (`HMat Double`, BLAS/HMat.hs:35): the same kernel instantiated for fp64, 1e-11 against the loop"""
    from tensor_ops_amd import capi
    from tensor_ops_amd.hipt import HipT
    T64 = HipT(0, dtype=np.float64)
    ws, X, Y, rng = problem(sizes, n + 10, 31, onehot=head == "softmax")
    ws = [(w.astype(np.float64), b.astype(np.float64)) for w, b in ws]
    X, Y = X.astype(np.float64), Y.astype(np.float64)
    order = rng.permutation(len(X))[:n]
    want = online_ref(ws, X, Y, order, rate, head)
    st, dw, db = run_entry(T64, ws, X, Y, order, rate, head)
    assert st == 0, capi.lib().to_last_error()
    for (w, b), gw, gb in zip(want, dw, db):
        assert gw.numpy().dtype == np.float64
        assert rel_err(gw.numpy(), w) < 1e-11 and rel_err(gb.numpy(), b) < 1e-11
