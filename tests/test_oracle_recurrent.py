"""Oracle self-checks for the Recurrent / AutoEncoder restatements (CPU): the BPTT gradient of the
unrolled TOp (Recurrent.hs:265-324, :392-463) against central finite differences of the forward
loss, the order conventions, and the exported host-mirror symbols."""
import os
import re
import subprocess

import numpy as np

from oracle import autoencoder as AE, neuralnet as NN, recurrent as R, top as TO
from oracle.tensor import OTensor

T = OTensor(np.float64)
RNG = np.random.default_rng(0x7e500003)


def fc_vals(o, i):
    return (0.5 * RNG.standard_normal(o), 0.5 * RNG.standard_normal((o, o)),
            0.5 * RNG.standard_normal((o, i)), 0.5 * RNG.standard_normal(o))


def ff_vals(o, i):
    return (0.5 * RNG.standard_normal((o, i)), 0.5 * RNG.standard_normal(o))


def fd(f, p, eps=1e-6):
    g = np.zeros_like(p)
    for idx in np.ndindex(p.shape):
        a, b = p.copy(), p.copy()
        a[idx] += eps
        b[idx] -= eps
        g[idx] = (f(a) - f(b)) / (2 * eps)
    return g


def _check_net(net, n, i, o, loss):
    xs = [RNG.uniform(-1, 1, i) for _ in range(n)]
    ys = [RNG.uniform(0.1, 0.9, o) for _ in range(n)]
    gI, gS, gP = R.netGrad(T, loss, xs, ys, net)
    assert len(gI) == n and len(gS) == net.n_s and len(gP) == net.n_p
    for k, p in enumerate(net.params):
        def f(q, k=k):
            ps = list(net.params)
            ps[k] = q
            return R.total_loss(T, loss, xs, ys, R.Network(net.op, net.state, ps, net.i_shape))
        np.testing.assert_allclose(gP[k], fd(f, p), rtol=1e-6, atol=1e-8)
    for k, s in enumerate(net.state):
        def f(q, k=k):
            ss = list(net.state)
            ss[k] = q
            return R.total_loss(T, loss, xs, ys, R.Network(net.op, ss, net.params, net.i_shape))
        np.testing.assert_allclose(gS[k], fd(f, s), rtol=1e-6, atol=1e-8)
    for t in range(n):  # input cotangents come back in REVERSED time order (Recurrent.hs:283)
        def f(q, t=t):
            x2 = list(xs)
            x2[t] = q
            return R.total_loss(T, loss, x2, ys, net)
        np.testing.assert_allclose(gI[n - 1 - t], fd(f, xs[t]), rtol=1e-6, atol=1e-8)


def test_single_fullyConnected_layer_bptt():
    net = R.net_then(R.fullyConnected(NN.actLogistic, *fc_vals(3, 2)), NN.actLogistic())
    _check_net(net, 4, 2, 3, NN.squaredError())


def test_two_recurrent_layers_and_a_stateless_one():
    net = R.genNet([(fc_vals(4, 3), NN.actLogistic, NN.actLogistic),
                    (ff_vals(5, 4), lambda: NN.actMap(NN.logistic), None)],
                   (fc_vals(2, 5), NN.actLogistic), NN.actSoftmax)
    assert net.n_s == 2 and net.n_p == 8
    # `~*~` keeps states as ss2 ++ ss1 and params as ps1 ++ ps2 (Recurrent.hs:216-221)
    assert [np.shape(s) for s in net.state] == [(2,), (4,)]
    assert [np.shape(p) for p in net.params][:3] == [(4, 4), (4, 3), (4,)]
    _check_net(net, 3, 3, 2, NN.crossEntropy())


def test_zero_and_one_step_sequences():
    net = R.net_then(R.fullyConnected(NN.actLogistic, *fc_vals(2, 2)), NN.actLogistic())
    gI, gS, gP = R.netGrad(T, NN.squaredError(), [], [], net)      # rollup Z_ = konst 0 (:441)
    assert gI == [] and all(np.all(np.asarray(g) == 0) for g in gS + gP)
    _check_net(net, 1, 2, 2, NN.squaredError())                     # rollup (S_ Z_) = loss (:442)


def test_runNetwork_threads_the_state():
    s, wS, w, b = fc_vals(3, 2)
    net = R.fullyConnected(NN.actLogistic, s, wS, w, b)
    x0, x1 = RNG.uniform(-1, 1, 2), RNG.uniform(-1, 1, 2)
    y0, n1 = R.runNetwork(T, net, x0)
    z0 = w @ x0 + wS @ s + b
    np.testing.assert_allclose(y0, z0, rtol=1e-14)                  # output = pre-activation (:108-118)
    np.testing.assert_allclose(n1.state[0], 1 / (1 + np.exp(-z0)), rtol=1e-14)
    y1, _ = R.runNetwork(T, n1, x1)
    np.testing.assert_allclose(y1, w @ x1 + wS @ (1 / (1 + np.exp(-z0))) + b, rtol=1e-14)


def test_trainNetwork_uses_both_rates():
    net = R.net_then(R.fullyConnected(NN.actLogistic, *fc_vals(3, 2)), NN.actLogistic())
    xs = [RNG.uniform(-1, 1, 2) for _ in range(3)]
    ys = [RNG.uniform(0, 1, 3) for _ in range(3)]
    _, gS, gP = R.netGrad(T, NN.squaredError(), xs, ys, net)
    new = R.trainNetwork(T, NN.squaredError(), 0.3, 0.05, xs, ys, net)
    np.testing.assert_allclose(new.state[0], net.state[0] - 0.3 * gS[0], rtol=1e-14)
    for a, p, g in zip(new.params, net.params, gP):
        np.testing.assert_allclose(a, p - 0.05 * g, rtol=1e-14)


def test_autoencoder_gradient_and_objective():
    enc = NN.genNet([ff_vals(3, 6)], NN.actLogistic, NN.actLogistic)
    dec = NN.genNet([ff_vals(6, 3)], NN.actLogistic, NN.actLogistic)
    e = AE.Encoder(enc, dec)
    x = RNG.uniform(0, 1, 6)
    loss = NN.squaredError()
    l0 = float(AE.testEncoder(T, loss, e, x))
    assert abs(l0 - float(np.sum((AE.encodeDecode(T, e, x) - x) ** 2))) < 1e-14
    np.testing.assert_allclose(AE.decode(T, e, AE.encode(T, e, x)), AE.encodeDecode(T, e, x), rtol=1e-14)
    g_e, g_d = AE.encGrad(T, loss, x, e)
    for side, grads in (("enc", g_e), ("dec", g_d)):
        net = getattr(e, side)
        for k, p in enumerate(net.params):
            def f(q, k=k, side=side, net=net):
                ps = list(net.params)
                ps[k] = q
                n2 = NN.Network(net.op, ps)
                e2 = AE.Encoder(n2, e.dec) if side == "enc" else AE.Encoder(e.enc, n2)
                return float(AE.testEncoder(T, loss, e2, x))
            np.testing.assert_allclose(grads[k], fd(f, p), rtol=1e-6, atol=1e-9)
    e2 = AE.trainEncoder(T, loss, 0.1, x, e)
    np.testing.assert_allclose(e2.enc.params[0], enc.params[0] - 0.1 * g_e[0], rtol=1e-14)
    assert float(AE.testEncoder(T, loss, e2, x)) < l0


def test_swap_n_is_its_own_inverse_in_the_backward():
    op = TO.swap_n(2, 3)
    xs = [np.full((1,), float(i)) for i in range(5)]
    ys = TO.runTOp(op, T, xs)
    assert [float(y[0]) for y in ys] == [2, 3, 4, 0, 1]
    back = op.grad(T, xs, ys)
    assert [float(y[0]) for y in back] == [0, 1, 2, 3, 4]


def test_host_mirror_exports_every_declared_symbol(repo_root):
    hdr = open(os.path.join(repo_root, "tensor-ops_amd", "host", "tensorops_host.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(toh_[A-Za-z0-9_]+)\s*\(", hdr)))
    lib = os.path.join(repo_root, "tensor-ops_amd", "libtensorops_host.so")
    if not os.path.exists(lib):
        import importlib.util
        spec = importlib.util.spec_from_file_location("_b", os.path.join(repo_root, "tensor-ops_amd", "build.py"))
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
        b.build()
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib]).decode()
    exported = set(re.findall(r" T (toh_[A-Za-z0-9_]+)", out))
    assert not [n for n in names if n not in exported]
    from tensor_ops_amd import tops
    assert set(tops.SIGNATURES) | {"toh_last_error"} == set(names)
