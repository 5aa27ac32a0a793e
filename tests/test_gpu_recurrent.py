"""GPU parity of the host mirror's Recurrent / AutoEncoder layers (Recurrent.hs, AutoEncoder.hs over
the C ABI) against the oracle restatement: BPTT gradients where every parameter is used n times
(cotangent accumulation through `&&&`'s sumT), fp32 at 1e-5 and fp64 at 1e-11, single sequences
and hidden batches of independent sequences (gradients = sums over sequences)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import autoencoder as AE, neuralnet as NN, recurrent as R  # noqa: E402
from oracle.tensor import OTensor  # noqa: E402

O = OTensor(np.float64)
RNG = np.random.default_rng(0x7e500004)


@pytest.fixture(params=["f32", "f64"])
def TH(request):
    """(device backend, host mirror, tolerance) with ElemT set for the duration of the test"""
    from tensor_ops_amd import tops
    from tensor_ops_amd.hipt import HipT
    tops.hlib()
    dt = np.float32 if request.param == "f32" else np.float64
    tops.set_elem_dtype(dt)
    yield HipT(0, dtype=dt), tops, (1e-5 if request.param == "f32" else 1e-11)
    tops.set_elem_dtype(np.float32)


def rel_err(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    den = np.linalg.norm(want.ravel())
    return np.linalg.norm((got - want).ravel()) / (den if den > 0 else 1.0)


def fc_vals(o, i):
    return (0.5 * RNG.standard_normal(o), 0.5 * RNG.standard_normal((o, o)),
            0.5 * RNG.standard_normal((o, i)), 0.5 * RNG.standard_normal(o))


def ff_vals(o, i):
    return (0.5 * RNG.standard_normal((o, i)), 0.5 * RNG.standard_normal(o))


OACT = {"actLogistic": NN.actLogistic, "actMapLogistic": lambda: NN.actMap(NN.logistic),
        "actSoftmax": NN.actSoftmax}


def build(T, H, layers, out_layer, out_act):
    """the same network twice: oracle and host mirror (device values uploaded from the same arrays)"""
    def dev(vals):
        return tuple(T.put(v) for v in vals)
    net_o = R.genNet([(v, OACT[a], OACT[s] if s else None) for v, a, s in layers],
                     (out_layer[0], OACT[out_layer[1]] if out_layer[1] else None), OACT[out_act])
    net_h = H.rnn_genNet([(dev(v), a, s) for v, a, s in layers], (dev(out_layer[0]), out_layer[1]), out_act)
    return net_o, net_h


CASES = [
    # (i, layers [(o, act, state_act|None)], out (o, state_act|None), out_act, loss, n_steps)
    (2, [], (3, "actLogistic"), "actLogistic", "squaredError", 4),
    (3, [(4, "actLogistic", "actLogistic"), (5, "actMapLogistic", None)], (2, "actLogistic"), "actSoftmax",
     "crossEntropy", 3),
    (6, [(8, "actLogistic", "actLogistic")], (4, None), "actSoftmax", "crossEntropy", 5),
    (4, [], (4, "actLogistic"), "actLogistic", "squaredError", 1),
]


def make(T, H, case):
    i, layers, (o, s_out), out_act, loss, n = case
    ls, prev = [], i
    for (w, a, s) in layers:
        ls.append((fc_vals(w, prev) if s else ff_vals(w, prev), a, s))
        prev = w
    out_layer = (fc_vals(o, prev) if s_out else ff_vals(o, prev), s_out)
    net_o, net_h = build(T, H, ls, out_layer, out_act)
    oloss = {"squaredError": NN.squaredError, "crossEntropy": NN.crossEntropy}[loss]()
    return net_o, net_h, oloss, loss, i, o, n


@pytest.mark.parametrize("case", CASES, ids=lambda c: "i%d_n%d_%s" % (c[0], c[5], c[4]))
def test_bptt_single_sequence(TH, case):
    T, H, tol = TH
    net_o, net_h, oloss, loss, i, o, n = make(T, H, case)
    assert [s.shape for s in net_h.state] == [np.shape(s) for s in net_o.state]
    assert [p.shape for p in net_h.params] == [np.shape(p) for p in net_o.params]
    xs = [RNG.uniform(-1, 1, i) for _ in range(n)]
    ys = [RNG.uniform(0.1, 0.9, o) for _ in range(n)]
    # runNetwork threads the state
    cur_o, cur_h = net_o, net_h
    for x in xs:
        yo, cur_o = R.runNetwork(O, cur_o, x)
        yh, cur_h = H.rnn_runNetwork(cur_h, T.put(x))
        assert rel_err(yh.numpy(), yo) < tol
    for a, b in zip(cur_h.state, cur_o.state):
        assert rel_err(a.numpy(), b) < tol
    gI, gS, gP = R.netGrad(O, oloss, xs, ys, net_o)
    dxs, dys = [T.put(x) for x in xs], [T.put(y) for y in ys]
    with T.memo():
        hI, hS, hP = H.rnn_netGrad(net_h, loss, dxs, dys)
    for a, b in zip(hI + hS + hP, gI + gS + gP):
        assert rel_err(a.numpy(), b) < tol
    # trainNetwork' with distinct rates for the initial state and the parameters
    new_o = R.trainNetwork(O, oloss, 0.3, 0.05, xs, ys, net_o)
    new_h = H.rnn_trainNetwork(net_h, loss, 0.3, 0.05, dxs, dys)
    for a, b in zip(new_h.state + new_h.params, new_o.state + new_o.params):
        assert rel_err(a.numpy(), b) < tol


@pytest.mark.parametrize("case", CASES[:3], ids=lambda c: "i%d_n%d_%s" % (c[0], c[5], c[4]))
def test_bptt_hidden_batch_of_sequences(TH, case):
    """B independent sequences under the hidden batch dimension: state and parameter cotangents are
    the sums over sequences (SURVEY.md 8(d) rule), inputs' stay per sequence."""
    T, H, tol = TH
    net_o, net_h, oloss, loss, i, o, n = make(T, H, case)
    B = 7
    xs = [RNG.uniform(-1, 1, (B, i)) for _ in range(n)]
    ys = [RNG.uniform(0.1, 0.9, (B, o)) for _ in range(n)]
    want_s, want_p = R.batched_grads(O, oloss, xs, ys, net_o)
    dxs = [T.put(x, batched=True) for x in xs]
    dys = [T.put(y, batched=True) for y in ys]
    with T.memo():
        hI, hS, hP = H.rnn_netGrad(net_h, loss, dxs, dys)
    for a, b in zip(hS + hP, want_s + want_p):
        assert a.batch == 0
        assert rel_err(a.numpy(), b) < 5 * tol
    # per-sequence input cotangents of sequence 3, reversed time order
    gI, _, _ = R.netGrad(O, oloss, [x[3] for x in xs], [y[3] for y in ys], net_o)
    for a, b in zip(hI, gI):
        assert a.batch == B and rel_err(a.numpy()[3], b) < 5 * tol


def test_launch_count_scales_linearly_with_steps(TH):
    """the unrolled graph is O(n) under the memo scope (CSE of the recomputed prefixes, Types.hs:155);
    without it the nested recomputation grows much faster"""
    T, H, tol = TH
    net_o, net_h, oloss, loss, i, o, _ = make(T, H, CASES[0])

    def launches(n, memo):
        xs = [T.put(RNG.uniform(-1, 1, i)) for _ in range(n)]
        ys = [T.put(RNG.uniform(0, 1, o)) for _ in range(n)]
        H.rnn_netGrad(net_h, loss, xs, ys)  # warm the expression cache
        l0 = T.stats()["launches"]
        if memo:
            with T.memo():
                _, gs, gp = H.rnn_netGrad(net_h, loss, xs, ys, want_inputs=False)
                T.force_many(gs + gp)       # (closing a scope demands nothing: force what the step produces)
        else:
            H.rnn_netGrad(net_h, loss, xs, ys, want_inputs=False)
        return T.stats()["launches"] - l0
    m2, m4, m8 = launches(2, True), launches(4, True), launches(8, True)
    assert m8 - m4 <= 2.2 * (m4 - m2) + 8, (m2, m4, m8)
    assert launches(4, False) > m4


def test_autoencoder(TH):
    T, H, tol = TH
    we, wd = ff_vals(3, 6), ff_vals(6, 3)
    enc_o = NN.genNet([we], NN.actLogistic, NN.actLogistic)
    dec_o = NN.genNet([wd], NN.actLogistic, NN.actLogistic)
    e_o = AE.Encoder(enc_o, dec_o)
    e_h = H.Encoder(H.genNet([tuple(T.put(v) for v in we)], "actLogistic", "actLogistic"),
                    H.genNet([tuple(T.put(v) for v in wd)], "actLogistic", "actLogistic"))
    x = RNG.uniform(0, 1, 6)
    dx = T.put(x)
    assert rel_err(e_h.encode(dx).numpy(), AE.encode(O, e_o, x)) < tol
    assert rel_err(e_h.encodeDecode(dx).numpy(), AE.encodeDecode(O, e_o, x)) < tol
    assert rel_err(e_h.decode(e_h.encode(dx)).numpy(), AE.encodeDecode(O, e_o, x)) < tol
    assert rel_err(e_h.testEncoder("squaredError", dx).numpy(), AE.testEncoder(O, NN.squaredError(), e_o, x)) < tol
    g_e, g_d = AE.encGrad(O, NN.squaredError(), x, e_o)
    h_e, h_d = e_h.encGrad("squaredError", dx)
    for a, b in zip(h_e + h_d, g_e + g_d):
        assert rel_err(a.numpy(), b) < tol
    n_o = AE.trainEncoder(O, NN.squaredError(), 0.1, x, e_o)
    n_h = e_h.trainEncoder("squaredError", 0.1, dx)
    for a, b in zip(n_h.enc.params + n_h.dec.params, n_o.enc.params + n_o.dec.params):
        assert rel_err(a.numpy(), b) < tol
    # hidden batch: gradients of a batch of inputs = sum of per-input gradients
    X = RNG.uniform(0, 1, (9, 6))
    h_e, h_d = e_h.encGrad("squaredError", T.put(X, batched=True))
    acc = None
    for b in range(9):
        g = AE.encGrad(O, NN.squaredError(), X[b], e_o)
        g = [np.asarray(v) for v in g[0] + g[1]]
        acc = g if acc is None else [p + q for p, q in zip(acc, g)]
    for a, b in zip(h_e + h_d, acc):
        assert rel_err(a.numpy(), b) < 5 * tol


def test_bptt_launches_with_and_without_row_programs(repo_root):
    """launches of one BPTT gradient (Recurrent.hs:265-324: a fullyConnected logistic layer into a
    softmax layer of 24 outputs, crossEntropy at every one of 4 time steps, 16 sequences) and of an auto-encoder gradient
    with squaredError over the whole 96-wide input (AutoEncoder.hs:87-142), with the row programs on and off
    (TOPS_ROWPROG: read once, hence two processes).  Same numbers, fewer launches: each time step's loss head --
    wider than the 16 lanes of the GEMM epilogue's closed forms -- is one compiled row kernel."""
    import json
    import os
    import subprocess
    import sys
    code = r'''
import json, numpy as np
from tensor_ops_amd import tops as H
from tensor_ops_amd.hipt import HipT
T = HipT(0); H.hlib()
rng = np.random.default_rng(9)
i, h, o, n, B = 12, 20, 24, 4, 16
fc = tuple(T.put(v) for v in (0.5 * rng.standard_normal(h), 0.5 * rng.standard_normal((h, h)), 0.5 * rng.standard_normal((h, i)), 0.5 * rng.standard_normal(h)))
ff = tuple(T.put(v) for v in (0.5 * rng.standard_normal((o, h)), 0.5 * rng.standard_normal(o)))
net = H.rnn_genNet([(fc, "actLogistic", "actLogistic")], (ff, None), "actSoftmax")
xs = [T.put(rng.uniform(-1, 1, (B, i)), batched=True) for _ in range(n)]
ys = [T.put(rng.uniform(0.1, 0.9, (B, o)), batched=True) for _ in range(n)]
def bptt():
    with T.memo():
        _, gs, gp = H.rnn_netGrad(net, "crossEntropy", xs, ys, want_inputs=False)
        T.force_many(gs + gp)
    return gs + gp
bptt()
l0 = T.stats()["launches"]; g = bptt(); l_bptt = T.stats()["launches"] - l0
w = 96
enc = H.genNet([(T.put(0.3 * rng.standard_normal((30, w))), T.put(0.3 * rng.standard_normal(30)))], "actLogistic", "actLogistic")
dec = H.genNet([(T.put(0.3 * rng.standard_normal((w, 30))), T.put(0.3 * rng.standard_normal(w)))], "actLogistic", "actLogistic")
ae = H.Encoder(enc, dec)
x = T.put(rng.uniform(0, 1, (32, w)), batched=True)
def aeg():
    with T.memo():
        ge, gd = ae.encGrad("squaredError", x)
        T.force_many(ge + gd)
    return ge + gd
aeg()
l0 = T.stats()["launches"]; ga = aeg(); l_ae = T.stats()["launches"] - l0
print(json.dumps({"bptt": l_bptt, "ae": l_ae, "sum": [float(np.abs(t.numpy()).sum()) for t in g + ga]}))
'''
    res = {}
    for flag in ("1", "0"):
        env = dict(os.environ, TOPS_ROWPROG=flag, PYTHONPATH=repo_root)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=repo_root)
        assert r.returncode == 0, r.stderr[-3000:]
        res[flag] = json.loads(r.stdout.strip().splitlines()[-1])
    print("launches per BPTT gradient (4 steps): %d with row programs, %d without; auto-encoder gradient: %d / %d"
          % (res["1"]["bptt"], res["0"]["bptt"], res["1"]["ae"], res["0"]["ae"]))
    assert res["1"]["bptt"] <= 75 < res["0"]["bptt"] and res["1"]["ae"] <= 6 < res["0"]["ae"], res   # (round 2: 133 and 13)
    for a, b in zip(res["1"]["sum"], res["0"]["sum"]):
        assert abs(a - b) <= 1e-5 * max(abs(a), abs(b)), res
