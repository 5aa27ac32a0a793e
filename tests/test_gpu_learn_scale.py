"""The batched gradTOp + SGD step at data-set scale (round 6, last): the routes tools/gemm_scan.py and tools/step_scan.py moved --
a tile per wave with bias + logistic, the cotangent's `act'` epilogue on a short K, weight gradients as stream-K / a split over
workgroups with `W - r dW` as the GEMM's own epilogue in place, the bias gradient as a column sum of its own, a rank-32 update --
checked against the same arithmetic in numpy fp64 (the reference's networkGradient summed over the batch, NeuralNet.hs:15-77 /
FeedForward.hs:57-235, as oracle/hmat.py restates it per sample; vectorised here because 20,000 samples through the per-sample
oracle would take minutes).  fp32: 1e-5 relative (north_star's bar); fp64: 1e-11."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def reference_step(ws, X, Y, rate):
    acts = [X]
    for li, (W, b) in enumerate(ws):
        z = acts[-1] @ W.T + b
        if li < len(ws) - 1:
            acts.append(1.0 / (1.0 + np.exp(-z)))
        else:
            e = np.exp(z - z.max(axis=1, keepdims=True))
            acts.append(e / e.sum(axis=1, keepdims=True))
    dz = acts[-1] * Y.sum(axis=1, keepdims=True) - Y
    out = [None] * len(ws)
    for li in range(len(ws) - 1, -1, -1):
        W, b = ws[li]
        gW, gb = dz.T @ acts[li], dz.sum(axis=0)
        if li > 0:
            dz = (dz @ W) * acts[li] * (1.0 - acts[li])
        out[li] = (W - rate * gW, b - rate * gb)
    return out


@pytest.mark.parametrize("dt", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("dims,B", [((784, 300, 100, 10), 20000), ((784, 256, 10), 8192), ((1024, 1024, 10), 4096), ((4096, 4096, 10), 32),
                                    ((300, 100, 10), 60000)])
def test_step_at_scale_matches_fp64_arithmetic(dt, dims, B):
    from tensor_ops_amd import tops as H
    from tensor_ops_amd.hipt import HipT
    T = HipT(0, dtype=np.float64) if dt == np.float64 else HipT(0)
    H.hlib()
    H.set_elem_dtype(dt)
    try:
        rng = np.random.default_rng(B + sum(dims))
        ws = [(0.5 * rng.standard_normal((o, i)) / np.sqrt(i), 0.5 * rng.standard_normal(o)) for i, o in zip(dims[:-1], dims[1:])]
        X = rng.uniform(0, 1, (B, dims[0])); Y = np.zeros((B, dims[-1])); Y[np.arange(B), rng.integers(0, dims[-1], B)] = 1.0
        ws_dt = [(w.astype(dt), b.astype(dt)) for w, b in ws]
        net = H.genNet([(T.put(w), T.put(b)) for w, b in ws_dt], "actMapLogistic", "actSoftmax")
        rate = 0.05 / B
        tr = H.Trainer(net, "crossEntropy", rate, T.put(X.astype(dt), batched=True), T.put(Y.astype(dt), batched=True), use_graph=False)
        tr.step()
        want = reference_step([(w.astype(np.float64), b.astype(np.float64)) for w, b in ws_dt], X.astype(dt).astype(np.float64), Y, rate)
        got = [p.numpy().astype(np.float64) for p in tr.net.params]
        tol = 1e-5 if dt == np.float32 else 1e-11
        for li, (W, b) in enumerate(want):
            # the update itself is what is compared: (p' - p) against (want - p), so a step that did nothing cannot pass
            dW_got, dW_want = got[2 * li] - ws_dt[li][0].astype(np.float64), W - ws_dt[li][0].astype(np.float64)
            db_got, db_want = got[2 * li + 1] - ws_dt[li][1].astype(np.float64), b - ws_dt[li][1].astype(np.float64)
            assert np.linalg.norm(dW_want) > 0
            if dt == np.float64:
                assert np.linalg.norm(dW_got - dW_want) <= tol * np.linalg.norm(dW_want) + 1e-15, (li, "W")
                assert np.linalg.norm(db_got - db_want) <= tol * np.linalg.norm(db_want) + 1e-15, (li, "b")
            # (fp32: the parameters to 1e-5 -- the update is a small difference of fp32 numbers and carries their rounding)
            assert np.linalg.norm(got[2 * li] - W) <= tol * np.linalg.norm(W), (li, "W'")
            assert np.linalg.norm(got[2 * li + 1] - b) <= tol * np.linalg.norm(b), (li, "b'")
        if dt == np.float32:   # ... and the fp32 update itself to 1e-3 of its own size
            for li, (W, b) in enumerate(want):
                d_got, d_want = got[2 * li] - ws_dt[li][0].astype(np.float64), W - ws_dt[li][0].astype(np.float64)
                assert np.linalg.norm(d_got - d_want) <= 2e-3 * np.linalg.norm(d_want), (li, "dW")
    finally:
        H.set_elem_dtype(np.float32)
