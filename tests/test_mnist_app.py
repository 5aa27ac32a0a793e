"""tensor-ops-mnist on the HIP backend (host/apps/mnist.cpp, app/MNIST.hs).

CPU: the IDX loader against files written here with numpy (`--check-data` parses and
reports without touching a GPU) and its error paths.  GPU: `trainAll` (per-sample online SGD,
app/MNIST.hs:390-393) against the oracle's plain-C HMat sequence, and the app end to end on a
synthetic data set of MNIST's format.
"""
import os
import struct
import subprocess

import numpy as np
import pytest


def app(repo_root):
    path = os.path.join(repo_root, "tensor-ops_amd", "tensor-ops-mnist-hip")
    if not os.path.exists(path):
        import importlib.util
        spec = importlib.util.spec_from_file_location("_b", os.path.join(repo_root, "tensor-ops_amd", "build.py"))
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
        b.build()
    return path


def write_idx(d, prefix, images, labels):
    n, r, c = images.shape
    with open(os.path.join(d, prefix + "-images-idx3-ubyte"), "wb") as f:
        f.write(struct.pack(">IIII", 0x803, n, r, c))
        f.write(images.astype(np.uint8).tobytes())
    with open(os.path.join(d, prefix + "-labels-idx1-ubyte"), "wb") as f:
        f.write(struct.pack(">II", 0x801, len(labels)))
        f.write(labels.astype(np.uint8).tobytes())


def test_idx_loader_reads_what_numpy_wrote(repo_root, tmp_path):
    rng = np.random.default_rng(5)
    tr_i, tr_l = rng.integers(0, 256, (37, 28, 28)), rng.integers(0, 10, 37)
    te_i, te_l = rng.integers(0, 256, (11, 28, 28)), rng.integers(0, 10, 11)
    write_idx(tmp_path, "train", tr_i, tr_l)
    write_idx(tmp_path, "t10k", te_i, te_l)
    out = subprocess.run([app(repo_root), "--data", str(tmp_path), "--check-data"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("n=")]
    for line, im, lb in zip(lines, (tr_i, te_i), (tr_l, te_l)):
        hist = ",".join(str(int((lb == c).sum())) for c in range(10))
        assert line == "n=%d rows=28 cols=28 pixel_sum=%d labels=%s" % (len(lb), int(im.sum()), hist)


def test_idx_loader_error_paths(repo_root, tmp_path):
    binp = app(repo_root)
    out = subprocess.run([binp, "--data", str(tmp_path), "--check-data"], capture_output=True, text=True)
    assert out.returncode == 1 and "not found" in out.stderr
    rng = np.random.default_rng(6)
    write_idx(tmp_path, "train", rng.integers(0, 256, (5, 28, 28)), rng.integers(0, 10, 5))
    write_idx(tmp_path, "t10k", rng.integers(0, 256, (5, 28, 28)), rng.integers(0, 10, 4))  # count mismatch
    out = subprocess.run([binp, "--data", str(tmp_path), "--check-data"], capture_output=True, text=True)
    assert out.returncode == 1 and "Could not combine" in out.stderr
    with open(os.path.join(tmp_path, "t10k-images-idx3-ubyte"), "r+b") as f:
        f.write(struct.pack(">I", 0x12345678))                                                # bad magic
    out = subprocess.run([binp, "--data", str(tmp_path), "--check-data"], capture_output=True, text=True)
    assert out.returncode == 1 and "Could not decode image" in out.stderr
    out = subprocess.run([binp, "--induce", "12"], capture_output=True, text=True)
    assert out.returncode == 2 and "out of range" in out.stderr


# ---- GPU ---------------------------------------------------------------------------------------------
def _setup(n, i, h, o, seed):
    rng = np.random.default_rng(seed)
    ws = [(0.5 * rng.standard_normal((h, i)), 0.5 * rng.standard_normal(h)),
          (0.5 * rng.standard_normal((o, h)), 0.5 * rng.standard_normal(o))]
    X = rng.uniform(0, 1, (n, i))
    Y = np.zeros((n, o))
    Y[np.arange(n), rng.integers(0, o, n)] = 1.0
    return ws, X, Y


def rel_err(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.linalg.norm((got - want).ravel()) / max(np.linalg.norm(want.ravel()), 1e-300)


def _online_stats():
    import ctypes as C
    from tensor_ops_amd import capi
    a, b = C.c_int64(), C.c_int64()
    capi.check(capi.lib().to_online_sgd_stats(C.byref(a), C.byref(b)))
    return a.value, b.value


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,fused,tol", [("f32", True, 1e-5), ("f32", False, 1e-5), ("f64", True, 1e-11)])
def test_trainAll_is_the_reference_online_sgd(dtype, fused, tol):
    """`foldl' trainNetwork` over 48 samples in a shuffled order == oracle/hmat_path.c's per-sample loop"""
    from oracle import hmat
    from tensor_ops_amd import tops as H
    from tensor_ops_amd.hipt import HipT
    H.hlib()
    dt = np.float32 if dtype == "f32" else np.float64
    H.set_elem_dtype(dt)
    try:
        T = HipT(0, dtype=dt)
        ws, X, Y = _setup(64, 20, 12, 5, 9)
        order = np.random.default_rng(1).permutation(64)[:48]
        want, _ = hmat.train_online(X[order], Y[order], ws[0][0], ws[0][1], ws[1][0], ws[1][1], 0.1)
        net = H.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
        s0 = _online_stats()
        got = H.trainAll(net, "crossEntropy", 0.1, T.put(X, batched=True), T.put(Y, batched=True),
                         order=list(order), use_fused=fused)
        for a, b in zip(got.params, want):
            assert rel_err(a.numpy(), b) < tol
        # with the library's fusion on the captured one-sample step is recognised as an ffLayer stack's and the 48
        # samples go through the persistent kernel (csrc/online_sgd.hip), in either precision; fusion off replays the step
        s1 = _online_stats()
        assert (s1[0] - s0[0], s1[1] - s0[1]) == ((1, 48) if fused else (0, 0))
        # the same through one `trainNetwork` call per sample (no graph, no staging buffer)
        cur = net
        for k in order[:8]:
            cur = H.trainNetwork(cur, "crossEntropy", 0.1, T.put(X[k]), T.put(Y[k]))
        eight = H.trainAll(net, "crossEntropy", 0.1, T.put(X, batched=True), T.put(Y, batched=True),
                           order=list(order[:8]), use_fused=fused)
        for a, b in zip(eight.params, cur.params):
            assert rel_err(a.numpy(), b.numpy()) < tol
        # the input network is untouched (values are immutable at the boundary)
        for p, (w, b) in zip(zip(net.params[0::2], net.params[1::2]), ws):
            assert rel_err(p[0].numpy(), w) < 1e-6 and rel_err(p[1].numpy(), b) < 1e-6
    finally:
        H.set_elem_dtype(np.float32)


@pytest.mark.gpu
def test_batch_gather_and_slice():
    from tensor_ops_amd.capi import TensorOpsError
    from tensor_ops_amd.hipt import HipT
    for dt in (np.float32, np.float64):
        T = HipT(0, dtype=dt)
        rng = np.random.default_rng(2)
        for width in (784, 10, 7):
            X = rng.uniform(0, 1, (50, width)).astype(dt)
            dX = T.put(X, batched=True)
            idx = list(rng.permutation(50)) + [3, 3, 49]
            assert np.array_equal(T.batch_gather(dX, idx).numpy(), X[idx])
            sl = T.batch_slice(dX, 13, 20)
            assert sl.batch == 20 and np.array_equal(sl.numpy(), X[13:33])
            assert np.array_equal(T.batch_sum(sl).numpy().shape, (width,))
            np.testing.assert_allclose(T.batch_sum(sl).numpy(), X[13:33].sum(0), rtol=1e-5)
        with pytest.raises(TensorOpsError):
            T.batch_gather(dX, [50])
        with pytest.raises(TensorOpsError):
            T.batch_slice(dX, 40, 11)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--white"], ["--f64"], ["--minibatch", "50"]])
def test_app_learns_the_synthetic_set(repo_root, extra):
    out = subprocess.run([app(repo_root), "--synthetic", "2000,400", "--layers", "[32,16]", "--batch", "500",
                          "--rate", "0.05", "--epochs", "2"] + extra,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr
    lines = out.stdout.splitlines()
    assert lines[0].startswith("Synthetic data") and "Loaded data." in lines and "Data processed." in lines
    assert any(l.startswith("rate: 0.05") and "layers: [32,16]" in l for l in lines)
    assert lines.count("[Epoch 1]") == 1 and lines.count("[Epoch 2]") == 1
    n_classes = 11 if "--white" in extra else 10
    if "--white" in extra:
        assert "white noise class enabled" in lines
        assert any(l == "Training on 2200 samples in batches of 500 ..." for l in lines)
    val = [float(l.split()[1].rstrip("%")) for l in lines if l.startswith("Validation:")]
    assert len(val) >= 8 and val[-1] < 15.0 and val[-1] < val[0], val
    # confusion matrix: n_classes rows "[r] c0 c1 ..." after every batch, entries sum to the set size
    k = max(i for i, l in enumerate(lines) if l.startswith("Validation:"))
    rows = lines[k + 1:k + 1 + n_classes]
    assert [r.split()[0] for r in rows] == ["[%d]" % i for i in range(n_classes)]
    total = sum(int(v) for r in rows for v in r.split()[1:])
    assert total == (440 if "--white" in extra else 400)


@pytest.mark.gpu
def test_app_induces_a_digit(repo_root):
    out = subprocess.run([app(repo_root), "--synthetic", "1500,200", "--layers", "[32]", "--batch", "1500",
                          "--rate", "0.05", "--noconfusion", "--induce", "3", "--induce-iters", "300",
                          "--max-batches", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr
    lines = out.stdout.splitlines()
    assert "inducing: 3" in lines
    k = max(i for i, l in enumerate(lines) if l.startswith("Validation:"))
    art = lines[k + 1:k + 29]
    assert len(art) == 28 and all(len(r) == 56 for r in art)          # 28 rows, pixels doubled (:433-437)
    probs = [float(v) for v in lines[k + 29].split("/")]
    assert len(probs) == 10 and int(np.argmax(probs)) == 3 and probs[3] > 0.9   # the induced image is a "3"
