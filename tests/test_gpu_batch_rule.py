"""The batch rule is a lowering, not an optional fusion.  The reference's DSL, called on batched
data, emits one outer product per sample where `gradTOp` returns (`gmul (transp x) dtdz`, TOp.hs:86-88): B*o*i numbers
whose only use is their sum over the batch.  In EVERY mode -- the default, `TOPS_LAZY_FUSE=0` (recorded, planned one
launch per op), `TOPS_LAZY=0` (every call eager, no scope needed), the trainer's own `use_fused=False` -- `to_batch_sum`
of that value is `to_gmul_batch_sum` (one GEMM with K = B) and the per-sample tensor is never allocated: config 3's would
be 822 MB, the pool must stay under 64 MB.  Parity against the per-sample C oracle at 1e-5."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WORK = r'''
import json, sys, numpy as np
import ctypes as C
from tensor_ops_amd import tops, capi
from tensor_ops_amd.hipt import HipT
from oracle import hmat
import bench
T = HipT(0); tops.hlib()
ws, X, Y = bench.synth(0, 1024)
want, _ = hmat.batched_grads(X, Y, ws[0][0], ws[0][1], ws[1][0], ws[1][1], recompute=False)
net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
# (the reference's per-sample rate over a summed gradient of 1024 rows leaves the range where the unfused softmax -- like
#  the reference's -- stays finite: the step length of bench.py, rate / rows)
tr = tops.Trainer(net, "crossEntropy", 0.02 / 1024, T.put(X, batched=True), T.put(Y, batched=True), use_graph=False,
                  use_fused=(sys.argv[1] == "fused"))
tr.grad()
_, g_ptr, n = tr.flat()
flat = np.empty(n, dtype=np.float32)
h = capi.c_tensor(); d = (C.c_int64 * 1)(n)
capi.check(capi.lib().to_wrap(C.c_void_p(g_ptr), 0, 1, d, 0, C.byref(h)))
capi.check(capi.lib().to_download(h, flat.ctypes.data_as(C.c_void_p), flat.nbytes))
capi.lib().to_release(h)
errs, off = [], 0
for w in want:
    k = int(np.prod(w.shape)); g = flat[off:off + k].reshape(w.shape).astype(np.float64); off += (k + 3) // 4 * 4
    errs.append(float(np.linalg.norm(g - w) / np.linalg.norm(w)))
tr.apply(); tr.grad(); tr.apply()
st = T.stats()
print(json.dumps({"errs": errs, "pool_bytes": int(st["pool_bytes"]), "launches": int(tr.launches_per_step),
                  "finite": bool(all(np.isfinite(p.numpy()).all() for p in tr.net.params))}))
'''


def _run(repo_root, mode, env):
    r = subprocess.run([sys.executable, "-c", WORK, mode], env=dict(os.environ, PYTHONPATH=repo_root, **env), cwd=repo_root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("name,mode,env", [
    ("default", "fused", {}),
    ("TOPS_LAZY_FUSE=0", "fused", {"TOPS_LAZY_FUSE": "0"}),
    ("TOPS_LAZY=0", "fused", {"TOPS_LAZY": "0"}),
    ("trainer use_fused=False (to_set_lazy(0) per step)", "eager", {}),
    ("everything off", "eager", {"TOPS_LAZY": "0", "TOPS_LAZY_FUSE": "0", "TOPS_ROWPROG": "0", "TOPS_PLAN_CACHE": "0"}),
    # forward + loss head joined inside each XCD (gemm_small_seam_kernel; off by default, it measured slower): same numbers
    ("TOPS_STEP_SEAM=1", "fused", {"TOPS_STEP_SEAM": "1"}),
    ("TOPS_STEP_SEAM=2", "fused", {"TOPS_STEP_SEAM": "2"}),
], ids=["default", "fuse_off", "lazy_off", "trainer_unfused", "all_off", "seam_last_arriver", "seam_owner_waits"])
def test_c3_never_materialises_the_per_sample_outer_products(repo_root, name, mode, env):
    got = _run(repo_root, mode, env)
    assert got["finite"]
    assert max(got["errs"]) < 1e-5, (name, got)
    assert got["pool_bytes"] < 64 << 20, (name, got)      # 1024 x 256 x 784 floats would be 822 MB on their own
    if name == "default":
        assert got["launches"] == 3
    if name.startswith("TOPS_STEP_SEAM"):
        assert got["launches"] == 2


def test_the_rule_holds_for_eager_calls_outside_any_scope():
    """to_gmul of two batched operands with nothing contracted returns at once and allocates nothing; to_batch_sum of it
    (also through scaleT and a sumT of two of them: the `&&&` of shared weights) is the K = B GEMM, exact on integers;
    asking for its elements still works; asking for more than TOPS_OUTER_MAX_BYTES is refused with a message."""
    from tensor_ops_amd.hipt import HipT
    from tensor_ops_amd.capi import TensorOpsError
    T = HipT(0)
    rng = np.random.default_rng(5)
    B, o, i = 512, 96, 200
    dz = rng.integers(-2, 3, (B, o)).astype(np.float32)
    x = rng.integers(-2, 3, (B, i)).astype(np.float32)
    dDz, dX = T.put(dz, batched=True), T.put(x, batched=True)
    T.sync()
    p0, l0 = T.stats()["pool_bytes"], T.stats()["launches"]
    per = T.gmul(1, 0, 1, dDz, dX)                      # [B] x (o, i): recorded
    assert T.stats()["launches"] == l0
    g = T.batch_sum(per)
    assert np.array_equal(g.numpy(), dz.T @ x)
    two = T.batch_sum(T.sumT([T.scaleT(2.0, per), T.gmul(1, 0, 1, dDz, dX)], (o, i)))
    assert np.array_equal(two.numpy(), 3 * (dz.T @ x))
    assert T.stats()["pool_bytes"] - p0 < 4 * B * o * i   # never the per-sample tensor
    assert np.array_equal(per.numpy(), np.einsum("bo,bi->boi", dz, x))   # its elements, when someone wants them
    big = T.gmul(1, 0, 1, T.genRand((3000,), "uniform", -1, 1, 1, batch=4096), T.genRand((3000,), "uniform", -1, 1, 2, batch=4096))
    gs = T.batch_sum(big)                               # 4096 x 3000 x 3000 floats = 147 GB: only ever its sum
    assert gs.numpy().shape == (3000, 3000)
    with pytest.raises(TensorOpsError, match="per-sample outer products"):
        big.numpy()


def test_caller_owned_operands_are_never_read_late_outside_a_scope():
    """the deferred outer product reads its operands when it is produced.  The library orders its OWN writes
    after such readers, but memory it does not own (to_wrap: a torch buffer) can change behind its back -- so outside a
    scope a product of wrapped operands is computed at once (launches at the call, a value that no longer depends on
    the operands' memory); the same product of the library's own tensors is only recorded."""
    import ctypes as C
    from tensor_ops_amd import capi
    from tensor_ops_amd.hipt import DT, HipT
    T = HipT(0)
    rng = np.random.default_rng(9)
    B, o, i = 64, 8, 12
    dz = rng.integers(-2, 3, (B, o)).astype(np.float32)
    x = rng.integers(-2, 3, (B, i)).astype(np.float32)
    want = np.einsum("bo,bi->boi", dz, x)
    own_dz, own_x = T.put(dz, batched=True), T.put(x, batched=True)      # the memory belongs to these two handles ...

    def wrap(t, n):                                                       # ... and these only point at it (non-owning)
        h = capi.c_tensor()
        d = (C.c_int64 * 1)(n)
        capi.check(capi.lib().to_wrap(C.c_void_p(t.ptr), 0, 1, d, B, C.byref(h)))
        return DT(h)
    T.sync()
    l0 = T.stats()["launches"]
    rec = T.gmul(1, 0, 1, own_dz, own_x)
    assert T.stats()["launches"] == l0            # the library's own tensors: recorded, nothing launched
    w_dz, w_x = wrap(own_dz, o), wrap(own_x, i)
    per = T.gmul(1, 0, 1, w_dz, w_x)
    assert T.stats()["launches"] > l0             # caller-owned memory: computed now
    T.sync()
    l1 = T.stats()["launches"]
    assert np.array_equal(per.numpy(), want)
    assert T.stats()["launches"] == l1            # (a download of an existing value: no kernel)
    assert np.array_equal(rec.numpy(), want)
