"""Every `TOPS_*` switch a product build reads is product surface (65 switches became 14 -- the A/B knobs
live in development builds only, csrc/common.hpp): one
fixed workload -- integer GEMMs through every route, an elementwise closure, two batched training steps of a small
network with a recognised loss head, one with a wide head (row program), a per-sample online-SGD stream, config 5's
shape class with the map fused -- is run in a fresh process per setting and must give the default's numbers (bit-exact
where the default is exact, 1e-5 elsewhere).  Settings are also combined: everything that turns an optimisation OFF at
once, and the A/B alternatives at once."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

WORKLOAD = r'''
import json, numpy as np
from tensor_ops_amd import tops as H
from tensor_ops_amd.hipt import HipT, logistic_closure
T = HipT(0); H.hlib()
rng = np.random.default_rng(77)
out = {}
# exact: integer GEMMs through the big-tile, mid-size, ragged, K-tail, skinny-K and small routes
exact = []
for m, k, n in [(1024, 1024, 1024), (512, 256, 512), (1024, 512, 768), (1000, 1000, 1000), (300, 131, 260), (65536, 64, 256), (48, 1024, 40), (2048, 2048, 2048), (1536, 200, 1536), (512, 2048, 512), (640, 640, 640)]:   # (the last two: several workgroups per tile)
    a = rng.integers(-2, 3, (m, k)).astype(np.float32); b = rng.integers(-2, 3, (k, n)).astype(np.float32)
    got = T.gmul(1, 1, 1, T.put(a), T.put(b)).numpy()
    exact.append(bool(np.array_equal(got, a @ b)))
# one extent 1: matVec / vecMat / an outer product / a tall column sum beyond the small-GEMM kernel's range (csrc/gemv.hip)
A = rng.integers(-2, 3, (1536, 4096)).astype(np.float32); xk = rng.integers(-2, 3, 4096).astype(np.float32); ym = rng.integers(-2, 3, 1536).astype(np.float32)
dA = T.put(A)
exact.append(bool(np.array_equal(T.matVec(dA, T.put(xk)).numpy(), A @ xk)))
exact.append(bool(np.array_equal(T.vecMat(T.put(ym), dA).numpy(), ym @ A)))
exact.append(bool(np.array_equal(T.matVec(T.transp(dA), T.put(ym)).numpy(), A.T @ ym)))
exact.append(bool(np.array_equal(T.outerV(T.put(ym), T.put(xk)).numpy(), np.outer(ym, xk))))
exact.append(bool(np.array_equal(T.sumRows(dA).numpy(), A.sum(axis=0))))
out["gemm_exact"] = exact
x = rng.uniform(-2, 2, (257, 129)).astype(np.float32)
out["lift"] = float(np.abs(T.liftT(lambda v: v[0] * v[0] / (1.5 + v[0] * v[0]), [T.put(x)], key="sw").numpy()).sum())
def net_problem(i, h, o, B):
    ws = [(0.5 * rng.standard_normal((h, i)), 0.5 * rng.standard_normal(h)), (0.5 * rng.standard_normal((o, h)), 0.5 * rng.standard_normal(o))]
    X = rng.uniform(0, 1, (B, i)); Y = np.zeros((B, o)); Y[np.arange(B), rng.integers(0, o, B)] = 1
    return ws, X, Y
for name, (i, h, o, B), hidden in (("head10", (96, 48, 10, 256), "actMapLogistic"), ("head24_tanh", (40, 28, 24, 64), "actMapTanh")):
    ws, X, Y = net_problem(i, h, o, B)
    net = H.genNet([(T.put(w), T.put(b)) for w, b in ws], hidden, "actSoftmax")
    tr = H.Trainer(net, "crossEntropy", 0.01 / B, T.put(X, batched=True), T.put(Y, batched=True), use_graph=False)
    tr.step(); tr.step()
    out[name] = [float(np.abs(p.numpy()).sum()) for p in tr.net.params]
# config 3 itself: 784 -> 256 -> 10 at 1024 rows, two steps (rate / rows: the step length of bench.py)
ws, X, Y = net_problem(784, 256, 10, 1024)
net = H.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
tr = H.Trainer(net, "crossEntropy", 0.02 / 1024, T.put(X, batched=True), T.put(Y, batched=True), use_graph=False)
tr.step(); tr.step()
out["c3"] = [float(np.abs(p.numpy()).sum()) for p in tr.net.params]
del tr
ws, X, Y = net_problem(30, 16, 6, 80)
net = H.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
got = H.trainAll(net, "crossEntropy", 0.05, T.put(X, batched=True), T.put(Y, batched=True), n=80)
out["online"] = [float(np.abs(p.numpy()).sum()) for p in got.params]
a = rng.integers(-2, 3, (256, 256, 64)).astype(np.float32); b = rng.integers(-2, 3, (64, 512)).astype(np.float32)
with T.memo():
    r = T.force(T.liftT(T.expr(logistic_closure, 1, key="swl"), [T.gmul(2, 1, 1, T.put(a), T.put(b))]))
out["c5_fused"] = float(r.numpy().astype(np.float64).sum())
# sibling products and lifts of one plan (the inner boundary's config 5 in small: 256 `gemm`s with one right operand, 256 lifts)
As = [T.put(a[i]) for i in range(256)]; Bd = T.put(b[:, :256].copy())
with T.memo():
    Cs = [T.gmul(1, 1, 1, x, Bd) for x in As]
    Ls = [T.liftT(T.expr(logistic_closure, 1, key="swl"), [c]) for c in Cs]
    T.force_many(Cs + Ls)
exact.append(bool(all(np.array_equal(c.numpy(), a[i] @ b[:, :256]) for i, c in enumerate(Cs))))
out["siblings"] = float(sum(l.numpy().astype(np.float64).sum() for l in Ls))
print(json.dumps(out))
'''

# The product's switches (csrc/common.hpp): every one of them, each way it can be set, alone -- and
# all of the "off" settings at once.  The watchdog / path / size switches get values that must change nothing.
PRODUCT = [
    ("TOPS_LAZY", "0"), ("TOPS_LAZY_FUSE", "0"), ("TOPS_LAZY_DEBUG", "1"), ("TOPS_EXPR_JIT", "0"), ("TOPS_ROWPROG", "0"),
    ("TOPS_PLAN_CACHE", "0"), ("TOPS_STEP_SEAM", "1"), ("TOPS_STEP_SEAM", "2"), ("TOPS_STEP_SEAM", "3"), ("TOPS_ONLINE_KERNEL", "0"), ("TOPS_ONLINE_GRAPH", "0"),
    ("TOPS_REPLAY_LIST_MAX", "0"), ("TOPS_OUTER_MAX_BYTES", "1073741824"), ("TOPS_RCCL_LIB", "/opt/rocm/lib/librccl.so"),
    ("TOPS_P2P_TIMEOUT_S", "5"), ("TOPS_ONLINE_TIMEOUT_S", "5"), ("TOPS_PINNED_STAGING", "0"),
    ("TOPS_GEMM_KW_KSPLIT", "0"), ("TOPS_LOSS_HEAD_MATCH", "0"), ("TOPS_SIBLING_BATCH", "0"), ("TOPS_GEMV", "0"),
]
OFF = {k: v for k, v in PRODUCT if v == "0"}
OFF["TOPS_STEP_SEAM"] = "1"   # (an optimisation that is off by default: "everything off" leaves the others off and turns it on)
# The A/B knobs of a development build (TOPS_BUILD_AB=1 python tensor-ops_amd/build.py): a product build does not read them
# (to_build_info), so there they are not routes at all and are not walked.
AB_OFF = {"TOPS_GEMM_W4": "0", "TOPS_GEMM_W4_128": "0",
          "TOPS_GEMM_W4_SPLITK": "0", "TOPS_GEMM_W4_EDGE": "0", "TOPS_GEMM_STREAMK": "0", "TOPS_GEMM_SKINNYK": "0",
          "TOPS_GEMM_PERSISTENT": "0", "TOPS_GEMM_WIDE_STORE": "0", "TOPS_GEMM_NT_STORE": "0", "TOPS_GEMM_UNALIGNED": "0",
          "TOPS_SMALL_PAIR": "0", "TOPS_SMALL_ONESHOT": "0", "TOPS_SMALL_ONESHOT8": "0", "TOPS_SMALL_XCD": "0",
          "TOPS_STEP_RANK1": "0", "TOPS_STEP_FUSE_TAIL": "0", "TOPS_SKINNYK_XCD_PAIRS": "0", "TOPS_SKINNYK_STAGGER": "0",
          "TOPS_GEMM_KW": "0", "TOPS_GEMM64_KW": "0", "TOPS_GEMM64_SKINNYK": "0", "TOPS_GEMM_STREAMK_HYBRID": "0"}
AB_ALT = {"TOPS_SKINNYK_V": "1", "TOPS_SKINNYK_NT": "0", "TOPS_STEP_CHAIN": "1", "TOPS_EW_MODE": "1", "TOPS_GEMM_STREAMK": "2",
          "TOPS_SMALL_NW": "4", "TOPS_GEMM_KW": "2", "TOPS_GEMM_KW_TILE": "3", "TOPS_GEMM_KW_NI": "3", "TOPS_GEMM_KW_SPLIT": "0", "TOPS_GEMM_KW_PAIR": "3"}
SETTINGS = [("default", {})] + [(k + "=" + v, {k: v}) for k, v in PRODUCT] + \
           [("everything_off", {k: v for k, v in OFF.items() if k != "TOPS_LAZY"}), ("everything_off_eager", dict(OFF))]
AB_SETTINGS = [(k + "=" + v, {k: v}) for k, v in sorted(AB_OFF.items())] + [(k + "=" + v, {k: v}) for k, v in sorted(AB_ALT.items())] + \
              [("ab_everything_off", dict(AB_OFF, **{k: v for k, v in OFF.items() if k != "TOPS_LAZY"})), ("ab_alternatives", dict(AB_ALT))]
_results = {}


def run(env_extra, repo_root):
    env = dict(os.environ, PYTHONPATH=repo_root, **env_extra)
    r = subprocess.run([sys.executable, "-c", WORKLOAD], env=env, capture_output=True, text=True, timeout=600, cwd=repo_root)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def _ab_build():
    import ctypes as C
    from tensor_ops_amd import capi
    v = C.c_int(0)
    capi.check(capi.lib().to_build_info(C.byref(v)))
    return v.value == 1


def test_the_switch_list_is_the_one_the_library_documents(repo_root):
    """csrc/common.hpp names the product switches; every `getenv("TOPS_...")` left in csrc/ and host/ is one of them (the
    rest go through ab_getenv, compiled out of a product build) and every one of them is walked below."""
    import re
    names = set()
    for d in ("tensor-ops_amd/csrc", "tensor-ops_amd/host", "tensor-ops_amd/host/tensorops", "tensor-ops_amd/host/apps"):
        for f in os.listdir(os.path.join(repo_root, d)):
            if f.endswith((".cpp", ".hip", ".hpp", ".h")):
                names |= set(re.findall(r'(?<![a-z_])getenv\("(TOPS_[A-Z0-9_]+)"\)', open(os.path.join(repo_root, d, f)).read()))
    assert names == {k for k, _ in PRODUCT}, sorted(names ^ {k for k, _ in PRODUCT})
    assert len(names) <= 19   # (round 6: + the loss-head recognition's switch VERDICT r5 asked for, + the sibling batches', + gemv.hip's)
    doc = open(os.path.join(repo_root, "tensor-ops_amd", "csrc", "common.hpp")).read()
    for k in names:
        assert k in doc, k


@pytest.mark.parametrize("name,env", SETTINGS + AB_SETTINGS, ids=[s[0] for s in SETTINGS + AB_SETTINGS])
def test_switch_gives_the_defaults_numbers(repo_root, name, env):
    if (name, env) in AB_SETTINGS and not _ab_build():
        pytest.skip("an A/B knob of a development build (TOPS_BUILD_AB=1): this product build does not read it")
    if "default" not in _results:
        _results["default"] = run({}, repo_root)
    want = _results["default"]
    assert all(want["gemm_exact"])
    if name == "default":
        return
    got = run(env, repo_root)
    assert all(got["gemm_exact"]), (name, got["gemm_exact"])
    for key in ("lift", "c5_fused", "siblings"):
        assert abs(got[key] - want[key]) <= 2e-6 * abs(want[key]), (name, key, got[key], want[key])
    for key in ("head10", "head24_tanh", "online", "c3"):
        for a, b in zip(got[key], want[key]):
            assert abs(a - b) <= 1e-5 * abs(b), (name, key, got[key], want[key])
