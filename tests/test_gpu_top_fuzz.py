"""Random compositions of the TOp DSL (`>>>`, firstOp, secondOp, `***`, `&&&`, shuffle, replicate, sumOp, drop/take, map,
zip, add, scale ...; Types.hs:139-264, TOp.hs:106-381): the oracle's polymorphic closures are run with the numpy backend
and with the HIP backend, `runTOp` and `gradTOp'` with random cotangents -- outside a scope (one launch per class-method
call) and inside one (the calls are recorded and planned by the library)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ad, neuralnet as NN, top as TO  # noqa: E402
from oracle.tensor import OTensor  # noqa: E402

SEED = 0x7e5000f5
RTOL = 1e-5
O = OTensor(np.float64)


@pytest.fixture(scope="module")
def T():
    from tensor_ops_amd.hipt import HipT
    return HipT(0)


UN = [("logistic", NN.logistic), ("sin", lambda x: ad.sin(x)), ("tanh", lambda x: ad.tanh(x)),
      ("sq", lambda x: x * x + 0.5), ("expm", lambda x: ad.exp(-(x * x)))]
BI = [("mul", lambda x, y: x * y), ("mix", lambda x, y: x * y + ad.sin(x)), ("sub", lambda x, y: x - 0.5 * y),
      ("quot", lambda x, y: x / (2.0 + y * y))]


def recipe(rng):
    """The random decisions, as data: [(kind, params, k_before)], plus the input and output arity."""
    k = int(rng.integers(1, 4))
    k0, out = k, []
    for _ in range(int(rng.integers(3, 9))):
        kind = str(rng.choice(["map", "zip", "dup", "swap", "add", "scale", "fanout", "par", "shuffle", "second", "sumop",
                               "swapn", "drop", "take", "rep"]))
        u, b2 = int(rng.integers(len(UN))), int(rng.integers(len(BI)))
        if kind == "map":
            out.append((kind, u, k)); k2 = k
        elif kind == "zip" and k >= 2:
            out.append((kind, b2, k)); k2 = k - 1
        elif kind == "dup" and k <= 4:
            out.append((kind, 0, k)); k2 = k + 1
        elif kind == "rep" and k <= 3:
            out.append((kind, 0, k)); k2 = k + 2
        elif kind == "swap" and k >= 2:
            out.append((kind, 0, k)); k2 = k
        elif kind == "swapn" and k >= 2:
            out.append((kind, int(rng.integers(1, k)), k)); k2 = k
        elif kind == "add" and k >= 2:
            out.append((kind, 0, k)); k2 = k - 1
        elif kind == "scale":
            out.append((kind, float(rng.choice([-1.0, 0.5, 2.5])), k)); k2 = k
        elif kind == "fanout" and k <= 4:
            out.append((kind, u, k)); k2 = k + 1
        elif kind in ("par", "second") and k >= 2:
            out.append((kind, u, k)); k2 = k
        elif kind == "shuffle":
            idx = [int(i) for i in rng.integers(0, k, size=int(rng.integers(1, min(k + 2, 5) + 1)))]
            out.append((kind, idx, k)); k2 = len(idx)
        elif kind == "sumop" and k >= 2:
            out.append((kind, 0, k)); k2 = 1
        elif kind in ("drop", "take") and k >= 2:
            out.append((kind, 0, k)); k2 = k - 1
        else:
            continue
        k = k2
    return out, k0, k


class OracleLib:       # the oracle's combinators (shape evidence passed where Haskell infers it)
    def __init__(self, n): self.n = n
    def idOp(self, k): return TO.idOp(k)
    def then(self, op, nxt): return TO.compose(nxt, op)
    def first(self, o, n_pass): return TO.first(o, n_pass)
    def second(self, n_skip, o): return TO.secondOp(n_skip, o)
    def map(self, f): return TO.map_(f)
    def zip(self, f): return TO.zip_(f)
    def dup(self): return TO.duplicate()
    def rep(self, m): return TO.replicate(m)
    def swap(self): return TO.swap()
    def swapn(self, a, k): return TO.swap_n(a, k - a)
    def add(self): return TO.add()
    def scale(self, a): return TO.scale(a)
    def negate(self): return TO.negate()
    def fanout(self, a, b): return TO.fanout(a, b, [(self.n,)])
    def par(self, a, b): return TO.par(a, b)
    def shuffle(self, idx, k): return TO.shuffle(idx, [(self.n,)] * k)
    def sumop(self, k): return TO.sumOp(k, (self.n,))
    def drop(self, k): return TO.drop(1, [(self.n,)] * k)
    def take(self, k): return TO.take(k - 1, [(self.n,)] * k)


class MirrorLib:       # the C++ host mirror through its C ABI (tensor-ops_amd/tops.py)
    def __init__(self, H, n): self.H, self.n = H, n
    def idOp(self, k): return self.H.idOp(k)
    def then(self, op, nxt): return op >> nxt
    def first(self, o, n_pass): return self.H.firstOp(o, n_pass)
    def second(self, n_skip, o): return self.H.secondOp(n_skip, o)
    def map(self, f): return self.H.map_(f)
    def zip(self, f): return self.H.zip_(f)
    def dup(self): return self.H.duplicate()
    def rep(self, m): return self.H.replicate(m)
    def swap(self): return self.H.swap()
    def swapn(self, a, k): return self.H.shuffle(list(range(a, k)) + list(range(a)), k)   # swap' = a re-ordering
    def add(self): return self.H.add()
    def scale(self, a): return self.H.scale(a)
    def negate(self): return self.H.negate()
    def fanout(self, a, b): return self.H.fanout(a, b)
    def par(self, a, b): return self.H.par(a, b)
    def shuffle(self, idx, k): return self.H.shuffle(idx, k)
    def sumop(self, k): return self.H.sumOp(k, (self.n,))
    def drop(self, k): return self.H.drop(1, k)
    def take(self, k): return self.H.take(k - 1, k)


def realise(rec, k0, L):
    op = L.idOp(k0)
    for kind, par_, k in rec:
        if kind == "map":
            nxt = L.first(L.map(UN[par_][1]), k - 1)
        elif kind == "zip":
            nxt = L.first(L.zip(BI[par_][1]), k - 2)
        elif kind == "dup":
            nxt = L.first(L.dup(), k - 1)
        elif kind == "rep":
            nxt = L.first(L.rep(3), k - 1)
        elif kind == "swap":
            nxt = L.first(L.swap(), k - 2)
        elif kind == "swapn":
            nxt = L.swapn(par_, k)
        elif kind == "add":
            nxt = L.first(L.add(), k - 2)
        elif kind == "scale":
            nxt = L.first(L.scale(par_), k - 1)
        elif kind == "fanout":
            nxt = L.first(L.fanout(L.map(UN[par_][1]), L.scale(3.0)), k - 1)
        elif kind == "par":
            nxt = L.par(L.map(UN[par_][1]), L.first(L.negate(), k - 2))
        elif kind == "second":
            nxt = L.second(1, L.first(L.map(UN[par_][1]), k - 2))
        elif kind == "shuffle":
            nxt = L.shuffle(par_, k)
        elif kind == "sumop":
            nxt = L.sumop(k)
        elif kind == "drop":
            nxt = L.drop(k)
        else:
            nxt = L.take(k)
        op = L.then(op, nxt)
    return op


def build(rng, n):
    rec, k0, k = recipe(rng)
    return realise(rec, k0, OracleLib(n)), k0, k, [r[0] for r in rec], rec


def rel_err(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    den = np.linalg.norm(want.ravel())
    return np.linalg.norm((got - want).ravel()) / (den if den > 1e-3 else 1.0)


@pytest.mark.parametrize("scoped", [False, True])
@pytest.mark.parametrize("case", range(int(os.environ.get("TOPS_FUZZ_CASES", "100"))))
def test_random_compositions_run_and_differentiate_like_the_oracle(T, case, scoped):
    rng = np.random.default_rng(SEED + case)
    n = int(rng.choice([1, 7, 64, 300]))
    op, k_in, k_out, desc, _ = build(rng, n)
    xs = [rng.uniform(-1, 1, size=n) for _ in range(k_in)]
    ds = [rng.uniform(-1, 1, size=n) for _ in range(k_out)]
    ys_o = TO.runTOp(op, O, xs)
    g_o = op.grad(O, list(xs), ds)
    dxs, dds = [T.put(x) for x in xs], [T.put(d) for d in ds]
    if scoped:
        with T.memo():
            ys_t = TO.runTOp(op, T, dxs)
            g_t = op.grad(T, dxs, dds)
    else:
        ys_t = TO.runTOp(op, T, dxs)
        g_t = op.grad(T, dxs, dds)
    assert len(ys_t) == len(ys_o) == k_out and len(g_t) == len(g_o) == k_in, desc
    for a, b in zip(ys_t, ys_o):
        assert rel_err(a.numpy(), b) < RTOL, desc
    for a, b in zip(g_t, g_o):
        assert rel_err(a.numpy(), b) < RTOL, desc


@pytest.fixture(scope="module")
def H():
    from tensor_ops_amd import tops
    tops.hlib()
    return tops


@pytest.mark.parametrize("case", range(int(os.environ.get("TOPS_FUZZ_CASES", "100"))))
def test_random_compositions_on_the_host_mirror(T, H, case):
    """The same recipes built from the C++ mirror's combinators (host/tensorops/top.hpp through toh_*): its runTOp and
    gradTOp' against the oracle's."""
    rng = np.random.default_rng(SEED + case)
    n = int(rng.choice([1, 7, 64, 300]))
    op, k_in, k_out, desc, rec = build(rng, n)
    hop = realise(rec, k_in, MirrorLib(H, n))
    xs = [rng.uniform(-1, 1, size=n) for _ in range(k_in)]
    ds = [rng.uniform(-1, 1, size=n) for _ in range(k_out)]
    ys_o = TO.runTOp(op, O, xs)
    g_o = op.grad(O, list(xs), ds)
    dxs, dds = [T.put(x) for x in xs], [T.put(d) for d in ds]
    ys_h = hop.run(dxs)
    g_h = hop.grad(dxs, dds)
    assert len(ys_h) == k_out and len(g_h) == k_in, desc
    for a, b in zip(ys_h, ys_o):
        assert rel_err(a.numpy(), b) < RTOL, desc
    for a, b in zip(g_h, g_o):
        assert rel_err(a.numpy(), b) < RTOL, desc
