"""Differential test of the planner (csrc/lazy.cpp): random graphs of class-method calls -- contractions with
biases, scales, closures (smooth and piecewise), sums, row sums, broadcasts, batch sums, transposed views, shared
subterms, weight-gradient shapes next to their row sums, `p - r*g` updates, results copied into parameter storage --
are run twice on the same inputs: recorded inside a fusion scope (results demanded in a random order, some never,
some only after the scope closed) and with deferral switched off (one launch per call).  Deferral must never
change WHAT a value is: every demanded result agrees to fp32 round-off (1e-5), shapes exactly.  Both element types."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x7e5000f2


@pytest.fixture(scope="module", params=["f32", "f64"])
def T(request):
    from tensor_ops_amd.hipt import HipT
    return HipT(0, dtype=np.float32 if request.param == "f32" else np.float64)


def set_lazy(on):
    from tensor_ops_amd import capi
    prev = C.c_int()
    capi.check(capi.lib().to_set_lazy(int(on), C.byref(prev)))
    return prev.value


def closures():
    from tensor_ops_amd import hipt
    return [
        ("logistic", 1, lambda v: 1.0 / (1.0 + hipt.exp(-v[0]))),
        ("tanh", 1, lambda v: hipt.tanh(v[0])),
        ("affine1", 1, lambda v: 0.5 * v[0] - 0.25),
        ("clamp", 1, lambda v: hipt.minimum(hipt.maximum(v[0], -0.7), 0.9)),
        ("softplusish", 1, lambda v: hipt.log(1.0 + hipt.exp(v[0]))),
        ("mul", 2, lambda v: v[0] * v[1]),
        ("sgd", 2, lambda v: v[0] - 0.125 * v[1]),
        ("dlogistic", 2, lambda v: v[0] * ((1.0 / (1.0 + hipt.exp(-v[1]))) * (1.0 - 1.0 / (1.0 + hipt.exp(-v[1]))))),
        ("mix3", 3, lambda v: v[0] * v[1] + 0.5 * v[2]),
        # (appended: the index-based picks of build_program keep their meaning)
        ("exp", 1, lambda v: hipt.exp(v[0])),
        ("recip", 1, lambda v: 1.0 / v[0]),
        ("sub", 2, lambda v: v[0] - v[1]),
        ("sub9", 2, lambda v: v[0] - 0.9 * v[1]),
        ("logse", 2, lambda v: -2.0 * (v[1] - 1.0 / (1.0 + hipt.exp(-v[0]))) * (1.0 / (1.0 + hipt.exp(-v[0])))
         * (1.0 - 1.0 / (1.0 + hipt.exp(-v[0])))),
        ("logse9", 2, lambda v: -1.8 * (v[1] - 1.0 / (1.0 + hipt.exp(-v[0]))) * (1.0 / (1.0 + hipt.exp(-v[0])))
         * (1.0 - 1.0 / (1.0 + hipt.exp(-v[0])))),
    ]


def build_program(rng, extended=False):
    """A random straight-line program over a pool of values; returns the list of steps (pure data).
    extended: also matrix-matrix products, closures over transposed views of recorded matrices, n-ary sums and
    constants (a separate switch so that the plain family's case numbers keep their programs)."""
    B = int(rng.choice([1, 3, 40, 130]))
    n_in, n_h, n_o = int(rng.choice([5, 33, 96])), int(rng.choice([4, 20, 64])), int(rng.choice([3, 10, 17]))
    steps, kinds = [], {}      # kinds[name] = ("vec", n, batched) | ("mat", r, c) | ("scal", batched)

    def new(name, kind):
        kinds[name] = kind
        return name
    leaves = {"x": ("vec", n_in, True), "y": ("vec", n_o, True), "W1": ("mat", n_h, n_in), "b1": ("vec", n_h, False),
              "W2": ("mat", n_o, n_h), "b2": ("vec", n_o, False)}
    kinds.update(leaves)
    cl = closures()
    for k in range(int(rng.integers(6, 22))):
        vecs = [n for n, v in kinds.items() if v[0] == "vec"]
        ops = ["matvec", "addbias", "lift1", "lift2", "scale", "sumrows_outer", "wgrad", "back", "sum2", "update"]
        if extended:
            ops += ["matmat", "lift_transp", "sum3", "konst_add", "matmat", "lift_transp"]
        op = rng.choice(ops)
        name = "v%d" % k
        if op == "matmat":               # (r x c) . (c x q), either operand possibly a transposed view
            mats = [(m, km[1], km[2], False) for m, km in kinds.items() if km[0] == "mat"]
            mats += [(m, c_, r_, True) for m, r_, c_, _ in mats]
            cand = [(a, b) for a in mats for b in mats if a[2] == b[1]]
            if not cand:
                continue
            a, b = cand[int(rng.integers(len(cand)))]
            steps.append(("gmul", name, (1, 1, 1), ("T", a[0]) if a[3] else a[0], ("T", b[0]) if b[3] else b[0], False))
            new(name, ("mat", a[1], b[2]))
            continue
        if op == "lift_transp":          # a closure over the transposed view of a matrix
            mats = [m for m, km in kinds.items() if km[0] == "mat"]
            m = mats[int(rng.integers(len(mats)))]
            c = [c for c in cl if c[1] == 1][int(rng.integers(3))]
            steps.append(("lift", name, c[0], [("T", m)]))
            new(name, ("mat", kinds[m][2], kinds[m][1]))
            continue
        if op == "sum3":
            cand = [(a, b, c_) for a in vecs for b in vecs for c_ in vecs
                    if kinds[a][1] == kinds[b][1] == kinds[c_][1] and kinds[a][2] == kinds[b][2] == kinds[c_][2]]
            if not cand:
                continue
            a, b, c_ = cand[int(rng.integers(len(cand)))]
            steps.append(("sum", name, [a, b, c_]))
            new(name, kinds[a])
            continue
        if op == "konst_add":
            v = vecs[int(rng.integers(len(vecs)))]
            if kinds[v][2]:
                continue
            steps.append(("konst", name + "k", kinds[v][1], float(rng.choice([0.0, 1.0, -0.5]))))
            new(name + "k", ("vec", kinds[v][1], False))
            steps.append(("sum", name, [v, name + "k"]))
            new(name, kinds[v])
            continue
        if op == "matvec":
            cand = [(w, v) for w, kw in kinds.items() if kw[0] == "mat" for v in vecs if kinds[v][1] == kw[2]]
            if not cand:
                continue
            w, v = cand[int(rng.integers(len(cand)))]
            steps.append(("gmul", name, (1, 1, 0), w, v, False))
            new(name, ("vec", kinds[w][1], kinds[v][2]))
        elif op == "addbias":
            cand = [(v, b) for v in vecs for b in vecs if kinds[v][1] == kinds[b][1] and v != b]
            if not cand:
                continue
            v, b = cand[int(rng.integers(len(cand)))]
            steps.append(("sum", name, [v, b]))
            new(name, ("vec", kinds[v][1], kinds[v][2] or kinds[b][2]))
        elif op == "lift1":
            v = vecs[int(rng.integers(len(vecs)))]
            c = [c for c in cl if c[1] == 1][int(rng.integers(5))]
            steps.append(("lift", name, c[0], [v]))
            new(name, kinds[v])
        elif op in ("lift2", "sum2"):
            cand = [(a, b) for a in vecs for b in vecs if kinds[a][1] == kinds[b][1]]
            a, b = cand[int(rng.integers(len(cand)))]
            if op == "sum2":
                steps.append(("sum", name, [a, b]))
            else:
                c = [c for c in cl if c[1] == 2][int(rng.integers(3))]
                steps.append(("lift", name, c[0], [a, b]))
            new(name, ("vec", kinds[a][1], kinds[a][2] or kinds[b][2]))
        elif op == "scale":
            v = vecs[int(rng.integers(len(vecs)))]
            steps.append(("scale", name, float(rng.choice([-1.0, 0.5, 2.0])), v))
            new(name, kinds[v])
        elif op == "sumrows_outer":      # softmax-style: s = sumRows v ; r = recip-ish ; outer r v
            v = vecs[int(rng.integers(len(vecs)))]
            steps.append(("sumrows", name + "s", v))
            new(name + "s", ("scal", kinds[v][2]))
            steps.append(("lift", name + "r", "affine1", [name + "s"]))
            new(name + "r", ("scal", kinds[v][2]))
            steps.append(("gmul", name, (0, 0, 1), name + "r", v, False))
            new(name, kinds[v])
        elif op == "wgrad":              # dW = sum_b dz (x) a, db = sum_b dz
            bat = [v for v in vecs if kinds[v][2]]
            if len(bat) < 2:
                continue
            dz, a = bat[int(rng.integers(len(bat)))], bat[int(rng.integers(len(bat)))]
            steps.append(("gmul", name, (1, 0, 1), dz, ("T", a), True))
            new(name, ("mat", kinds[dz][1], kinds[a][1]))
            steps.append(("batchsum", name + "b", dz))
            new(name + "b", ("vec", kinds[dz][1], False))
        elif op == "back":               # dh = W^T dz
            cand = [(w, v) for w, kw in kinds.items() if kw[0] == "mat" for v in vecs if kinds[v][1] == kw[1]]
            if not cand:
                continue
            w, v = cand[int(rng.integers(len(cand)))]
            steps.append(("gmul", name, (1, 1, 0), ("T", w), v, False))
            new(name, ("vec", kinds[w][2], kinds[v][2]))
        elif op == "update":             # p' = p - r*g for a parameter-shaped unbatched value
            mats = [m for m, km in kinds.items() if km[0] == "mat"]
            cand = [(p, g) for p in mats for g in mats if kinds[p] == kinds[g] and p != g]
            if not cand:
                continue
            p, g = cand[int(rng.integers(len(cand)))]
            steps.append(("lift", name, "sgd", [p, g]))
            new(name, kinds[p])
    sizes = {"B": B}
    return leaves, steps, kinds, sizes


def build_program2(rng):
    """A two-layer training step written out call by call -- forward, one of several loss heads (the two the library
    has closed forms for, near misses of both that it must NOT take for them, a scaled one, none), the backward pass,
    weight and bias gradients, updates -- batched or, like the reference's own per-sample call, unbatched; with extra
    consumers sprinkled over the intermediate values so that what a fused launch swallows is still wanted."""
    B = int(rng.choice([0, 0, 1, 5, 64]))
    bat = B > 0
    n_in, n_h, n_o = int(rng.choice([7, 40, 100])), int(rng.choice([6, 32])), int(rng.choice([3, 10, 16]))
    leaves = {"x": ("vec", n_in, bat), "y": ("vec", n_o, bat), "W1": ("mat", n_h, n_in), "b1": ("vec", n_h, False),
              "W2": ("mat", n_o, n_h), "b2": ("vec", n_o, False)}
    kinds = dict(leaves)
    steps = []

    def add(st, kind):
        steps.append(st)
        kinds[st[1]] = kind
        return st[1]

    def extra(v):          # sometimes another reader of v
        if rng.random() < 0.3:
            k = len(steps)
            how = rng.choice(["tanh", "scale", "self"])
            if how == "tanh":
                add(("lift", "e%d" % k, "tanh", [v]), kinds[v])
            elif how == "scale":
                add(("scale", "e%d" % k, 0.5, v), kinds[v])
            else:
                add(("lift", "e%d" % k, "mul", [v, v]), kinds[v])
    z1 = add(("gmul", "z1", (1, 1, 0), "W1", "x", False), ("vec", n_h, bat))
    a1 = add(("sum", "a1", [z1, "b1"]), ("vec", n_h, bat))
    extra(a1)
    h = add(("lift", "h", "logistic", [a1]), ("vec", n_h, bat))
    extra(h)
    z2 = add(("gmul", "z2", (1, 1, 0), "W2", h, False), ("vec", n_o, bat))
    a2 = add(("sum", "a2", [z2, "b2"]), ("vec", n_o, bat)) if rng.random() < 0.8 else z2
    extra(a2)
    head = str(rng.choice(["smce", "smce", "smce_near", "logse", "logse_near", "smce_scaled", "plain"]))
    if head.startswith("smce"):
        e = add(("lift", "e", "exp", [a2]), ("vec", n_o, bat))
        s_ = add(("sumrows", "s", e), ("scal", bat))
        r = add(("lift", "r", "recip", [s_]), ("scal", bat))
        p = add(("gmul", "p", (0, 0, 1), r, e, False), ("vec", n_o, bat))
        extra(p)
        sy = add(("sumrows", "sy", "y"), ("scal", bat))
        t = add(("gmul", "t", (0, 0, 1), sy, p, False), ("vec", n_o, bat))
        dz2 = add(("lift", "dz2", "sub9" if head == "smce_near" else "sub", [t, "y"]), ("vec", n_o, bat))
        if head == "smce_scaled":
            dz2 = add(("scale", "dz2s", 2.0, dz2), ("vec", n_o, bat))
    elif head.startswith("logse"):
        dz2 = add(("lift", "dz2", "logse9" if head == "logse_near" else "logse", [a2, "y"]), ("vec", n_o, bat))
    else:
        dz2 = add(("lift", "dz2", "sub", [a2, "y"]), ("vec", n_o, bat))
    extra(dz2)
    dh = add(("gmul", "dh", (1, 1, 0), ("T", "W2"), dz2, False), ("vec", n_h, bat))
    dz1 = add(("lift", "dz1", "dlogistic", [dh, a1]), ("vec", n_h, bat))
    extra(dz1)
    for dz, a, W, b, tag in ((dz2, h, "W2", "b2", "2"), (dz1, "x", "W1", "b1", "1")):
        if bat:
            gW = add(("gmul", "gW" + tag, (1, 0, 1), dz, ("T", a), True), kinds[W])
            gb = add(("batchsum", "gb" + tag, dz), kinds[b])
        else:
            gW = add(("gmul", "gW" + tag, (1, 0, 1), dz, a, False), kinds[W])
            gb = dz
        if rng.random() < 0.8:
            add(("lift", "nW" + tag, "sgd", [W, gW]), kinds[W])
        if rng.random() < 0.8:
            add(("lift", "nb" + tag, "sgd", [b, gb]), kinds[b])
    # sometimes the update lands in the parameters' own storage (one call or one per tensor) and a second forward
    # pass reads them: what was recorded against the OLD values must still see the old values
    news = [(n[1:], n) for n in kinds if n.startswith(("nW", "nb"))]       # ("W2", "nW2") ...
    if news and rng.random() < 0.6:
        sel = [pr for pr in news if rng.random() < 0.8] or news[:1]
        if rng.random() < 0.5:
            steps.append(("copy_many", "cm", [d for d, _ in sel], [s_ for _, s_ in sel]))
        else:
            for d, s_ in sel:
                steps.append(("copy", "c" + d, d, s_))
        q1 = add(("gmul", "q1", (1, 1, 0), "W1", "x", False), ("vec", n_h, bat))
        qa = add(("sum", "qa", [q1, "b1"]), ("vec", n_h, bat))
        qh = add(("lift", "qh", "logistic", [qa]), ("vec", n_h, bat))
        q2 = add(("gmul", "q2", (1, 1, 0), "W2", qh, False), ("vec", n_o, bat))
        add(("sum", "q3", [q2, "b2"]), ("vec", n_o, bat))
    return leaves, steps, kinds, {"B": max(B, 1), "head": head, "batched": bat}


def run_program(T, leaves, steps, inputs, demand_order, lazy, late):
    from tensor_ops_amd import capi
    cl = {c[0]: c for c in closures()}
    prev = set_lazy(lazy)
    try:
        env = {}
        for name, kind in leaves.items():
            env[name] = T.put(inputs[name], batched=(kind[0] == "vec" and kind[2]))

        def val(ref):
            if isinstance(ref, tuple):
                return T.transp(env[ref[1]])
            return env[ref]
        results = {}
        with T.memo():
            for st in steps:
                if st[0] == "gmul":
                    _, name, (lm, lo, ln), a, b, red = st
                    f = T.gmul_batch_sum if red else T.gmul
                    env[name] = f(lm, lo, ln, val(a), val(b))
                elif st[0] == "sum":
                    xs = [env[v] for v in st[2]]
                    env[st[1]] = T.sumT(xs, xs[0].shape)
                elif st[0] == "lift":
                    c = cl[st[2]]
                    env[st[1]] = T.liftT(c[2], [val(v) for v in st[3]], key=("fuzz", c[0]))
                elif st[0] == "konst":
                    env[st[1]] = T.konst((st[2],), st[3])
                elif st[0] == "scale":
                    env[st[1]] = T.scaleT(st[2], env[st[3]])
                elif st[0] == "sumrows":
                    env[st[1]] = T.sumRows(env[st[2]])
                elif st[0] == "force":      # the host looks at a value in the middle of the recording
                    results[st[1]] = env[st[2]].numpy()
                elif st[0] == "drop":       # ... or lets go of one it will not use again
                    del env[st[2]]
                elif st[0] == "copy":
                    capi.check(capi.lib().to_copy_into(env[st[2]].h, env[st[3]].h))
                elif st[0] == "copy_many":
                    n = len(st[2])
                    d = (capi.c_tensor * n)(*[env[v].h for v in st[2]])
                    s_ = (capi.c_tensor * n)(*[env[v].h for v in st[3]])
                    capi.check(capi.lib().to_copy_into_many(n, d, s_))
                elif st[0] == "batchsum":
                    h = capi.c_tensor()
                    capi.check(capi.lib().to_batch_sum(env[st[2]].h, C.byref(h)))
                    from tensor_ops_amd.hipt import DT
                    env[st[1]] = DT(h)
            for name in demand_order:
                if name not in late:
                    results[name] = env[name].numpy()
        for name in demand_order:
            if name in late:
                results[name] = env[name].numpy()     # asked for after the scope closed
        return results
    finally:
        set_lazy(prev)


# (a long sweep: TOPS_FUZZ_CASES=12000, 24,000 graphs with both element types, a minute on the GPU.  Case 3562 is kept
#  by name: `v * logistic'(v)` next to `logistic v` -- the rewrite to d*h(1-h) once dropped v's second consumer and the
#  GEMM that makes v handed out only logistic v)
@pytest.mark.parametrize("case", sorted(set(range(int(os.environ.get("TOPS_FUZZ_CASES", "120")))) | {3562}))
def test_recorded_graphs_equal_eager_execution(T, case):
    rng = np.random.default_rng(SEED + case)
    leaves, steps, kinds, sizes = build_program(rng)
    if not steps:
        pytest.skip("empty program")
    B = sizes["B"]
    inputs = {}
    for name, kind in leaves.items():
        if kind[0] == "vec":
            shape = ((B,) if kind[2] else ()) + (kind[1],)
        else:
            shape = (kind[1], kind[2])
        inputs[name] = rng.uniform(-1, 1, size=shape)
    produced = [st[1] for st in steps]
    k = int(rng.integers(1, len(produced) + 1))
    demand = [produced[i] for i in rng.permutation(len(produced))[:k]]
    late = set(d for d in demand if rng.random() < 0.25)
    eager = run_program(T, leaves, steps, inputs, demand, False, late)
    lazy = run_program(T, leaves, steps, inputs, demand, True, late)
    tol = 1e-5 if T.dtype == np.float32 else 1e-11
    for name in demand:
        a, b = eager[name].astype(np.float64), lazy[name].astype(np.float64)
        assert a.shape == b.shape, (name, a.shape, b.shape)
        fin = np.isfinite(a)          # (a closure may overflow fp32 on these inputs: then it must do so in both runs)
        assert np.array_equal(fin, np.isfinite(b)) and np.array_equal(a[~fin], b[~fin], equal_nan=True), (case, name)
        a, b = np.where(fin, a, 0.0), np.where(fin, b, 0.0)
        den = max(np.linalg.norm(a.ravel()), 1e-30)
        assert np.linalg.norm((a - b).ravel()) / den < tol or np.allclose(a, b, rtol=0, atol=tol), (case, name, steps)


@pytest.mark.parametrize("case", range(int(os.environ.get("TOPS_FUZZ_CASES", "120"))))
def test_recorded_training_steps_equal_eager_execution(T, case):
    rng = np.random.default_rng(SEED + 1000003 + case)
    leaves, steps, kinds, sizes = build_program2(rng)
    B = sizes["B"]
    inputs = {}
    for name, kind in leaves.items():
        if kind[0] == "vec":
            shape = ((B,) if kind[2] else ()) + (kind[1],)
        else:
            shape = (kind[1], kind[2])
        inputs[name] = rng.uniform(-1, 1, size=shape) if name != "y" else rng.uniform(0, 1, size=shape)
    produced = [st[1] for st in steps if st[0] not in ("copy", "copy_many")]
    wanted = [n for n in produced if n.startswith(("nW", "nb", "gW", "gb", "dz", "e", "q", "dh"))] or produced
    k = int(rng.integers(1, len(wanted) + 1))
    demand = [wanted[i] for i in rng.permutation(len(wanted))[:k]]
    late = set(d for d in demand if rng.random() < 0.25)
    eager = run_program(T, leaves, steps, inputs, demand, False, late)
    lazy = run_program(T, leaves, steps, inputs, demand, True, late)
    tol = 1e-5 if T.dtype == np.float32 else 1e-11
    for name in demand:
        a, b = eager[name].astype(np.float64), lazy[name].astype(np.float64)
        assert a.shape == b.shape, (name, a.shape, b.shape)
        assert np.isfinite(a).all() and np.isfinite(b).all(), (case, name)
        den = max(np.linalg.norm(a.ravel()), 1e-30)
        assert np.linalg.norm((a - b).ravel()) / den < tol or np.allclose(a, b, rtol=0, atol=tol), (case, name, sizes, steps)


def _inputs_of(st):
    if st[0] == "gmul":
        return [r[1] if isinstance(r, tuple) else r for r in (st[3], st[4])]
    if st[0] == "sum":
        return list(st[2])
    if st[0] == "lift":
        return [r[1] if isinstance(r, tuple) else r for r in st[3]]
    if st[0] in ("scale",):
        return [st[3]]
    if st[0] in ("sumrows", "batchsum"):
        return [st[2]]
    if st[0] == "copy":
        return [st[2], st[3]]
    if st[0] == "copy_many":
        return list(st[2]) + list(st[3])
    return []


@pytest.mark.parametrize("case", range(int(os.environ.get("TOPS_FUZZ_CASES", "120"))))
def test_recorded_programs_with_values_forced_and_dropped_midway(T, case):
    """Both families again, with the host looking at values in the middle of the recording (a flush of part of the
    graph, the rest continues on what exists now) and releasing handles it no longer needs (liveness decides what a
    fused launch must still hand out)."""
    rng = np.random.default_rng(SEED + 2000003 + case)
    fam2 = bool(case % 2)
    leaves, steps, kinds, sizes = build_program2(rng) if fam2 else build_program(rng, extended=(case % 4 == 2))
    if not steps:
        pytest.skip("empty program")
    B = sizes["B"]
    inputs = {}
    for name, kind in leaves.items():
        shape = (((B,) if kind[2] else ()) + (kind[1],)) if kind[0] == "vec" else (kind[1], kind[2])
        inputs[name] = rng.uniform(0, 1, size=shape) if (fam2 and name == "y") else rng.uniform(-1, 1, size=shape)
    produced = [st[1] for st in steps if st[0] not in ("copy", "copy_many")]
    k = int(rng.integers(1, len(produced) + 1))
    demand = [produced[i] for i in rng.permutation(len(produced))[:k]]
    late = set(d for d in demand if rng.random() < 0.25)
    last_use = {}
    for i, st in enumerate(steps):
        for v in _inputs_of(st):
            last_use[v] = i
    out, nf = [], 0
    for i, st in enumerate(steps):
        out.append(st)
        made = [s2[1] for s2 in steps[:i + 1] if s2[0] not in ("copy", "copy_many")]
        if rng.random() < 0.12:
            out.append(("force", "force%d" % nf, made[int(rng.integers(len(made)))]))
            nf += 1
        for v in made:
            if v not in demand and last_use.get(v, -1) <= i and rng.random() < 0.3 and ("drop", "", v) not in out \
                    and not any(o[0] == "force" and o[2] == v for o in out[-1:]):
                out.append(("drop", "", v))
    # a dropped value must not be forced later
    dropped, steps2 = set(), []
    for st in out:
        if st[0] == "drop":
            dropped.add(st[2])
        if st[0] == "force" and st[2] in dropped:
            continue
        steps2.append(st)
    forced = [st[1] for st in steps2 if st[0] == "force"]
    eager = run_program(T, leaves, steps2, inputs, demand, False, late)
    lazy = run_program(T, leaves, steps2, inputs, demand, True, late)
    tol = 1e-5 if T.dtype == np.float32 else 1e-11
    for name in demand + forced:
        a, b = eager[name].astype(np.float64), lazy[name].astype(np.float64)
        assert a.shape == b.shape, (name, a.shape, b.shape)
        fin = np.isfinite(a)
        assert np.array_equal(fin, np.isfinite(b)) and np.array_equal(a[~fin], b[~fin], equal_nan=True), (case, name)
        a, b = np.where(fin, a, 0.0), np.where(fin, b, 0.0)
        den = max(np.linalg.norm(a.ravel()), 1e-30)
        assert np.linalg.norm((a - b).ravel()) / den < tol or np.allclose(a, b, rtol=0, atol=tol), (case, name, sizes, steps2)


def test_four_threads_record_concurrently(T):
    """Scopes, memo tables and the pending-op list under four host threads recording different programs at once
    (a Haskell RTS with -N calls in from several capabilities): every thread gets the values the single-threaded
    eager run got."""
    import threading
    progs = []
    for case in range(24):
        rng = np.random.default_rng(SEED + 3000003 + case)
        leaves, steps, kinds, sizes = build_program2(rng) if case % 2 else build_program(rng, extended=True)
        if not steps:
            continue
        B = sizes["B"]
        inputs = {}
        for name, kind in leaves.items():
            shape = (((B,) if kind[2] else ()) + (kind[1],)) if kind[0] == "vec" else (kind[1], kind[2])
            inputs[name] = rng.uniform(0, 1, size=shape) if name == "y" else rng.uniform(-1, 1, size=shape)
        produced = [st[1] for st in steps if st[0] not in ("copy", "copy_many")]
        demand = [produced[i] for i in rng.permutation(len(produced))[:max(1, len(produced) // 2)]]
        progs.append((leaves, steps, inputs, demand, run_program(T, leaves, steps, inputs, demand, False, set())))
    prev = set_lazy(True)
    errors = []

    def work(k):
        try:
            for rep in range(3):
                for leaves, steps, inputs, demand, want in progs[k::4]:
                    got = run_program(T, leaves, steps, inputs, demand, True, set(demand[::3]))
                    for name in demand:
                        a, b = want[name].astype(np.float64), got[name].astype(np.float64)
                        fin = np.isfinite(a)
                        if a.shape != b.shape or not np.array_equal(fin, np.isfinite(b)):
                            errors.append((k, name, "shape / finiteness"))
                            continue
                        a, b = np.where(fin, a, 0.0), np.where(fin, b, 0.0)
                        tol = 1e-5 if T.dtype == np.float32 else 1e-11
                        if not (np.linalg.norm((a - b).ravel()) <= tol * max(np.linalg.norm(a.ravel()), 1e-30) or
                                np.allclose(a, b, rtol=0, atol=tol)):
                            errors.append((k, name, float(np.abs(a - b).max())))
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))
    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    set_lazy(prev)
    assert not errors, errors[:5]


def test_the_sweep_exercised_the_fusion_rules(T):
    """(runs after the sweep above) the random graphs did reach the planner's rules: launches were fused and recorded
    ops were left without storage of their own."""
    from tensor_ops_amd import capi
    a = [C.c_int64() for _ in range(4)]
    capi.check(capi.lib().to_lazy_stats(*[C.byref(v) for v in a]))
    recorded, fused, elided, flushes = [v.value for v in a]
    assert recorded > 500 and fused > 30 and elided > 20 and flushes > 100, (recorded, fused, elided, flushes)
