"""Random closures for liftT -- trees over the whole symbolic vocabulary (+ - * / neg recip exp log sqrt abs signum sin
cos tanh pow max min, constants), arity 1..3 -- evaluated by the library (run-time specialised kernel, pre-fused functor
when the classifier recognises one, or the bytecode VM with TOPS_EXPR_JIT=0) and by numpy in double (1e-5 / 1e-11, widened only by numpy's own drift in the element type).
usage: expr_fuzz.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd import hipt
from tensor_ops_amd.hipt import HipT, Sym

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
NP1 = {"exp": np.exp, "log": np.log, "sqrt": np.sqrt, "sin": np.sin, "cos": np.cos, "tanh": np.tanh,
       "recip": lambda x: 1.0 / x, "abs": np.abs, "signum": np.sign}


def un(name, x):
    if isinstance(x, Sym):
        return x.__tops_unary__(name)
    return NP1[name](x)


def mx(a, b):
    return hipt.maximum(a, b) if isinstance(a, Sym) or isinstance(b, Sym) else np.maximum(a, b)


def mn(a, b):
    return hipt.minimum(a, b) if isinstance(a, Sym) or isinstance(b, Sym) else np.minimum(a, b)


def gen(depth, arity):
    """returns (function of the argument list, description)"""
    r = rng.random()
    if depth == 0 or r < 0.15:
        if rng.random() < 0.25:
            c = float(rng.choice([0.0, 1.0, -1.0, 0.5, 2.0, -0.25, 3.0]))
            return (lambda v, c=c: c + 0.0 * v[0]), "%g" % c
        i = int(rng.integers(arity))
        return (lambda v, i=i: v[i]), "x%d" % i
    k = rng.choice(["add", "sub", "mul", "div", "neg", "recip", "exp", "log", "sqrt", "abs", "signum", "sin", "cos", "tanh",
                    "pow", "max", "min", "affine"])
    a, da = gen(depth - 1, arity)
    if k in ("add", "sub", "mul", "max", "min"):
        b, db = gen(depth - 1, arity)
        f = {"add": lambda x, y: x + y, "sub": lambda x, y: x - y, "mul": lambda x, y: x * y, "max": mx, "min": mn}[k]
        return (lambda v: f(a(v), b(v))), "%s(%s,%s)" % (k, da, db)
    if k == "div":
        b, db = gen(depth - 1, arity)
        return (lambda v: a(v) / (un("abs", b(v)) + 0.5)), "(%s / (|%s|+.5))" % (da, db)
    if k == "neg":
        return (lambda v: -a(v)), "-(%s)" % da
    if k == "recip":
        return (lambda v: un("recip", un("abs", a(v)) + 0.5)), "recip(|%s|+.5)" % da
    if k == "exp":
        return (lambda v: un("exp", mn(a(v), 4.0))), "exp(min(%s,4))" % da
    if k in ("log", "sqrt"):
        return (lambda v: un(k, un("abs", a(v)) + 0.5)), "%s(|%s|+.5)" % (k, da)
    if k in ("sin", "cos"):   # (bounded argument: sin(400) in fp32 is conditioned 400 times worse than its input)
        return (lambda v: un(k, mn(mx(a(v), -8.0), 8.0))), "%s(clamp8(%s))" % (k, da)
    if k in ("abs", "signum", "tanh"):
        return (lambda v: un(k, a(v))), "%s(%s)" % (k, da)
    if k == "pow":
        e = float(rng.choice([2.0, 0.5, -1.0, 3.0, 1.5]))
        return (lambda v: (un("abs", a(v)) + 0.5) ** e), "(|%s|+.5)^%g" % (da, e)
    c0, c1 = float(rng.choice([0.5, -1.0, 2.0, -0.125])), float(rng.choice([0.0, 1.0, -0.25]))
    return (lambda v: c0 * a(v) + c1), "(%g*%s%+g)" % (c0, da, c1)


bad = 0
Ts = {np.float32: HipT(0), np.float64: HipT(0, dtype=np.float64)}
for case in range(n_cases):
    arity = int(rng.integers(1, 4))
    f, desc = gen(int(rng.integers(1, 6)), arity)
    dt = np.float32 if rng.random() < 0.7 else np.float64
    T = Ts[dt]
    shape = tuple(int(x) for x in rng.choice([1, 3, 17, 257, 1000], size=int(rng.integers(1, 3))))
    xs = [rng.uniform(-2, 2, shape).astype(dt) for _ in range(arity)]
    if rng.random() < 0.3:      # values where abs / signum / max / min have their kinks
        xs[0].ravel()[:: 3] = 0.0
    with np.errstate(all="ignore"):
        want = np.broadcast_to(np.asarray(f([x.astype(np.float64) for x in xs]), dtype=np.float64), shape)
    try:
        got = T.liftT(f, [T.put(x) for x in xs], key=("exprfuzz", seed, case)).numpy().astype(np.float64)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("ERROR", case, desc, repr(e)[:200])
        continue
    # 1e-5 (fp32) / 1e-11 (fp64) of max(|value|, 1), plus -- a chain of five fp32 transcendentals legitimately drifts
    # further -- four times what numpy's own evaluation in the element type differs from the double one
    tol = 1e-5 if dt == np.float32 else 1e-11
    with np.errstate(all="ignore"):
        same = np.broadcast_to(np.asarray(f([x for x in xs])), shape).astype(np.float64)
    fin = np.isfinite(want)
    scale = np.maximum(np.abs(want), 1.0)
    band = tol * scale + 4.0 * np.where(np.isfinite(same), np.abs(same - want), 0.0)
    ok = got.shape == want.shape and np.array_equal(fin, np.isfinite(got)) and np.all(np.abs(got - want)[fin] <= band[fin])
    if not ok:
        bad += 1
        idx = np.argwhere(~(np.abs(got - want) <= band))[:3] if got.shape == want.shape else []
        print("MISMATCH", case, dt.__name__, desc, shape, [(tuple(i), want[tuple(i)], got[tuple(i)]) for i in idx])
print("cases", n_cases, "mismatches", bad)
