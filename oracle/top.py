"""Oracle (test infrastructure) -- the `TOp` DSL restated in Python.

`TOp ns ms` is a pair of closures polymorphic in the tensor backend
(src/TensorOps/Types.hs:122-125).  Here the backend dictionary is passed
explicitly: `run(T, xs)` / `grad(T, xs, ds)` with `T` an `oracle.tensor.OTensor`
(or anything with the same methods).  `n_in`/`n_out` stand for the type-level
`Known Length ns` / `Known Length ms` evidence used by `firstOp` & co.

Products (`Prod t ns`) are Python lists.
"""
import numpy as np

from . import ad


class Lazy:
    """A lazily evaluated `Prod` -- Haskell thunk semantics for the recomputed
    forward pass `f1 xs` inside `g3` (Types.hs:155): a backward closure that
    ignores its `xs` argument (e.g. `add`, TOp.hs:218) never forces it."""

    def __init__(self, thunk):
        self._thunk = thunk
        self._val = None

    def force(self):
        if self._thunk is not None:
            self._val = list(self._thunk())
            self._thunk = None
        return self._val

    def __getitem__(self, i):
        if isinstance(i, slice):
            return Lazy(lambda: self.force()[i])
        return self.force()[i]

    def __len__(self):
        return len(self.force())

    def __iter__(self):
        return iter(self.force())


def _list(xs):
    return xs.force() if isinstance(xs, Lazy) else list(xs)


class TOp:
    def __init__(self, n_in, n_out, run, grad):
        self.n_in = n_in
        self.n_out = n_out
        self.run = run      # (T, xs) -> ys
        self.grad = grad    # (T, xs, dys) -> dxs

    # `>>>` of Control.Category: self then other
    def __rshift__(self, other):
        return compose(other, self)


def runTOp(op, T, xs):
    ys = op.run(T, list(xs))
    assert len(ys) == op.n_out
    return ys


def gradTOp(op, T, xs):
    """`gradTOp` (Types.hs:127-132): seed the single scalar output with 1 --
    `only (getI $ generateA (\\_ -> I 1))`, i.e. built through `generateA` (:132)."""
    assert op.n_out == 1
    return op.grad(T, list(xs), [T.generate((), lambda _i: 1)])


# ---- Category / product combinators (Types.hs:135-264) ----------------------
def idOp(n):
    return TOp(n, n, lambda T, xs: _list(xs), lambda T, xs, ds: list(ds))


def compose(f2, f1):
    """`TOp f2 g2 . TOp f1 g1` (Types.hs:139-156):
    f3 = f2 . f1 ;  g3 xs ds = g1 xs (g2 (f1 xs) ds)  -- f1 xs is RECOMPUTED."""
    assert f1.n_out == f2.n_in, (f1.n_out, f2.n_in)
    return TOp(f1.n_in, f2.n_out,
               lambda T, xs: f2.run(T, f1.run(T, xs)),
               lambda T, xs, ds: f1.grad(T, xs, f2.grad(T, Lazy(lambda: f1.run(T, xs)), ds)))


def firstOp(op):
    """`firstOp` (Types.hs:165-182): act on the leading n_in, pass the rest."""
    def f(T, xs):
        return op.run(T, xs[:op.n_in]) + _list(xs[op.n_in:])

    def g(T, xs, ds):
        return op.grad(T, xs[:op.n_in], ds[:op.n_out]) + list(ds[op.n_out:])
    # arity of the pass-through part is only known at call time (type var `os`)
    return _Open(op, f, g, front=True)


def secondOp(n_skip, op):
    """`secondOp` (Types.hs:184-201): pass the leading `n_skip`, act on the rest."""
    def f(T, xs):
        return _list(xs[:n_skip]) + op.run(T, xs[n_skip:])

    def g(T, xs, ds):
        return list(ds[:n_skip]) + op.grad(T, xs[n_skip:], ds[n_skip:])
    return TOp(n_skip + op.n_in, n_skip + op.n_out, f, g)


class _Open(TOp):
    """A `firstOp` whose pass-through length `os` is fixed on first use
    (Haskell infers it from the composition site)."""

    def __init__(self, op, f, g, front):
        super().__init__(None, None, f, g)
        self.inner = op

    def close(self, n_pass):
        return TOp(self.inner.n_in + n_pass, self.inner.n_out + n_pass, self.run, self.grad)


def first(op, n_pass):
    """`firstOp @os` with the pass-through arity made explicit."""
    return firstOp(op).close(n_pass)


def then_first(t1, t2):
    """`t1 *>> t2 = firstOp t1 >>> t2` (Types.hs:204-209)."""
    n_pass = t2.n_in - t1.n_out
    assert n_pass >= 0
    return compose(t2, first(t1, n_pass))


def par(o1, o2):
    """`***` (Types.hs:221-241)."""
    def f(T, xs):
        return o1.run(T, xs[:o1.n_in]) + o2.run(T, xs[o1.n_in:])

    def g(T, xs, ds):
        return (o1.grad(T, xs[:o1.n_in], ds[:o1.n_out]) +
                o2.grad(T, xs[o1.n_in:], ds[o1.n_out:]))
    return TOp(o1.n_in + o2.n_in, o1.n_out + o2.n_out, f, g)


def fanout(o1, o2, shapes):
    """`&&&` (Types.hs:243-264): backward sums the two cotangents with
    `sumT [gxy, gxz]` (:259).  `shapes` = the `SingI as` evidence."""
    assert o1.n_in == o2.n_in

    def f(T, xs):
        return o1.run(T, xs) + o2.run(T, xs)

    def g(T, xs, ds):
        g1 = o1.grad(T, xs, ds[:o1.n_out])
        g2 = o2.grad(T, xs, ds[o1.n_out:])
        return [T.sumT([a, b], s) for s, a, b in zip(shapes, g1, g2)]
    return TOp(o1.n_in, o1.n_out + o2.n_out, f, g)


# ---- op vocabulary (src/TensorOps/TOp.hs) ------------------------------------
class VFunc:
    """`VFunc n` (Types.hs:114-117): function and gradient, both polymorphic."""

    def __init__(self, f, g):
        self.f = f   # list of n values -> value
        self.g = g   # list of n values -> list of n partials


def gradLift(T, vf, xs, dtdy):
    """`TT.gradLift` (src/TensorOps/Tensor.hs:119-129): for each input i one
    (n+1)-ary liftT  \\(d :* x) -> d * (vfGrad f x)_i."""
    xs = _list(xs)
    n = len(xs)
    return [T.liftT(lambda dx, i=i: dx[0] * vf.g(dx[1:])[i], [dtdy] + list(xs))
            for i in range(n)]


def liftOp(n, vf, shape=None):
    """`liftOp` (TOp.hs:42-54).  n == 0 is the `UØ` branch (a constant)."""
    if n == 0:
        return TOp(0, 1, lambda T, xs: [T.konst(shape, vf.f([]))], lambda T, xs, ds: [])
    return TOp(n, 1,
               lambda T, xs: [T.liftT(vf.f, _list(xs))],
               lambda T, xs, ds: gradLift(T, vf, xs, ds[0]))


def gmul(len_m, len_o, len_n):
    """`TO.gmul` (TOp.hs:56-94).
    fwd  : T.gmul lM lO lN x y                                   (:68)
    dx   : T.gmul lM lN lO dtdz (transp y)                       (:81)
    dy   : T.gmul (rev lO) (rev lM) lN (transp x) dtdz           (:86-88)"""
    def f(T, xs):
        return [T.gmul(len_m, len_o, len_n, xs[0], xs[1])]

    def g(T, xs, ds):
        x, y = xs
        dtdz = ds[0]
        dx = T.gmul(len_m, len_n, len_o, dtdz, T.transp(y))
        dy = T.gmul(len_o, len_m, len_n, T.transp(x), dtdz)
        return [dx, dy]
    return TOp(2, 1, f, g)


def transpOp():
    """`transpOp` (TOp.hs:97-104)."""
    return TOp(1, 1, lambda T, xs: [T.transp(xs[0])], lambda T, xs, ds: [T.transp(ds[0])])


def shuffle(idx, in_shapes):
    """`shuffle` (TOp.hs:106-131): outputs = inputs selected by `idx`;
    backward: each input gets `sumT` of the cotangents of the outputs that
    selected it (possibly none -> zero tensor)."""
    def f(T, xs):
        return [xs[i] for i in idx]

    def g(T, xs, ds):
        return [T.sumT([d for k, d in zip(idx, ds) if k == i], in_shapes[i])
                for i in range(len(in_shapes))]
    return TOp(len(in_shapes), len(idx), f, g)


def sumRows():
    """`TO.sumRows` (TOp.hs:151-159): backward broadcasts dtdz into every row
    via `mapRows (LS LZ) (\\_ -> dtdz) x`."""
    return TOp(1, 1,
               lambda T, xs: [T.sumRows(xs[0])],
               lambda T, xs, ds: [T.mapRows(1, lambda _r: ds[0], xs[0])])


def sumOp(n, shape):
    """`sumOp` (TOp.hs:161-169)."""
    return TOp(n, 1,
               lambda T, xs: [T.sumT(_list(xs), shape)],
               lambda T, xs, ds: [ds[0] for _ in range(n)])


def scale(alpha):
    """`scale` (TOp.hs:171-177)."""
    return TOp(1, 1,
               lambda T, xs: [T.scaleT(alpha, xs[0])],
               lambda T, xs, ds: [T.scaleT(alpha, ds[0])])


def konst(n, shape, x):
    """`konst` (TOp.hs:185-192)."""
    return TOp(0, n, lambda T, xs: [T.konst(shape, x) for _ in range(n)], lambda T, xs, ds: [])


def negate():
    """`negate = scale (-1)` (TOp.hs:194-196)."""
    return scale(-1)


def map_(f, f_prime=None):
    """`map' f f'` (TOp.hs:198-206) / `map f = map' f (diff f)` (:209-213)."""
    if f_prime is None:
        f_prime = ad.diff(f)
    return liftOp(1, VFunc(lambda xs: f(xs[0]), lambda xs: [f_prime(xs[0])]))


def add(shape=None):
    """`add` (TOp.hs:215-221): forward `sumT [x,y]`."""
    return TOp(2, 1,
               lambda T, xs: [T.sumT([xs[0], xs[1]], np.shape(xs[0]))],
               lambda T, xs, ds: [ds[0], ds[0]])


def add3():
    """`add3` (TOp.hs:223-229)."""
    return TOp(3, 1,
               lambda T, xs: [T.sumT(_list(xs), np.shape(xs[0]))],
               lambda T, xs, ds: [ds[0], ds[0], ds[0]])


def zipN(n, f, f_grad=None):
    """`zipN' u f f'` (TOp.hs:232-239) / `zipN u f = zipN' u f (grad f)` (:241-247)."""
    if f_grad is None:
        f_grad = ad.grad(f)
    return liftOp(n, VFunc(f, f_grad))


def zip_(f, f_grad=None):
    """`zip` / `zip'` (TOp.hs:249-266)."""
    if f_grad is None:
        return zipN(2, lambda xs: f(xs[0], xs[1]))
    return zipN(2, lambda xs: f(xs[0], xs[1]), lambda xs: list(f_grad(xs[0], xs[1])))


def zip3(f, f_grad=None):
    """`zip3` / `zip3'` (TOp.hs:268-285)."""
    if f_grad is None:
        return zipN(3, lambda xs: f(*xs))
    return zipN(3, lambda xs: f(*xs), lambda xs: list(f_grad(*xs)))


def replicate(n):
    """`replicate` (TOp.hs:287-293): backward `sumT` of the n cotangents."""
    return TOp(1, n,
               lambda T, xs: [xs[0] for _ in range(n)],
               lambda T, xs, ds: [T.sumT(list(ds), np.shape(xs[0]))])


def duplicate():
    """`duplicate` (TOp.hs:295-302)."""
    return TOp(1, 2,
               lambda T, xs: [xs[0], xs[0]],
               lambda T, xs, ds: [T.sumT([ds[0], ds[1]], np.shape(xs[0]))])


def inner(len_m, len_n):
    """`inner lM lN = gmul lM (LS LZ) lN` (TOp.hs:304-311)."""
    return gmul(len_m, 1, len_n)


def outer(len_m, len_n):
    """`outer lM lN = gmul lM LZ lN` (TOp.hs:313-320)."""
    return gmul(len_m, 0, len_n)


def dot():
    """`dot = inner LZ LZ` (TOp.hs:322-325)."""
    return inner(0, 0)


def matVec():
    """`matVec = inner (LS LZ) LZ` (TOp.hs:327-331)."""
    return inner(1, 0)


def vecMat():
    """`vecMat = inner LZ (LS LZ)` (TOp.hs:333-337)."""
    return inner(0, 1)


def matMat():
    """`matMat = inner (LS LZ) (LS LZ)` (TOp.hs:339-343)."""
    return inner(1, 1)


def swap():
    """`swap` (TOp.hs:346-351)."""
    return TOp(2, 2,
               lambda T, xs: [xs[1], xs[0]],
               lambda T, xs, ds: [ds[1], ds[0]])


def swap_n(n_front, n_back):
    """`swap'` (TOp.hs:353-360) = `shuffleF swapProd swapProd`: ns ++ ms -> ms ++ ns,
    a pure re-ordering both ways (no `sumT`)."""
    n = n_front + n_back
    return TOp(n, n,
               lambda T, xs: _list(xs[n_front:]) + _list(xs[:n_front]),
               lambda T, xs, ds: list(ds[n_back:]) + list(ds[:n_back]))


def drop(n_drop, shapes):
    """`drop` (TOp.hs:362-370): dropped inputs get `sumT []` = zeros."""
    n = len(shapes)

    def g(T, xs, ds):
        return [T.sumT([], shapes[i]) for i in range(n_drop)] + list(ds)
    return TOp(n, n - n_drop, lambda T, xs: _list(xs[n_drop:]), g)


def take(n_take, shapes):
    """`take` (TOp.hs:372-381)."""
    n = len(shapes)

    def g(T, xs, ds):
        return list(ds) + [T.sumT([], shapes[i]) for i in range(n_take, n)]
    return TOp(n, n_take, lambda T, xs: _list(xs[:n_take]), g)
