"""Oracle (test infrastructure) -- forward-mode automatic differentiation.

The reference derives elementwise gradients with the third-party `ad` package
(`Numeric.AD`: `diff` at src/TensorOps/TOp.hs:212, `grad` at :246; cabal
dependency `ad`, tensor-ops.cabal:49, pinned only by stackage lts-7.2 =
ad-4.3.2.1; NOT vendored under /root/reference).  `diff f x` is the derivative
of a scalar function by forward-mode dual numbers and `grad f xs` the gradient
(reverse mode in `ad`; mathematically the same vector).  We restate the
published algorithm: dual numbers `a + b eps`, eps^2 = 0, with constants
carrying no tangent.  `ad`'s exact internal operation order is not reproduced
bit for bit -- floating-point parity tolerances (1e-5 rel fp32, 1e-12 fp64)
cover the difference.

Polymorphic scalar functions (`forall a. RealFloat a => a -> a`,
src/TensorOps/Types.hs:114-117) are written against the module-level
`exp/log/sqrt/...` below, which dispatch on `Dual` or numpy values.
"""
import numpy as np


class Dual:
    __slots__ = ("p", "t")
    __array_priority__ = 1000  # make numpy defer to our reflected operators

    def __init__(self, p, t):
        self.p = p
        self.t = t

    @staticmethod
    def lift(x):
        return x if isinstance(x, Dual) else Dual(x, None)

    # tangent None == structurally zero (a lifted constant)
    def _tan(self):
        return 0.0 if self.t is None else self.t

    def __neg__(self):
        return Dual(-self.p, None if self.t is None else -self.t)

    def __add__(self, o):
        o = Dual.lift(o)
        if self.t is None and o.t is None:
            t = None
        elif self.t is None:
            t = o.t
        elif o.t is None:
            t = self.t
        else:
            t = self.t + o.t
        return Dual(self.p + o.p, t)

    __radd__ = __add__

    def __sub__(self, o):
        return self + (-Dual.lift(o))

    def __rsub__(self, o):
        return Dual.lift(o) + (-self)

    def __mul__(self, o):
        o = Dual.lift(o)
        if self.t is None and o.t is None:
            t = None
        elif self.t is None:
            t = self.p * o.t
        elif o.t is None:
            t = self.t * o.p
        else:
            t = self.t * o.p + self.p * o.t
        return Dual(self.p * o.p, t)

    __rmul__ = __mul__

    def __truediv__(self, o):
        o = Dual.lift(o)
        q = self.p / o.p
        if self.t is None and o.t is None:
            t = None
        elif o.t is None:
            t = self.t / o.p
        elif self.t is None:
            t = -(q / o.p) * o.t
        else:
            t = self.t / o.p - (q / o.p) * o.t
        return Dual(q, t)

    def __rtruediv__(self, o):
        return Dual.lift(o) / self

    def __pow__(self, k):
        if isinstance(k, Dual):
            return exp(k * log(self))
        p = self.p ** k
        t = None if self.t is None else k * self.p ** (k - 1) * self.t
        return Dual(p, t)


def _unary(name, f, df):
    """Dispatch on Dual / symbolic (anything with `__tops_unary__`) / numpy."""
    def g(x):
        if isinstance(x, Dual):
            y = g(x.p)
            return Dual(y, None if x.t is None else df(x.p, y) * x.t)
        hook = getattr(x, "__tops_unary__", None)
        if hook is not None:
            return hook(name)
        return f(x)
    return g


exp = _unary("exp", np.exp, lambda x, y: y)
log = _unary("log", np.log, lambda x, y: 1.0 / x)
sqrt = _unary("sqrt", np.sqrt, lambda x, y: 0.5 / y)
sin = _unary("sin", np.sin, lambda x, y: cos(x))
cos = _unary("cos", np.cos, lambda x, y: -sin(x))
tanh = _unary("tanh", np.tanh, lambda x, y: 1.0 - y * y)
signum = _unary("signum", np.sign, lambda x, y: 0.0 * x)
abs_ = _unary("abs", np.abs, lambda x, y: signum(x))


def _one_like(x):
    return x.one_like() if hasattr(x, "one_like") else np.ones_like(x)


def _zero_like(x):
    return x.zero_like() if hasattr(x, "zero_like") else np.zeros_like(x)


def recip(x):
    """`recip` of the `Fractional` class: 1 / x."""
    return 1.0 / x


def diff(f):
    """`Numeric.AD.diff` (used by `TO.map`, src/TensorOps/TOp.hs:209-213)."""
    def df(x):
        r = Dual.lift(f(Dual(x, _one_like(x))))
        return _zero_like(x) if r.t is None else r.t + _zero_like(x)
    return df


def grad(f):
    """`Numeric.AD.grad` (used by `TO.zipN`, src/TensorOps/TOp.hs:241-247):
    gradient of `f :: Vec n a -> a` as a list of n partials."""
    def gf(xs):
        xs = list(xs)
        out = []
        for i in range(len(xs)):
            args = [Dual(x, _one_like(x) if j == i else None) for j, x in enumerate(xs)]
            r = Dual.lift(f(args))
            base = _zero_like(xs[i])
            out.append(base if r.t is None else r.t + base)
        return out
    return gf
