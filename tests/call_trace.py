"""Test infrastructure: the class-method call stream of a TOp program as a dataflow graph, from two sources.

* `TracingTensor` -- the oracle's numpy backend (`oracle/tensor.py`) with every `class Tensor` method
  (src/TensorOps/Types.hs:52-109) logged.  Running `oracle/top.py` (the restatement of the reference's DSL)
  on it yields the calls the reference's `TOp` closures make, with the identities of their operands.
* `parse_mirror_log` -- the log the C++ host mirror writes (tensor-ops_amd/host/tensorops/trace.hpp), i.e. the
  calls that reach the C ABI -- the ones `instance Tensor HipT` (hs/TensorOps/Backend/HipTensor.hs) would send.

`canonical()` turns either into a set of structural hashes: a call is identified by its method, its static
arguments and (recursively) the calls that produced its operands, leaves by their position among the program's
inputs.  Evaluation order and repetition do not enter (the oracle is strict Python and computes cotangents
nobody asks for, the mirror is call-by-need like Haskell): the oracle's set is restricted to what the demanded
results depend on, and the two sets must then be EQUAL -- every call the mirror issues is one the reference's
closures make on the same operands, and nothing the results need is obtained any other way.
"""
import hashlib
import itertools

import numpy as np

from oracle.tensor import OTensor

POINTS = (lambda i: 0.3 + 0.17 * i, lambda i: 0.7 + 0.29 * i)  # the closure fingerprint's two points (trace.hpp); positive: log, sqrt


def num(v):
    return "%.6e" % float(v)


class Rec:
    __slots__ = ("method", "params", "ins", "out", "dims")

    def __init__(self, method, params, ins, out, dims):
        self.method, self.params, self.ins, self.out, self.dims = method, tuple(params), tuple(ins), out, tuple(dims)

    def __repr__(self):
        return f"{self.method}({','.join(self.params)}) {list(self.ins)} -> {self.out} {self.dims}"


class TracingTensor(OTensor):
    """`OTensor` + a log of its class-method calls."""

    def __init__(self, dtype=np.float64):
        super().__init__(dtype)
        self.recs = []
        self._ids = {}
        self._keep = []

    # identities: numpy arrays by object, kept alive while the log is
    def name(self, arr, vid):
        self._ids[id(arr)] = vid
        self._keep.append(arr)

    def leaves(self, arrs):
        for i, a in enumerate(arrs):
            self.name(a, i)
        self._next = len(arrs)

    def _id(self, arr):
        v = self._ids.get(id(arr))
        if v is None:
            v = self._next
            self._next += 1
            self.name(arr, v)
            self.recs.append(Rec("L", (), (), v, np.shape(arr)))
        return v

    def _log(self, method, params, ins, out):
        out = np.array(out, copy=True)  # a fresh identity for every result
        in_ids = [self._id(x) for x in ins]
        v = self._next
        self._next += 1
        self.name(out, v)
        self.recs.append(Rec(method, params, in_ids, v, out.shape))
        return out

    def id_of(self, arr):
        return self._ids[id(arr)]

    # -- the class methods ---------------------------------------------------------------------------
    def liftT(self, f, xs):
        xs = list(xs)
        fp = [str(len(xs))] + [num(f([p(i) for i in range(len(xs))])) for p in POINTS]
        return self._log("liftT", fp, xs, super().liftT(f, xs))

    def gmul(self, lm, lo, ln, x, y):
        return self._log("gmul", (str(lm), str(lo), str(ln)), [x, y], super().gmul(lm, lo, ln, x, y))

    def sumT(self, xs, shape):
        xs = list(xs)
        return self._log("sumT", (str(len(xs)),), xs, super().sumT(xs, shape))

    def scaleT(self, alpha, x):
        return self._log("scaleT", (num(alpha),), [x], super().scaleT(alpha, x))

    def transp(self, x):
        return self._log("transp", (), [x], super().transp(x))

    def sumRows(self, x):
        return self._log("sumRows", (), [x], super().sumRows(x))

    def _rows(self, method, length, f, x, with_index):
        x = np.asarray(x)
        lead = x.shape[:length]
        rows = []
        for i in itertools.product(*[range(d) for d in lead]):
            row = self._log("row", [str(k) for k in i], [x], x[i])
            rows.append(np.asarray(f(i, row) if with_index else f(row), dtype=self.dtype))
        out = np.stack(rows).reshape(lead + rows[0].shape) if rows else np.zeros(lead, self.dtype)
        return self._log(method, (str(length),), [x] + rows, out)

    def mapRows(self, len_n, f, x):
        return self._rows("mapRows", len_n, f, x, False)

    def ixRows(self, len_m, f, x):
        return self._rows("ixRows", len_m, f, x, True)

    def diag(self, rank, x):
        return self._log("diag", (str(rank),), [x], super().diag(rank, x))

    def getDiag(self, x):
        return self._log("getDiag", (), [x], super().getDiag(x))

    def generate(self, shape, f):
        out = super().generate(shape, f)
        return self._log("generateA", [num(v) for v in np.ravel(out)], [], out)

    def konst(self, shape, x):
        return self._log("konst", (num(x),), [], super().konst(shape, x))


def parse_mirror_log(text):
    recs = []
    for line in text.splitlines():
        if not line:
            continue
        f = line.split("\t")
        if f[0] == "L":
            dims = f[2].split("|")[0]
            recs.append(Rec("L", (), (), int(f[1]), [int(d) for d in dims.split("x")] if dims else []))
            continue
        method, params, ins, out, shape = f
        dims = shape.split("|")[0]
        recs.append(Rec(method, params.split(",") if params else (), [int(i) for i in ins.split(",")] if ins else (),
                        int(out), [int(d) for d in dims.split("x")] if dims else []))
    return recs


def _h(obj):
    return hashlib.sha1(repr(obj).encode()).hexdigest()[:16]


def canonical(recs, n_leaves, roots=None):
    """{structural hash: description} of the calls in `recs` (restricted to what `roots` depend on)."""
    val = {i: ("leaf", i) for i in range(n_leaves)}
    alias = {}
    by_out = {}
    order = []
    for r in recs:
        if r.method == "L":
            val.setdefault(r.out, ("L", r.dims))
            continue
        # the batching extension: gmul_batch_sum is gmul with the sum over samples fused, batch_sum
        # that sum alone; per sample they are gmul and the identity.  `sumT [x] = x`.
        ident = r.method == "batch_sum" or (r.method == "sumT" and len(r.ins) == 1)
        if ident:
            if r.out not in val:
                val[r.out] = val[r.ins[0]]
                alias[r.out] = r.ins[0]
            continue
        method = "gmul" if r.method == "gmul_batch_sum" else r.method
        h = _h((method, r.params, tuple(val[i] for i in r.ins), r.dims))
        if r.out not in val:
            val[r.out] = h
            by_out[r.out] = r
        order.append((h, r, method))
    if roots is not None:
        need, stack = set(), list(roots)
        while stack:
            v = stack.pop()
            v = alias.get(v, v)
            while v in alias:
                v = alias[v]
            if v in need:
                continue
            need.add(v)
            if v in by_out:
                stack.extend(by_out[v].ins)
        order = [(h, r, m) for h, r, m in order if r.out in need]
    return {h: f"{m}({','.join(r.params)}) -> {tuple(r.dims)}" for h, r, m in order}


def diff(want, got):
    """human-readable difference of two canonical sets ('' when equal)"""
    lines = []
    for h in sorted(set(want) - set(got)):
        lines.append("  the reference's closures call, the mirror does not:  " + want[h])
    for h in sorted(set(got) - set(want)):
        lines.append("  the mirror calls, the reference's closures do not:   " + got[h])
    return "\n".join(lines)
