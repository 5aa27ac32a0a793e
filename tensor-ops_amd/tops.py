"""ctypes view of the C++ host mirror (host/tensorops_host.h): the `TOp` DSL and the
Learn layer with the reference's names, running on the HIP backend.

Harness plumbing: tests build TOps here exactly the way the reference builds them
(`TO.matVec`, `>>>`, `firstOp`, `genNet`, `trainNetwork` ...) and compare with the
oracle.  Closures are handed over as SSA programs recorded by `hipt.Sym`.
"""
import ctypes as C
import os

from . import capi, hipt
from .capi import c_tensor, check as _check_to, dims_arr
from .hipt import DT

HOST_LIB_PATH = os.path.join(capi.HERE, "libtensorops_host.so")
c_op = C.c_void_p
c_net = C.c_void_p
c_trainer = C.c_void_p
c_rnn = C.c_void_p
i32p = C.POINTER(C.c_int32)
f64p = C.POINTER(C.c_double)
tp = C.POINTER(c_tensor)

SIGNATURES = {
    "toh_op_named": [C.c_char_p, C.c_int, C.c_double, C.POINTER(c_op)],
    "toh_op_gmul": [C.c_int, C.c_int, C.c_int, C.POINTER(c_op)],
    "toh_op_map": [C.c_int, i32p, C.c_int, f64p, C.POINTER(c_op)],
    "toh_op_map_with": [C.c_int, i32p, C.c_int, f64p, C.c_int, i32p, C.c_int, f64p, C.POINTER(c_op)],
    "toh_op_zipN": [C.c_int, C.c_int, i32p, C.c_int, f64p, C.POINTER(c_op)],
    "toh_op_zipN_with": [C.c_int, C.c_int, i32p, C.c_int, f64p, i32p, C.POINTER(i32p), i32p, C.POINTER(f64p),
                         C.POINTER(c_op)],
    "toh_op_sumOp": [C.c_int, C.c_int, capi.i64p, C.POINTER(c_op)],
    "toh_op_konst": [C.c_int, C.c_int, capi.i64p, C.c_double, C.POINTER(c_op)],
    "toh_op_shuffle": [C.c_int, C.c_int, i32p, C.POINTER(c_op)],
    "toh_op_drop": [C.c_int, C.c_int, C.POINTER(c_op)],
    "toh_op_take": [C.c_int, C.c_int, C.POINTER(c_op)],
    "toh_op_compose": [c_op, c_op, C.POINTER(c_op)],
    "toh_op_first": [c_op, C.c_int, C.POINTER(c_op)],
    "toh_op_second": [C.c_int, c_op, C.POINTER(c_op)],
    "toh_op_then_first": [c_op, c_op, C.POINTER(c_op)],
    "toh_op_par": [c_op, c_op, C.POINTER(c_op)],
    "toh_op_fanout": [c_op, c_op, C.POINTER(c_op)],
    "toh_op_arity": [c_op, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "toh_op_release": [c_op],
    "toh_run": [c_op, C.c_int, tp, tp],
    "toh_grad": [c_op, C.c_int, tp, tp, i32p, tp],
    "toh_gradTOp": [c_op, C.c_int, tp, i32p, tp],
    "toh_genNet": [C.c_int, tp, tp, C.c_int, C.c_int, C.POINTER(c_net)],
    "toh_genNet_rand": [C.c_int, capi.i64p, C.c_int, C.c_int, C.c_uint64, C.POINTER(c_net)],
    "toh_buildNet": [c_op, C.c_int, tp, C.POINTER(c_net)],
    "toh_net_seq": [c_net, c_net, C.POINTER(c_net)],
    "toh_net_after_op": [c_op, c_net, C.POINTER(c_net)],
    "toh_net_then_op": [c_net, c_op, C.POINTER(c_net)],
    "toh_net_release": [c_net],
    "toh_net_n_params": [c_net, C.POINTER(C.c_int)],
    "toh_net_params": [c_net, tp],
    "toh_runNetwork": [c_net, c_tensor, tp],
    "toh_netGrad": [c_net, C.c_int, c_tensor, c_tensor, C.c_int, tp],
    "toh_trainNetwork": [c_net, C.c_int, C.c_double, c_tensor, c_tensor, C.POINTER(c_net)],
    "toh_trainer_create": [c_net, C.c_int, C.c_double, c_tensor, c_tensor, C.c_int, C.c_int,
                           C.POINTER(c_trainer)],
    "toh_trainer_flat_size": [c_net, capi.i64p],
    "toh_trainer_create_ext": [c_net, C.c_int, C.c_double, c_tensor, c_tensor, C.c_int, C.c_int,
                               C.c_void_p, C.c_void_p, C.POINTER(c_trainer)],
    "toh_trainer_create_opts": [c_net, C.c_int, C.c_double, c_tensor, c_tensor, C.c_int, C.c_void_p,
                                C.c_void_p, C.POINTER(c_trainer)],
    "toh_trainer_is_fused": [c_trainer, C.POINTER(C.c_int)],
    "toh_trainer_is_graph": [c_trainer, C.POINTER(C.c_int)],
    "toh_trainer_release": [c_trainer],
    "toh_trainer_grad": [c_trainer],
    "toh_trainer_apply": [c_trainer],
    "toh_trainer_step": [c_trainer],
    "toh_trainer_flat": [c_trainer, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), capi.i64p],
    "toh_trainer_net": [c_trainer, C.POINTER(c_net)],
    "toh_trainer_launches_per_step": [c_trainer, capi.i64p],
    "toh_trainer_step_launches": [c_trainer, capi.i64p],
    "toh_trainAll": [c_net, C.c_int, C.c_double, c_tensor, c_tensor, C.c_int64, capi.i64p, C.c_int,
                     C.POINTER(c_net)],
    "toh_trace_begin": [C.c_int, tp],
    "toh_trace_end": [C.c_char_p, C.c_int64, capi.i64p],
    # Recurrent.hs
    "toh_rnn_fullyConnected": [C.c_int, c_tensor, c_tensor, c_tensor, c_tensor, C.POINTER(c_rnn)],
    "toh_rnn_fullyConnected_rand": [C.c_int, C.c_int64, C.c_int64, C.c_uint64, C.POINTER(c_rnn)],
    "toh_rnn_ffLayer": [c_tensor, c_tensor, C.POINTER(c_rnn)],
    "toh_rnn_stateless": [c_net, C.POINTER(c_rnn)],
    "toh_rnn_seq": [c_rnn, c_rnn, C.POINTER(c_rnn)],
    "toh_rnn_then_act": [c_rnn, C.c_int, C.POINTER(c_rnn)],
    "toh_rnn_then_op": [c_rnn, c_op, C.POINTER(c_rnn)],
    "toh_rnn_after_op": [c_op, c_rnn, C.POINTER(c_rnn)],
    "toh_rnn_release": [c_rnn],
    "toh_rnn_counts": [c_rnn, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "toh_rnn_state": [c_rnn, tp],
    "toh_rnn_params": [c_rnn, tp],
    "toh_rnn_run": [c_rnn, c_tensor, tp, C.POINTER(c_rnn)],
    "toh_rnn_netGrad": [c_rnn, C.c_int, C.c_int, tp, tp, tp, tp, tp],
    "toh_rnn_trainNetwork": [c_rnn, C.c_int, C.c_double, C.c_double, C.c_int, tp, tp, C.POINTER(c_rnn)],
    # AutoEncoder.hs
    "toh_ae_encode": [c_net, c_net, c_tensor, tp],
    "toh_ae_decode": [c_net, c_net, c_tensor, tp],
    "toh_ae_encodeDecode": [c_net, c_net, c_tensor, tp],
    "toh_ae_testEncoder": [c_net, c_net, C.c_int, c_tensor, tp],
    "toh_ae_encGrad": [c_net, c_net, C.c_int, c_tensor, tp],
    "toh_ae_trainEncoder": [c_net, c_net, C.c_int, C.c_double, c_tensor, C.POINTER(c_net), C.POINTER(c_net)],
}

ACT = {"actLogistic": 0, "actMapLogistic": 1, "actSoftmax": 2, "actMapTanh": 3}
LOSS = {"squaredError": 0, "crossEntropy": 1}

_hlib = None


def hlib():
    global _hlib
    if _hlib is None:
        capi.lib()  # the kernels first: no fallback
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError("libtensorops_host.so is not built (run `python tensor-ops_amd/build.py`)")
        L = C.CDLL(HOST_LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int32
        L.toh_last_error.argtypes = []
        L.toh_last_error.restype = C.c_char_p
        _hlib = L
    return _hlib


def set_elem_dtype(dtype):
    """Select the host mirror's `ElemT` (np.float32 default, np.float64 = HMat's element type).
    Values the mirror builds itself (konst, genNet_rand, the trainer's flat buffers) use it."""
    import numpy as np
    from . import capi
    dt = np.dtype(dtype)
    code = {np.dtype(np.float32): capi.TO_F32, np.dtype(np.float64): capi.TO_F64}[dt]
    check(capi.lib().to_set_default_dtype(code))


def check(status):
    if status != 0:
        raise capi.TensorOpsError(status, hlib().toh_last_error().decode(errors="replace"))


def _ssa(f, n):
    """Record `f` on symbolic inputs -> (code array, n_instr, consts array, n_consts)."""
    tape = hipt._Tape(n)
    r = f([hipt.Sym(tape, i) for i in range(n)])
    if not isinstance(r, hipt.Sym):
        r = hipt.Sym(tape, tape.const(r))
    if r.v != n + len(tape.code) - 1:
        z = tape.const(0.0)
        tape.code.append((hipt.X_ADD, r.v, z))
    flat = (C.c_int32 * max(3 * len(tape.code), 1))(*[int(x) for ins in tape.code for x in ins])
    cs = (C.c_double * max(len(tape.consts), 1))(*tape.consts)
    return flat, len(tape.code), cs, len(tape.consts)


def _tarr(ts):
    return (c_tensor * max(len(ts), 1))(*[t.h if t is not None else None for t in ts])


class Trace:
    """Log the class-method calls the mirror issues inside the `with` block (host/tensorops/trace.hpp);
    `leaves` (device tensors) name the program's inputs.  `.text` holds the log afterwards."""

    def __init__(self, leaves):
        self.leaves = list(leaves)
        self.text = None

    def __enter__(self):
        check(hlib().toh_trace_begin(len(self.leaves), _tarr(self.leaves)))
        return self

    def __exit__(self, *a):
        n = C.c_int64()
        check(hlib().toh_trace_end(None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value + 1)
        check(hlib().toh_trace_end(buf, n.value + 1, C.byref(n)))
        self.text = buf.value.decode()


class Op:
    """A `TOp ns ms`."""

    def __init__(self, h):
        self.h = h

    def __del__(self):
        try:
            if self.h:
                hlib().toh_op_release(self.h)
        except Exception:
            pass

    @property
    def arity(self):
        a, b = C.c_int(), C.c_int()
        check(hlib().toh_op_arity(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def __rshift__(self, other):  # `>>>`
        return _mk(hlib().toh_op_compose, self.h, other.h)

    def run(self, xs):  # runTOp
        out = (c_tensor * max(self.arity[1], 1))()
        check(hlib().toh_run(self.h, len(xs), _tarr(xs), out))
        return [DT(out[i]) for i in range(self.arity[1])]

    def grad(self, xs, ds, want=None):  # gradTOp'
        n = len(xs)
        out = (c_tensor * max(n, 1))()
        w = (C.c_int32 * max(n, 1))(*[int(bool(x)) for x in want]) if want is not None else None
        check(hlib().toh_grad(self.h, n, _tarr(xs), _tarr(ds), w, out))
        return [DT(out[i]) if out[i] else None for i in range(n)]

    def gradTOp(self, xs, want=None):  # gradTOp
        n = len(xs)
        out = (c_tensor * max(n, 1))()
        w = (C.c_int32 * max(n, 1))(*[int(bool(x)) for x in want]) if want is not None else None
        check(hlib().toh_gradTOp(self.h, n, _tarr(xs), w, out))
        return [DT(out[i]) if out[i] else None for i in range(n)]


def _mk(fn, *args):
    h = c_op()
    check(fn(*args, C.byref(h)))
    return Op(h)


def named(name, iarg=0, darg=0.0):
    return _mk(hlib().toh_op_named, name.encode(), iarg, darg)


# vocabulary, named as in src/TensorOps/TOp.hs
def gmul(lm, lo, ln): return _mk(hlib().toh_op_gmul, lm, lo, ln)
def inner(lm, ln): return gmul(lm, 1, ln)
def outer(lm, ln): return gmul(lm, 0, ln)
def dot(): return named("dot")
def matVec(): return named("matVec")
def vecMat(): return named("vecMat")
def matMat(): return named("matMat")
def add(): return named("add")
def add3(): return named("add3")
def duplicate(): return named("duplicate")
def replicate(n): return named("replicate", n)
def swap(): return named("swap")
def negate(): return named("negate")
def scale(a): return named("scale", 0, float(a))
def sumRows(): return named("sumRows")
def transpOp(): return named("transpOp")
def idOp(n): return named("idOp", n)
def softmax(): return named("softmax")
def squaredError(): return named("squaredError")
def crossEntropy(): return named("crossEntropy")


def map_(f, f_prime=None):
    """`TO.map f` (derivative by forward-mode AD on the host) / `TO.map' f f'`."""
    code, n, cs, nc = _ssa(lambda v: f(v[0]), 1)
    if f_prime is None:
        return _mk(hlib().toh_op_map, n, code, nc, cs)
    code2, n2, cs2, nc2 = _ssa(lambda v: f_prime(v[0]), 1)
    return _mk(hlib().toh_op_map_with, n, code, nc, cs, n2, code2, nc2, cs2)


def zipN(n, f):
    code, ni, cs, nc = _ssa(f, n)
    return _mk(hlib().toh_op_zipN, n, ni, code, nc, cs)


def zipN_with(n, f, grads):
    """`zipN'` (TOp.hs:232-239): f and an explicit gradient, `grads(v)` = the n partial derivatives."""
    code, ni, cs, nc = _ssa(f, n)
    parts = [_ssa(lambda v, i=i: grads(v)[i], n) for i in range(n)]
    n_g = (C.c_int32 * n)(*[p[1] for p in parts])
    nc_g = (C.c_int32 * n)(*[p[3] for p in parts])
    g = (i32p * n)(*[C.cast(p[0], i32p) for p in parts])
    c_g = (f64p * n)(*[C.cast(p[2], f64p) for p in parts])
    op = _mk(hlib().toh_op_zipN_with, n, ni, code, nc, cs, n_g, g, nc_g, c_g)
    op._keep = parts
    return op


def zip_(f): return zipN(2, lambda v: f(v[0], v[1]))
def zip_with(f, g): return zipN_with(2, lambda v: f(v[0], v[1]), lambda v: g(v[0], v[1]))   # zip'
def zip3_with(f, g): return zipN_with(3, lambda v: f(v[0], v[1], v[2]), lambda v: g(v[0], v[1], v[2]))   # zip3'
def zip3(f): return zipN(3, lambda v: f(v[0], v[1], v[2]))


def sumOp(n, shape):
    d, r = dims_arr(shape)
    return _mk(hlib().toh_op_sumOp, n, r, d)


def konst(n, shape, x):
    d, r = dims_arr(shape)
    return _mk(hlib().toh_op_konst, n, r, d, float(x))


def shuffle(idx, n_in):
    a = (C.c_int32 * max(len(idx), 1))(*idx)
    return _mk(hlib().toh_op_shuffle, n_in, len(idx), a)


def drop(n_drop, n): return _mk(hlib().toh_op_drop, n_drop, n)
def take(n_take, n): return _mk(hlib().toh_op_take, n_take, n)
def firstOp(o, n_pass): return _mk(hlib().toh_op_first, o.h, n_pass)
def secondOp(n_skip, o): return _mk(hlib().toh_op_second, n_skip, o.h)
def then_first(a, b): return _mk(hlib().toh_op_then_first, a.h, b.h)   # a *>> b
def par(a, b): return _mk(hlib().toh_op_par, a.h, b.h)                 # a *** b
def fanout(a, b): return _mk(hlib().toh_op_fanout, a.h, b.h)           # a &&& b


class Net:
    """A `Network t i o` (FeedForward.hs:57-61)."""

    def __init__(self, h):
        self.h = h

    def __del__(self):
        try:
            if self.h:
                hlib().toh_net_release(self.h)
        except Exception:
            pass

    @property
    def params(self):
        n = C.c_int()
        check(hlib().toh_net_n_params(self.h, C.byref(n)))
        out = (c_tensor * max(n.value, 1))()
        check(hlib().toh_net_params(self.h, out))
        return [DT(out[i]) for i in range(n.value)]


def buildNet(op, params):
    """`buildNet` (FeedForward.hs:68-73); `liftNet op = buildNet op []` (:110-113)."""
    h = c_net()
    check(hlib().toh_buildNet(op.h, len(params), _tarr(params), C.byref(h)))
    return Net(h)


def liftNet(op): return buildNet(op, [])


def net_seq(a, b):          # a ~*~ b (:82-90)
    h = c_net()
    check(hlib().toh_net_seq(a.h, b.h, C.byref(h)))
    return Net(h)


def net_after(f, n):        # f ~* n (:96-101)
    h = c_net()
    check(hlib().toh_net_after_op(f.h, n.h, C.byref(h)))
    return Net(h)


def net_then(n, f):         # n *~ f (:103-108)
    h = c_net()
    check(hlib().toh_net_then_op(n.h, f.h, C.byref(h)))
    return Net(h)


def nmap(f, n): return net_then(n, map_(f))   # (:115-121)


def networkGradient(net, loss, x, y): return netGrad(net, loss, x, y, want_x=False)[1:]   # (:166-176)


def genNet(weights, hidden_act, out_act):
    """`genNet` with given weights [(W1,b1),...] (device tensors)."""
    ws = _tarr([w for w, _ in weights])
    bs = _tarr([b for _, b in weights])
    h = c_net()
    check(hlib().toh_genNet(len(weights), ws, bs, ACT[hidden_act], ACT[out_act], C.byref(h)))
    return Net(h)


def genNet_rand(sizes, hidden_act, out_act, seed):
    d, r = dims_arr(sizes)
    h = c_net()
    check(hlib().toh_genNet_rand(r, d, ACT[hidden_act], ACT[out_act], seed, C.byref(h)))
    return Net(h)


def runNetwork(net, x):
    h = c_tensor()
    check(hlib().toh_runNetwork(net.h, x.h, C.byref(h)))
    return DT(h)


def netGrad(net, loss, x, y, want_x=True):
    n = len(net.params)
    out = (c_tensor * (n + 1))()
    check(hlib().toh_netGrad(net.h, LOSS[loss], x.h, y.h, int(want_x), out))
    return [DT(out[i]) if out[i] else None for i in range(n + 1)]


def trainNetwork(net, loss, rate, x, y):
    h = c_net()
    check(hlib().toh_trainNetwork(net.h, LOSS[loss], float(rate), x.h, y.h, C.byref(h)))
    return Net(h)


def trainAll(net, loss, rate, X, Y, order=None, n=None, use_memo=True, use_fused=True):
    """`foldl' trainNetwork` over rows of resident batched X, Y in `order` (app/MNIST.hs:390-393)."""
    flags = (1 if use_memo else 0) | (4 if use_fused else 0)
    if order is None:
        cnt, arr = (X.batch if n is None else n), None
    else:
        cnt = len(order)
        arr = (C.c_int64 * max(cnt, 1))(*[int(v) for v in order])
    h = c_net()
    check(hlib().toh_trainAll(net.h, LOSS[loss], float(rate), X.h, Y.h, cnt, arr, flags, C.byref(h)))
    return Net(h)


class Trainer:
    """Replayed batched gradTOp step over fixed (X, Y) batch buffers."""

    def __init__(self, net, loss, rate, x, y, use_memo=True, use_graph=True, ext_params=None,
                 ext_grads=None, use_fused=True, fresh_thunks=False):
        h = c_trainer()
        flags = (1 if use_memo else 0) | (2 if use_graph else 0) | (4 if use_fused else 0) | (8 if fresh_thunks else 0)
        check(hlib().toh_trainer_create_opts(net.h, LOSS[loss], float(rate), x.h, y.h, flags,
                                             ext_params, ext_grads, C.byref(h)))
        self.h = h
        self._keep = (x, y)

    @property
    def fused(self):
        v = C.c_int()
        check(hlib().toh_trainer_is_fused(self.h, C.byref(v)))
        return bool(v.value)

    @property
    def graph(self):
        v = C.c_int()
        check(hlib().toh_trainer_is_graph(self.h, C.byref(v)))
        return bool(v.value)

    @staticmethod
    def flat_size(net):
        v = C.c_int64()
        check(hlib().toh_trainer_flat_size(net.h, C.byref(v)))
        return v.value

    def __del__(self):
        try:
            if self.h:
                hlib().toh_trainer_release(self.h)
        except Exception:
            pass

    def grad(self):
        check(hlib().toh_trainer_grad(self.h))

    def apply(self):
        check(hlib().toh_trainer_apply(self.h))

    def step(self):
        """grad + apply; on the pre-fused path the update runs inside the weight-gradient launches."""
        check(hlib().toh_trainer_step(self.h))

    def flat(self):
        p, g, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(hlib().toh_trainer_flat(self.h, C.byref(p), C.byref(g), C.byref(n)))
        return p.value, g.value, n.value

    @property
    def net(self):
        h = c_net()
        check(hlib().toh_trainer_net(self.h, C.byref(h)))
        return Net(h)

    @property
    def launches_per_step(self):
        """kernel launches of one grad()"""
        v = C.c_int64()
        check(hlib().toh_trainer_launches_per_step(self.h, C.byref(v)))
        return v.value

    @property
    def step_launches(self):
        """kernel launches of one step() (0 before the first step)"""
        v = C.c_int64()
        check(hlib().toh_trainer_step_launches(self.h, C.byref(v)))
        return v.value


# ---- Recurrent.hs ------------------------------------------------------------------------------------
class RNet:
    """A recurrent `Network t i o` (Recurrent.hs:66-72): op, initial state, parameters."""

    def __init__(self, h):
        self.h = h

    def __del__(self):
        try:
            if self.h:
                hlib().toh_rnn_release(self.h)
        except Exception:
            pass

    def _counts(self):
        a, b = C.c_int(), C.c_int()
        check(hlib().toh_rnn_counts(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    @property
    def state(self):
        n = self._counts()[0]
        out = (c_tensor * max(n, 1))()
        check(hlib().toh_rnn_state(self.h, out))
        return [DT(out[i]) for i in range(n)]

    @property
    def params(self):
        n = self._counts()[1]
        out = (c_tensor * max(n, 1))()
        check(hlib().toh_rnn_params(self.h, out))
        return [DT(out[i]) for i in range(n)]

    # n1 ~*~ n2
    def seq(self, other):
        h = c_rnn()
        check(hlib().toh_rnn_seq(self.h, other.h, C.byref(h)))
        return RNet(h)

    # n *~ getAct act
    def then_act(self, act):
        h = c_rnn()
        check(hlib().toh_rnn_then_act(self.h, ACT[act], C.byref(h)))
        return RNet(h)

    def then_op(self, op):
        h = c_rnn()
        check(hlib().toh_rnn_then_op(self.h, op.h, C.byref(h)))
        return RNet(h)


def rnn_fullyConnected(state_act, s, w_state, w, b):
    h = c_rnn()
    check(hlib().toh_rnn_fullyConnected(ACT[state_act], s.h, w_state.h, w.h, b.h, C.byref(h)))
    return RNet(h)


def rnn_fullyConnected_rand(state_act, i, o, seed):
    h = c_rnn()
    check(hlib().toh_rnn_fullyConnected_rand(ACT[state_act], i, o, seed, C.byref(h)))
    return RNet(h)


def rnn_ffLayer(w, b):
    h = c_rnn()
    check(hlib().toh_rnn_ffLayer(w.h, b.h, C.byref(h)))
    return RNet(h)


def rnn_stateless(net):
    h = c_rnn()
    check(hlib().toh_rnn_stateless(net.h, C.byref(h)))
    return RNet(h)


def rnn_genNet(layers, out_layer, out_act):
    """`genNet` (Recurrent.hs:140-164): layers = [(values, act, state_act|None)], out_layer =
    (values, state_act|None); values = (s, W', W, b) or (W, b).  (l *~ f') ~*~ go xs."""
    def mk(vals, s_act):
        return rnn_fullyConnected(s_act, *vals) if s_act is not None else rnn_ffLayer(*vals)
    if not layers:
        vals, s_act = out_layer
        return mk(vals, s_act).then_act(out_act)
    (vals, act, s_act), rest = layers[0], layers[1:]
    return mk(vals, s_act).then_act(act).seq(rnn_genNet(rest, out_layer, out_act))


def rnn_runNetwork(net, x):
    y, nxt = c_tensor(), c_rnn()
    check(hlib().toh_rnn_run(net.h, x.h, C.byref(y), C.byref(nxt)))
    return DT(y), RNet(nxt)


def rnn_netGrad(net, loss, xs, ys, want_inputs=True):
    """(gI, gS, gP); gI in the reference's order (reversed time)."""
    ns, np_ = net._counts()
    n = len(xs)
    gi = (c_tensor * max(n, 1))()
    gs = (c_tensor * max(ns, 1))()
    gp = (c_tensor * max(np_, 1))()
    check(hlib().toh_rnn_netGrad(net.h, LOSS[loss], n, _tarr(xs), _tarr(ys), gi if want_inputs else None, gs, gp))
    return ([DT(gi[i]) for i in range(n)] if want_inputs else None,
            [DT(gs[i]) for i in range(ns)], [DT(gp[i]) for i in range(np_)])


def rnn_trainNetwork(net, loss, rate_state, rate_params, xs, ys):
    h = c_rnn()
    check(hlib().toh_rnn_trainNetwork(net.h, LOSS[loss], float(rate_state), float(rate_params), len(xs),
                                      _tarr(xs), _tarr(ys), C.byref(h)))
    return RNet(h)


# ---- AutoEncoder.hs ------------------------------------------------------------------------------------
class Encoder:
    """`Encoder t i o` (AutoEncoder.hs:37-40)."""

    def __init__(self, enc, dec):
        self.enc, self.dec = enc, dec

    def _call(self, fn, *args):
        h = c_tensor()
        check(fn(self.enc.h, self.dec.h, *args, C.byref(h)))
        return DT(h)

    def encode(self, x): return self._call(hlib().toh_ae_encode, x.h)
    def decode(self, y): return self._call(hlib().toh_ae_decode, y.h)
    def encodeDecode(self, x): return self._call(hlib().toh_ae_encodeDecode, x.h)
    def testEncoder(self, loss, x): return self._call(hlib().toh_ae_testEncoder, LOSS[loss], x.h)

    def encGrad(self, loss, x):
        ne, nd = len(self.enc.params), len(self.dec.params)
        out = (c_tensor * (ne + nd))()
        check(hlib().toh_ae_encGrad(self.enc.h, self.dec.h, LOSS[loss], x.h, out))
        g = [DT(out[i]) for i in range(ne + nd)]
        return g[:ne], g[ne:]

    def trainEncoder(self, loss, rate, x):
        a, b = c_net(), c_net()
        check(hlib().toh_ae_trainEncoder(self.enc.h, self.dec.h, LOSS[loss], float(rate), x.h,
                                         C.byref(a), C.byref(b)))
        return Encoder(Net(a), Net(b))
