"""Oracle (test infrastructure; parity unpinned -- see oracle/__init__.py) -- restatement of
src/TensorOps/Learn/NeuralNet/Recurrent.hs: stateful networks, their composition, the
BPTT unrolling (`unroll`/`rollup`) and `trainNetwork'`.

Type-level evidence (`Sing ss`, `Sing ps`, `SingI i`) becomes explicit shapes.
Products are Python lists; the order conventions of the reference are kept exactly
(inputs are fed REVERSED in time, outputs come back reversed, Recurrent.hs:289-293).
"""
import numpy as np

from . import neuralnet as NN
from . import top as TO


class Network:
    """`Network t i o` (Recurrent.hs:66-72): op : ([i] : ss ++ ps) -> ([o] : ss)."""

    def __init__(self, op, state, params, i_shape):
        self.op = op
        self.state = list(state)
        self.params = list(params)
        self.i_shape = tuple(i_shape)

    @property
    def n_s(self):
        return len(self.state)

    @property
    def n_p(self):
        return len(self.params)


def fullyConnected_op(act):
    """`fc` (Recurrent.hs:108-118) on [x, s, W', W, b] -> [z, act z] with z = W x + W' s + b:
    the OUTPUT is the pre-activation sum, the new STATE its activation."""
    inner = (TO.first(TO.swap() >> TO.matVec(), 2)      # firstOp @'[ '[o,i], '[o] ] (swap >>> matVec)
             >> TO.first(TO.swap(), 1))                 # firstOp @'[ '[o] ] swap
    return (TO.secondOp(1, inner)
            >> TO.first(TO.swap() >> TO.matVec(), 2)    # firstOp @'[ '[o], '[o] ]
            >> TO.add3()
            >> TO.duplicate()
            >> TO.secondOp(1, act()))


def fullyConnected(act, s, w_state, w, b):
    """`fullyConnected` (Recurrent.hs:91-119) with given values (reference: normalDistr 0 0.5)."""
    return Network(fullyConnected_op(act), [s], [w_state, w, b], (np.shape(w)[1],))


def stateless(ffnet, i_shape):
    """`stateless` (Recurrent.hs:126-131): FF.N sP o p -> N SNil sP o Ø p."""
    return Network(ffnet.op, [], ffnet.params, i_shape)


def ffLayer(w, b):
    """Recurrent.hs:133-138."""
    return stateless(NN.ffLayer(w, b), (np.shape(w)[1],))


def seq_net(n1, n2):
    """`~*~` (Recurrent.hs:170-222): states ss2 ++ ss1, params ps1 ++ ps2."""
    s1, s2, p1, p2 = n1.n_s, n2.n_s, n1.n_p, n2.n_p
    op = (TO.secondOp(1, TO.first(TO.swap_n(s2, s1 + p1), p2))
          >> TO.first(n1.op, s2 + p2)
          >> TO.secondOp(1, TO.swap_n(s1, s2 + p2))
          >> TO.first(n2.op, s1))
    return Network(op, n2.state + n1.state, n1.params + n2.params, n1.i_shape)


def net_then(n, f):
    """`*~` (Recurrent.hs:247-252): N (o >>> firstOp f)."""
    return Network(n.op >> TO.first(f, n.n_s), n.state, n.params, n.i_shape)


def then_net(f, n, i_shape=None):
    """`~*` (Recurrent.hs:240-245): N (f *>> o)."""
    return Network(TO.then_first(f, n.op), n.state, n.params, i_shape or n.i_shape)


def genNet(layers, out_layer, out_act):
    """`genNet` (Recurrent.hs:140-164).  `layers` = [(layer_values, act, state_act|None), ...]
    hidden layers; `out_layer` = (layer_values, state_act|None); layer_values =
    (s, W', W, b) for a fullyConnected layer, (W, b) for a stateless ffLayer.
    go []            = final *~ f
    go ((f',fS'):xs) = (l *~ f') ~*~ go xs          (infixl 5 *~, infixr 4 ~*~)"""
    def mk(vals, s_act):
        return fullyConnected(s_act, *vals) if s_act is not None else ffLayer(*vals)
    if not layers:
        vals, s_act = out_layer
        return net_then(mk(vals, s_act), out_act())
    (vals, act, s_act), rest = layers[0], layers[1:]
    return seq_net(net_then(mk(vals, s_act), act()), genNet(rest, out_layer, out_act))


def runNetwork(T, net, x):
    """Recurrent.hs:224-232: (y, network with the new state)."""
    out = TO.runTOp(net.op, T, [x] + net.state + net.params)
    return out[0], Network(net.op, out[1:], net.params, net.i_shape)


def _shapes(net):
    return [np.shape(s) for s in net.state], [np.shape(p) for p in net.params]


def unroll(net_op, i_shape, s_shapes, p_shapes, n):
    """`unroll` (Recurrent.hs:392-431): Replicate n [i] ++ ss ++ ps -> ss ++ Replicate n [o].
    The LAST of the n inputs is consumed first; its output lands LAST."""
    ls, lp = len(s_shapes), len(p_shapes)
    if n == 0:
        return TO.take(ls, list(s_shapes) + list(p_shapes))
    m = n - 1
    step = (TO.fanout(net_op, TO.drop(1 + ls, [i_shape] + list(s_shapes) + list(p_shapes)),
                      [i_shape] + list(s_shapes) + list(p_shapes))
            >> TO.swap_n(1, ls + lp))
    return TO.secondOp(m, step) >> TO.first(unroll(net_op, i_shape, s_shapes, p_shapes, m), 1)


def rollup(loss, n):
    """`rollup` (Recurrent.hs:434-463): Replicate n [o] ++ Replicate n [o] -> [[]],
    total = rollup(m) + loss(last output, first target)   (`TO.add` = sumT [rest, this])."""
    if n == 0:
        return TO.konst(1, (), 0)
    if n == 1:
        return loss
    m = n - 1
    return (TO.secondOp(m, TO.first(loss, m) >> TO.swap_n(1, m))
            >> TO.first(rollup(loss, m), 1)
            >> TO.add())


def netGrad(T, loss, xs, ys, net):
    """`netGrad` (Recurrent.hs:265-324).  Returns (gI, gS, gP); gI is in the order of the
    REVERSED inputs, exactly like the reference (`prodToVec' I n grI`, :283)."""
    n = len(xs)
    s_sh, p_sh = _shapes(net)
    ls, lp = len(s_sh), len(p_sh)
    o_shapes = None  # `drop lS` needs the shapes of what it drops (states) and keeps (outputs)
    unrolled_raw = unroll(net.op, net.i_shape, s_sh, p_sh, n)
    # shapes of the outputs are only needed for cotangent zeros of DROPPED entries = the states
    unrolled = unrolled_raw >> TO.drop(ls, list(s_sh) + [None] * n)
    o_prime = TO.first(unrolled, n) >> rollup(loss, n)
    inp = list(reversed(list(xs))) + net.state + net.params + list(ys)
    g = TO.gradTOp(o_prime, T, inp)[:n + ls + lp]
    del o_shapes
    return g[:n], g[n:n + ls], g[n + ls:]


def trainNetwork(T, loss, r_s, r_p, xs, ys, net):
    """`trainNetwork'` (Recurrent.hs:326-356): separate rates for the initial state and params."""
    _, g_s, g_p = netGrad(T, loss, xs, ys, net)
    r_s, r_p = T.dtype.type(r_s), T.dtype.type(r_p)
    s2 = [T.liftT(lambda og: og[0] - r_s * og[1], [a, g]) for a, g in zip(net.state, g_s)]
    p2 = [T.liftT(lambda og: og[0] - r_p * og[1], [a, g]) for a, g in zip(net.params, g_p)]
    return Network(net.op, s2, p2, net.i_shape)


def total_loss(T, loss, xs, ys, net):
    """forward value of the BPTT objective (sum over time of loss(o_t, y_t))"""
    cur, tot = net, 0.0
    for x, y in zip(xs, ys):
        o, cur = runNetwork(T, cur, x)
        tot += float(np.asarray(TO.runTOp(loss, T, [o, y])[0]))
    return tot


def batched_grads(T, loss, xs_b, ys_b, net):
    """new capability (SURVEY.md 8(d) rule): B independent sequences at FIXED state/params,
    gradients summed in float64.  xs_b[t] has shape (B, i)."""
    B = np.shape(xs_b[0])[0]
    acc_s = acc_p = None
    for b in range(B):
        _, gs, gp = netGrad(T, loss, [x[b] for x in xs_b], [y[b] for y in ys_b], net)
        gs = [np.asarray(a, np.float64) for a in gs]
        gp = [np.asarray(a, np.float64) for a in gp]
        acc_s = gs if acc_s is None else [a + c for a, c in zip(acc_s, gs)]
        acc_p = gp if acc_p is None else [a + c for a, c in zip(acc_p, gp)]
    return acc_s, acc_p
