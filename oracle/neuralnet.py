"""Oracle (test infrastructure) -- `TensorOps.Learn.NeuralNet{,.FeedForward}`.

Follows src/TensorOps/Learn/NeuralNet.hs:15-77 and
src/TensorOps/Learn/NeuralNet/FeedForward.hs:57-235.
"""
import numpy as np

from . import ad
from . import top as TO


# ---- NeuralNet.hs ---------------------------------------------------------------
def logistic(x):
    """NeuralNet.hs:42-44."""
    return 1 / (1 + ad.exp(-x))


def logistic_prime(x):
    """NeuralNet.hs:46-50."""
    lx = logistic(x)
    return lx * (1 - lx)


def actMap(f):
    """NeuralNet.hs:21-25: derivative by forward-mode AD (`TO.map f`)."""
    return TO.map_(f)


def actMap_(f, f_prime):
    """NeuralNet.hs:27-32 (`actMap'`)."""
    return TO.map_(f, f_prime)


def actLogistic():
    """NeuralNet.hs:38-40."""
    return actMap_(logistic, logistic_prime)


def softmax():
    """NeuralNet.hs:52-59:
    map exp >>> duplicate >>> firstOp (sumRows >>> map recip) >>> outer LZ (LS LZ)."""
    # `>>>` is infixr 1 (Control.Category): a >>> (b >>> (c >>> d)).  Association does
    # not change values, but it changes how often `f1 xs` is recomputed (Types.hs:155).
    return (TO.map_(ad.exp)
            >> (TO.duplicate()
                >> (TO.first(TO.sumRows() >> TO.map_(ad.recip), 1)
                    >> TO.outer(0, 1))))


def actSoftmax():
    """NeuralNet.hs:34-36."""
    return softmax()


def squaredError():
    """NeuralNet.hs:61-68: negate *>> add >>> duplicate >>> dot."""
    # `*>>` is infixr 0, `>>>` infixr 1: negate *>> (add >>> (duplicate >>> dot))
    return TO.then_first(TO.negate(), TO.add() >> (TO.duplicate() >> TO.dot()))


def crossEntropy():
    """NeuralNet.hs:71-77: map log *>> dot >>> negate.  Second input is the target."""
    # map log *>> (dot >>> negate)
    return TO.then_first(TO.map_(ad.log), TO.dot() >> TO.negate())


# ---- FeedForward.hs --------------------------------------------------------------
class Network:
    """`Network t i o` (FeedForward.hs:57-61): op : ([i] : ps) -> [[o]], params."""

    def __init__(self, op, params):
        self.op = op
        self.params = list(params)


def seq_net(n1, n2):
    """`~*~` (FeedForward.hs:82-90): N (o1 *>> o2) (p1 ++ p2)."""
    return Network(TO.then_first(n1.op, n2.op), n1.params + n2.params)


def net_then(n, f):
    """`*~` (FeedForward.hs:103-108): N (o >>> f) p."""
    return Network(n.op >> f, n.params)


def then_net(f, n):
    """`~*` (FeedForward.hs:96-101): N (f *>> o) p."""
    return Network(TO.then_first(f, n.op), n.params)


def liftNet(op):
    """`liftNet` (FeedForward.hs:110-113): a parameterless network."""
    return Network(op, [])


def nmap(f, n):
    """`nmap` (FeedForward.hs:115-121): n *~ TO.map f."""
    return net_then(n, TO.map_(f))


def ffLayer_op():
    """`ffLayer'` (FeedForward.hs:209-213):
    firstOp (swap >>> matVec) >>> add   on  [x, W, b]."""
    return TO.first(TO.swap() >> TO.matVec(), 1) >> TO.add()


def ffLayer(w, b):
    """`ffLayer` (FeedForward.hs:201-214) with the weights given (the reference
    draws them from `normalDistr 0 0.5`; RNG streams are not reproducible, so
    initial weights are INPUTS to every parity test)."""
    return Network(ffLayer_op(), [w, b])


def genNet(weights, hidden_act, out_act):
    """`genNet` (FeedForward.hs:216-235):
    go []          = ffLayer *~ f
    go ((x,f'):xs) = (ffLayer *~ f') ~*~ go xs
    `weights` = [(W1,b1), ..., (Wk,bk)], k-1 hidden layers."""
    (w, b), rest = weights[0], weights[1:]
    if not rest:
        return net_then(ffLayer(w, b), out_act())
    return seq_net(net_then(ffLayer(w, b), hidden_act()), genNet(rest, hidden_act, out_act))


def runNetwork(T, net, x):
    """FeedForward.hs:123-129."""
    return TO.runTOp(net.op, T, [x] + net.params)[0]


def netGrad(T, loss, x, y, net):
    """`netGrad` (FeedForward.hs:178-199): gradTOp (o *>> loss) (x :< p >: y),
    keep the cotangents of x and the params (drop y's)."""
    op = TO.then_first(net.op, loss)
    inp = [x] + net.params + [y]
    g = TO.gradTOp(op, T, inp)
    return g[:1 + len(net.params)]


def networkGradient(T, loss, x, y, net):
    """FeedForward.hs:166-176: parameter gradients only (`tail'`)."""
    return netGrad(T, loss, x, y, net)[1:]


def trainNetwork(T, loss, r, x, y, net):
    """`trainNetwork` (FeedForward.hs:131-148): p' = zip (\\o g -> o - r*g) p grads."""
    grads = netGrad(T, loss, x, y, net)[1:]
    r = T.dtype.type(r)
    new = [T.liftT(lambda og: og[0] - r * og[1], [p, g]) for p, g in zip(net.params, grads)]
    return Network(net.op, new)


def induceNetwork(T, loss, r, y, net, x):
    """`induceNetwork` (FeedForward.hs:150-164): gradient step on the INPUT."""
    gx = netGrad(T, loss, x, y, net)[0]
    r = T.dtype.type(r)
    return T.liftT(lambda og: og[0] - r * og[1], [x, gx])


# ---- batched gradTOp (new capability; definition = SURVEY.md section 8(d)) -------------
def batched_param_grads(T, loss, xs, ys, net):
    """G = sum_b networkGradient(x_b, y_b) at FIXED params, summed in float64."""
    acc = None
    for x, y in zip(xs, ys):
        g = networkGradient(T, loss, x, y, net)
        g = [np.asarray(a, dtype=np.float64) for a in g]
        acc = g if acc is None else [a + b for a, b in zip(acc, g)]
    return acc


def batched_losses(T, loss, xs, ys, net):
    op = TO.then_first(net.op, loss)
    return np.array([float(TO.runTOp(op, T, [x] + net.params + [y])[0]) for x, y in zip(xs, ys)])
