"""Oracle (test infrastructure; parity unpinned) -- restatement of
src/TensorOps/Learn/NeuralNet/AutoEncoder.hs over the FeedForward networks of oracle/neuralnet.py."""
from . import neuralnet as NN
from . import top as TO


class Encoder:
    """`Encoder t i o` (AutoEncoder.hs:37-40)."""

    def __init__(self, enc, dec):
        self.enc = enc
        self.dec = dec


def encode(T, e, x):            # :42-48
    return NN.runNetwork(T, e.enc, x)


def decode(T, e, y):            # :50-56
    return NN.runNetwork(T, e.dec, y)


def encoderNet(e):              # :83-87   e >>> d  =  e ~*~ d
    return NN.seq_net(e.enc, e.dec)


def encodeDecode(T, e, x):      # :58-63
    return NN.runNetwork(T, encoderNet(e), x)


def _objective(loss, op_e, n_pe, op_d, n_pd):
    """AutoEncoder.hs:130-137 (and :73-79): duplicate x, run enc then dec on one copy,
    swap so the reconstruction is the prediction and x the target, apply the loss."""
    return (TO.first(TO.duplicate(), n_pe + n_pd)
            >> TO.secondOp(1, TO.first(op_e, n_pd) >> op_d)
            >> TO.swap()
            >> loss)


def testEncoder(T, loss, e, x):  # :65-81 (through encoderNet's composed op)
    net = encoderNet(e)
    op = (TO.first(TO.duplicate(), len(net.params)) >> TO.secondOp(1, net.op) >> TO.swap() >> loss)
    return TO.runTOp(op, T, [x] + net.params)[0]


def encGrad(T, loss, x, e):      # :112-142
    op = _objective(loss, e.enc.op, len(e.enc.params), e.dec.op, len(e.dec.params))
    g = TO.gradTOp(op, T, [x] + e.enc.params + e.dec.params)[1:]
    return g[:len(e.enc.params)], g[len(e.enc.params):]


def trainEncoder(T, loss, r, x, e):  # :89-110
    g_e, g_d = encGrad(T, loss, x, e)
    r = T.dtype.type(r)
    step = lambda p, g: T.liftT(lambda og: og[0] - r * og[1], [p, g])  # noqa: E731
    return Encoder(NN.Network(e.enc.op, [step(p, g) for p, g in zip(e.enc.params, g_e)]),
                   NN.Network(e.dec.op, [step(p, g) for p, g in zip(e.dec.params, g_d)]))
