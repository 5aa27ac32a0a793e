"""CPU-side checks of the call-trace yardstick (tests/call_trace.py): the tracing backend changes no value, the
mirror's log format parses, and the canonical form ignores order, repetition and the batching extension."""
import numpy as np

import call_trace as CT
from oracle import neuralnet as NN

RNG = np.random.default_rng(0x7e500008)
OACT = {"actMapLogistic": lambda: NN.actMap(NN.logistic), "actSoftmax": NN.actSoftmax}


def ff_weights(sizes):
    return [(0.5 * RNG.standard_normal((o, i)), 0.5 * RNG.standard_normal(o)) for i, o in zip(sizes[:-1], sizes[1:])]


def one_hot(o):
    y = np.zeros(o)
    y[RNG.integers(0, o)] = 1.0
    return y


def test_tracing_backend_is_the_oracle():
    """the tracing wrapper changes no value (CPU-only sanity of the test's own yardstick)"""
    from oracle.tensor import OTensor
    sizes = [5, 4, 3]
    ws = ff_weights(sizes)
    x, y = RNG.uniform(0, 1, size=5), one_hot(3)
    nets = [NN.genNet(ws, OACT["actMapLogistic"], OACT["actSoftmax"]) for _ in range(2)]
    Tr = CT.TracingTensor()
    Tr.leaves([])
    a = NN.netGrad(OTensor(np.float64), NN.crossEntropy(), x, y, nets[0])
    b = NN.netGrad(Tr, NN.crossEntropy(), x, y, nets[1])
    for p, q in zip(a, b):
        assert np.array_equal(p, q)


def test_canonical_form_ignores_order_repetition_and_batching():
    # z = gmul(W, x); h = liftT f z -- once in program order, once with the gmul repeated (a recomputed forward pass,
    # Types.hs:155), the batched spelling of the contraction, a batch_sum and a `sumT [x]` in between
    a = "\n".join(["gmul\t1,1,0\t1,0\t3\t4|0", "liftT\t1,5.0e-01,6.0e-01\t3\t4\t4|0"])
    b = "\n".join(["gmul_batch_sum\t1,1,0\t1,0\t7\t4|0", "batch_sum\t\t7\t8\t4|0", "sumT\t1\t8\t8\t4|0",
                   "gmul\t1,1,0\t1,0\t9\t4|8", "liftT\t1,5.0e-01,6.0e-01\t8\t5\t4|8"])
    ca, cb = (CT.canonical(CT.parse_mirror_log(t), 2) for t in (a, b))
    assert set(ca) == set(cb) and len(ca) == 2
    # ... and tells different operands apart
    c = a.replace("1,0\t3", "0,1\t3")
    assert set(CT.canonical(CT.parse_mirror_log(c), 2)) != set(ca)


def test_demand_restriction_drops_what_nobody_asked_for():
    Tr = CT.TracingTensor()
    x, w = RNG.standard_normal(3), RNG.standard_normal((2, 3))
    Tr.leaves([x, w])
    z = Tr.gmul(1, 1, 0, w, x)
    Tr.scaleT(2.0, z)                       # computed by a strict host, never demanded
    keep = Tr.liftT(lambda v: v[0] * v[0], [z])
    full = CT.canonical(Tr.recs, 2)
    need = CT.canonical(Tr.recs, 2, roots=[Tr.id_of(keep)])
    assert len(full) == 3 and len(need) == 2 and set(need) < set(full)
