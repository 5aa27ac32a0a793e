"""`BTensor v b` restated over a BLAS dictionary (oracle/btensor.py) and instantiated with the reference's own instance,
HMat (src/TensorOps/BLAS/HMat.hs:103-231): its dispatcher (`gmulB` / `gmulBLAS` / `dispatchBLAS` / `naiveGMul`,
src/TensorOps/Backend/BTensor.hs:141-175,592-716) equals the authoritative definition `Nested.gmul'`
(src/Data/Nested.hs:451-473) on every (#ms, #os, #ns) class, and the other class methods (`liftBTensor` :345-369,
`sumBTensor` :754-773, `transpBTensor` :740-752, matrix `+` through `gemm ... eye` :110-113) equal the nested-vector
backend's.  The same `BTensorOps` runs over `to_blas_*` handles in tests/test_gpu_btensor_route.py."""
import numpy as np
import pytest

from oracle import nested, neuralnet as NN, top as TO
from oracle.btensor import BTensorOps, BTensorT, Counting, HMatB
from oracle.tensor import OTensor

# (ms, os, ns): every branch of gmulB / gmulBLAS -- the eight dispatchBLAS cases, scaleB of a matrix, the mapped
# gemv / gemm over trailing matrices (config 5's shape class), trace(gemm), and each way into naiveGMul
CLASSES = [
    ((), (), ()), ((), (), (3,)), ((), (4,), ()), ((), (4,), (3,)), ((5,), (), ()), ((5,), (), (3,)), ((5,), (4,), ()), ((5,), (4,), (3,)),
    ((2, 3), (), ()), ((2, 3), (), (4,)), ((2, 2, 3), (), ()), ((2, 2, 3), (), (2,)),
    ((2, 3), (4,), ()), ((2, 3), (4,), (5,)), ((2, 3, 2), (4,), (5,)), ((2, 2, 2, 3), (2,), ()),
    ((), (3, 4), ()), ((5,), (3, 4), ()), ((2, 5), (3, 4), ()), ((2,), (3, 4), (2,)),
    ((2,), (2, 3, 2), ()), ((), (2, 2, 2), (3,)),
    ((2,), (3,), (2, 2)), ((), (), (2, 3)), ((2, 2), (), (2, 3)), ((3,), (2,), (2, 2, 2)),
]
ROUTE = {"dispatch": "axpy dot ger gemv gemm".split()}


def ints(rng, shape):
    return rng.integers(-3, 4, size=shape).astype(np.float64)


@pytest.mark.parametrize("ms,os_,ns", CLASSES)
def test_dispatch_equals_the_nested_definition(ms, os_, ns):
    rng = np.random.default_rng(len(ms) * 100 + len(os_) * 10 + len(ns) + 7)
    a, b = ints(rng, ms + os_), ints(rng, tuple(reversed(os_)) + ns)
    blas = Counting(HMatB())
    ops = BTensorOps(blas)
    got = ops.to_array(ops.gmul(len(ms), len(os_), len(ns), ops.from_array(a), ops.from_array(b)))
    want = nested.gmul(len(ms), len(os_), len(ns), a, b)
    assert got.shape == want.shape == ms + ns
    assert np.array_equal(got, want), (ms, os_, ns, blas.calls)


def test_which_blas_calls_each_class_makes():
    """the dispatch table itself (BTensor.hs:149-174, 609-616, 660-713): one call where BLAS can take the contraction,
    one per trailing matrix where it is mapped, scalar traversals where it is 'naive'"""
    rng = np.random.default_rng(3)

    def calls(ms, os_, ns):
        blas = Counting(HMatB())
        ops = BTensorOps(blas)
        ops.gmul(len(ms), len(os_), len(ns), ops.from_array(ints(rng, ms + os_)), ops.from_array(ints(rng, tuple(reversed(os_)) + ns)))
        return {k: v for k, v in blas.calls.items() if k in ("axpy", "dot", "ger", "gemv", "gemm", "scaleB", "traceB", "transpB", "iElemsB")}
    assert calls((), (), (3,)) == {"axpy": 1}
    assert calls((), (4,), ()) == {"dot": 1}
    assert calls((), (4,), (3,)) == {"gemv": 1, "transpB": 1}
    assert calls((5,), (), (3,)) == {"ger": 1}
    assert calls((5,), (4,), ()) == {"gemv": 1}
    assert calls((5,), (4,), (3,)) == {"gemm": 1}
    assert calls((2, 3), (), ()) == {"scaleB": 1}
    assert calls((6, 3), (4,), (5,)) == {"gemm": 6}              # config 5's class: one GEMM per leading index (:703-710)
    assert calls((2, 3, 2), (4,), ()) == {"gemv": 6}
    assert calls((5,), (3, 4), ()) == {"gemm": 5, "traceB": 5}   # (:611-613)
    naive = calls((2,), (3,), (2, 2))                            # |ns| = 2: naiveGMul (:616) -- no contraction call at all:
    assert not {"gemv", "dot", "ger"} & set(naive) and naive["scaleB"] == 6   # 2 rows x 3 scaled slices of B, and the six
    assert naive["gemm"] == 6                                    # `+` of 2x2 matrices, each a GEMM with `eye` (:113)


def test_matrix_plus_is_gemm_with_eye_and_vector_plus_is_axpy():
    rng = np.random.default_rng(4)
    blas = Counting(HMatB())
    ops = BTensorOps(blas)
    A, B = ints(rng, (3, 4)), ints(rng, (3, 4))
    assert np.array_equal(ops.to_array(ops.add(ops.from_array(A), ops.from_array(B))), A + B)
    assert blas.calls.get("gemm") == 1 and blas.calls.get("eye") == 1          # (:113)
    x, y = ints(rng, (5,)), ints(rng, (5,))
    assert np.array_equal(ops.to_array(ops.sub(ops.from_array(x), ops.from_array(y))), x - y)
    assert blas.calls.get("axpy") == 1                                           # (:116)
    T3, U3 = ints(rng, (2, 3, 4)), ints(rng, (2, 3, 4))
    assert np.array_equal(ops.to_array(ops.sumT([ops.from_array(T3), ops.from_array(U3), ops.from_array(T3)])), 2 * T3 + U3)
    assert blas.calls["gemm"] == 1 + 2 * 2                                       # two additions of two nested matrices each


def test_the_other_class_methods_equal_the_nested_backend():
    rng = np.random.default_rng(5)
    O, ops = OTensor(np.float64), BTensorOps(HMatB())
    for shape in [(), (5,), (3, 4), (2, 3, 4), (2, 2, 3, 2)]:
        x, y = ints(rng, shape), ints(rng, shape)
        X, Y = ops.from_array(x), ops.from_array(y)
        f = lambda v: v[0] * v[1] - 2 * v[0]
        assert np.array_equal(ops.to_array(ops.liftT(f, [X, Y])), O.liftT(f, [x, y])), shape
        assert np.array_equal(ops.to_array(ops.scaleT(3.0, X)), O.scaleT(3.0, x))
        assert np.array_equal(ops.to_array(ops.transp(X)), O.transp(x))
        if shape:
            assert np.array_equal(ops.to_array(ops.sumRows(X)), O.sumRows(x)), shape
            T = BTensorT(HMatB())
            g = lambda r: T.scaleT(2.0, r)
            assert np.array_equal(T.get(T.mapRows(1, g, X)), 2 * x)
            assert np.array_equal(T.get(T.ixRows(1, lambda i, r: T.scaleT(float(i[0]), r), X)), x * np.arange(shape[0]).reshape((-1,) + (1,) * (len(shape) - 1)))
        for i in np.ndindex(*shape):
            assert ops.index(i, X) == x[i]
            break
    v = ints(rng, (3,))
    for rank in (1, 2, 3):
        assert np.array_equal(ops.to_array(ops.diag(rank, ops.from_array(v))), O.diag(rank, v))
    assert np.array_equal(ops.to_array(ops.getDiag(ops.from_array(ints(rng, (3, 3))))).shape, (3,))
    d3 = O.diag(3, v)
    assert np.array_equal(ops.to_array(ops.getDiag(ops.from_array(d3))), v)


def test_c1_step_through_btensor_over_hmat_equals_the_nested_backend():
    """BASELINE config 1 (2 -> 16 -> 1, logistic, squaredError): runTOp + gradTOp + the SGD update with every class method
    going through BTensor's dispatcher over HMat -- the reference's own CPU route -- against the nested-vector oracle"""
    rng = np.random.default_rng(0x7e500001)
    O, T = OTensor(np.float64), BTensorT(Counting(HMatB()))
    ws = [(0.5 * rng.standard_normal((16, 2)), 0.5 * rng.standard_normal(16)), (0.5 * rng.standard_normal((1, 16)), 0.5 * rng.standard_normal(1))]
    net_o = NN.genNet(ws, NN.actLogistic, NN.actLogistic)
    net_t = NN.Network(net_o.op, [T.put(p) for p in net_o.params])
    x, y = rng.uniform(-1, 1, 2), np.array([1.0])
    want = NN.trainNetwork(O, NN.squaredError(), 1.0, x, y, net_o)
    got = NN.trainNetwork(T, NN.squaredError(), 1.0, T.put(x), T.put(y), net_t)
    for a, b in zip(got.params, want.params):
        assert np.allclose(T.get(a), b, rtol=1e-12, atol=1e-14)
    c = T.ops.b.calls
    assert c["gemv"] >= 4 and c["ger"] == 2, c                                   # matVec forward (recomputed), ger for dW
