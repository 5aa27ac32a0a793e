"""Committed known-answer vectors (tests/golden/gmul_einsum.npz, recipe: make_golden.py,
independent `numpy.einsum` formulation): the oracle on CPU, the HIP backend on the GPU."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gmul_einsum.npz"))
N = int(G["n_cases"])


@pytest.mark.parametrize("k", range(N))
def test_oracle_matches_golden(k):
    from oracle import nested
    lm, lo, ln = (int(v) for v in G["l%d" % k])
    a, b, c = G["a%d" % k], G["b%d" % k], G["c%d" % k]
    assert np.array_equal(nested.gmul(lm, lo, ln, a, b), c)
    assert np.array_equal(nested.gmul_literal(lm, lo, ln, a, b), c)
    assert np.array_equal(nested.gmul_flat(lm, lo, ln, a, b), c)
    assert np.array_equal(nested.transpose(a), G["t%d" % k])


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(N))
def test_hip_matches_golden_bit_exact(k):
    from tensor_ops_amd.hipt import HipT
    T = HipT(0)
    lm, lo, ln = (int(v) for v in G["l%d" % k])
    a, b, c = G["a%d" % k], G["b%d" % k], G["c%d" % k]
    got = T.gmul(lm, lo, ln, T.put(a), T.put(b)).numpy()
    assert got.shape == c.shape and np.array_equal(got, c.astype(np.float32))
    assert np.array_equal(T.transp(T.put(a)).numpy(), G["t%d" % k].astype(np.float32))
