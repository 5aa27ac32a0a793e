"""Generates tests/golden/gmul_einsum.npz: known-answer vectors for `gmul`/`transp` computed by
an INDEPENDENT formulation -- explicit `numpy.einsum` subscripts written from the definition
C[m..,n..] = sum_o A[m..,o1..oq] * B[oq..o1,n..] (src/Data/Nested.hs:465-472) -- not by the
oracle.  Small-integer data, so every expected value is exact in fp32 and fp64.
Run: python tests/golden/make_golden.py   (the .npz is committed; this script is its recipe)."""
import os
import string

import numpy as np

CASES = [((3,), (4,), (2,)), ((2, 3), (4,), (5,)), ((2,), (3, 4), (2,)), ((2, 3), (2, 3), ()),
         ((2,), (2, 3, 2), (3, 2)), ((2, 3), (), (4,)), ((), (5,), ()), ((), (), (3,)), ((4,), (), ()),
         ((), (), ()), ((2, 2, 2), (3,), (2, 2)), ((5,), (4, 3, 2), ()), ((), (6, 2), (7,)),
         ((3, 2), (2, 2), (2, 3))]


def subscripts(lm, lo, ln):
    L = string.ascii_lowercase
    m, o, n = L[:lm], L[lm:lm + lo], L[lm + lo:lm + lo + ln]
    return "%s%s,%s%s->%s%s" % (m, o, o[::-1], n, m, n)


def main():
    rng = np.random.default_rng(0x7E500001)
    out = {}
    for k, (ms, os_, ns) in enumerate(CASES):
        a = rng.integers(-4, 5, size=ms + os_).astype(np.float64)
        b = rng.integers(-4, 5, size=tuple(reversed(os_)) + ns).astype(np.float64)
        c = np.einsum(subscripts(len(ms), len(os_), len(ns)), a, b)
        out["a%d" % k], out["b%d" % k], out["c%d" % k] = a, b, np.asarray(c)
        out["l%d" % k] = np.array([len(ms), len(os_), len(ns)])
        out["t%d" % k] = np.einsum("%s->%s" % (string.ascii_lowercase[:a.ndim],
                                               string.ascii_lowercase[:a.ndim][::-1]), a) if a.ndim else a
    out["n_cases"] = np.array(len(CASES))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gmul_einsum.npz"), **out)


if __name__ == "__main__":
    main()
