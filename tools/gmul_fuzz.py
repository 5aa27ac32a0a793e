"""Random `gmul lM lO lN` calls -- ranks 0..3 on each side, hidden batch on either operand or both, the batch-summed
form, fp32 and fp64 -- against numpy einsum on small integers (bit-exact).  usage: gmul_fuzz.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd.hipt import HipT
from tools.mismatch_report import same
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
Ts = {np.float32: HipT(0), np.float64: HipT(0, dtype=np.float64)}
bad = 0
for case in range(n_cases):
    dt = np.float32 if rng.random() < 0.7 else np.float64
    T = Ts[dt]
    lm, lo, ln = int(rng.integers(0, 4)), int(rng.integers(0, 4)), int(rng.integers(0, 4))
    if lm + lo > 4 or lo + ln > 4:
        continue
    big = rng.random() < 0.2
    d = lambda: int(rng.integers(1, 40 if big else 7))
    ms, os_, ns = [d() for _ in range(lm)], [d() for _ in range(lo)], [d() for _ in range(ln)]
    ba, bb = bool(rng.integers(2)), bool(rng.integers(2))
    B = int(rng.choice([1, 2, 5, 33]))
    red = (ba and bb) and rng.random() < 0.4
    a = rng.integers(-2, 3, ([B] if ba else []) + ms + os_).astype(dt)
    b = rng.integers(-2, 3, ([B] if bb else []) + os_[::-1] + ns).astype(dt)
    L = "abcdefghijklmnop"
    mi, oi, ni = L[:lm], L[lm:lm + lo], L[lm + lo:lm + lo + ln]
    sa, sb = ("z" if ba else "") + mi + oi, ("z" if bb else "") + oi[::-1] + ni
    so = ("" if red or not (ba or bb) else "z") + mi + ni
    want = np.einsum("%s,%s->%s" % (sa, sb, so), a.astype(np.float64), b.astype(np.float64)).astype(dt)
    A, Bt = T.put(a, batched=ba), T.put(b, batched=bb)
    try:
        got = (T.gmul_batch_sum if red else T.gmul)(lm, lo, ln, A, Bt).numpy()
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("ERROR", case, dt.__name__, (lm, lo, ln), ms, os_, ns, "batch", ba, bb, B, "red", red, repr(e)[:200])
        continue
    got = np.asarray(got).reshape(want.shape) if got.size == want.size else got
    if not same(got, want, tool='gmul_fuzz', case=case, dtype=dt.__name__, lm=lm, lo=lo, ln=ln, ms=ms, os=os_, ns=ns, ba=ba, bb=bb, B=B, red=red):
        bad += 1
        print("MISMATCH", case, dt.__name__, (lm, lo, ln), ms, os_, ns, "batch", ba, bb, B, "red", red, got.shape, want.shape)
print("cases", n_cases, "mismatches", bad)
