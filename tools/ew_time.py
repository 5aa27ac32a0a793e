"""map logistic / add over n elements: GB/s.  usage: ew_time.py n [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensor_ops_amd.hipt import HipT, logistic_closure
T = HipT(0)
e = T.expr(logistic_closure, 1, key="ewt")
for n in [int(x) for x in sys.argv[1:]]:
    a = T.genRand((n,), "uniform", -1, 1, 1); b = T.genRand((n,), "uniform", -1, 1, 2)
    for name, f, by in (("logistic", lambda: T.liftT(e, [a]), 8.0 * n), ("add", lambda: T.sumT([a, b], (n,)), 12.0 * n)):
        for _ in range(20): f()
        T.sync(); T.timer_start()
        for _ in range(50): f()
        ms = T.timer_stop() / 50
        print("%-9s n %11d  %8.4f ms  %7.1f GB/s" % (name, n, ms, by / ms / 1e6), flush=True)
    del a, b
