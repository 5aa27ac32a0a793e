import sys, os, time, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(25, exit=True)
import numpy as np
from tensor_ops_amd import tops as H
from tensor_ops_amd.hipt import HipT
T = HipT(0); H.hlib()
rng = np.random.default_rng(1)
def fc(o,i): return tuple(T.put(v) for v in (0.5*rng.standard_normal(o),0.5*rng.standard_normal((o,o)),0.5*rng.standard_normal((o,i)),0.5*rng.standard_normal(o)))
def ff(o,i): return tuple(T.put(v) for v in (0.5*rng.standard_normal((o,i)),0.5*rng.standard_normal(o)))
which = sys.argv[1]
n = int(sys.argv[2])
if which == "a":
    net = H.rnn_genNet([], (fc(3,2), "actLogistic"), "actLogistic"); i, o, loss = 2, 3, "squaredError"
elif which == "b":
    net = H.rnn_genNet([(fc(4,3), "actLogistic", "actLogistic")], (fc(2,4), "actLogistic"), "actLogistic"); i, o, loss = 3, 2, "squaredError"
elif which == "c":
    net = H.rnn_genNet([(fc(4,3), "actLogistic", "actLogistic"), (ff(5,4), "actMapLogistic", None)], (fc(2,5), "actLogistic"), "actSoftmax"); i, o, loss = 3, 2, "crossEntropy"
xs = [T.put(rng.uniform(-1,1,i)) for _ in range(n)]
ys = [T.put(rng.uniform(0.1,0.9,o)) for _ in range(n)]
print("built", flush=True)
t = time.time(); l0 = T.stats()["launches"]
cur = net
for x in xs:
    y, cur = H.rnn_runNetwork(cur, x)
T.sync(); print("run", round(time.time()-t,3), T.stats()["launches"]-l0, flush=True)
t = time.time(); l0 = T.stats()["launches"]
with T.memo():
    g = H.rnn_netGrad(net, loss, xs, ys)
T.sync(); print("grad memo", round(time.time()-t,3), T.stats()["launches"]-l0, flush=True)
t = time.time(); l0 = T.stats()["launches"]
g = H.rnn_netGrad(net, loss, xs, ys)
T.sync(); print("grad nomemo", round(time.time()-t,3), T.stats()["launches"]-l0, flush=True)
