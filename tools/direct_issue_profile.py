"""Where the host time of a directly-issued (un-replayed) batched step goes, by function: Trainer.step(use_graph=False)
under a SIGPROF sampler (tools/probes/sigprof.c), stacks resolved against the two libraries' symbol tables with nm.
Prints self / inclusive shares and the implied microseconds of the step; writes gpurun_out/r05_direct_profile.txt.
    python tools/direct_issue_profile.py [softmax|logistic] [steps]"""
import bisect
import collections
import ctypes as C
import os
import subprocess
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tensor_ops_amd import tops  # noqa: E402
from tensor_ops_amd.hipt import HipT  # noqa: E402

head = sys.argv[1] if len(sys.argv) > 1 else "softmax"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
so = os.path.join(ROOT, "gpurun_out", "sigprof.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "probes", "sigprof.c"), "-ldl"])
prof = C.CDLL(so)

T = HipT(0)
ws, X, Y = bench.synth(0, 1024)
dX, dY = T.put(X, batched=True), T.put(Y, batched=True)
acts, loss = (("actMapLogistic", "actSoftmax"), "crossEntropy") if head == "softmax" else (("actLogistic", "actLogistic"), "squaredError")
net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], *acts)
tr = tops.Trainer(net, loss, bench.RATE / 1024, dX, dY, use_graph=False)
for _ in range(300):
    tr.step()
T.sync()
f = tr.step
t0 = time.perf_counter()
for _ in range(3000):
    f()
T.sync()
us_plain = (time.perf_counter() - t0) / 3000 * 1e6
prof.sigprof_start(4000)
t0 = time.perf_counter()
for _ in range(steps):
    f()
t1 = time.perf_counter()
n = prof.sigprof_stop()
T.sync()
us = (t1 - t0) / steps * 1e6
raw = os.path.join(ROOT, "gpurun_out", "r05_direct_stacks.txt")
prof.sigprof_dump(raw.encode(), 2)

tables = {}


def table(lib):
    if lib not in tables:
        syms = []
        try:
            out = subprocess.run(["nm", "-C", "--defined-only", lib], capture_output=True, text=True).stdout
            out += subprocess.run(["nm", "-C", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
            for line in out.splitlines():
                p = line.split(None, 2)
                if len(p) == 3 and p[1] in "tTwWiu":
                    syms.append((int(p[0], 16), p[2]))
        except OSError:
            pass
        syms.sort()
        tables[lib] = ([a for a, _ in syms], [s for _, s in syms])
    return tables[lib]


def resolve(fr):
    lib, _, off = fr.rpartition("+")
    off = int(off, 16)
    base = os.path.basename(lib)
    if not os.path.exists(lib):
        return base
    addrs, names = table(lib)
    i = bisect.bisect_right(addrs, off) - 1
    if i < 0:
        return base
    nm = names[i]
    return (nm if len(nm) < 150 else nm[:147] + "...") + "  [" + base + "]"


self_c, incl_c = collections.Counter(), collections.Counter()
group_c = collections.Counter()
for line in open(raw):
    frames = [resolve(x) for x in line.strip().split(";") if x]
    if not frames:
        continue
    self_c[frames[0]] += 1
    for s in set(frames):
        incl_c[s] += 1
    # coarse attribution: the innermost frame that belongs to one of the three libraries owns the sample (libc / libstdc++
    # frames above it -- malloc, free, memcpy -- are charged to their caller)
    owner = "python/ctypes"
    for s in frames:                                   # innermost first: first hit wins
        if "libamdhip64" in s or "libhsa" in s or "libamd_comgr" in s:
            owner = "HIP runtime (launch, events)"
            break
        if "[libtensorops_hip.so]" in s:
            owner = "libtensorops_hip (entry points, planner)"
            break
        if "[libtensorops_host.so]" in s:
            owner = "libtensorops_host (the TOp mirror)"
            break
    group_c[owner] += 1

tot = sum(self_c.values())
lines = ["direct step, %s head: %.2f us/step sampled (%.2f unsampled), %d samples over %d steps" % (head, us, us_plain, tot, steps), "", "-- who owns the samples (innermost library on the stack) --"]
for k, v in group_c.most_common():
    lines.append("%6.2f%%  %6.2f us  %s" % (100 * v / tot, us * v / tot, k))
lines += ["", "-- inclusive, top 60 --"]
for k, v in incl_c.most_common(60):
    lines.append("%6.2f%%  %6.2f us  %s" % (100 * v / tot, us * v / tot, k))
lines += ["", "-- self, top 50 --"]
for k, v in self_c.most_common(50):
    lines.append("%6.2f%%  %6.2f us  %s" % (100 * v / tot, us * v / tot, k))
txt = "\n".join(lines)
print(txt)
open(os.path.join(ROOT, "gpurun_out", "r05_direct_profile_%s.txt" % head), "w").write(txt + "\n")
os.remove(raw)
