"""One data-parallel rank: batched gradTOp on its shard -> all-reduce of the flat gradient -> SGD update, a few
steps; writes its final flat parameter vector.  Launched once per rank by tests/test_gpu_multi.py (one GPU per
rank) and tests/test_gpu_p2p.py (two ranks sharing ONE GPU: the peer-to-peer exchange only).
  env: RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT ; argv: mode(p2p|rccl) out_dir rows_per_rank steps same_gpu(0|1)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

mode, out_dir, rows, steps, same_gpu = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

dist.init_process_group(backend="gloo")   # bootstrap only: carries ids / IPC handles
import bench  # noqa: E402
from tensor_ops_amd import capi, tops  # noqa: E402
from tensor_ops_amd.dist import DataParallel, init_direct_comm, init_p2p  # noqa: E402
from tensor_ops_amd.hipt import DT, HipT  # noqa: E402

dev = 0 if same_gpu else rank
T = HipT(dev)
L = capi.lib()
ws, _, _ = bench.synth(0, 8)
_, X, Y = bench.synth(rank, rows)          # this rank's shard: rows [rank*rows, (rank+1)*rows) of the global batch
net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
rate = 0.02 / (rows * world)
tr = tops.Trainer(net, "crossEntropy", rate, T.put(X, batched=True), T.put(Y, batched=True), use_graph=False)
p_ptr, g_ptr, n = tr.flat()


def wrap(ptr):
    h = capi.c_tensor()
    d = (C.c_int64 * 1)(n)
    capi.check(L.to_wrap(C.c_void_p(ptr), capi.TO_F32, 1, d, 0, C.byref(h)))
    return DT(h)


G, P = wrap(g_ptr), wrap(p_ptr)
if mode == "p2p":
    err = init_p2p(rank, world, n)
    assert err is None, err
    dp = DataParallel(None, tr.grad, tr.apply, world, direct_handle=G, p2p_params=P, p2p_rate=rate)
else:
    init_direct_comm(rank, world)
    dp = DataParallel(None, tr.grad, tr.apply, world, direct_handle=G)
for _ in range(steps):
    dp.step()
T.sync()
st, code = C.c_int(), C.c_int()
capi.check(L.to_p2p_status(C.byref(st), C.byref(code)))
assert code.value == 0, "p2p exchange timed out: %d" % code.value
np.save(os.path.join(out_dir, "params_%d.npy" % rank), P.numpy())
dist.barrier()
dist.destroy_process_group()
