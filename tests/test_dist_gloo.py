"""The N>1 path on CPU: world_size-2 `gloo` processes run the data-parallel step logic of
tensor-ops_amd/dist.py (row sharding, ONE all-reduce(sum) of the flat gradient buffer,
identical update on every rank) with the oracle's C restatement standing in for the
per-rank gradient kernel; the result must equal the single-process full-batch step and
the replicas must stay bit-identical."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import tensor_ops_amd  # noqa: F401
    from tensor_ops_amd.dist import DataParallel, init_process_group, shard_rows
    from oracle import hmat
    import bench
    init_process_group("gloo")
    B = 16
    ws, _, _ = bench.synth(0, 1)
    lo, hi = shard_rows(B * world, rank, world)
    _, Xg, Yg = _global_batch(B * world)
    X, Y = Xg[lo:hi], Yg[lo:hi]
    sizes = [ws[0][0].size, ws[0][1].size, ws[1][0].size, ws[1][1].size]
    offs = np.cumsum([0] + [(s + 3) // 4 * 4 for s in sizes])
    flat_p = torch.zeros(int(offs[-1]), dtype=torch.float64)
    flat_g = torch.zeros(int(offs[-1]), dtype=torch.float64)
    params = [ws[0][0], ws[0][1], ws[1][0], ws[1][1]]
    for o, p in zip(offs, params):
        flat_p[int(o):int(o) + p.size] = torch.from_numpy(p.ravel().copy())

    def views():
        return [flat_p[int(o):int(o) + s].numpy().reshape(p.shape) for o, s, p in zip(offs, sizes, params)]

    def grad_fn():
        g, _ = hmat.batched_grads(X, Y, *views(), recompute=False)
        for o, gi in zip(offs, g):
            flat_g[int(o):int(o) + gi.size] = torch.from_numpy(gi.ravel())

    def apply_fn():
        flat_p.sub_(0.02 * flat_g)

    dp = DataParallel(flat_g, grad_fn, apply_fn, world)
    for _ in range(3):
        dp.step()
    np.save(os.path.join(out_dir, "p%d.npy" % rank), flat_p.numpy())
    torch.distributed.destroy_process_group()


def _global_batch(n):
    rng = np.random.default_rng(1234)
    X = rng.uniform(0, 1, size=(n, 784))
    Y = np.zeros((n, 10))
    Y[np.arange(n), rng.integers(0, 10, size=n)] = 1.0
    return None, X, Y


def test_shard_rows():
    sys.path.insert(0, ROOT)
    import tensor_ops_amd  # noqa: F401
    from tensor_ops_amd.dist import shard_rows
    assert [shard_rows(8192, r, 8) for r in (0, 7)] == [(0, 1024), (7168, 8192)]
    with pytest.raises(ValueError):
        shard_rows(10, 0, 4)


def test_data_parallel_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    p0, p1 = np.load(tmp_path / "p0.npy"), np.load(tmp_path / "p1.npy")
    assert np.array_equal(p0, p1), "replicas diverged"
    # single process, full batch, same three steps
    sys.path.insert(0, ROOT)
    import bench
    from oracle import hmat
    ws, _, _ = bench.synth(0, 1)
    _, X, Y = _global_batch(32)
    params = [ws[0][0].copy(), ws[0][1].copy(), ws[1][0].copy(), ws[1][1].copy()]
    for _ in range(3):
        g, _l = hmat.batched_grads(X, Y, *params, recompute=False)
        params = [p - 0.02 * gi for p, gi in zip(params, g)]
    off = 0
    for p in params:
        np.testing.assert_allclose(p0[off:off + p.size].reshape(p.shape), p, rtol=1e-12, atol=1e-14)
        off += (p.size + 3) // 4 * 4


# ---- the set-up logic bench.py runs before a multi-rank measurement, on CPU ---------------------------------------
class _GlooTransports:
    """Stand-in for tensor-ops_amd.dist.HipCollectives: flat buffers are CPU torch tensors, both "transports" are gloo
    all-reduces -- optionally broken in the ways a first run on real hardware could break them."""

    def __init__(self, dist, bufs, direct="ok", p2p="ok", rank=0):
        self.dist, self.bufs, self.direct, self.p2p, self.rank = dist, bufs, direct, p2p, rank
        self.calls = []
        self.n_comm = 0

    def wrap(self, ptr, n):
        return self.bufs[ptr]

    def init_direct(self, rank, world):
        self.calls.append("init_direct")
        if self.direct == "raise_on_rank1" and rank == 1:
            raise RuntimeError("ncclCommInitRank: invalid usage (ranks share one device)")

    def init_p2p(self, rank, world, n):
        self.calls.append("init_p2p")
        return "to_p2p_create: no fine-grained memory" if self.p2p == "unavailable" else None

    def _sum(self, t, wrong=False):
        self.dist.all_reduce(t)
        if wrong:
            t.add_(1.0)
        return 0

    def comm_allreduce(self, t):
        self.calls.append("comm")
        self.n_comm += 1
        if self.direct == "slow_on_rank1" and self.rank == 1:
            import time
            time.sleep(0.01)      # (only ONE rank sees RCCL slow: the decision must still be the same everywhere)
        st = self._sum(t, self.direct == "wrong_sum")
        if self.direct == "fails_while_timed_on_rank0" and self.rank == 0 and self.n_comm == 3:
            return 7              # (the probe passed; a later call reports an error on one rank only)
        return st

    def p2p_allreduce(self, t):
        self.calls.append("p2p")
        if self.p2p == "slow":
            import time
            time.sleep(0.01)
        return self._sum(t)

    def sync(self):
        pass


class _Flat:
    """a CPU tensor with the `data_ptr()` the set-up code keys handles on"""

    def __init__(self, t, key):
        self.t, self.key = t, key

    def data_ptr(self):
        return self.key

    def fill_(self, v):
        self.t.fill_(v)

    def zero_(self):
        self.t.zero_()

    def __eq__(self, v):
        return self.t == v


def _setup_worker(rank, world, port, out_dir, scenario):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import json
    import torch
    import tensor_ops_amd  # noqa: F401
    from tensor_ops_amd.dist import init_process_group, setup_collectives
    dist = init_process_group("gloo")
    n = 1024
    g, p = torch.zeros(n), torch.zeros(n)
    api = _GlooTransports(dist, {1: g, 2: p}, rank=rank, **scenario["api"])
    res = {}
    try:
        got = setup_collectives(api, dist, rank, world, _Flat(g, 1), _Flat(p, 2), n, scenario["want"], timing_iters=3)
        res = {"collective": got["collective"], "us": got["collective_us"], "direct": got["direct"] is not None,
               "p2p_params": got["p2p_params"] is not None, "calls": api.calls}
    except SystemExit as e:
        res = {"exit": str(e)}
    json.dump(res, open(os.path.join(out_dir, "r%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


SCENARIOS = {
    "both_fine": {"want": "direct", "api": {}},
    "rccl_refuses_on_one_rank": {"want": "direct", "api": {"direct": "raise_on_rank1"}},
    "rccl_sums_wrongly": {"want": "direct", "api": {"direct": "wrong_sum"}},
    "p2p_asked_but_unavailable": {"want": "p2p", "api": {"p2p": "unavailable"}},
    "p2p_asked": {"want": "p2p", "api": {}},
    # --collective auto (bench.py's default): the transport that MEASURED faster, the same one on every rank
    "auto_p2p_is_faster": {"want": "auto", "api": {"direct": "slow_on_rank1"}},
    "auto_rccl_is_faster": {"want": "auto", "api": {"p2p": "slow"}},
    "auto_rccl_unavailable": {"want": "auto", "api": {"direct": "raise_on_rank1"}},
    "auto_p2p_unavailable": {"want": "auto", "api": {"p2p": "unavailable"}},
    # a transport that passes its probe and then fails while it is being timed (one rank only) costs its leg, not the run
    "rccl_fails_while_timed": {"want": "auto", "api": {"direct": "fails_while_timed_on_rank0"}},
}


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_collective_setup_control_flow(tmp_path, name):
    """bench.py's multi-rank set-up (tensor-ops_amd/dist.py::setup_collectives) had never executed anywhere before the
    driver's first 8-GPU run.  Two gloo ranks walk it here with stand-in transports: every rank must reach the same
    decision whatever fails where, nobody may be left waiting in a collective, and the bench line's fields come out."""
    import json
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_setup_worker, args=(r, world, port, str(tmp_path), SCENARIOS[name])) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, name
    r0, r1 = (json.load(open(tmp_path / ("r%d.json" % r))) for r in range(world))
    if name == "p2p_asked_but_unavailable":
        assert "exit" in r0 and "exit" in r1 and "peer-to-peer" in r0["exit"]
        return
    assert r0["collective"] == r1["collective"] and r0["direct"] == r1["direct"] and r0["p2p_params"] == r1["p2p_params"]
    us = r0["us"]
    assert us["payload_bytes"] == 4096
    if name == "both_fine":
        assert r0["collective"] == "direct" and us["rccl_to_comm_allreduce_sum"] > 0 and us["p2p_one_shot_to_p2p_allreduce_sum"] > 0
    elif name in ("rccl_refuses_on_one_rank", "rccl_sums_wrongly"):
        # the C-ABI RCCL leg is dropped on BOTH ranks, the step falls back to the peer-to-peer exchange
        assert r0["collective"] == "p2p" and r0["p2p_params"] and "direct_unavailable" in us and "direct_unavailable" in r1["us"]
        assert "rccl_to_comm_allreduce_sum" not in us and us["fallback"].startswith("p2p")
    elif name == "p2p_asked":
        assert r0["collective"] == "p2p" and r0["p2p_params"]
    elif name == "auto_p2p_is_faster":
        # rank 1's RCCL calls take 10 ms, rank 0's do not: both ranks see the slowest rank's latency and choose alike
        assert r0["collective"] == "p2p" and r0["p2p_params"] and us["auto_chose"].startswith("p2p") and r1["us"]["auto_chose"] == us["auto_chose"]
        assert us["rccl_to_comm_allreduce_sum"] == r1["us"]["rccl_to_comm_allreduce_sum"] > 5000
    elif name == "auto_rccl_is_faster":
        assert r0["collective"] == "direct" and not r0["p2p_params"] and r0["direct"] and us["auto_chose"].startswith("rccl")
    elif name == "auto_rccl_unavailable":
        assert r0["collective"] == "p2p" and r0["p2p_params"] and "direct_unavailable" in us
    elif name == "auto_p2p_unavailable":
        assert r0["collective"] == "direct" and r0["direct"] and "p2p_unavailable" in us and us["auto_chose"].startswith("rccl")
    elif name == "rccl_fails_while_timed":
        assert r0["collective"] == "p2p" and r0["p2p_params"] and "while it was being timed" in us["direct_unavailable"]
        assert "while it was being timed" in r1["us"]["direct_unavailable"] and "rccl_to_comm_allreduce_sum" not in us
