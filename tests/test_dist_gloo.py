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
