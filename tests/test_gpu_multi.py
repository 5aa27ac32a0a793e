"""Data-parallel batched gradTOp on the HIP path with more than one rank (BASELINE config 4):
rows sharded contiguously, ONE all-reduce of the flat weight gradient per step, identical update everywhere.

* test_p2p_two_ranks_on_one_gpu runs wherever there is ONE GPU: two processes share it, map each other's exchange
  buffers through hipIpc and run the one-shot peer-to-peer all-reduce (csrc/p2p.hip) with the SGD update fused.
* test_dp_ranks_on_separate_gpus needs >= 2 GPUs (skipped otherwise): min(n, 8) ranks, both collectives (RCCL on
  the C ABI, peer-to-peer), against the single-GPU full-batch step: 1e-5, replicas bit-identical."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def device_count():
    import ctypes as C
    from tensor_ops_amd import capi
    n = C.c_int()
    capi.check(capi.lib().to_device_count(C.byref(n)))
    return n.value


def full_batch_reference(world, rows, steps):
    """the same steps on ONE GPU over the whole global batch"""
    import bench
    from tensor_ops_amd import tops
    from tensor_ops_amd.hipt import HipT
    T = HipT(0)
    ws, _, _ = bench.synth(0, 8)
    shards = [bench.synth(r, rows) for r in range(world)]
    X = np.concatenate([s[1] for s in shards])
    Y = np.concatenate([s[2] for s in shards])
    net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
    tr = tops.Trainer(net, "crossEntropy", 0.02 / (rows * world), T.put(X, batched=True), T.put(Y, batched=True),
                      use_graph=False)
    for _ in range(steps):
        tr.grad()
        tr.apply()
    return [p.numpy() for p in tr.net.params]


def run_ranks(mode, world, rows, steps, same_gpu, repo_root):
    port = 29600 + os.getpid() % 300
    with tempfile.TemporaryDirectory() as out:
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", TOPS_P2P_TIMEOUT_S="20")
            procs.append(subprocess.Popen([sys.executable, os.path.join(repo_root, "tools", "dp_worker.py"), mode, out,
                                           str(rows), str(steps), "1" if same_gpu else "0"],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        logs = []
        for p in procs:
            try:
                o, _ = p.communicate(timeout=240)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            logs.append(o.decode(errors="replace"))
        for p, lg in zip(procs, logs):
            assert p.returncode == 0, lg[-3000:]
        return [np.load(os.path.join(out, "params_%d.npy" % r)) for r in range(world)]


def check(flat_by_rank, want_params):
    for f in flat_by_rank[1:]:
        assert np.array_equal(f, flat_by_rank[0])          # replicas stay bit-identical
    off = 0
    for w in want_params:
        got = flat_by_rank[0][off:off + w.size].reshape(w.shape)
        err = np.linalg.norm((got.astype(np.float64) - w).ravel()) / np.linalg.norm(w.astype(np.float64).ravel())
        assert err < RTOL, err
        off += (w.size + 3) // 4 * 4


def test_p2p_two_ranks_on_one_gpu(repo_root):
    rows, steps = 256, 3
    got = run_ranks("p2p", 2, rows, steps, True, repo_root)
    check(got, full_batch_reference(2, rows, steps))


@pytest.mark.parametrize("mode", ["rccl", "p2p"])
def test_dp_ranks_on_separate_gpus(repo_root, mode):
    n = device_count()
    if n < 2:
        pytest.skip("needs >= 2 GPUs (this box has %d)" % n)
    world = min(n, 8)
    rows, steps = 1024, 3                                    # BASELINE config 4: 1024 rows per GPU
    got = run_ranks(mode, world, rows, steps, False, repo_root)
    check(got, full_batch_reference(world, rows, steps))


def test_bench_line_from_two_ranks_sharing_the_gpu(repo_root):
    """`bench.py --gpus 2 --same-gpu` exactly as the driver launches an N>1 run (torch.distributed.run, one process per
    rank, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment): the whole multi-rank branch -- shard
    synthesis, transport set-up and probes, the timed regions with their barriers and the MAX over ranks, the JSON line
    -- runs end to end before an 8-GPU box ever sees it.  The ranks time-share GPU 0, so the number is not a scaling
    result; what is checked is that the path works and the line is well-formed."""
    import json
    port = 29700 + os.getpid() % 200
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(repo_root, "bench.py"),
                        "--gpus", "2", "--same-gpu", "--steps", "5", "--warmup", "2", "--regions", "3"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=repo_root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 2048 and d["config"]["parallelism"] == "dp2"
    us = d["config"]["collective_us_alone"]
    assert us["p2p_one_shot_to_p2p_allreduce_sum"] > 0 and us["payload_bytes"] == 814128
    assert "direct_unavailable" in us and "peer-to-peer" in d["config"]["collective"]
    assert d["step"]["params_finite_after_timed_region"] is True
    assert d["cpu_baseline"] is None                                # (an N = 1 leg)
    # N > 1: the exchange as a roofline object, the step as one captured launch list, the scaling figure against this
    # run's own N = 1 rate -- and the transport `auto` chose is the one the step uses
    rf = d["roofline"]
    assert rf["bound"] == "xgmi-latency" and rf["payload_bytes"] == 814128 and rf["peak"] == 153.0
    assert rf["us_alone"] == us["p2p_one_shot_to_p2p_allreduce_sum"] and 0 < rf["frac"] < 1 and 0 < rf["frac_of_7_links"] < rf["frac"]
    assert rf["bytes_out_per_rank"] == 814128 and "p2p" in rf["kernel"]
    assert us["auto_chose"].startswith("p2p")
    assert d["step"]["whole_step_captured_as_one_launch_list"] is True
    assert d["n1_steps_per_s_this_run"] > 0 and abs(d["weak_scaling_efficiency"] - d["value"] / (2 * d["n1_steps_per_s_this_run"])) < 1e-3
    # the line is also config-4 parity evidence: three steps of the timed step object from the
    # initial parameters against a single-GPU replay on the full batch, replicas bit-identical
    c4 = d["c4_parity"]
    assert c4["ok"] is True and c4["steps"] == 3 and c4["replicas_bit_identical"] is True
    assert d["c4_parity_rel_err"] == c4["rel_err"] < 1e-5
    assert "rccl_nranks" in d and d["rccl_nranks"] is None          # (two ranks on one device: RCCL refuses, no communicator)


def test_bench_line_with_the_rccl_leg_forced_at_world_one(repo_root):
    """the RCCL leg of the N > 1 line has never run on hardware (one GPU per box).  `bench.py --force-dist
    --collective direct` makes a one-rank world: the library loads RCCL, builds a communicator (to_comm_init), the step's
    all-reduce goes through `to_comm_allreduce_sum` on the library's stream, and the line takes the N > 1 branch -- the
    exchange as a roofline object, `rccl_nranks` from ncclCommCount -- so a test has parsed that leg before an 8-GPU box does."""
    import json
    port = 29500 + os.getpid() % 200
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(repo_root, "bench.py"), "--gpus", "1", "--force-dist", "--collective", "direct",
                        "--no-aux", "--steps", "5", "--warmup", "2", "--regions", "3"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=repo_root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["step"]["params_finite_after_timed_region"] is True
    assert d["rccl_nranks"] == 1                                              # RCCL's own count of the communicator
    assert "to_comm_allreduce_sum" in d["config"]["collective"]
    us = d["config"]["collective_us_alone"]
    assert us["rccl_to_comm_allreduce_sum"] > 0 and us["payload_bytes"] == 814128
    rf = d["roofline"]
    assert rf["bound"] == "xgmi-latency" and rf["payload_bytes"] == 814128 and rf["peak"] == 153.0 and rf["unit"] == "GB/s"
    assert rf["us_alone"] == us["rccl_to_comm_allreduce_sum"] and "RCCL" in rf["kernel"]
    assert rf["bytes_out_per_rank"] == 0 and rf["achieved"] == 0.0            # (a world of one moves nothing: the fields, not a rate)
    assert d["cpu_baseline"] is None and d["c4_parity"] is None
