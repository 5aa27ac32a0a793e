"""Data-parallel batched gradTOp: one process per GPU, batch rows sharded contiguously,
ONE all-reduce(sum) of the flat weight-gradient buffer per step.

`torch.distributed` is plumbing: backend "nccl" is RCCL over xGMI on the GPU box, "gloo"
in the CPU tests.  No other collective touches the data path: samples never interact
(the net is per-sample, FeedForward.hs:57-61), parameters are replicated and every rank
applies the identical update, so replicas stay bit-identical.
"""
import os


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), \
        int(os.environ.get("WORLD_SIZE", 1))


def shard_rows(global_rows, rank, world):
    """Contiguous row range [lo, hi) of rank `rank` (C4: 8192 rows -> 8 x 1024)."""
    if global_rows % world != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_rows, world))
    per = global_rows // world
    return rank * per, (rank + 1) * per


def init_process_group(backend):
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend)
    return dist


def init_direct_comm(rank, world):
    """The C-ABI collective (to_comm_*, RCCL loaded by the library): rank 0 creates the unique id, any
    torch.distributed backend (gloo is enough) carries its 128 bytes to the other ranks."""
    import ctypes as C
    from . import capi
    buf = (C.c_char * 128)()
    if rank == 0:
        capi.check(capi.lib().to_comm_unique_id(buf))
    if world > 1:
        import torch
        import torch.distributed as dist
        t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        dist.broadcast(t, src=0)
        buf = (C.c_char * 128).from_buffer_copy(bytes(t.tolist()))
    capi.check(capi.lib().to_comm_init(rank, world, buf))


def init_p2p(rank, world, n_elems, dtype_code=0):
    """The one-shot peer-to-peer all-reduce (to_p2p_*, csrc/p2p.hip): every rank creates its exchange buffer, the
    64-byte IPC handles are all-gathered over torch.distributed (any backend; gloo is enough), every rank maps
    its peers' buffers.  Returns None on success, else the reason -- and it is the SAME answer on every rank:
    a rank that fails still takes part in every collective of the set-up, so nobody is left waiting for it."""
    import ctypes as C
    from . import capi
    L = capi.lib()
    mine = (C.c_char * 64)()
    err = None
    if L.to_p2p_create(n_elems, dtype_code, world, mine) != 0:
        err = "to_p2p_create: " + L.to_last_error().decode(errors="replace")
    if world > 1:
        import torch
        import torch.distributed as dist
        t = torch.frombuffer(bytearray(mine.raw), dtype=torch.uint8).clone()
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        blob = b"".join(bytes(o.tolist()) for o in out)
        ok = torch.tensor([0 if err else 1], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if not ok.item():
            L.to_p2p_shutdown()
            return err or "a peer could not create its exchange buffer"
    else:
        blob = mine.raw
        if err:
            return err
    handles = (C.c_char * (64 * world)).from_buffer_copy(blob)
    if L.to_p2p_connect(rank, handles) != 0:
        err = "to_p2p_connect: " + L.to_last_error().decode(errors="replace")
    if world > 1:
        ok = torch.tensor([0 if err else 1], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if not ok.item():
            return err or "a peer could not map the exchange buffers"
    return err


class HipCollectives:
    """The two C-ABI transports as `setup_collectives` drives them (a CPU test substitutes a gloo-backed stand-in to
    run the same control flow without a GPU)."""

    def __init__(self, T, one_device_per_rank=True):
        import ctypes as C
        from . import capi
        from .hipt import DT
        self._C, self._capi, self._DT, self._T = C, capi, DT, T
        self._rccl_ok = one_device_per_rank

    def wrap(self, ptr, n):
        C, capi = self._C, self._capi
        h = capi.c_tensor()
        d = (C.c_int64 * 1)(n)
        capi.check(capi.lib().to_wrap(C.c_void_p(ptr), capi.TO_F32, 1, d, 0, C.byref(h)))
        return self._DT(h)

    def init_direct(self, rank, world):
        if not self._rccl_ok:   # (RCCL refuses two ranks on one device; do not find out by trying)
            raise RuntimeError("ranks share one device: RCCL needs one device per rank")
        init_direct_comm(rank, world)

    def init_p2p(self, rank, world, n):
        return init_p2p(rank, world, n)

    def comm_allreduce(self, handle):
        return self._capi.lib().to_comm_allreduce_sum(handle.h)

    def p2p_allreduce(self, handle):
        return self._capi.lib().to_p2p_allreduce_sum(handle.h)

    def sync(self):
        self._T.sync()


def setup_collectives(api, dist, rank, world, flat_g, flat_p, nflat, want, timing_iters=200):
    """Bring up the all-reduce transports of a multi-rank run and decide which one the step uses.

    Both C-ABI transports (RCCL loaded by the library, the one-shot peer-to-peer exchange) are set up and probed with
    a known vector before anything is timed on them, so that the bench line can carry the stand-alone latency of each
    (SURVEY.md 8(e)).  A transport that cannot be set up, or returns a wrong sum, costs its leg, not the run -- and
    every rank takes the same way out (each decision is agreed through a MIN all-reduce over `dist`).
    `want`: "auto" | "direct" | "p2p" | "torch".  "auto" takes the transport that MEASURED faster here (the slowest
    rank's latency of each, agreed through a MAX all-reduce, so every rank decides alike); the others name one and fall
    back RCCL -> p2p -> torch only when it cannot be set up.
    Returns a dict: direct (handle of the flat gradient, or None), p2p_params
    (handle of the flat parameters when the step uses the p2p exchange), torch_group, collective (the transport the step
    will use), collective_us (latencies / reasons)."""
    import time
    import torch

    def agree(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    out = {"direct": None, "p2p_params": None, "torch_group": None, "collective": want, "collective_us": None}
    if want == "torch":
        return out
    direct = api.wrap(flat_g.data_ptr(), nflat)
    expect = float(world * (world + 1) // 2)
    direct_err = None
    try:
        api.init_direct(rank, world)
    except Exception as e:  # noqa: BLE001
        direct_err = "set-up: %r" % (e,)
    if agree(direct_err is None):
        flat_g.fill_(float(rank + 1))
        st = api.comm_allreduce(direct)
        api.sync()
        if not agree(st == 0 and bool((flat_g == expect).all().item())):
            direct_err = "probe all-reduce failed (status %d)" % st
    elif direct_err is None:
        direct_err = "a peer could not set it up"
    p2p_err = api.init_p2p(rank, world, nflat)   # the same answer on every rank
    if p2p_err is None:
        flat_g.fill_(float(rank + 1))
        st = api.p2p_allreduce(direct)
        api.sync()
        if not agree(st == 0 and bool((flat_g == expect).all().item())):
            p2p_err = "probe exchange failed (status %d)" % st
    if want == "p2p" and p2p_err is not None:
        raise SystemExit("--collective p2p: the peer-to-peer exchange could not be set up: %s" % p2p_err)
    if want == "p2p":
        out["p2p_params"] = api.wrap(flat_p.data_ptr(), nflat)
    us = {}
    legs = []
    if direct_err is None:
        legs.append(("rccl_to_comm_allreduce_sum", api.comm_allreduce))
    else:
        us["direct_unavailable"] = direct_err
    if p2p_err is None:
        legs.append(("p2p_one_shot_to_p2p_allreduce_sum", api.p2p_allreduce))
    else:
        us["p2p_unavailable"] = p2p_err
    for name, fn in legs:
        flat_g.zero_()
        # (the calls are statements, not `assert`s: python -O strips an assert together with the call inside it -- and a
        #  failure on one rank must not leave the others waiting in the barrier: the status is agreed before anything
        #  is reported)
        bad = 0
        for _ in range(min(20, timing_iters)):
            st = fn(direct)           # (always called: a rank that stopped calling would strand its peers in the collective)
            bad = bad or st
        dist.barrier()
        api.sync()
        t0 = time.perf_counter()
        for _ in range(timing_iters):
            st = fn(direct)
            bad = bad or st
        api.sync()
        took = (time.perf_counter() - t0) / timing_iters * 1e6
        if agree(bad == 0):
            # the latency every rank will live with is the slowest rank's
            t = torch.tensor([took], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            us[name] = round(float(t.item()), 2)
        else:
            reason = "a call failed while it was being timed (status %d on rank %d)" % (bad, rank) if bad else "a call failed on a peer while it was being timed"
            if name.startswith("rccl"):
                direct_err = reason
                us["direct_unavailable"] = reason
            else:
                p2p_err = reason
                us["p2p_unavailable"] = reason
    us["payload_bytes"] = nflat * 4
    out["direct"] = direct
    if want == "auto":
        t_rccl = us.get("rccl_to_comm_allreduce_sum") if direct_err is None else None
        t_p2p = us.get("p2p_one_shot_to_p2p_allreduce_sum") if p2p_err is None else None
        if t_rccl is None and t_p2p is None:
            want = "direct"          # neither C-ABI transport: the fallback chain below ends at torch
        elif t_p2p is not None and (t_rccl is None or t_p2p <= t_rccl):
            want = "p2p"
        else:
            want = "direct"
        us["auto_chose"] = {"direct": "rccl_to_comm_allreduce_sum", "p2p": "p2p_one_shot_to_p2p_allreduce_sum"}[want]
        out["collective"] = want
        if want == "p2p":
            out["p2p_params"] = api.wrap(flat_p.data_ptr(), nflat)
    if direct_err is not None and want == "direct":
        if p2p_err is None:
            # RCCL through the C ABI is not available (e.g. ranks sharing one GPU): the peer-to-peer exchange carries it
            out["p2p_params"] = api.wrap(flat_p.data_ptr(), nflat)
            out["collective"] = "p2p"
            us["fallback"] = "p2p one-shot exchange"
        else:
            # fall back to torch.distributed's RCCL process group for the step's all-reduce
            out["torch_group"] = dist.new_group(backend="nccl")
            out["direct"] = None
            out["collective"] = "torch"
            us["fallback"] = "torch.distributed nccl group"
    out["collective_us"] = us
    return out


class DataParallel:
    """step() = local summed gradients -> all-reduce(sum) on the flat buffer -> SGD update.

    grad_fn()  enqueues the local G_r into `flat_grads` (a torch tensor view)
    apply_fn() applies p <- p - rate * G on the flat parameter buffer
    """

    def __init__(self, flat_grads, grad_fn, apply_fn, world, force=False, direct_handle=None, step_fn=None,
                 p2p_params=None, p2p_rate=0.0, group=None):
        """`direct_handle`: a library handle (hipt.DT) of the flat gradient buffer -> the all-reduce goes
        through the C ABI (to_comm_allreduce_sum) instead of torch.distributed.
        `p2p_params`: a library handle of the flat PARAMETER buffer -> the exchange is the one-shot peer-to-peer
        all-reduce with the update `p - rate * sum` applied in the same launch (to_p2p_allreduce_sgd); apply_fn
        is then not called.
        `step_fn`: grad + update as one call (Trainer.step: the update fused into the gradient launches),
        used when there is nothing to all-reduce (a single rank)."""
        self.step_fn = step_fn
        self.flat_grads = flat_grads
        self.grad_fn = grad_fn
        self.apply_fn = apply_fn
        self.world = 2 if (force and world == 1) else world  # force: exercise the collective at world 1
        self.direct = direct_handle
        self.p2p_params = p2p_params
        self.p2p_rate = float(p2p_rate)
        self.group = group   # torch.distributed process group of the all-reduce (None = the default group)
        self._graph = None   # the captured step (capture())
        if self.world > 1 and self.direct is None:
            import torch.distributed as dist
            self._dist = dist
        if self.direct is not None:
            from . import capi
            self._capi = capi

    def capture(self):
        """The N > 1 step as ONE replayable launch list: local gradients, the exchange, the update are issued once
        between to_graph_begin / to_graph_end (grad_fn must issue directly -- a Trainer with use_graph=False) and every
        later step() is a single to_graph_launch.  The peer-to-peer exchange is a plain kernel whose epoch lives on the
        device; RCCL's all-reduce is captured as a graph node (RCCL supports stream capture).  torch.distributed's
        collective is not capturable from here: such a step stays replay + host call + launch.  Returns True when the
        step is captured."""
        if self.world == 1 or self._graph is not None or self.direct is None:
            return self._graph is not None
        import ctypes as C
        capi = self._capi
        capi.check(capi.lib().to_graph_begin())
        try:
            self._issue()
        except Exception:
            g = capi.c_graph()
            capi.lib().to_graph_end(C.byref(g))
            if g:
                capi.lib().to_graph_release(g)
            raise
        g = capi.c_graph()
        capi.check(capi.lib().to_graph_end(C.byref(g)))
        self._graph = g
        return True

    def release(self):
        if self._graph is not None:
            self._capi.lib().to_graph_release(self._graph)
            self._graph = None

    def step(self):
        if self._graph is not None:
            self._capi.check(self._capi.lib().to_graph_launch(self._graph))
            return
        self._issue()

    def _issue(self):
        if self.world == 1 and self.step_fn is not None:
            self.step_fn()
            return
        self.grad_fn()
        if self.world > 1 and self.p2p_params is not None:
            self._capi.check(self._capi.lib().to_p2p_allreduce_sgd(self.p2p_params.h, self.direct.h,
                                                                   self.p2p_rate, 0))
            return
        if self.world > 1:
            # 203,530 floats = 814 KB: latency-bound; one collective on one flat buffer
            if self.direct is not None:
                self._capi.check(self._capi.lib().to_comm_allreduce_sum(self.direct.h))
            else:
                self._dist.all_reduce(self.flat_grads, op=self._dist.ReduceOp.SUM, group=self.group)
        self.apply_fn()
