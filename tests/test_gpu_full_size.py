"""Parity at BASELINE.json's FULL sizes through size-independent checks: sampled rows against
an fp64 CPU computation of the same rows, linearity, shard-sum identities, and the C oracle
for the full batched step.  Tolerance 1e-5 relative (fp32, north_star)."""
import ctypes as C

import numpy as np
import pytest

from tools.mismatch_report import same

pytestmark = pytest.mark.gpu

SEED = 0x7e500001
RTOL = 1e-5


@pytest.fixture(scope="module")
def T():
    from tensor_ops_amd.hipt import HipT
    return HipT(0)


def rel_err(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.linalg.norm((got - want).ravel()) / np.linalg.norm(want.ravel())


def test_c2_gmul_4096(T):
    """config 2: gmul '[4096,4096] x '[4096,4096]."""
    rng = np.random.default_rng(SEED)
    n = 4096
    a = rng.uniform(-1, 1, size=(n, n)).astype(np.float32)
    b = rng.uniform(-1, 1, size=(n, n)).astype(np.float32)
    da, db = T.put(a), T.put(b)
    c = T.gmul(1, 1, 1, da, db).numpy()
    rows = rng.choice(n, size=48, replace=False)
    want = a[rows].astype(np.float64) @ b.astype(np.float64)
    assert rel_err(c[rows], want) < RTOL
    cols = rng.choice(n, size=48, replace=False)
    want = a.astype(np.float64) @ b[:, cols].astype(np.float64)
    assert rel_err(c[:, cols], want) < RTOL
    # linearity in the first operand: (2a) b == 2 (a b) exactly in binary fp
    c2 = T.gmul(1, 1, 1, T.scaleT(2.0, da), db).numpy()
    assert np.array_equal(c2, 2 * c)
    # transposed views give the same GEMM: (b^T a^T)^T
    ct = T.gmul(1, 1, 1, T.transp(db), T.transp(da)).numpy()
    assert rel_err(ct.T[rows], c[rows]) < 1e-6


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,k,n", [(4096, 288, 4096), (4096, 304, 4352), (4100, 288, 4096), (4097, 304, 4097),
                                   (2048, 1024, 2048), (3072, 320, 3072), (2304, 1040, 2560), (1024, 4096, 1024),
                                   (1024, 1024, 1024), (1536, 1536, 1536), (1280, 528, 1920), (640, 2048, 512),
                                   (1000, 1008, 1000), (1100, 528, 900), (260, 256, 388), (1000, 1000, 1000),
                                   (2000, 640, 2000), (1001, 512, 1003), (4000, 288, 4000), (3900, 304, 4060),
                                   (640, 640, 640), (896, 200, 1408), (1792, 136, 1984), (1004, 333, 708), (1408, 1030, 1408),
                                   (8192, 344, 8200), (4096, 16, 4096), (4096, 48, 4096), (4352, 1024, 4352)])
def test_c2_kernel_every_layout_bit_exact_on_integers(T, ta, tb, m, k, n):
    """The full-tile GEMM kernel (four waves of 128x128, row-/column-owning 16-byte fragments: a lane's
    accumulators belong to permuted rows/columns that the epilogue maps back) on all four operand layouts, the
    WHOLE output compared: small-integer data, so every product and partial sum is exact in fp32 whatever the
    summation order.  (4100 rows / 4352 columns: a block of whole rounds of tiles goes to that kernel, the border
    strips elsewhere; 4097: rows that are only dword-aligned still take the 16-byte loads and the LDS DMA;
    1024 x 4096 x 1024: 16 tiles, the K loop split sixteen ways over blockIdx.y and summed by a second pass;
    2048^2 (64 tiles), 3072^2 (144 tiles) and 2304 x 2560 (90 tiles): stream-K, every workgroup an equal share of the k-tile stream,
    partial tiles added up in workgroup order by the fix-up pass;
    1024^3, 1536^3, 1280 x 528 x 1920, 640 x 2048 x 512, 2048^2: the same pinned body on 128x128 tiles -- four waves of
    64x64, 8-byte owning fragments -- with the K loop split two to four ways;
    1000 x 1008 x 1000, 1100 x 528 x 900, 260 x 256 x 388, 2000 x 640 x 2000: extents that are multiples of 4 but not of
    128 -- the edge tiles stay on the pinned body (a lane beyond the extent re-reads the last valid row / column,
    stores are guarded); 1000^3 adds a K tail (second launch); 1001 x 512 x 1003: not even multiples of 4 -- the block
    of whole tiles on the pinned kernel, the border strips elsewhere; 4000 x 288 x 4000 and 3900 x 304 x 4060: 256 tiles
    of 256x256 counting the edge tiles = one whole round, run whole on the 256-tile kernel.
    Round 3: 100 .. 1024 tiles of 64x64 (640^3 .. 2048^2, 1000^3 with its K tail, the last row of shapes: ragged M / N, K
    tails of 8, 13 and 6 inside the kernel) run on gemm_kwave.hip -- one workgroup per tile, the K loop split over its
    waves, partial tiles summed in LDS in wave order; 8192 x 344 x 8200: thousands of tiles and a K of a few hundred --
    the same kernel with a tile per WAVE and no split, ragged N and a K tail of 8 included;
    4096 x 16 x 4096 and 4096 x 48 x 4096: one and three k-tiles on the pinned body -- fewer tiles than LDS images, and
    the first in-loop DMA off the scalar base; 4352 x 1024 x 4352: 289 tiles, hybrid stream-K -- one whole round of
    tiles straight into C, the last 33 tiles as a stream over all workgroups.)"""
    rng = np.random.default_rng(SEED + 7 + 2 * ta + tb)
    a = rng.integers(-2, 3, size=(m, k)).astype(np.float32)
    b = rng.integers(-2, 3, size=(k, n)).astype(np.float32)
    da = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
    db = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
    got = T.gmul(1, 1, 1, da, db).numpy()
    want = (a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32)
    assert same(got, want, a=a, b=b, m=m, k=k, n=n, ta=ta, tb=tb)   # (a failure names tiles, waves and k-ranges)


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,k,n", [(768, 784, 768), (768, 790, 772), (500, 2056, 520), (512, 2048, 512), (384, 4096, 384),
                                   (256, 4100, 256), (132, 3000, 1028), (1024, 1024, 512), (704, 704, 704), (768, 4096, 768)])
def test_few_tiles_several_workgroups_per_tile_bit_exact_on_integers(T, ta, tb, m, k, n):
    """Round 4: fewer 64x64 tiles than CUs -- gemm_kwave.hip with KS = 2, 3, 4, 6 or 8 workgroups per tile (kw_ksplit), all
    of a tile's workgroups on one XCD, partial tiles through that XCD's L2, the last arriver adds them in k order:
    768 x 784 x 768 and 768 x 790 x 772 three ways (the second with a K tail of 6 and a ragged last tile column),
    500 x 2056 x 520 three ways with ragged tiles both ways and a K tail of 8, 512 x 2048 x 512 four ways, 384 x 4096 x 384
    six ways, 256 x 4100 x 256 eight ways, 132 x 3000 x 1028 (three tile rows, the last of four rows), 1024 x 1024 x 512 and
    704^3 two ways, 768 x 4096 x 768 three ways with two workgroups per CU.  Whole output, exact on small integers."""
    rng = np.random.default_rng(SEED + 77 + 2 * ta + tb)
    a = rng.integers(-2, 3, size=(m, k)).astype(np.float32)
    b = rng.integers(-2, 3, size=(k, n)).astype(np.float32)
    da = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
    db = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
    want = (a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32)
    for rep in range(3):   # (the counters must be back at zero for the next launch)
        l0 = T.stats()["launches"]
        got = T.gmul(1, 1, 1, da, db).numpy()
        assert T.stats()["launches"] - l0 == 1
        assert same(got, want, a=a, b=b, m=m, k=k, n=n, ta=ta, tb=tb)


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,k,n", [(1088, 1088, 1088), (1152, 2048, 1152), (1150, 2056, 1156), (1472, 1482, 1470), (1792, 1800, 1792),
                                   (1100, 1100, 1100)])
def test_more_tiles_than_cus_stream_k_bit_exact_on_integers(T, ta, tb, m, k, n):
    """Round 6: 257 .. 1024 tiles of 64x64 whose last round of the 256 CUs would be mostly empty -- gemm_kwave.hip as STREAM-K
    (kw_streamk): 512 workgroups, each an equal share of the stream "tile 0's k-tiles, tile 1's k-tiles, ...", one run of the
    wave-split K loop per tile a share touches, partial tiles handed over write-through and added in k order by the last
    contributor to arrive.  1088^3 (289 tiles), 1152 x 2048 x 1152 (324 tiles, a share is 81 of a tile's 128 k-tiles),
    1150 x 2056 x 1156 (ragged tiles both ways, a K tail of 8: added by the run that ends a tile), 1472 x 1482 x 1470 (529
    tiles, K tail of 10, ragged), 1792 x 1800 x 1792 (784 tiles), 1100^3.  Whole output, exact on small integers, three
    launches each (the counters must be back at zero for the next one)."""
    if (ta and m % 4) or (not tb and n % 4):
        pytest.skip("an m- / n-contiguous operand needs whole quads")
    rng = np.random.default_rng(SEED + 177 + 2 * ta + tb)
    a = rng.integers(-2, 3, size=(m, k)).astype(np.float32)
    b = rng.integers(-2, 3, size=(k, n)).astype(np.float32)
    da = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
    db = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
    want = (a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32)
    for rep in range(3):
        l0 = T.stats()["launches"]
        got = T.gmul(1, 1, 1, da, db).numpy()
        assert T.stats()["launches"] - l0 == 1
        assert same(got, want, a=a, b=b, m=m, k=k, n=n, ta=ta, tb=tb)


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,k,n", [(768, 768, 768), (768, 790, 772), (1280, 1280, 1280), (1280, 520, 1276), (768, 1024, 1024), (1024, 1040, 768),
                                   (640, 656, 640), (832, 840, 832), (768, 4096, 768), (1024, 1040, 512), (512, 1024, 1024),
                                   (1088, 1096, 1088), (1024, 1030, 1276), (1280, 1024, 1020)])
def test_tile_menu_of_16x16_blocks_bit_exact_on_integers(T, ta, tb, m, k, n):
    """Round 6: gemm_kw16.hip -- the wave-split design on the tile whose COUNT fits the 256 CUs, built from 16x16x4 MFMA blocks:
    768^3 = 256 tiles of 48x48 (144 of 64x64 split three ways before: 66 -> 86 TF), 1280^3 = 256 tiles of 80x80 (104 -> 118 TF),
    768 x K x 1024 = 256 tiles of 48x64 and 1024 x K x 768 of 64x48, 1024 x K x 512 = 256 tiles of 64x32 and 512 x K x 1024 of 32x64
    (82 -> 94 TF), 640^3 / 832^3 on 48x48, 1088^3 = 238 tiles of 64x80 (87 -> 97 TF), 1024 x K x 1280 = 256 of 64x80 and
    1280 x K x 1024 of 80x64 (94 -> 110 TF); with K tails of 6, 8 and 16 inside the
    kernel (a wave with an odd run of k-tiles ends on a zeroed ghost tile), ragged last tile columns, a long K.  Whole output,
    exact on small integers, three launches each, all four operand layouts."""
    if (ta and m % 4) or (not tb and n % 4):
        pytest.skip("an m- / n-contiguous operand needs whole quads")
    rng = np.random.default_rng(SEED + 277 + 2 * ta + tb)
    a = rng.integers(-2, 3, size=(m, k)).astype(np.float32)
    b = rng.integers(-2, 3, size=(k, n)).astype(np.float32)
    da = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
    db = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
    want = (a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32)
    for rep in range(3):
        l0 = T.stats()["launches"]
        got = T.gmul(1, 1, 1, da, db).numpy()
        assert T.stats()["launches"] - l0 == 1
        assert same(got, want, a=a, b=b, m=m, k=k, n=n, ta=ta, tb=tb)


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,k,n", [(10000, 300, 2048), (784, 300, 10000), (6144, 528, 4096), (5120, 520, 5120), (12288, 256, 1024),
                                   (4000, 264, 4000), (2052, 1030, 10000)])
def test_more_than_1024_tiles_the_big_tiles_do_not_fit_bit_exact_on_integers(T, ta, tb, m, k, n):
    """Round 6, last (tools/gemm_scan.py): more than 1,024 tiles of 64x64 whose extents the 256x256 tiles do not fit -- a ragged
    M or N (10000, 784), a K that is no multiple of 16 or below 512, one and a half rounds of big tiles (6144 x K x 4096) -- go to
    gemm_kwave.hip, a tile per WAVE (K < 512; two or more rounds of 2,048 tiles without a mostly empty last one) or per
    workgroup (5120 x K x 5120 = 25 rounds of 256 exactly).  Whole output, exact on small integers, one launch, all four layouts."""
    if (ta and m % 4) or (not tb and n % 4):
        pytest.skip("an m- / n-contiguous operand needs whole quads")
    rng = np.random.default_rng(SEED + 377 + 2 * ta + tb)
    a = rng.integers(-2, 3, size=(m, k)).astype(np.float32)
    b = rng.integers(-2, 3, size=(k, n)).astype(np.float32)
    da = T.transp(T.put(np.ascontiguousarray(a.T))) if ta else T.put(a)
    db = T.transp(T.put(np.ascontiguousarray(b.T))) if tb else T.put(b)
    want = a @ b   # (exact in fp32 too: integers below 2^24)
    l0 = T.stats()["launches"]
    got = T.gmul(1, 1, 1, da, db).numpy()
    assert T.stats()["launches"] - l0 == 1
    assert np.array_equal(got, want)


@pytest.mark.parametrize("batched_b", [True, False])
def test_full_tile_kernel_with_a_hidden_batch(T, batched_b):
    """The same kernel under a hidden batch (blockIdx.z walks the samples; 16 x (1024/256)^2 = 256 tiles):
    per-sample products of integer data, bit-exact."""
    rng = np.random.default_rng(SEED + 21)
    B, m, k, n = 16, 1024, 304, 1024
    a = rng.integers(-2, 3, size=(B, m, k)).astype(np.float32)
    b = rng.integers(-2, 3, size=((B, k, n) if batched_b else (k, n))).astype(np.float32)
    got = T.gmul(1, 1, 1, T.put(a, batched=True), T.put(b, batched=batched_b))
    assert got.batch == B
    want = np.matmul(a.astype(np.float64), b.astype(np.float64)).astype(np.float32)
    assert np.array_equal(got.numpy(), want)


def test_c5_stays_on_the_streaming_kernel(T):
    """A route guard with a clock on it: config 5a ('[512,512,64] x '[64,512]) takes 0.15 ms on gemm_skinnyk3_kernel and 0.22 ms on
    the wave-split kernel, whose widened rules (round 6, last) would accept it -- run_gemm asks the streaming kernel first.  The
    bound is loose (0.19 ms) and taken as the best of three timed batches."""
    a = T.genRand((512, 512, 64), "uniform", -1.0, 1.0, SEED + 91)
    b = T.genRand((64, 512), "uniform", -1.0, 1.0, SEED + 92)
    for _ in range(100):
        T.gmul(2, 1, 1, a, b)
    best = 1e9
    for _ in range(3):
        T.sync(); T.timer_start()
        for _ in range(100):
            T.gmul(2, 1, 1, a, b)
        best = min(best, T.timer_stop() / 100)
    assert best < 0.19, best


def test_c5_rank3_gmul_and_mapped_logistic(T):
    """config 5: gmul '[512,512,64] x '[64,512] then map logistic over the 512^3 result."""
    from tensor_ops_amd.hipt import logistic_closure
    rng = np.random.default_rng(SEED + 5)
    a = rng.uniform(-1, 1, size=(512, 512, 64)).astype(np.float32)
    b = rng.uniform(-1, 1, size=(64, 512)).astype(np.float32)
    c = T.gmul(2, 1, 1, T.put(a), T.put(b))
    assert c.shape == (512, 512, 512)
    l = T.liftT(T.expr(logistic_closure, 1, key="full_logi"), [c])
    ch = c.numpy()
    i = rng.choice(512, size=6, replace=False)
    want = np.einsum("xjk,kn->xjn", a[i].astype(np.float64), b.astype(np.float64))
    assert rel_err(ch[i], want) < RTOL
    lh = l.numpy()
    assert np.max(np.abs(lh[i] - 1 / (1 + np.exp(-want)))) < 2e-6
    assert lh.min() > 0.0 and lh.max() < 1.0
    # checksum of checksums: sum over everything vs fp64 of the downloaded GEMM result
    tot = float(T.sumRows(T.sumRows(T.sumRows(l))).numpy())
    ref = float((1 / (1 + np.exp(-ch.astype(np.float64)))).sum())
    assert abs(tot - ref) < 1e-5 * ref
    # the same two calls recorded in one fusion scope: ONE launch (logistic in the GEMM's epilogue, C stored
    # once), same values
    st = T.stats()["launches"]
    dA, dB = T.put(a), T.put(b)
    with T.memo():
        lf = T.force(T.liftT(T.expr(logistic_closure, 1, key="full_logi"), [T.gmul(2, 1, 1, dA, dB)]))
    assert T.stats()["launches"] - st == 1
    lfh = lf.numpy()
    assert np.max(np.abs(lfh[i] - 1 / (1 + np.exp(-want)))) < 2e-6
    assert np.max(np.abs(lfh - lh)) < 2e-6


def _c3(rank=0, batch=1024):
    import bench
    return bench.synth(rank, batch)


def _flat_grads(tr):
    from tensor_ops_amd import capi
    _, g_ptr, n = tr.flat()
    flat = np.empty(n, dtype=np.float32)
    h = capi.c_tensor()
    d = (C.c_int64 * 1)(n)
    capi.check(capi.lib().to_wrap(C.c_void_p(g_ptr), 0, 1, d, 0, C.byref(h)))
    capi.check(capi.lib().to_download(h, flat.ctypes.data_as(C.c_void_p), flat.nbytes))
    capi.lib().to_release(h)
    return flat


def _split(flat, shapes):
    out, off = [], 0
    for s in shapes:
        n = int(np.prod(s))
        out.append(flat[off:off + n].reshape(s))
        off += (n + 3) // 4 * 4
    return out


SHAPES = [(256, 784), (256,), (10, 256), (10,)]


@pytest.mark.parametrize("fused", [True, False])
def test_c3_full_batch_step_against_c_oracle(T, fused):
    """config 3: ffLayer 784->256->10 batched gradTOp at B = 1024 vs the per-sample C oracle."""
    from oracle import hmat
    from tensor_ops_amd import tops
    ws, X, Y = _c3()
    want, _ = hmat.batched_grads(X, Y, ws[0][0], ws[0][1], ws[1][0], ws[1][1], recompute=False)
    net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
    tr = tops.Trainer(net, "crossEntropy", 0.02, T.put(X, batched=True), T.put(Y, batched=True),
                      use_fused=fused)
    tr.grad()
    if fused:
        # forward, loss head + dz_1, and ONE paired launch of the two weight gradients
        assert tr.launches_per_step == 3
    for g, w in zip(_split(_flat_grads(tr), SHAPES), want):
        assert rel_err(g, w) < RTOL
    tr.apply()
    for p, w0, g in zip(tr.net.params, [ws[0][0], ws[0][1], ws[1][0], ws[1][1]], want):
        assert rel_err(p.numpy(), w0 - 0.02 * g) < RTOL


@pytest.mark.parametrize("graph", [True, False])
def test_c3_steps_on_distinct_batches_through_trainers_that_share_one_parameter_buffer(T, graph):
    """bench.py's `steady_state.distinct_batches` leg and its headline both run trainers on caller-owned flat buffers
    (`ext_params` / `ext_grads`); the leg has eight of them, one resident batch each, taking turns on ONE parameter buffer.
    Here: four batches of 1024 distinct rows, eight steps round robin (each trainer replaying its own captured step, or
    issuing it directly), against the per-sample C oracle stepping the same parameters through the same batches in the same
    order -- `foldl' trainBatch` with a different batch every step (FeedForward.hs:131-148 on a batch), 1e-5."""
    from oracle import hmat
    from tensor_ops_amd import tops
    ws, _, _ = _c3()
    batches = [_c3(100 + i)[1:] for i in range(4)]
    rate = 0.5 / 1024
    net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
    nflat = tops.Trainer.flat_size(net)
    flat_p, flat_g = T.konst((nflat,), 0.0), T.konst((nflat,), 0.0)     # library-owned here; torch-owned in bench.py
    T.sync()
    trs = [tops.Trainer(net, "crossEntropy", rate, T.put(X, batched=True), T.put(Y, batched=True), use_graph=graph,
                        ext_params=flat_p.ptr, ext_grads=flat_g.ptr) for X, Y in batches]
    # (every trainer copies the initial parameters into the shared buffer when it is made: the same numbers four times)
    params = [np.asarray(a, dtype=np.float64) for a in (ws[0][0], ws[0][1], ws[1][0], ws[1][1])]
    for k in range(8):
        X, Y = batches[k % 4]
        g, _ = hmat.batched_grads(X, Y, *params, recompute=False)
        params = [a - rate * np.asarray(gi, dtype=np.float64) for a, gi in zip(params, g)]
        trs[k % 4].step()
    T.sync()
    got = _split(flat_p.numpy(), SHAPES)
    for a, w in zip(got, params):
        assert rel_err(a, w) < RTOL
    # and all four trainers see the same parameters: their views are views of the one buffer
    for tr in trs[1:]:
        for a, b in zip(tr.net.params, trs[0].net.params):
            assert np.array_equal(a.numpy(), b.numpy())


def test_c4_shard_sum_equals_full_batch(T):
    """config 4 on one GPU: the sum of the 8 per-shard gradients (what the all-reduce forms)
    equals the single-device batch-8192 gradient to 1e-5 (summation order differs)."""
    from tensor_ops_amd import tops
    import bench
    ws, _, _ = bench.synth(0, 1)
    shards = [bench.synth(r, 1024)[1:] for r in range(8)]
    net = tops.genNet([(T.put(w), T.put(b)) for w, b in ws], "actMapLogistic", "actSoftmax")
    acc = None
    for X, Y in shards:
        tr = tops.Trainer(net, "crossEntropy", 0.02, T.put(X, batched=True), T.put(Y, batched=True),
                          use_graph=False)
        tr.grad()
        g = _flat_grads(tr).astype(np.float64)
        acc = g if acc is None else acc + g
        del tr
    Xf = np.concatenate([s[0] for s in shards])
    Yf = np.concatenate([s[1] for s in shards])
    tr = tops.Trainer(net, "crossEntropy", 0.02, T.put(Xf, batched=True), T.put(Yf, batched=True),
                      use_graph=False)
    tr.grad()
    full = _flat_grads(tr)
    for a, b in zip(_split(acc, SHAPES), _split(full, SHAPES)):
        assert rel_err(b, a) < RTOL


@pytest.mark.parametrize("dt", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("m,k,n", [(1024, 1024, 1024), (768, 768, 768), (1088, 1088, 1088), (1024, 8200, 1024), (10000, 300, 2048), (100, 8192, 300),
                                   (512, 2048, 512), (1000, 1000, 1000)])
def test_gemm_with_beta_c_on_the_wave_split_routes(dt, m, k, n):
    """`gemm alpha a b (Just (beta, c))` (BLAS.hs:117-123) -- and the planner's `W - r dW` -- on the routes of gemm_kwave.hip /
    gemm_kw16.hip / gemm_kwave_f64.hip (a workgroup per tile, several per tile, stream-K, a wave per tile, the tile menu): beta * C
    joins the sum in the kernel's final reduction (round 6, last; before, a beta sent the product to the compiler-scheduled
    bodies).  Exact on small integers; one launch."""
    from tensor_ops_amd.hipb import HipB
    B = HipB(0, dtype=dt)
    rng = np.random.default_rng(m + k + n)
    a = rng.integers(-2, 3, (m, k)).astype(dt); b = rng.integers(-2, 3, (k, n)).astype(dt); c = rng.integers(-5, 6, (m, n)).astype(dt)
    da, db, dc = B.T.put(a), B.T.put(b), B.T.put(c)
    l0 = B.T.stats()["launches"]
    got = B.gemm(2.0, da, db, (-3.0, dc)).numpy()
    assert B.T.stats()["launches"] - l0 <= (1 if dt == np.float32 else 3)
    assert np.array_equal(got.astype(np.float64), 2.0 * (a.astype(np.float64) @ b.astype(np.float64)) - 3.0 * c)
