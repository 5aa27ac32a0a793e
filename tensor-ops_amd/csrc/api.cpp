// extern "C" entry points (include/tensorops_hip.h) and the host-side planning
// that turns `class Tensor` / `class BLAS` calls into kernel launches.
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>

#include "ops.hpp"

struct to_graph_s {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  std::vector<to_tensor> kept;  // every tensor created while capturing stays reserved
  // the captured step as a list of kernel launches, when that is all it consists of (see common.hpp, LaunchRec)
  std::vector<std::unique_ptr<to::LaunchRec>> launches;
  bool replay_list = false;
  std::vector<to::StepDesc> desc;  // what the planner made of the captured step (to_graph_online_sgd)
};

namespace to {

static thread_local std::string g_err;
static std::vector<std::unique_ptr<LaunchRec>> g_capture_launches;  // of the capture in progress
static std::vector<StepDesc> g_capture_desc;

static uint64_t bits(double d) {
  uint64_t u;
  std::memcpy(&u, &d, 8);
  return u;
}

static to_tensor track(to_tensor t) { return t; }  // capture bookkeeping lives in new_tensor/new_view

hipStream_t S() { return rt().stream; }

void require_init() { TO_CHECK(rt().inited, TO_ERR_STATE, "to_init has not been called"); }

void no_capture(const char* what) {
  TO_CHECK(!rt().capturing, TO_ERR_STATE, std::string(what) + " is not allowed during graph capture");
}

// ---- group collapsing ----------------------------------------------------------------------
struct Group {
  int64_t extent = 1;
  int64_t stride = 0;
  bool ok = true;
};

static Group collapse(int n, const int64_t* dims, const int64_t* strides) {
  Group g;
  bool have = false;
  int64_t last_stride = 0;
  for (int i = 0; i < n; ++i) {
    if (dims[i] == 1) continue;
    if (have && last_stride != strides[i] * dims[i]) g.ok = false;
    last_stride = strides[i];
    g.extent *= dims[i];
    have = true;
  }
  for (int i = 0; i < n; ++i)
    if (dims[i] == 0) g.extent = 0;
  g.stride = have ? last_stride : 0;
  return g;
}

// A sub-block of the output (rows [r0, r0+m), columns [c0, c0+n)) as a problem of its own.
static GemmProblem gemm_block(const GemmProblem& p, int64_t r0, int64_t m, int64_t c0, int64_t n) {
  GemmProblem q = p;
  const int64_t es = p.dtype == TO_F64 ? 8 : 4;
  auto adv = [es](const void* ptr, int64_t elems) -> const void* {
    return ptr ? static_cast<const void*>(static_cast<const char*>(ptr) + elems * es) : nullptr;
  };
  q.M = m; q.N = n;
  q.A = adv(p.A, r0 * p.a_sm);
  q.B = adv(p.B, c0 * p.b_sn);
  q.C = const_cast<void*>(adv(p.C, r0 * p.c_sm + c0));
  q.Cin = adv(p.Cin, r0 * p.c_sm + c0);
  q.bias = adv(p.bias, c0);
  q.dact = adv(p.dact, r0 * p.c_sm + c0);
  return q;
}

// Which kernel run_gemm hands a (non-empty, unsplit) problem to.
enum GemmRoute { ROUTE_SMALL, ROUTE_MFMA, ROUTE_F64, ROUTE_NAIVE };
static GemmRoute gemm_route(const GemmProblem& p) {
  const bool sliver = p.K >= 256 && p.M * p.N >= 256 && gemm_small_can(p) &&
                      ((p.M + 15) / 16) * ((p.N + 15) / 16) * p.batch <= 4096;
  if (gemm_mfma_worthwhile(p) && gemm_small_applicable(p)) return ROUTE_SMALL;  // few tiles, long K
  if (gemm_mfma_worthwhile(p) && (p.reduce_batch || p.batch <= 65535)) return p.dtype == TO_F64 ? ROUTE_F64 : ROUTE_MFMA;
  // a sliver (fewer than 8 rows or columns) with a long K -- e.g. the border strip of a split: one thread per
  // output element would walk K serially (4 x 4096 x 4096: 0.95 ms); the small-GEMM kernel splits K
  if (sliver) return ROUTE_SMALL;
  return ROUTE_NAIVE;
}

bool gemm_small_route(const GemmProblem& p) {
  if (p.M == 0 || p.N == 0 || p.K == 0 || p.batch != 1 || p.reduce_batch) return false;
  if (gemv_form(p)) return false;   // (a large matVec / vecMat / outer product: gemv.hip, through run_gemm)
  // (a few tiles under a very long K -- a weight gradient over a data set, 300 x 60000 x 784: one workgroup a tile here whatever K
  //  is, 400 us; 291 split over workgroups in gemm_kwave.hip, which carries alpha / beta C / bias / activation but no row sums)
  if (p.K >= 8192 && !p.rowsum && !p.loss_rows && !p.tail_out && gemm_kw_long_k(p)) return false;
  const int64_t t64 = ((p.M + 63) / 64) * ((p.N + 63) / 64);
  return gemm_small_applicable(p) || (t64 < 200 && gemm_small_can(p));
}

// The fused elementwise epilogue (alpha, beta*Cin, bias, act, dact) exists in the small-GEMM kernel (both element
// types), the tiled fp32 kernel and the one-thread-per-element fallback (where border strips and K tails of a problem
// run_gemm splits may land) and both wave-split kernels (gemm_kwave*.hip); the tiled fp64 kernel has alpha/beta only.
// (lazy.cpp launches the small-GEMM kernel itself when gemm_small_route holds, and run_gemm otherwise.)
bool gemm_epilogue_ok(const GemmProblem& p) {
  if (p.M == 0 || p.N == 0 || p.K == 0 || p.batch == 0) return false;
  if (gemm_small_route(p) || gemm_skinnyk_applicable(p) || gemm_skinnyk64_applicable(p) || gemv_form(p)) return true;
  if (p.dtype == TO_F64) return gemm_kw64_applicable(p);
  const GemmRoute r = gemm_route(p);
  return r == ROUTE_SMALL || r == ROUTE_MFMA;
}

void run_gemm(const GemmProblem& p) {
  if (p.M == 0 || p.N == 0 || p.batch == 0) return;
  TO_CHECK(p.M <= 2147483647LL && p.N <= 2147483647LL && p.K <= 2147483647LL, TO_ERR_SHAPE,
           "collapsed GEMM extent exceeds 2^31-1");
  // Large GEMMs whose extents are no multiple of the 256x256 (fp64: 256x128) tile, or whose tile count is no multiple of
  // the 256 CUs: the fast full-tile kernel gets the largest block of (nearly) WHOLE ROUNDS of tiles, the two border
  // strips go their own way (smaller tiles, split-K, the small-GEMM kernel).  A ragged last round of big tiles
  // costs a whole round: 4100x4096x4096 took 2.38 ms against 0.95 ms for 4096^3.
  // A K that is no multiple of the 16-deep k-tile puts EVERY tile on the guarded (bounds-checked, scalar-load) path:
  // 1000^3 ran at 25 TF.  Run the multiple-of-16 part unguarded and add the K tail (< 16) in a second, tiny launch
  // (C = alpha A2.B2 + 1 C).  Only for linear epilogues; the summation order changes within the 1e-5 bar.
  // a few hundred 64x64 tiles: the K loop split over the waves of each tile's workgroup (K tails included)
  // one extent is 1: matVec / vecMat / an outer product beyond the small-GEMM kernel's range -- HBM-bound streaming kernels
  if (gemv_form(p, true)) {
    launch_gemv(p, S());
    return;
  }
  {   // (development builds, TOPS_T32_FIRST=1: the four-wave 32x32-tile kernel ahead of the wave-split one, for A/B runs)
    static const int t32_first = [] { const char* e = ab_getenv("TOPS_T32_FIRST"); return e ? atoi(e) : 0; }();
    if (t32_first && gemm_t32_applicable(p)) {
      launch_gemm_t32(p, S());
      return;
    }
  }
  // short K, B small enough to live in LDS, a long stream of rows (config 5): the barrier-free streaming kernel -- ahead of the
  // wave-split kernel, whose rule for "more than 1,024 tiles the big tiles do not fit" would take these too (0.150 -> 0.217 ms)
  if (gemm_skinnyk_applicable(p)) {
    launch_gemm_skinnyk(p, S());
    return;
  }
  if (gemm_skinnyk64_applicable(p)) {
    launch_gemm_skinnyk64(p, S());
    return;
  }
  // ... on the tile shape whose count fits the CUs: 48x48 / 48x64 / 64x48 / 80x80 where that beats the 64x64 routes (768^3: 256
  // tiles of 48x48 instead of 144 of 64x64 split three ways)
  if (gemm_kw16_applicable(p)) {
    launch_gemm_kw16(p, S());
    return;
  }
  if (gemm_kw_applicable(p)) {
    launch_gemm_kw(p, S());
    return;
  }
  if (gemm_kw64_applicable(p)) {
    launch_gemm_kw64(p, S());
    return;
  }
  if (p.K % 16 != 0 && p.K >= 128 && p.M * p.N >= 65536 && !p.reduce_batch && !p.rowsum && !p.loss_rows && p.act == 0 &&
      !p.dact) {
    const int64_t es = p.dtype == TO_F64 ? 8 : 4, K0 = p.K / 16 * 16;
    GemmProblem head = p, tail = p;
    head.K = K0;
    head.bias = nullptr;  // the bias belongs to the launch that finishes the element
    tail.K = p.K - K0;
    tail.A = static_cast<const char*>(p.A) + K0 * p.a_sk * es;
    tail.B = static_cast<const char*>(p.B) + K0 * p.b_sk * es;
    tail.Cin = p.C;
    tail.beta = 1.0;
    run_gemm(head);
    run_gemm(tail);
    return;
  }
  const bool f64 = p.dtype == TO_F64;
  auto full_rounds = [f64](const GemmProblem& q) { return f64 ? gemm_f64_w4_full_rounds(q) : gemm_w4_full_rounds(q); };
  const int64_t TNW = f64 ? 128 : 256;  // tile width (fp64: 256 x 128 tiles)
  if (!f64 && gemm_w4_edge_whole(p) && gemm_route(p) == ROUTE_MFMA) {  // ragged, but whole rounds of tiles with its edge tiles
    launch_gemm_mfma(p, S());
    return;
  }
  if (!p.reduce_batch && !p.rowsum && !p.loss_rows && p.M >= 256 && p.N >= 256 && !full_rounds(p)) {
    const int64_t tm = p.M / 256, tn = p.N / TNW;
    int64_t best = 0, bm = 0, bn = 0;
    for (int64_t dm = 0; dm < 8 && dm < tm; ++dm)
      for (int64_t dn = 0; dn < 8 && dn < tn; ++dn) {
        const int64_t t = (tm - dm) * (tn - dn) * p.batch;
        if (t > best && full_rounds(gemm_block(p, 0, (tm - dm) * 256, 0, (tn - dn) * TNW))) { best = t; bm = tm - dm; bn = tn - dn; }
      }
    // worth it when the block carries at least half of the work
    if (best > 0 && 2 * bm * bn * 256 * TNW >= p.M * p.N) {
      const GemmProblem main = gemm_block(p, 0, bm * 256, 0, bn * TNW);
      if (full_rounds(main)) {
        run_gemm(main);
        if (bn * TNW < p.N) run_gemm(gemm_block(p, 0, p.M, bn * TNW, p.N - bn * TNW));       // right strip
        if (bm * 256 < p.M) run_gemm(gemm_block(p, bm * 256, p.M - bm * 256, 0, bn * TNW));  // bottom strip
        return;
      }
    }
  }
  // Mid sizes whose extents are no multiple of 4 (multiples of 4 run whole on the pinned 128-tile kernel, edge tiles
  // included): the block of whole 128x128 tiles goes to that kernel, the two border strips their own way (slivers
  // with a long K: the small-GEMM kernel).  Otherwise every tile is on the guarded scalar-load path.
  if (!f64 && p.batch == 1 && !p.reduce_batch && !p.rowsum && !p.loss_rows && p.K % 16 == 0 && p.M >= 384 && p.N >= 384 &&
      (p.M % 4 != 0 || p.N % 4 != 0) && p.beta == 0.0 && p.alpha == 1.0 && !p.bias && !p.dact && p.act == 0) {
    const int64_t bm = p.M / 128, bn = p.N / 128;
    if (bm * bn >= 16 && bm * bn <= 288 && 4 * bm * bn * 128 * 128 >= 3 * p.M * p.N) {
      run_gemm(gemm_block(p, 0, bm * 128, 0, bn * 128));
      if (bn * 128 < p.N) run_gemm(gemm_block(p, 0, p.M, bn * 128, p.N - bn * 128));          // right strip
      if (bm * 128 < p.M) run_gemm(gemm_block(p, bm * 128, p.M - bm * 128, 0, bn * 128));     // bottom strip
      return;
    }
  }
  switch (gemm_route(p)) {
    case ROUTE_SMALL: launch_gemm_small(p, S()); break;  // latency-bound shapes: in-workgroup split-K, no LDS staging
    case ROUTE_MFMA: launch_gemm_mfma(p, S()); break;
    case ROUTE_F64:
      TO_CHECK(!p.bias && !p.act && !p.dact, TO_ERR_STATE, "internal: fused epilogue routed to the tiled fp64 kernel");
      launch_gemm_f64(p, S());
      break;
    default: launch_gemm_naive(p, S()); break;
  }
}

// a : ms++os, b : Reverse os ++ ns.  reduce: sum the result over the hidden batch.
// The batch rule is a LOWERING (round 4): called on batched data the reference's `gmul (transp x) dtdz` (TOp.hs:86-88)
// is one outer product PER SAMPLE, B*o*i numbers whose only use in a gradient is their sum over the batch.  In every
// mode -- inside a scope, outside one, TOPS_LAZY=0, TOPS_LAZY_FUSE=0 -- such a product is recorded, not computed
// (persample_outer below), what a cotangent passes through on its way to `batchSum` (sumT of `&&&`, scaleT) records behind
// it, and to_batch_sum lowers the lot to to_gmul_batch_sum: ONE GEMM with K = B.  Whatever else asks for its elements
// produces it then -- unless it is larger than TOPS_OUTER_MAX_BYTES (default 8 GiB): a 4096-row batch through a
// 4096 -> 4096 layer would be 275 GB, and a refusal that says what to call instead beats an allocation failure.
static bool persample_outer(int lo, to_tensor a, to_tensor b, bool reduce) {
  return !reduce && lo == 0 && a->batch > 0 && b->batch > 0 && a->rank + b->rank > 0;
}
static int64_t outer_max_bytes() {
  static const int64_t v = [] { const char* e = getenv("TOPS_OUTER_MAX_BYTES"); return e ? atoll(e) : (int64_t)8 << 30; }();
  return v;
}
// (called where a contraction is about to be COMPUTED -- eagerly or by the planner -- never where it is only recorded)
static void outer_size_guard(const GmulPlan& gp, int lo, to_tensor a, to_tensor b, bool reduce, bool dry) {
  if (dry || !persample_outer(lo, a, b, reduce)) return;
  int64_t bytes = (int64_t)(gp.dtype == TO_F64 ? 8 : 4) * std::max<int64_t>(gp.out_batch, 1);
  for (int i = 0; i < gp.out_rank; ++i) bytes *= gp.odims[i];
  TO_CHECK(bytes <= outer_max_bytes(), TO_ERR_UNSUPPORTED,
           "gmul: the per-sample outer products of " + std::to_string(gp.out_batch) + " samples are " + std::to_string(bytes) +
               " bytes; sum them over the batch (to_batch_sum of this value, or to_gmul_batch_sum) instead of asking "
               "for their elements, or raise TOPS_OUTER_MAX_BYTES");
}
// a batched value that exists only as its recorded op (and is not a view of one)
static bool pending_batched(to_tensor t) { return t->batch > 0 && !t->ptr && !t->view_base && t->node != nullptr; }

void gmul_plan(GmulPlan& gp, int lm, int lo, int ln, to_tensor a_in, to_tensor b_in, bool reduce, bool dry) {
  TO_CHECK(lm >= 0 && lo >= 0 && ln >= 0, TO_ERR_ARG, "negative Length");
  TO_CHECK(a_in->rank == lm + lo, TO_ERR_SHAPE,
           "gmul: first operand " + shape_str(a_in) + " is not ms++os with |ms|=" +
               std::to_string(lm) + " |os|=" + std::to_string(lo));
  TO_CHECK(b_in->rank == lo + ln, TO_ERR_SHAPE,
           "gmul: second operand " + shape_str(b_in) + " is not Reverse os ++ ns with |os|=" +
               std::to_string(lo) + " |ns|=" + std::to_string(ln));
  TO_CHECK(lm + ln <= TO_MAX_RANK, TO_ERR_SHAPE, "gmul: result rank > 8");
  for (int j = 0; j < lo; ++j)
    TO_CHECK(a_in->dims[lm + j] == b_in->dims[lo - 1 - j], TO_ERR_SHAPE,
             "gmul: contracted dims differ: " + shape_str(a_in) + " vs " + shape_str(b_in));
  TO_CHECK(a_in->batch == 0 || b_in->batch == 0 || a_in->batch == b_in->batch, TO_ERR_SHAPE,
           "gmul: operands carry different batch sizes");
  TO_CHECK(a_in->dtype == b_in->dtype, TO_ERR_ARG, "gmul: operands have different dtypes");

  Holder& ha = gp.ha;
  Holder& hb = gp.hb;  // possibly materialised operands
  to_tensor a = a_in, b = b_in;
  // shape-only stand-ins for operands the real plan would replace (dry runs never touch memory)
  to_tensor_s sa, sb;
  auto stand_in = [](to_tensor_s& s, to_tensor like, int64_t batch) {
    s.rank = like->rank;
    s.dtype = like->dtype;
    int64_t st = 1;
    for (int i = like->rank - 1; i >= 0; --i) {
      s.dims[i] = like->dims[i];
      s.strides[i] = st;
      st *= like->dims[i];
    }
    s.batch = batch;
    s.bstride = st;
  };

  // reduce with only one (or no) batched operand: sum that operand first
  if (reduce && !(a->batch > 0 && b->batch > 0)) {
    if (a->batch > 0) {
      if (dry) { stand_in(sa, a, 0); a = &sa; gp.exact = false; }
      else { ha.t = batch_sum_impl(a); a = ha.t; }
    } else if (b->batch > 0) {
      if (dry) { stand_in(sb, b, 0); b = &sb; gp.exact = false; }
      else { hb.t = batch_sum_impl(b); b = hb.t; }
    }
    reduce = false;
  }

  // K dims of B listed in A's order (o_1..o_q): o_j sits at B position lo-1-j
  int64_t bk_dims[TO_MAX_RANK], bk_str[TO_MAX_RANK];
  auto b_kgroup = [&](to_tensor bb) {
    for (int j = 0; j < lo; ++j) {
      bk_dims[j] = bb->dims[lo - 1 - j];
      bk_str[j] = bb->strides[lo - 1 - j];
    }
    return collapse(lo, bk_dims, bk_str);
  };
  Group gM = collapse(lm, a->dims, a->strides);
  Group gKa = collapse(lo, a->dims + lm, a->strides + lm);
  if (!gM.ok || !gKa.ok) {
    if (dry) {
      stand_in(sa, a, a->batch);
      a = &sa;
      gp.exact = false;
    } else {
      to_tensor c = contiguous(a);
      if (ha.t) release(ha.t);
      ha.t = c;
      a = c;
    }
    gM = collapse(lm, a->dims, a->strides);
    gKa = collapse(lo, a->dims + lm, a->strides + lm);
  }
  Group gKb = b_kgroup(b);
  Group gN = collapse(ln, b->dims + lo, b->strides + lo);
  if (!gKb.ok || !gN.ok) {
    // pack B as [o_1..o_q, ns] (K in A's order): permuted view, then a packed copy
    int64_t pd[TO_MAX_RANK], ps[TO_MAX_RANK];
    for (int j = 0; j < lo; ++j) {
      pd[j] = b->dims[lo - 1 - j];
      ps[j] = b->strides[lo - 1 - j];
    }
    for (int j = lo; j < lo + ln; ++j) {
      pd[j] = b->dims[j];
      ps[j] = b->strides[j];
    }
    to_tensor c;
    if (dry) {
      to_tensor_s like;
      like.rank = lo + ln;
      like.dtype = b->dtype;
      for (int j = 0; j < lo + ln; ++j) like.dims[j] = pd[j];
      stand_in(sb, &like, b->batch);
      c = &sb;
      gp.exact = false;
    } else {
      Holder view(new_view(b, lo + ln, pd, ps, b->batch, b->bstride, 0));
      c = contiguous(view.t);
      if (hb.t) release(hb.t);
      hb.t = c;
    }
    b = c;
    // c is laid out with K already in A's order: describe it directly
    gKb = collapse(lo, c->dims, c->strides);
    gN = collapse(ln, c->dims + lo, c->strides + lo);
  }

  const int64_t M = gM.extent, K = gKa.extent, N = gN.extent;
  const int64_t Ba = a->batch, Bb = b->batch;
  const int64_t B = Ba > 0 ? Ba : Bb;

  gp.out_rank = lm + ln;
  for (int i = 0; i < lm; ++i) gp.odims[i] = a_in->dims[i];
  for (int i = 0; i < ln; ++i) gp.odims[lm + i] = b_in->dims[lo + i];
  gp.out_batch = reduce ? 0 : B;
  gp.dtype = a->dtype;

  GemmProblem& p = gp.p;
  p = GemmProblem{};
  p.dtype = a->dtype;
  p.alpha = 1.0;
  p.beta = 0.0;
  p.Cin = nullptr;
  p.C = nullptr;
  p.A = a->ptr;
  p.B = b->ptr;
  p.M = M; p.N = N; p.K = K;
  p.a_sm = gM.stride; p.a_sk = gKa.stride;
  p.b_sk = gKb.stride; p.b_sn = gN.stride;
  p.c_sm = N;
  p.batch = 1;
  p.a_sb = p.b_sb = p.c_sb = 0;
  p.reduce_batch = 0;

  if (K == 0 || (reduce && B == 0)) {  // empty contraction: zeros (`sum' [] = 0`)
    gp.zero = true;
    return;
  }

  if (reduce) {
    // out[m,n] = sum_b sum_k A_b[m,k] B_b[k,n]: fold the batch into K when the strides allow
    int64_t d2[TO_MAX_RANK + 1], sa2[TO_MAX_RANK + 1], sb2[TO_MAX_RANK + 1];
    d2[0] = B; sa2[0] = a->bstride; sb2[0] = b->bstride;
    d2[1] = K; sa2[1] = gKa.stride; sb2[1] = gKb.stride;
    Group fa = collapse(2, d2, sa2), fb = collapse(2, d2, sb2);
    if (fa.ok && fb.ok) {
      p.K = B * K;
      p.a_sk = fa.stride;
      p.b_sk = fb.stride;
    } else {
      p.reduce_batch = 1;
      p.batch = B;
      p.a_sb = a->bstride;
      p.b_sb = b->bstride;
    }
    return;
  }

  if (Ba == 0 && Bb == 0) {
    // one GEMM.  A matrix-vector product is laid out as the ROW x^T.A^T ([1 x M], like one sample of the batched
    // form below): the same flops, and the epilogues that work along a row -- bias, loss head -- apply to the
    // reference's own per-sample calls (app/MNIST.hs:390-396 trains sample by sample on unbatched tensors)
    if (N == 1 && ln == 0 && M > 1) {
      GemmProblem q = p;
      q.A = b->ptr; q.M = 1; q.a_sm = 0; q.a_sk = gKb.stride;
      q.B = a->ptr; q.N = M; q.b_sk = gKa.stride; q.b_sn = gM.stride;
      q.c_sm = M;
      p = q;
      gp.rows_are_samples = true;
    }
  } else if (Ba > 0 && Bb == 0) {
    int64_t d2[2] = {B, M}, s2[2] = {a->bstride, gM.stride};
    Group f = collapse(2, d2, s2);
    if (f.ok) {  // [B;M,K] x [K,N]: one GEMM with M' = B*M
      p.M = B * M;
      p.a_sm = f.stride;
      gp.rows_are_samples = M == 1;
    } else {
      p.batch = B; p.a_sb = a->bstride; p.b_sb = 0; p.c_sb = M * N;
    }
  } else if (Ba == 0 && Bb > 0) {
    int64_t d2[2] = {B, N}, s2[2] = {b->bstride, gN.stride};
    Group f = collapse(2, d2, s2);
    if (N == 1) {
      // matVec with a batched vector: C[b,m] = sum_k X[b,k] A^T[k,m]  (one GEMM, M' = B)
      GemmProblem q = p;
      q.A = b->ptr; q.M = B; q.a_sm = b->bstride; q.a_sk = gKb.stride;
      q.B = a->ptr; q.N = M; q.b_sk = gKa.stride; q.b_sn = gM.stride;
      q.c_sm = M;
      p = q;
      gp.rows_are_samples = true;
    } else if (M == 1 && f.ok) {  // vecMat against a batch folded into N
      p.N = B * N;
      p.b_sn = f.stride;
      p.c_sm = B * N;
    } else {
      p.batch = B; p.a_sb = 0; p.b_sb = b->bstride; p.c_sb = M * N;
    }
  } else {
    p.batch = B; p.a_sb = a->bstride; p.b_sb = b->bstride; p.c_sb = M * N;
  }
  outer_size_guard(gp, lo, a_in, b_in, reduce, dry);
}

to_tensor gmul_impl(int lm, int lo, int ln, to_tensor a_in, to_tensor b_in, bool reduce) {
  GmulPlan gp;
  gmul_plan(gp, lm, lo, ln, a_in, b_in, reduce, false);
  Holder hout(new_tensor(gp.out_rank, gp.odims, gp.out_batch, gp.dtype));
  if (gp.zero) {
    launch_fill(hout.t->dtype, hout.t->ptr, hout.t->total(), 0.0, S());
    return hout.take();
  }
  gp.p.C = hout.t->ptr;
  run_gemm(gp.p);
  return hout.take();
}

// ---- elementwise helpers --------------------------------------------------------------------
void lift_check(to_expr f, int n, const to_tensor* xs_in, int64_t* batch, int* dtype_out) {
  TO_CHECK(f != nullptr, TO_ERR_ARG, "null expression");
  TO_CHECK(n == f->arity, TO_ERR_ARG,
           "liftT: expression arity " + std::to_string(f->arity) + " != " + std::to_string(n) + " inputs");
  int64_t B = 0;
  const int dtype = n > 0 ? xs_in[0]->dtype : *dtype_out;
  for (int i = 0; i < n; ++i) {
    TO_CHECK(xs_in[i] != nullptr, TO_ERR_ARG, "null tensor");
    TO_CHECK(xs_in[i]->dtype == dtype, TO_ERR_ARG, "liftT: inputs have different dtypes");
    TO_CHECK(same_shape(xs_in[0], xs_in[i]), TO_ERR_SHAPE,
             "liftT: shapes differ: " + shape_str(xs_in[0]) + " vs " + shape_str(xs_in[i]));
    if (xs_in[i]->batch > 0) {
      TO_CHECK(B == 0 || B == xs_in[i]->batch, TO_ERR_SHAPE, "liftT: different batch sizes");
      B = xs_in[i]->batch;
    }
  }
  *batch = B;
  *dtype_out = dtype;
}

to_tensor lift_impl(to_expr f, int n, const to_tensor* xs_in, int rank_hint, const int64_t* dims_hint,
                    int dtype_hint) {
  int64_t B = 0;
  int dtype = dtype_hint;
  lift_check(f, n, xs_in, &B, &dtype);
  std::vector<Holder> hold(n);
  EwArgs a{};
  a.kind = f->kind;
  a.n = n;
  to_tensor out = n > 0 ? new_tensor(xs_in[0]->rank, xs_in[0]->dims, B, dtype)
                        : new_tensor(rank_hint, dims_hint, 0, dtype);
  Holder hout(out);
  a.dtype = dtype;
  a.out = out->ptr;
  a.total = out->total();
  for (int i = 0; i < n; ++i) {
    hold[i].t = contiguous(xs_in[i]);
    a.x[i] = hold[i].t->ptr;
    a.period[i] = (B > 0 && hold[i].t->batch == 0) ? hold[i].t->numel() : a.total;
    if (a.period[i] == 0) a.period[i] = 1;
  }
  for (int i = 0; i < 4; ++i) a.coef[i] = f->coef_d[i];
  a.c0 = f->c0_d;
  if (f->kind == EW_VM) expr_prepare(f, dtype);  // builds the specialised kernel / VM tables on first use
  a.d_code = f->d_code;
  a.d_consts = dtype == TO_F64 ? (const void*)f->d_consts_f64 : (const void*)f->d_consts_f32;
  a.n_instr = (int)(f->vm_code.size() / 4);
  a.n_slots = f->n_slots;
  a.result_slot = f->result_slot;
  a.jit = f->jit[dtype == TO_F64 ? 1 : 0];
  launch_ewise(a, S());
  return hout.take();
}

// the same launch on raw ranges: `total` elements of every operand, all full-size (a batch of sibling lifts whose operands
// lie one behind the other in memory, lazy.cpp)
void lift_launch_raw(to_expr f, int n, const void* const* xs, void* out, int64_t total, int dtype) {
  EwArgs a{};
  a.kind = f->kind;
  a.n = n;
  a.dtype = dtype;
  a.out = out;
  a.total = total;
  for (int i = 0; i < n; ++i) {
    a.x[i] = xs[i];
    a.period[i] = total > 0 ? total : 1;
  }
  for (int i = 0; i < 4; ++i) a.coef[i] = f->coef_d[i];
  a.c0 = f->c0_d;
  if (f->kind == EW_VM) expr_prepare(f, dtype);
  a.d_code = f->d_code;
  a.d_consts = dtype == TO_F64 ? (const void*)f->d_consts_f64 : (const void*)f->d_consts_f32;
  a.n_instr = (int)(f->vm_code.size() / 4);
  a.n_slots = f->n_slots;
  a.result_slot = f->result_slot;
  a.jit = f->jit[dtype == TO_F64 ? 1 : 0];
  launch_ewise(a, S());
}

to_tensor affine_impl(int n, const to_tensor* xs, const double* coef, double c) {
  to_expr_s e;
  e.arity = n;
  e.kind = EW_AFFINE;
  for (int i = 0; i < 4; ++i) e.coef_d[i] = i < n ? coef[i] : 0.0;
  e.c0_d = c;
  return lift_impl(&e, n, xs, 0, nullptr);
}

to_tensor kind_impl(int kind, int n, const to_tensor* xs) {
  to_expr_s e;
  e.arity = n;
  e.kind = kind;
  return lift_impl(&e, n, xs, 0, nullptr);
}

void map_rows_const_check(int len_n, to_tensor row, to_tensor like) {
  TO_CHECK(len_n >= 0 && len_n <= like->rank, TO_ERR_SHAPE, "mapRows: bad Length");
  TO_CHECK(row->rank == like->rank - len_n, TO_ERR_SHAPE,
           "mapRows: row " + shape_str(row) + " does not fit under " + shape_str(like));
  for (int i = 0; i < row->rank; ++i)
    TO_CHECK(row->dims[i] == like->dims[len_n + i], TO_ERR_SHAPE,
             "mapRows: row " + shape_str(row) + " does not fit under " + shape_str(like));
  TO_CHECK(row->batch == 0 || like->batch == 0 || row->batch == like->batch, TO_ERR_SHAPE,
           "mapRows: different batch sizes");
  TO_CHECK(row->dtype == like->dtype, TO_ERR_ARG, "mapRows: different dtypes");
}

// every ms-slice under `like`'s leading len_n dims := row (only like's SHAPE is read)
to_tensor map_rows_const_impl(int len_n, to_tensor row, to_tensor like) {
  Holder hr(contiguous(row));
  const int64_t B = row->batch > 0 ? row->batch : like->batch;
  Holder r(new_tensor(like->rank, like->dims, B, row->dtype));
  int64_t R = 1;
  for (int i = 0; i < len_n; ++i) R *= like->dims[i];
  const int64_t J = hr.t->numel();
  launch_bcast_axis(row->dtype, hr.t->ptr, r.t->ptr, B > 0 ? B : 1, R, J, hr.t->batch > 0 ? J : 0, S());
  return r.take();
}

void stack_check(int rank_m, const int64_t* dims_m, const to_tensor* rows, int64_t* nrows_out, int64_t* odims,
                 int64_t* batch) {
  TO_CHECK(rank_m >= 0 && rank_m <= TO_MAX_RANK, TO_ERR_ARG, "stack: bad rank");
  TO_CHECK(rank_m == 0 || dims_m, TO_ERR_ARG, "null argument: dims_m");
  int64_t nrows = 1;
  for (int i = 0; i < rank_m; ++i) {
    TO_CHECK(dims_m[i] >= 0, TO_ERR_ARG, "stack: negative extent");
    nrows *= dims_m[i];
  }
  TO_CHECK(nrows >= 1, TO_ERR_UNSUPPORTED, "stack of zero rows needs the row shape");
  TO_CHECK(rows != nullptr, TO_ERR_ARG, "null argument: rows");
  for (int64_t r = 0; r < nrows; ++r) {
    TO_CHECK(rows[r] != nullptr, TO_ERR_ARG, "null argument: rows[r]");
    TO_CHECK(same_shape(rows[0], rows[r]) && rows[0]->batch == rows[r]->batch, TO_ERR_SHAPE,
             "stack: rows differ in shape");
    TO_CHECK(rows[r]->dtype == rows[0]->dtype, TO_ERR_ARG, "stack: different dtypes");
  }
  TO_CHECK(rank_m + rows[0]->rank <= TO_MAX_RANK, TO_ERR_SHAPE, "stack: result rank > 8");
  for (int i = 0; i < rank_m; ++i) odims[i] = dims_m[i];
  for (int i = 0; i < rows[0]->rank; ++i) odims[rank_m + i] = rows[0]->dims[i];
  *nrows_out = nrows;
  *batch = rows[0]->batch;
}

// rows of a `mapRows` / `ixRows` traversal -> one tensor, 16 rows per launch
to_tensor stack_impl(int rank_m, const int64_t* dims_m, const to_tensor* rows) {
  int64_t nrows = 0, d[TO_MAX_RANK], B = 0;
  stack_check(rank_m, dims_m, rows, &nrows, d, &B);
  Holder o(new_tensor(rank_m + rows[0]->rank, d, B, rows[0]->dtype));
  const int64_t rowsz = rows[0]->numel(), w = (int64_t)rows[0]->esize() / 4;
  if (rowsz == 0 || o.t->total() == 0) return o.take();
  for (int64_t r0 = 0; r0 < nrows; r0 += 16) {
    const int m = (int)std::min<int64_t>(16, nrows - r0);
    std::vector<std::unique_ptr<Holder>> keep;
    const void* src[16];
    int64_t sb[16];
    for (int k = 0; k < m; ++k) {
      to_tensor x = rows[r0 + k];
      if (!x->inner_contiguous()) {
        keep.emplace_back(new Holder(contiguous(x)));
        x = keep.back()->t;
      }
      src[k] = x->ptr;
      sb[k] = (x->batch > 1 ? x->bstride : 0) * w;
    }
    launch_stack_rows(m, src, sb, o.t->at(r0 * rowsz), B > 0 ? B : 1, rowsz * w, nrows * rowsz * w, S());
  }
  return o.take();
}

to_tensor sum_impl(int n, const to_tensor* xs, int rank, const int64_t* dims, int dtype0) {
  if (n == 0) {
    to_tensor out = new_tensor(rank, dims, 0, dtype0);
    try {
      launch_fill(out->dtype, out->ptr, out->total(), 0.0, S());
    } catch (...) {
      release(out);
      throw;
    }
    return out;
  }
  if (n == 1) {
    retain(xs[0]);
    return xs[0];
  }
  // left fold, up to 4 operands per pass: ((x0+x1)+x2)+x3 ... same association as foldl1'
  const double ones[4] = {1, 1, 1, 1};
  Holder acc;
  int i = 0;
  while (i < n) {
    to_tensor group[4];
    int g = 0;
    if (acc.t) group[g++] = acc.t;
    while (g < 4 && i < n) group[g++] = xs[i++];
    to_tensor r = affine_impl(g, group, ones, 0.0);
    if (acc.t) release(acc.t);
    acc.t = r;
  }
  return acc.take();
}

to_tensor transp_impl(to_tensor x) {
  int64_t d[TO_MAX_RANK], s[TO_MAX_RANK];
  for (int i = 0; i < x->rank; ++i) {
    d[i] = x->dims[x->rank - 1 - i];
    s[i] = x->strides[x->rank - 1 - i];
  }
  return new_view(x, x->rank, d, s, x->batch, x->bstride, 0);
}

to_tensor sum_rows_impl(to_tensor x_in) {
  TO_CHECK(x_in->rank >= 1, TO_ERR_SHAPE, "sumRows needs rank >= 1, got " + shape_str(x_in));
  Holder hx(contiguous(x_in));
  to_tensor x = hx.t;
  const int64_t R = x->dims[0];
  int64_t J = 1;
  for (int i = 1; i < x->rank; ++i) J *= x->dims[i];
  to_tensor out = new_tensor(x->rank - 1, x->dims + 1, x->batch, x->dtype);
  Holder hout(out);
  const int64_t O = x->batch > 0 ? x->batch : 1;
  launch_sum_axis(x->dtype, x->ptr, out->ptr, O, R, J, R * J, J, 1, S());
  return hout.take();
}

to_tensor batch_sum_impl(to_tensor x_in) {
  if (x_in->batch == 0) {
    retain(x_in);
    return x_in;
  }
  Holder hx(contiguous(x_in));
  to_tensor x = hx.t;
  to_tensor out = new_tensor(x->rank, x->dims, 0, x->dtype);
  Holder hout(out);
  launch_sum_axis(x->dtype, x->ptr, out->ptr, 1, x->batch, x->numel(), 0, x->numel(), 1, S());
  return hout.take();
}

static double read_scalar(to_tensor t, int64_t offset) {
  no_capture("reading a scalar back to the host");
  double v64 = 0.0;
  float v32 = 0.f;
  void* dst = t->dtype == TO_F64 ? (void*)&v64 : (void*)&v32;
  device_to_host(dst, t->at(offset), t->esize(), S());
  return t->dtype == TO_F64 ? v64 : (double)v32;
}

static void check_dtype(int dtype) { TO_CHECK(dtype == TO_F32 || dtype == TO_F64, TO_ERR_ARG, "unknown dtype"); }

}  // namespace to

using namespace to;

// HIP's current device is per OS thread and to_init selects it on the initialising thread only: a Haskell
// capability (or any second thread) calling in would otherwise allocate and load modules on device 0.
static void bind_device() {
  static thread_local int bound = -1;
  to::Runtime& r = to::rt();
  if (r.inited && bound != r.device) {
    (void)hipSetDevice(r.device);
    bound = r.device;
  }
}

// host time spent inside the library's entry points (to_api_time): what a host's own per-step cost is NOT
static int64_t g_api_calls = 0;
// TOPS_API_COUNT=1: calls and nanoseconds per entry point, printed when the process exits (diagnostic)
static std::map<std::string, std::pair<int64_t, int64_t>>& api_counts() {
  static std::map<std::string, std::pair<int64_t, int64_t>> m;
  return m;
}
static bool api_count_on() {
  static const bool on = [] {
    const char* e = ab_getenv("TOPS_API_COUNT");
    const bool v = e && atoi(e) != 0;
    if (v)
      atexit([] {
        for (auto& kv : api_counts())
          std::fprintf(stderr, "[api] %-28s %10lld calls %12.1f us\n", kv.first.c_str(), (long long)kv.second.first,
                       kv.second.second / 1e3);
      });
    return v;
  }();
  return on;
}
// (the time-stamp counter, not clock_gettime: two calls of the latter per entry point were 2 us of a 40 us step)
// (x86-64: an invariant TSC, synchronised across cores, is assumed -- every x86 server part of the last decade; elsewhere the
//  compiler's cycle counter where it has one (aarch64: cntvct), else the steady clock.  Calibrated once against steady_clock.)
#if defined(__x86_64__) || defined(__i386__)
static inline uint64_t api_ticks() { return __builtin_ia32_rdtsc(); }
#elif defined(__aarch64__)
static inline uint64_t api_ticks() {
  uint64_t v;
  asm volatile("mrs %0, cntvct_el0" : "=r"(v));
  return v;
}
#else
static inline uint64_t api_ticks() {
  return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#endif
static double api_ns_per_tick() {
  static const double v = [] {
    const auto c0 = std::chrono::steady_clock::now();
    const uint64_t t0 = api_ticks();
    while (std::chrono::steady_clock::now() - c0 < std::chrono::microseconds(200)) {
    }
    const uint64_t t1 = api_ticks();
    return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - c0).count() / (double)(t1 - t0);
  }();
  return v;
}
static uint64_t g_api_ticks = 0;
struct ApiClock {
  const char* fn;
  uint64_t t0 = api_ticks();
  explicit ApiClock(const char* f) : fn(f) {}
  ~ApiClock() {
    const uint64_t dt = api_ticks() - t0;
    g_api_ticks += dt;
    ++g_api_calls;
    if (api_count_on()) {
      auto& c = api_counts()[fn];
      c.first++;
      c.second += (int64_t)((double)dt * api_ns_per_tick());
    }
  }
};

#define API_BEGIN                                          \
  std::lock_guard<std::recursive_mutex> guard_(to::lock()); \
  ApiClock clock_(__func__);                                \
  bind_device();                                            \
  try {
#define API_END                          \
  return TO_OK;                          \
  }                                      \
  catch (const to::Error& e) {           \
    to::g_err = e.what();                \
    return e.code;                       \
  }                                      \
  catch (const std::exception& e) {      \
    to::g_err = e.what();                \
    return TO_ERR_ARG;                   \
  }
// like API_END but falls through on success
#define API_END_CHECK                    \
  }                                      \
  catch (const to::Error& e) {           \
    to::g_err = e.what();                \
    return e.code;                       \
  }
#define NONNULL(p) TO_CHECK((p) != nullptr, TO_ERR_ARG, "null argument: " #p)

extern "C" {

const char* to_last_error(void) { return to::g_err.c_str(); }

to_status to_init(int device) {
  API_BEGIN
  Runtime& r = rt();
  if (r.inited) {
    TO_CHECK(r.device == device, TO_ERR_STATE, "already initialised on another device");
    return TO_OK;
  }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  TO_CHECK(e == hipSuccess && n > 0, TO_ERR_HIP,
           "no HIP device visible: this backend has no CPU fallback (hipGetDeviceCount: " +
               std::string(hipGetErrorString(e)) + ")");
  TO_CHECK(device >= 0 && device < n, TO_ERR_ARG, "device index out of range");
  TO_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  TO_HIP(hipGetDeviceProperties(&prop, device));
  TO_CHECK(std::string(prop.gcnArchName).rfind("gfx950", 0) == 0, TO_ERR_UNSUPPORTED,
           std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);
  TO_HIP(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
  r.own_stream = true;
  TO_HIP(hipEventCreate(&r.ev0));
  TO_HIP(hipEventCreate(&r.ev1));
  r.device = device;
  r.inited = true;
  gemm_small_seam_init();
  gemm_kw_pair_init();
  gemm_t32_init();
  API_END
}

to_status to_build_info(int* ab_knobs) {
  API_BEGIN
  NONNULL(ab_knobs);
#ifdef TOPS_AB_KNOBS
  *ab_knobs = 1;
#else
  *ab_knobs = 0;
#endif
  API_END
}

to_status to_shutdown(void) {
  API_BEGIN
  Runtime& r = rt();
  if (!r.inited) return TO_OK;
  (void)hipStreamSynchronize(r.stream);
  scope_reset_all();
  for (auto& fl : r.free_lists) {
    for (void* p : fl) (void)hipFree(p);
    fl.clear();
  }
  r.pool_bytes = 0;
  if (r.own_stream && r.stream) (void)hipStreamDestroy(r.stream);
  if (r.ev0) (void)hipEventDestroy(r.ev0);
  if (r.ev1) (void)hipEventDestroy(r.ev1);
  comm_shutdown();
  p2p_shutdown();
  staging_shutdown();
  expr_shutdown();
  if (r.side) {
    (void)hipStreamSynchronize(r.side);
    (void)hipStreamDestroy(r.side);
    for (auto& e : r.fork_ev)
      if (e) (void)hipEventDestroy(e);
    if (r.join_ev) (void)hipEventDestroy(r.join_ev);
  }
  r = Runtime();
  API_END
}

to_status to_device_count(int* out) {
  API_BEGIN
  NONNULL(out);
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  *out = n;
  API_END
}

to_status to_set_stream(void* hip_stream) {
  API_BEGIN
  require_init();
  no_capture("to_set_stream");
  Runtime& r = rt();
  TO_HIP(hipStreamSynchronize(r.stream));
  if (r.own_stream) {
    (void)hipStreamDestroy(r.stream);
    r.own_stream = false;
  }
  if (hip_stream) {
    r.stream = static_cast<hipStream_t>(hip_stream);
  } else {
    TO_HIP(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
    r.own_stream = true;
  }
  API_END
}

to_status to_get_stream(void** out) {
  API_BEGIN
  require_init();
  NONNULL(out);
  *out = rt().stream;
  API_END
}

to_status to_sync(void) {
  API_BEGIN
  require_init();
  no_capture("to_sync");
  // waits for what has been ENQUEUED; demanding values is to_force / to_force_many (see scope_end, lazy.cpp)
  TO_HIP(hipStreamSynchronize(S()));
  TO_CHECK(gemm_small_chain_status() == 0, TO_ERR_HIP,
           "a grid barrier of a chained step launch timed out: the results of that step are invalid; set TOPS_STEP_CHAIN=0");
  TO_CHECK(gemm_t32_take_failure() == 0, TO_ERR_HIP,
           "the joined forward + loss-head launch gave up waiting for a row block's tiles (are CUs masked below one round of the "
           "grid?): the outputs of that launch are invalid; the joined form is off for the rest of this process (TOPS_STEP_SEAM=0 "
           "turns it off from the start)");
  TO_CHECK(gemm_small_seam_take_failure() == 0, TO_ERR_HIP,
           "the joined forward + loss-head launch (TOPS_STEP_SEAM) gave up waiting for a row block: the outputs of that launch "
           "are invalid; the seam is off for the rest of this process");
  API_END
}

to_status to_stats(int64_t* live_handles, int64_t* pool_bytes, int64_t* kernel_launches) {
  API_BEGIN
  if (live_handles) *live_handles = rt().live_handles;
  if (pool_bytes) *pool_bytes = rt().pool_bytes;
  if (kernel_launches) *kernel_launches = rt().launches;
  API_END
}

// ---- handles -------------------------------------------------------------------------------
to_status to_alloc(int dtype, int rank, const int64_t* dims, int64_t batch, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(out);
  check_dtype(dtype);
  TO_CHECK(rank == 0 || dims, TO_ERR_ARG, "null dims");
  *out = track(new_tensor(rank, dims, batch, dtype));
  API_END
}

to_status to_wrap(void* device_ptr, int dtype, int rank, const int64_t* dims, int64_t batch,
                  to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(out);
  NONNULL(device_ptr);
  check_dtype(dtype);
  TO_CHECK(rank >= 0 && rank <= TO_MAX_RANK, TO_ERR_ARG, "rank must be 0..8");
  auto* t = new to_tensor_s();
  t->rank = rank;
  t->dtype = dtype;
  int64_t n = 1;
  for (int i = 0; i < rank; ++i) {
    t->dims[i] = dims[i];
    n *= dims[i];
  }
  int64_t s = 1;
  for (int i = rank - 1; i >= 0; --i) {
    t->strides[i] = s;
    s *= t->dims[i];
  }
  t->batch = batch;
  t->bstride = n;
  auto* b = new Buffer();
  b->ptr = device_ptr;
  b->owned = false;
  t->buf = b;
  t->ptr = device_ptr;
  static std::atomic<uint64_t> wrap_id{1ull << 62};
  t->id = wrap_id++;
  rt().live_handles++;
  *out = t;
  API_END
}

// Reference counts are atomic and a handle's shape never changes: the three calls a host makes most often (a
// `ForeignPtr` copy, its finaliser, a shape query) take no lock.  The LAST reference is always given up under the
// lock: freeing a handle touches the recorded graph.
to_status to_retain(to_tensor t) {
  if (!t) {
    to::g_err = "null argument: t";
    return TO_ERR_ARG;
  }
  t->refs.fetch_add(1);
  return TO_OK;
}

to_status to_release(to_tensor t) {
  if (!t) return TO_OK;
  int cur = t->refs.load();
  while (cur > 1)
    if (t->refs.compare_exchange_weak(cur, cur - 1)) return TO_OK;
  API_BEGIN
  release(t);
  API_END
}

to_status to_shape(to_tensor t, int* rank, int64_t* dims, int64_t* batch) {
  if (!t) {
    to::g_err = "null argument: t";
    return TO_ERR_ARG;
  }
  if (rank) *rank = t->rank;
  if (dims)
    for (int i = 0; i < t->rank; ++i) dims[i] = t->dims[i];
  if (batch) *batch = t->batch;
  return TO_OK;
}

to_status to_set_default_dtype(int dtype) {
  API_BEGIN
  check_dtype(dtype);
  rt().default_dtype = dtype;
  API_END
}

to_status to_default_dtype(int* dtype) {
  API_BEGIN
  NONNULL(dtype);
  *dtype = rt().default_dtype;
  API_END
}

to_status to_dtype(to_tensor t, int* dtype) {
  API_BEGIN
  NONNULL(t); NONNULL(dtype);
  *dtype = t->dtype;
  API_END
}

to_status to_is_contiguous(to_tensor t, int* out) {
  API_BEGIN
  NONNULL(t);
  NONNULL(out);
  *out = t->contiguous() ? 1 : 0;
  API_END
}

to_status to_data_ptr(to_tensor t, void** out) {
  API_BEGIN
  NONNULL(t);
  NONNULL(out);
  ensure(t);
  *out = t->ptr;
  API_END
}

to_status to_upload(to_tensor t, const void* host, int64_t nbytes) {
  API_BEGIN
  require_init();
  NONNULL(t);
  no_capture("to_upload");
  ensure(t);
  before_write(t);
  TO_CHECK(t->contiguous(), TO_ERR_ARG, "to_upload needs a contiguous tensor");
  TO_CHECK(nbytes == t->total() * (int64_t)t->esize(), TO_ERR_SHAPE,
           "to_upload: byte count does not match " + shape_str(t));
  if (nbytes) {
    NONNULL(host);
    host_to_device(t->ptr, host, (size_t)nbytes, S());
    t->id = fresh_id();  // new contents: memo entries keyed on the old value must not match
  }
  API_END
}

to_status to_download(to_tensor t, void* host, int64_t nbytes) {
  API_BEGIN
  require_init();
  NONNULL(t);
  no_capture("to_download");
  TO_CHECK(nbytes == t->total() * (int64_t)t->esize(), TO_ERR_SHAPE,
           "to_download: byte count does not match " + shape_str(t));
  if (nbytes) {
    NONNULL(host);
    ensure(t);
    Holder c(contiguous(t));
    device_to_host(host, c.t->ptr, (size_t)nbytes, S());
  }
  API_END
}

to_status to_from_host(int dtype, int rank, const int64_t* dims, int64_t batch, const void* host,
                       to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(out);
  check_dtype(dtype);
  {
    // `generateA (\_ -> I 1)`, the seed of gradTOp (src/TensorOps/Types.hs:127-132), arrives here: a value of up to 64
    // elements that are all the same number is a constant like to_fill's -- known to the planner inside a scope, set by a
    // fill launch instead of a blocking copy outside one (so it may also appear in a captured step)
    TO_CHECK(rank >= 0 && rank <= TO_MAX_RANK && (rank == 0 || dims), TO_ERR_ARG, "to_from_host: bad rank / null dims");
    int64_t n = batch > 0 ? batch : 1;
    for (int i = 0; i < rank; ++i) n *= dims[i];
    if (n >= 1 && n <= 64 && host) {
      auto at = [&](int64_t i) { return dtype == TO_F64 ? static_cast<const double*>(host)[i] : (double)static_cast<const float*>(host)[i]; };
      const size_t es = dtype == TO_F64 ? 8 : 4;
      bool uniform = at(0) == at(0);  // (not NaN; and bit for bit the same, so that -0 stays -0)
      for (int64_t i = 1; i < n && uniform; ++i)
        uniform = std::memcmp(static_cast<const char*>(host) + i * es, host, es) == 0;
      if (uniform) {
        if (lazy_active()) {
          NodeDesc d;
          d.op = N_FILL;
          d.alpha = at(0);
          *out = lazy_record(d, 0, nullptr, rank, dims, batch, dtype);
          return TO_OK;
        }
        Holder t(new_tensor(rank, dims, batch, dtype));
        launch_fill(dtype, t.t->ptr, t.t->total(), at(0), S());
        *out = track(t.take());
        return TO_OK;
      }
    }
  }
  no_capture("to_from_host");
  Holder t(new_tensor(rank, dims, batch, dtype));
  const int64_t nbytes = t.t->total() * (int64_t)t.t->esize();
  if (nbytes) {
    NONNULL(host);
    host_to_device(t.t->ptr, host, (size_t)nbytes, S());
  }
  *out = t.take();
  API_END
}

to_status to_fill(int dtype, int rank, const int64_t* dims, int64_t batch, double value,
                  to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(out);
  check_dtype(dtype);
  if (lazy_active()) {  // a recorded constant: the planner can see its value (the gradient seed of gradTOp)
    NodeDesc d;
    d.op = N_FILL;
    d.alpha = value;
    *out = lazy_record(d, 0, nullptr, rank, dims, batch, dtype);
    return TO_OK;
  }
  Holder t(new_tensor(rank, dims, batch, dtype));
  launch_fill(dtype, t.t->ptr, t.t->total(), value, S());
  *out = track(t.take());
  API_END
}

to_status to_rand(int dtype, int rank, const int64_t* dims, int64_t batch, int dist, double a,
                  double b, uint64_t seed, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(out);
  check_dtype(dtype);
  TO_CHECK(dist >= 0 && dist <= 4, TO_ERR_ARG,
           "dist must be 0 (uniform), 1 (normal), 2 (exponential), 3 (cauchy) or 4 (laplace)");
  TO_CHECK(dist != 2 || a > 0.0, TO_ERR_ARG, "exponential: the rate must be positive");
  Holder t(new_tensor(rank, dims, batch, dtype));
  launch_rand(dtype, t.t->ptr, t.t->total(), dist, a, b, seed, S());
  *out = track(t.take());
  API_END
}

// ---- class Tensor ------------------------------------------------------------------------------
// The pure methods below are RECORDED inside a fusion scope (to_memo_begin .. to_memo_end): they validate
// shapes, return a deferred handle at once, and lazy.cpp runs the recorded graph -- fused into GEMM epilogues
// where the kernels allow -- when a result is actually needed.  Outside a scope they run eagerly.
static to_tensor do_gmul(int len_m, int len_o, int len_n, to_tensor a, to_tensor b, bool reduce) {
  // Outside a scope the deferred outer product reads its operands LATER.  before_write only sees writes made through the
  // library, so caller-owned memory (to_wrap: a torch buffer the host may overwrite right after the call) is never read
  // late outside a scope: such a product is computed now, like every other eager call.  Inside a scope the
  // documented rule for wrapped memory applies (tensorops_hip.h, "Caller-owned memory").
  const bool wrapped = (a->buf && !a->buf->owned) || (b->buf && !b->buf->owned);
  if (lazy_active() || (persample_outer(len_o, a, b, reduce) && !wrapped)) {
    GmulPlan gp;
    gmul_plan(gp, len_m, len_o, len_n, a, b, reduce, true);  // validation + output shape, no memory touched
    NodeDesc d;
    d.op = N_GMUL;
    d.lm = len_m; d.lo = len_o; d.ln = len_n;
    d.reduce = reduce;
    const to_tensor in[2] = {a, b};
    return lazy_record(d, 2, in, gp.out_rank, gp.odims, gp.out_batch, gp.dtype);
  }
  ensure(a);
  ensure(b);
  return gmul_impl(len_m, len_o, len_n, a, b, reduce);
}

to_status to_gmul(int len_m, int len_o, int len_n, to_tensor a, to_tensor b, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(a); NONNULL(b); NONNULL(out);
  MemoKey key{{1, (uint64_t)len_m, (uint64_t)len_o, (uint64_t)len_n, a->id, b->id}};
  if (to_tensor hit = memo_find(key)) { *out = hit; return TO_OK; }
  to_tensor r = track(do_gmul(len_m, len_o, len_n, a, b, false));
  memo_put(key, r);
  *out = r;
  API_END
}

to_status to_gmul_batch_sum(int len_m, int len_o, int len_n, to_tensor a, to_tensor b,
                            to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(a); NONNULL(b); NONNULL(out);
  MemoKey key{{2, (uint64_t)len_m, (uint64_t)len_o, (uint64_t)len_n, a->id, b->id}};
  if (to_tensor hit = memo_find(key)) { *out = hit; return TO_OK; }
  to_tensor r = track(do_gmul(len_m, len_o, len_n, a, b, true));
  memo_put(key, r);
  *out = r;
  API_END
}

to_status to_lift(to_expr f, int n, const to_tensor* xs, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(f); NONNULL(out);
  TO_CHECK(n >= 1, TO_ERR_ARG, "to_lift needs n >= 1 (use to_fill for constants)");
  NONNULL(xs);
  MemoKey key{{3, f->uid}};
  for (int i = 0; i < n; ++i) {
    NONNULL(xs[i]);
    key.k.push_back(xs[i]->id);
  }
  if (to_tensor hit = memo_find(key)) { *out = hit; return TO_OK; }
  to_tensor r;
  if (lazy_active()) {
    int64_t B = 0;
    int dtype = TO_F32;
    lift_check(f, n, xs, &B, &dtype);
    NodeDesc d;
    d.op = N_LIFT;
    d.f = f;
    r = lazy_record(d, n, xs, xs[0]->rank, xs[0]->dims, B, dtype);
  } else {
    ensure_all(n, xs);
    r = track(lift_impl(f, n, xs, 0, nullptr));
  }
  memo_put(key, r);
  *out = r;
  API_END
}

to_status to_sum(int n, const to_tensor* xs, int rank, const int64_t* dims, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(out);
  TO_CHECK(n >= 0, TO_ERR_ARG, "negative count");
  MemoKey key{{4, (uint64_t)n}};
  int64_t B = 0;
  for (int i = 0; i < n; ++i) {
    NONNULL(xs[i]);
    key.k.push_back(xs[i]->id);
    TO_CHECK(same_shape(xs[0], xs[i]), TO_ERR_SHAPE,
             "sumT: shapes differ: " + shape_str(xs[0]) + " vs " + shape_str(xs[i]));
    TO_CHECK(xs[0]->dtype == xs[i]->dtype, TO_ERR_ARG, "sumT: different dtypes");
    if (xs[i]->batch > 0) {
      TO_CHECK(B == 0 || B == xs[i]->batch, TO_ERR_SHAPE, "sumT: different batch sizes");
      B = xs[i]->batch;
    }
  }
  if (n > 0) {
    if (to_tensor hit = memo_find(key)) { *out = hit; return TO_OK; }
  }
  to_tensor r;
  if (n == 1) {  // sumT [x] = x
    retain(xs[0]);
    r = xs[0];
  } else if (lazy_active()) {
    NodeDesc d;
    if (n == 0) {
      d.op = N_FILL;
      d.alpha = 0.0;
      r = lazy_record(d, 0, nullptr, rank, dims, 0, rt().default_dtype);
    } else {
      d.op = N_SUM;
      r = lazy_record(d, n, xs, xs[0]->rank, xs[0]->dims, B, xs[0]->dtype);
    }
  } else if (std::any_of(xs, xs + n, pending_batched)) {
    // (per-sample outer products on their way to batchSum -- the sumT of `&&&` / shared weights -- stay recorded)
    NodeDesc d;
    d.op = N_SUM;
    r = lazy_record(d, n, xs, xs[0]->rank, xs[0]->dims, B, xs[0]->dtype);
  } else {
    ensure_all(n, xs);
    r = track(sum_impl(n, xs, rank, dims, n > 0 ? xs[0]->dtype : rt().default_dtype));
  }
  if (n > 0) memo_put(key, r);
  *out = r;
  API_END
}

to_status to_scale(double alpha, to_tensor x, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(x); NONNULL(out);
  MemoKey key{{5, bits(alpha), x->id}};
  if (to_tensor hit = memo_find(key)) { *out = hit; return TO_OK; }
  to_tensor r;
  if (lazy_active() || pending_batched(x)) {  // (a recorded per-sample outer product stays recorded under scaleT)
    NodeDesc d;
    d.op = N_SCALE;
    d.alpha = alpha;
    r = lazy_record(d, 1, &x, x->rank, x->dims, x->batch, x->dtype);
  } else {
    ensure(x);
    r = track(affine_impl(1, &x, &alpha, 0.0));
  }
  memo_put(key, r);
  *out = r;
  API_END
}

to_status to_transp(to_tensor x, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(x); NONNULL(out);
  MemoKey key{{6, x->id}};
  if (to_tensor hit = memo_find(key)) { *out = hit; return TO_OK; }
  to_tensor r = track(transp_impl(x));  // a view; of a deferred value, a deferred view
  memo_put(key, r);
  *out = r;
  API_END
}

to_status to_sum_rows(to_tensor x, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(x); NONNULL(out);
  MemoKey key{{7, x->id}};
  if (to_tensor hit = memo_find(key)) { *out = hit; return TO_OK; }
  to_tensor r;
  if (lazy_active()) {
    TO_CHECK(x->rank >= 1, TO_ERR_SHAPE, "sumRows needs rank >= 1, got " + shape_str(x));
    NodeDesc d;
    d.op = N_SUM_ROWS;
    r = lazy_record(d, 1, &x, x->rank - 1, x->dims + 1, x->batch, x->dtype);
  } else {
    ensure(x);
    r = track(sum_rows_impl(x));
  }
  memo_put(key, r);
  *out = r;
  API_END
}

to_status to_map_rows_const(int len_n, to_tensor row, to_tensor like, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(row); NONNULL(like); NONNULL(out);
  map_rows_const_check(len_n, row, like);
  MemoKey key{{8, (uint64_t)len_n, row->id, like->id}};
  if (to_tensor hit = memo_find(key)) { *out = hit; return TO_OK; }
  to_tensor res;
  if (lazy_active()) {
    NodeDesc d;
    d.op = N_MAP_ROWS;
    d.len_n = len_n;
    // only like's SHAPE matters (it becomes the result's): recording it as an input would be a false dependency
    res = lazy_record(d, 1, &row, like->rank, like->dims, row->batch > 0 ? row->batch : like->batch, row->dtype);
  } else {
    ensure(row);  // only like's shape is read
    res = track(map_rows_const_impl(len_n, row, like));
  }
  memo_put(key, res);
  *out = res;
  API_END
}

to_status to_slice(to_tensor x, int len_m, const int64_t* index, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(x); NONNULL(out);
  TO_CHECK(len_m >= 0 && len_m <= x->rank, TO_ERR_SHAPE, "slice: bad Length");
  int64_t off = 0;
  for (int i = 0; i < len_m; ++i) {
    TO_CHECK(index[i] >= 0 && index[i] < x->dims[i], TO_ERR_SHAPE, "slice: index out of range");
    off += index[i] * x->strides[i];
  }
  MemoKey key{{10, (uint64_t)len_m, x->id}};
  for (int i = 0; i < len_m; ++i) key.k.push_back((uint64_t)index[i]);
  if (to_tensor hit = memo_find(key)) { *out = hit; return TO_OK; }
  to_tensor r = track(new_view(x, x->rank - len_m, x->dims + len_m, x->strides + len_m, x->batch,
                               x->bstride, off));
  memo_put(key, r);
  *out = r;
  API_END
}

to_status to_stack(int rank_m, const int64_t* dims_m, const to_tensor* rows, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(out);
  int64_t nrows = 0, d[TO_MAX_RANK], B = 0;
  stack_check(rank_m, dims_m, rows, &nrows, d, &B);
  const int rank = rank_m + rows[0]->rank;
  MemoKey key{{11, (uint64_t)rank_m}};
  for (int i = 0; i < rank_m; ++i) key.k.push_back((uint64_t)dims_m[i]);
  for (int64_t r = 0; r < nrows; ++r) key.k.push_back(rows[r]->id);
  if (to_tensor hit = memo_find(key)) { *out = hit; return TO_OK; }
  bool same = rank_m >= 1;
  for (int64_t r = 1; r < nrows && same; ++r) same = rows[r] == rows[0];
  to_tensor res;
  if (lazy_active()) {
    NodeDesc nd;
    nd.len_n = rank_m;
    if (same) {
      // `mapRows l (\_ -> d) x` (the gradient of TO.sumRows, src/TensorOps/TOp.hs:155-158): every row is the SAME value
      nd.op = N_MAP_ROWS;
      res = lazy_record(nd, 1, rows, rank, d, B, rows[0]->dtype);
    } else {
      nd.op = N_STACK;
      res = lazy_record(nd, (int)nrows, rows, rank, d, B, rows[0]->dtype);
    }
  } else {
    ensure_all((int)nrows, rows);
    if (same) {
      Holder like(new_deferred(rank, d, B, rows[0]->dtype));  // only its shape is read
      res = track(map_rows_const_impl(rank_m, rows[0], like.t));
    } else {
      res = track(stack_impl(rank_m, dims_m, rows));
    }
  }
  memo_put(key, res);
  *out = res;
  API_END
}

to_status to_diag(int rank, to_tensor x, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(x); NONNULL(out);
  TO_CHECK(x->rank == 1, TO_ERR_SHAPE, "diag takes a vector, got " + shape_str(x));
  TO_CHECK(rank >= 1 && rank <= TO_MAX_RANK, TO_ERR_ARG, "diag: rank must be 1..8");
  TO_CHECK(x->batch == 0, TO_ERR_UNSUPPORTED, "diag of a batched tensor");
  ensure(x);
  Holder c(contiguous(x));
  int64_t d[TO_MAX_RANK];
  for (int i = 0; i < rank; ++i) d[i] = x->dims[0];
  Holder o(new_tensor(rank, d, 0, x->dtype));
  launch_fill(x->dtype, o.t->ptr, o.t->total(), 0.0, S());
  launch_diag(x->dtype, c.t->ptr, o.t->ptr, x->dims[0], rank, S());
  *out = track(o.take());
  API_END
}

to_status to_get_diag(to_tensor x, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(x); NONNULL(out);
  TO_CHECK(x->rank >= 2, TO_ERR_SHAPE, "getDiag needs rank >= 2, got " + shape_str(x));
  TO_CHECK(x->batch == 0, TO_ERR_UNSUPPORTED, "getDiag of a batched tensor");
  int64_t step = 0;
  for (int i = 0; i < x->rank; ++i) {
    TO_CHECK(x->dims[i] == x->dims[0], TO_ERR_SHAPE, "getDiag needs equal dims, got " + shape_str(x));
    step += x->strides[i];
  }
  ensure(x);
  Holder o(new_tensor(1, x->dims, 0, x->dtype));
  launch_get_diag(x->dtype, x->ptr, o.t->ptr, x->dims[0], step, S());
  *out = track(o.take());
  API_END
}

to_status to_index(to_tensor x, const int64_t* index, int64_t sample, double* out) {
  API_BEGIN
  require_init();
  NONNULL(x); NONNULL(out);
  int64_t off = 0;
  for (int i = 0; i < x->rank; ++i) {
    TO_CHECK(index[i] >= 0 && index[i] < x->dims[i], TO_ERR_SHAPE, "(!): index out of range");
    off += index[i] * x->strides[i];
  }
  if (x->batch > 0) {
    TO_CHECK(sample >= 0 && sample < x->batch, TO_ERR_SHAPE, "(!): sample out of range");
    off += sample * x->bstride;
  }
  ensure(x);
  *out = read_scalar(x, off);
  API_END
}

static void arg_extreme(to_tensor x, int64_t* host_out, bool minimum, const char* who) {
  require_init();
  NONNULL(x); NONNULL(host_out);
  no_capture(who);
  TO_CHECK(x->rank == 1 && x->dims[0] >= 1, TO_ERR_SHAPE, "argMax takes a non-empty vector, got " + shape_str(x));
  ensure(x);
  const int64_t B = x->batch > 0 ? x->batch : 1;
  const int64_t nl = (B * 8 + 3) / 4;  // B int64 in a float-typed pool buffer
  Holder tmp(new_tensor(1, &nl, 0));
  launch_arg_max_rows(x->dtype, x->ptr, reinterpret_cast<long long*>(tmp.t->ptr), B, x->dims[0], x->bstride,
                      x->strides[0], S(), minimum);
  device_to_host(host_out, tmp.t->ptr, (size_t)B * sizeof(int64_t), S());
}

to_status to_arg_max(to_tensor x, int64_t* host_out) {
  API_BEGIN
  arg_extreme(x, host_out, false, "to_arg_max");
  API_END
}

to_status to_arg_min(to_tensor x, int64_t* host_out) {
  API_BEGIN
  arg_extreme(x, host_out, true, "to_arg_min");
  API_END
}

to_status to_one_hot(int dtype, int64_t n, double hot, double cold, int64_t batch,
                     const int64_t* host_idx, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(host_idx); NONNULL(out);
  check_dtype(dtype);
  no_capture("to_one_hot");
  TO_CHECK(n >= 1 && batch >= 0, TO_ERR_ARG, "oneHot: bad size");
  const int64_t B = batch > 0 ? batch : 1;
  for (int64_t b = 0; b < B; ++b)
    TO_CHECK(host_idx[b] >= 0 && host_idx[b] < n, TO_ERR_SHAPE, "oneHot: index out of range");
  const int64_t nl = (B * 8 + 3) / 4;
  Holder tmp(new_tensor(1, &nl, 0));
  host_to_device(tmp.t->ptr, host_idx, (size_t)B * sizeof(int64_t), S());
  Holder o(new_tensor(1, &n, batch, dtype));
  launch_one_hot(dtype, o.t->ptr, reinterpret_cast<const long long*>(tmp.t->ptr), B, n, hot, cold, S());
  TO_HIP(hipStreamSynchronize(S()));  // host_idx may be stack memory
  *out = o.take();
  API_END
}

// ---- class BLAS ------------------------------------------------------------------------------------
static void need_rank(to_tensor t, int r, const char* who) {
  TO_CHECK(t->rank == r, TO_ERR_SHAPE,
           std::string(who) + ": expected rank " + std::to_string(r) + ", got " + shape_str(t));
  TO_CHECK(t->batch == 0, TO_ERR_UNSUPPORTED, std::string(who) + ": BLAS-class entry points take unbatched handles");
}

to_status to_blas_axpy(double alpha, to_tensor x, to_tensor y_or_null, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(x); NONNULL(out);
  need_rank(x, 1, "axpy");
  ensure(x);
  if (y_or_null) {
    need_rank(y_or_null, 1, "axpy");
    ensure(y_or_null);
    to_tensor xs[2] = {x, y_or_null};
    const double c[2] = {alpha, 1.0};
    *out = track(affine_impl(2, xs, c, 0.0));
  } else {
    *out = track(affine_impl(1, &x, &alpha, 0.0));
  }
  API_END
}

to_status to_blas_dot(to_tensor x, to_tensor y, double* out) {
  API_BEGIN
  require_init();
  NONNULL(x); NONNULL(y); NONNULL(out);
  need_rank(x, 1, "dot");
  need_rank(y, 1, "dot");
  ensure(x);
  ensure(y);
  Holder r(gmul_impl(0, 1, 0, x, y, false));
  *out = read_scalar(r.t, 0);
  API_END
}

to_status to_blas_ger(to_tensor x, to_tensor y, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(x); NONNULL(y); NONNULL(out);
  need_rank(x, 1, "ger");
  need_rank(y, 1, "ger");
  ensure(x);
  ensure(y);
  *out = track(gmul_impl(1, 0, 1, x, y, false));
  API_END
}

// C = alpha * A[n,o] . B[o,m] + beta * Cin ; B may be a vector (m == 1)
static to_tensor blas_mm(double alpha, to_tensor a, to_tensor b, double beta, to_tensor c,
                         bool vec) {
  ensure(a);
  ensure(b);
  if (c) ensure(c);
  TO_CHECK(a->dims[1] == b->dims[0], TO_ERR_SHAPE,
           "gemm/gemv: inner dims differ: " + shape_str(a) + " vs " + shape_str(b));
  const int64_t n = a->dims[0], o = a->dims[1], m = vec ? 1 : b->dims[1];
  int64_t od[2] = {n, m};
  TO_CHECK(a->dtype == b->dtype && (!c || c->dtype == a->dtype), TO_ERR_ARG, "gemm/gemv: different dtypes");
  Holder out(new_tensor(vec ? 1 : 2, od, 0, a->dtype));
  Holder hc;
  if (c) {
    TO_CHECK(c->rank == (vec ? 1 : 2) && c->dims[0] == n && (vec || c->dims[1] == m), TO_ERR_SHAPE,
             "gemm/gemv: C has shape " + shape_str(c));
    hc.t = contiguous(c);
  }
  GemmProblem p{};
  p.dtype = a->dtype;
  p.A = a->ptr; p.B = b->ptr; p.C = out.t->ptr;
  p.M = n; p.N = m; p.K = o;
  p.a_sm = a->strides[0]; p.a_sk = a->strides[1];
  p.b_sk = b->strides[0]; p.b_sn = vec ? 1 : b->strides[1];
  p.c_sm = m;
  p.batch = 1;
  p.alpha = alpha;
  p.beta = c ? beta : 0.0;
  p.Cin = c ? hc.t->ptr : nullptr;
  if (o == 0) {
    if (c) {
      to_tensor r = affine_impl(1, &hc.t, &beta, 0.0);
      return r;
    }
    launch_fill(out.t->dtype, out.t->ptr, out.t->total(), 0.0, S());
    return out.take();
  }
  run_gemm(p);
  return out.take();
}

to_status to_blas_gemv(double alpha, to_tensor a, to_tensor x, double beta, to_tensor y_or_null,
                       to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(a); NONNULL(x); NONNULL(out);
  need_rank(a, 2, "gemv");
  need_rank(x, 1, "gemv");
  if (y_or_null) need_rank(y_or_null, 1, "gemv");
  *out = track(blas_mm(alpha, a, x, beta, y_or_null, true));
  API_END
}

to_status to_blas_gemm(double alpha, to_tensor a, to_tensor b, double beta, to_tensor c_or_null,
                       to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(a); NONNULL(b); NONNULL(out);
  need_rank(a, 2, "gemm");
  need_rank(b, 2, "gemm");
  if (c_or_null) need_rank(c_or_null, 2, "gemm");
  // `gemm 1 a b Nothing` inside a scope is a recorded gmul like any other (round 6): BTensor's `mapBTM` issues one per
  // trailing matrix (BTensor.hs:703-710), and only a recorded stream lets the planner see that they are siblings --
  // same B, same shapes -- and send them out as one launch (lazy.cpp, sibling batches)
  if (lazy_active() && alpha == 1.0 && !c_or_null) {
    TO_CHECK(a->dims[1] == b->dims[0], TO_ERR_SHAPE, "gemm/gemv: inner dims differ: " + shape_str(a) + " vs " + shape_str(b));
    TO_CHECK(a->dtype == b->dtype, TO_ERR_ARG, "gemm/gemv: different dtypes");
    MemoKey key{{1, 1ull, 1ull, 1ull, a->id, b->id}};
    if (to_tensor hit = memo_find(key)) { *out = hit; return TO_OK; }
    to_tensor r = track(do_gmul(1, 1, 1, a, b, false));
    memo_put(key, r);
    *out = r;
    return TO_OK;
  }
  *out = track(blas_mm(alpha, a, b, beta, c_or_null, false));
  API_END
}

to_status to_blas_scale(double alpha, to_tensor x, to_tensor* out) { return to_scale(alpha, x, out); }

to_status to_blas_add(to_tensor x, to_tensor y, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(x); NONNULL(y); NONNULL(out);
  TO_CHECK(same_shape(x, y), TO_ERR_SHAPE, "addB: shapes differ");
  ensure(x);
  ensure(y);
  to_tensor xs[2] = {x, y};
  const double c[2] = {1.0, 1.0};
  *out = track(affine_impl(2, xs, c, 0.0));
  API_END
}

to_status to_blas_index_row(int64_t i, to_tensor a, to_tensor* out) {
  {
    API_BEGIN
    NONNULL(a);
    need_rank(a, 2, "indexRowB");
    API_END_CHECK
  }
  return to_slice(a, 1, &i, out);
}

to_status to_blas_transp(to_tensor a, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(a); NONNULL(out);
  need_rank(a, 2, "transpB");
  *out = track(transp_impl(a));
  API_END
}

to_status to_blas_eye(int dtype, int64_t n, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(out);
  check_dtype(dtype);
  int64_t d[2] = {n, n};
  Holder ones(new_tensor(1, d, 0, dtype));
  launch_fill(dtype, ones.t->ptr, n, 1.0, S());
  Holder o(new_tensor(2, d, 0, dtype));
  launch_fill(dtype, o.t->ptr, n * n, 0.0, S());
  launch_diag(dtype, ones.t->ptr, o.t->ptr, n, 2, S());
  *out = track(o.take());
  API_END
}

to_status to_blas_trace(to_tensor a, double* out) {
  API_BEGIN
  require_init();
  NONNULL(a); NONNULL(out);
  need_rank(a, 2, "traceB");
  TO_CHECK(a->dims[0] == a->dims[1], TO_ERR_SHAPE, "traceB needs a square matrix");
  ensure(a);
  Holder r(new_tensor(0, nullptr, 0, a->dtype));
  launch_sum_axis(a->dtype, a->ptr, r.t->ptr, 1, a->dims[0], 1, 0, a->strides[0] + a->strides[1], 0, S());
  *out = read_scalar(r.t, 0);
  API_END
}

to_status to_blas_diag(to_tensor x, to_tensor* out) { return to_diag(2, x, out); }

to_status to_blas_get_diag(to_tensor a, to_tensor* out) {
  {
    API_BEGIN
    NONNULL(a);
    need_rank(a, 2, "getDiagB");
    API_END_CHECK
  }
  return to_get_diag(a, out);
}

to_status to_blas_sum(to_tensor x, double* out) {
  API_BEGIN
  require_init();
  NONNULL(x); NONNULL(out);
  TO_CHECK(x->rank == 1 || x->rank == 2, TO_ERR_SHAPE, "sumB takes a vector or matrix");
  ensure(x);
  Holder c(contiguous(x));
  Holder r(new_tensor(0, nullptr, 0, x->dtype));
  launch_sum_axis(x->dtype, c.t->ptr, r.t->ptr, 1, c.t->total(), 1, 0, 1, 0, S());
  *out = read_scalar(r.t, 0);
  API_END
}

// ---- expressions -------------------------------------------------------------------------------------
to_status to_expr_compile(int arity, int n_instr, const int32_t* code, int n_consts,
                          const double* consts, to_expr* out) {
  API_BEGIN
  require_init();
  NONNULL(out);
  *out = expr_compile(arity, n_instr, code, n_consts, consts);
  API_END
}

to_status to_expr_release(to_expr e) {
  API_BEGIN
  expr_release(e);  // ref-counted: recorded ops that still use it keep it alive
  API_END
}

to_status to_expr_kind(to_expr e, int* kind) {
  API_BEGIN
  NONNULL(e); NONNULL(kind);
  if (e->kind == EW_VM) expr_prepare(e, TO_F32);
  *kind = (e->kind == EW_VM && e->jit[0]) ? 100 : e->kind;  // 100: run-time specialised kernel
  API_END
}

// ---- batching ------------------------------------------------------------------------------------------
// The sum over the hidden batch.  The DSL's backward closures know nothing of batches: with batched data the
// cotangent of an unbatched parameter comes back batched -- for a weight matrix the per-sample outer products
// gmul (transp x) dtdz (src/TensorOps/TOp.hs:86-88), [B x o x i] -- and the host sums it where gradTOp returns
// (`batchSum`, hs/TensorOps/Backend/HipTensor.hs).  Inside a scope the sum is pushed into the recorded producer so
// that the per-sample value never exists: sum_b gmul(a_b, b_b) is ONE contraction with the batch folded into K
// (to_gmul_batch_sum), and the sum commutes with sumT and scaleT.
static to_tensor batch_sum_value(to_tensor x) {
  MemoKey key{{9, x->id}};
  if (to_tensor hit = memo_find(key)) return hit;
  to_tensor r = nullptr;
  NodeDesc nd;
  std::vector<to_tensor> in;
  if (x->batch == 0) {
    retain(x);
    r = x;
  } else if (lazy_active() || pending_batched(x)) {
    if (lazy_node_of(x, &nd, &in)) {
      if (nd.op == N_GMUL && !nd.reduce) {
        r = do_gmul(nd.lm, nd.lo, nd.ln, in[0], in[1], true);
      } else if (nd.op == N_SCALE) {
        Holder s(batch_sum_value(in[0]));
        NodeDesc d;
        d.op = N_SCALE;
        d.alpha = nd.alpha;
        r = lazy_record(d, 1, &s.t, s.t->rank, s.t->dims, 0, s.t->dtype);
      } else if (nd.op == N_SUM) {
        // sum_b (x_b + y_b + c) = sum_b x_b + sum_b y_b + B c: an unbatched operand is shared by all B samples.  The common
        // one is the zero tensor `sumT []` of `drop` / `take` / `&&&` (TOp.hs:362-381): it disappears.
        const double B = (double)x->batch;
        std::vector<std::unique_ptr<Holder>> parts;
        std::vector<to_tensor> ps;
        for (to_tensor i : in) {
          NodeDesc id;
          std::vector<to_tensor> iin;
          if (i->batch > 0) {
            parts.emplace_back(new Holder(batch_sum_value(i)));
          } else if (lazy_node_of(i, &id, &iin) && id.op == N_FILL) {
            if (id.alpha == 0.0) continue;
            NodeDesc d;
            d.op = N_FILL;
            d.alpha = B * id.alpha;
            parts.emplace_back(new Holder(lazy_record(d, 0, nullptr, i->rank, i->dims, 0, i->dtype)));
          } else {
            NodeDesc d;
            d.op = N_SCALE;
            d.alpha = B;
            parts.emplace_back(new Holder(lazy_record(d, 1, &i, i->rank, i->dims, 0, i->dtype)));
          }
          ps.push_back(parts.back()->t);
        }
        if (ps.size() == 1) {
          retain(ps[0]);
          r = ps[0];
        } else if (ps.size() >= 2) {
          NodeDesc d;
          d.op = N_SUM;
          r = lazy_record(d, (int)ps.size(), ps.data(), x->rank, x->dims, 0, x->dtype);
        }
      }
    }
    if (!r) {
      NodeDesc d;
      d.op = N_BATCH_SUM;
      r = lazy_record(d, 1, &x, x->rank, x->dims, 0, x->dtype);
    }
    if (!lazy_active()) ensure(r);  // (an eager call: the sum exists when it returns; the per-sample value never does)
  } else {
    ensure(x);
    r = track(batch_sum_impl(x));
  }
  memo_put(key, r);
  return r;
}

to_status to_batch_sum(to_tensor x, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(x); NONNULL(out);
  *out = batch_sum_value(x);
  API_END
}

to_status to_batch_bcast(to_tensor x, int64_t batch, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(x); NONNULL(out);
  TO_CHECK(x->batch == 0 && batch > 0, TO_ERR_ARG, "batch_bcast takes an unbatched tensor and B > 0");
  ensure(x);
  Holder c(contiguous(x));
  Holder o(new_tensor(x->rank, x->dims, batch, x->dtype));
  launch_bcast_axis(x->dtype, c.t->ptr, o.t->ptr, 1, batch, c.t->numel(), 0, S());
  *out = track(o.take());
  API_END
}

to_status to_batch_select(to_tensor x, int64_t sample, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(x); NONNULL(out);
  TO_CHECK(x->batch > 0 && sample >= 0 && sample < x->batch, TO_ERR_SHAPE, "batch_select: sample out of range");
  *out = track(new_view(x, x->rank, x->dims, x->strides, 0, x->numel(), sample * x->bstride));
  API_END
}

to_status to_batch_slice(to_tensor x, int64_t start, int64_t count, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(x); NONNULL(out);
  TO_CHECK(x->batch > 0 && start >= 0 && count >= 1 && start + count <= x->batch, TO_ERR_SHAPE,
           "batch_slice: range out of bounds");
  *out = track(new_view(x, x->rank, x->dims, x->strides, count, x->bstride, start * x->bstride));
  API_END
}

to_status to_batch_gather(to_tensor x, int64_t n_idx, const int64_t* host_idx, to_tensor* out) {
  API_BEGIN
  require_init();
  NONNULL(x); NONNULL(host_idx); NONNULL(out);
  no_capture("to_batch_gather");
  TO_CHECK(x->batch > 0 && n_idx >= 1, TO_ERR_SHAPE, "batch_gather: needs a batched tensor and at least one index");
  for (int64_t k = 0; k < n_idx; ++k)
    TO_CHECK(host_idx[k] >= 0 && host_idx[k] < x->batch, TO_ERR_SHAPE, "batch_gather: index out of range");
  ensure(x);
  Holder c(contiguous(x));
  // The indices of a minibatch (up to 8,192 of them) ride through the pinned upload ring, stream-ordered: the host's copy is taken
  // before this returns and nothing waits for the GPU (tools/ops_scan2.py: 37 us a call -> the gather's own few, with the
  // synchronisation this used to end on); more indices: the synchronous staged transfer.
  Holder tmp;
  const long long* didx;
  if ((size_t)n_idx * sizeof(int64_t) <= 65536) {
    didx = static_cast<const long long*>(table_upload(host_idx, (size_t)n_idx * sizeof(int64_t), S()));
  } else {
    const int64_t nl = (n_idx * 8 + 3) / 4;
    tmp.t = new_tensor(1, &nl, 0);
    host_to_device(tmp.t->ptr, host_idx, (size_t)n_idx * sizeof(int64_t), S());   // (synchronous: host_idx may be stack memory)
    didx = reinterpret_cast<const long long*>(tmp.t->ptr);
  }
  Holder o(new_tensor(x->rank, x->dims, n_idx, x->dtype));
  launch_gather_rows(c.t->ptr, o.t->ptr, didx, n_idx, x->numel() * (int64_t)x->esize(), S());
  *out = track(o.take());
  API_END
}

// ---- memo / graph ----------------------------------------------------------------------------------------
to_status to_memo_begin(void) {
  API_BEGIN
  require_init();
  scope_begin();
  API_END
}

to_status to_memo_end(void) {
  API_BEGIN
  scope_end();  // launches what the host still holds of this thread's recorded results, drops the memo table
  API_END
}

to_status to_force(to_tensor t) {
  API_BEGIN
  require_init();
  NONNULL(t);
  ensure(t);
  API_END
}

to_status to_force_many(int n, const to_tensor* ts) {
  API_BEGIN
  require_init();
  TO_CHECK(n >= 0, TO_ERR_ARG, "negative count");
  if (n > 0) NONNULL(ts);
  for (int i = 0; i < n; ++i) NONNULL(ts[i]);
  ensure_all(n, ts);  // ONE plan for all of them: a step's new parameters share launches
  API_END
}

to_status to_set_lazy(int on, int* previous) {
  API_BEGIN
  const int prev = lazy_set(on);
  if (previous) *previous = prev;
  API_END
}

to_status to_set_loss_head_match(int on, int* previous) {
  API_BEGIN
  const int prev = lazy_set_loss_head_match(on);
  if (previous) *previous = prev;
  API_END
}

to_status to_lazy_stats(int64_t* recorded, int64_t* fused_launches, int64_t* elided, int64_t* flushes) {
  API_BEGIN
  if (recorded) *recorded = lazy_stat(0);
  if (fused_launches) *fused_launches = lazy_stat(1);
  if (elided) *elided = lazy_stat(2);
  if (flushes) *flushes = lazy_stat(3);
  API_END
}

to_status to_lazy_time(int64_t* plan_ns, int64_t* flush_ns) {
  API_BEGIN
  if (plan_ns) *plan_ns = lazy_stat(4);
  if (flush_ns) *flush_ns = lazy_stat(5);
  API_END
}

to_status to_plan_cache_stats(int64_t* hits, int64_t* misses, int64_t* entries) {
  API_BEGIN
  lazy_cache_stats(hits, misses, entries);
  API_END
}

to_status to_plan_cache_clear(void) {
  API_BEGIN
  lazy_cache_clear();
  API_END
}

to_status to_transfer_stats(int64_t* staged_calls, int64_t* staged_bytes, int64_t* direct_calls, int64_t* direct_bytes) {
  API_BEGIN
  const TransferStats st = transfer_stats();
  if (staged_calls) *staged_calls = st.staged_calls;
  if (staged_bytes) *staged_bytes = st.staged_bytes;
  if (direct_calls) *direct_calls = st.direct_calls;
  if (direct_bytes) *direct_bytes = st.direct_bytes;
  API_END
}

to_status to_api_time(int64_t* ns, int64_t* calls) {
  API_BEGIN
  if (ns) *ns = (int64_t)((double)g_api_ticks * api_ns_per_tick());
  if (calls) *calls = g_api_calls;
  API_END
}

to_status to_graph_begin(void) {
  API_BEGIN
  require_init();
  no_capture("to_graph_begin");
  TO_HIP(hipStreamBeginCapture(S(), hipStreamCaptureModeRelaxed));
  rt().capturing = true;
  rt().capture_kept.clear();
  g_capture_launches.clear();
  set_launch_recorder(&g_capture_launches);
  g_capture_desc.clear();
  lazy_describe_into(&g_capture_desc);
  API_END
}

to_status to_graph_end(to_graph* out) {
  API_BEGIN
  NONNULL(out);
  TO_CHECK(rt().capturing, TO_ERR_STATE, "to_graph_end without to_graph_begin");
  // (like the end of a scope, the end of a capture demands nothing: what the step produces is forced or copied into
  //  place INSIDE the capture; deferred handles a garbage-collected host merely still holds must not become part of
  //  the replayed step)
  set_launch_recorder(nullptr);
  lazy_describe_into(nullptr);
  rt().capturing = false;
  auto* g = new to_graph_s();
  g->launches.swap(g_capture_launches);
  g->desc.swap(g_capture_desc);
  g->kept.assign(rt().capture_kept.begin(), rt().capture_kept.end());
  rt().capture_kept.clear();
  hipError_t e = hipStreamEndCapture(S(), &g->graph);
  if (e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
  if (e == hipSuccess) {
    // A short step made of kernels only, every one of them issued through launch_k: replay = issue them again.
    static const int list_max = [] { const char* v = getenv("TOPS_REPLAY_LIST_MAX"); return v ? atoi(v) : 12; }();
    size_t n_nodes = 0;
    if (hipGraphGetNodes(g->graph, nullptr, &n_nodes) == hipSuccess && n_nodes == g->launches.size() &&
        n_nodes >= 1 && (int)n_nodes <= list_max) {
      std::vector<hipGraphNode_t> nodes(n_nodes);
      bool all_kernels = hipGraphGetNodes(g->graph, nodes.data(), &n_nodes) == hipSuccess;
      for (size_t i = 0; i < n_nodes && all_kernels; ++i) {
        hipGraphNodeType ty;
        all_kernels = hipGraphNodeGetType(nodes[i], &ty) == hipSuccess && ty == hipGraphNodeTypeKernel;
      }
      g->replay_list = all_kernels;
    }
    if (!g->replay_list) g->launches.clear();
  }
  if (e != hipSuccess) {
    for (to_tensor t : g->kept) release(t);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    fail(TO_ERR_HIP, std::string("graph capture failed: ") + hipGetErrorString(e));
  }
  *out = g;
  API_END
}

to_status to_graph_launch(to_graph g) {
  API_BEGIN
  NONNULL(g);
  no_capture("to_graph_launch");
  lazy_flush_all();  // a replay rewrites the captured buffers: recorded ops must have read them first
  if (g->replay_list) {
    for (const auto& l : g->launches) l->replay(S());
    TO_HIP(hipGetLastError());
  } else {
    TO_HIP(hipGraphLaunch(g->exec, S()));
  }
  API_END
}

to_status to_graph_info(to_graph g, int* n_launches, int* replays_as_launch_list) {
  API_BEGIN
  NONNULL(g);
  size_t n = 0;
  if (g->graph) (void)hipGraphGetNodes(g->graph, nullptr, &n);
  if (n_launches) *n_launches = (int)n;
  if (replays_as_launch_list) *replays_as_launch_list = g->replay_list ? 1 : 0;
  API_END
}

to_status to_graph_release(to_graph g) {
  API_BEGIN
  if (g) {
    (void)hipStreamSynchronize(S());
    for (to_tensor t : g->kept) release(t);
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
  }
  API_END
}

to_status to_sgd_step_inplace(to_tensor p, to_tensor g, double rate) {
  API_BEGIN
  require_init();
  NONNULL(p); NONNULL(g);
  TO_CHECK(same_shape(p, g) && p->batch == g->batch, TO_ERR_SHAPE,
           "sgd: " + shape_str(p) + " vs " + shape_str(g));
  TO_CHECK(p->contiguous() && g->contiguous(), TO_ERR_ARG, "sgd needs contiguous tensors");
  TO_CHECK(p->dtype == g->dtype, TO_ERR_ARG, "sgd: different dtypes");
  ensure(p);
  ensure(g);
  before_write(p);
  p->id = fresh_id();
  launch_sgd(p->dtype, p->ptr, g->ptr, rate, p->total(), S());
  API_END
}

static void copy_check(to_tensor d, to_tensor sr) {
  NONNULL(d); NONNULL(sr);
  TO_CHECK(same_shape(d, sr) && d->batch == sr->batch, TO_ERR_SHAPE,
           "copy_into: " + shape_str(d) + " vs " + shape_str(sr));
  TO_CHECK(d->contiguous(), TO_ERR_ARG, "copy_into needs a contiguous destination");
  TO_CHECK(d->dtype == sr->dtype, TO_ERR_ARG, "copy_into: different dtypes");
}

to_status to_copy_into(to_tensor dst, to_tensor src) {
  API_BEGIN
  require_init();
  copy_check(dst, src);
  lazy_copy_into(1, &dst, &src);
  API_END
}

to_status to_copy_into_many(int n, const to_tensor* dsts, const to_tensor* srcs) {
  API_BEGIN
  require_init();
  NONNULL(dsts); NONNULL(srcs);
  TO_CHECK(n >= 0, TO_ERR_ARG, "copy_into_many: negative count");
  for (int i = 0; i < n; ++i) copy_check(dsts[i], srcs[i]);
  lazy_copy_into(n, dsts, srcs);
  API_END
}

// GEMM with fused epilogue on packed row-major operands: C[M,N] = A.B (+bias, act, dact)
// returns true when `rowsum` (sum_k A[m,k]) was produced by the same launch
struct LossHead {  // loss gradient fused into the last layer's GEMM epilogue when the kernel can
  int kind = 0;    // GemmProblem::loss_rows
  const void* target = nullptr;
  void* loss_out = nullptr;
  bool done = false;  // set when the launch produced dz instead of z
  // optional fused tail (GemmProblem::tail_*): the previous layer's cotangent for the same rows
  const void* tail_w = nullptr;
  const void* tail_h = nullptr;
  void* tail_out = nullptr;
  int tail_n = 0;
  bool tail_done = false;
};
static bool fused_gemm(int dtype, const void* A, int64_t a_sm, int64_t a_sk, const void* B, int64_t b_sk,
                       int64_t b_sn, void* C, int64_t M, int64_t N, int64_t K, const void* bias,
                       int act, const void* dact, void* rowsum = nullptr, hipStream_t stream = nullptr,
                       LossHead* head = nullptr) {
  GemmProblem p{};
  p.dtype = dtype;
  p.A = A; p.B = B; p.C = C;
  p.M = M; p.N = N; p.K = K;
  p.a_sm = a_sm; p.a_sk = a_sk; p.b_sk = b_sk; p.b_sn = b_sn; p.c_sm = N;
  p.batch = 1;
  p.alpha = 1.0; p.beta = 0.0;
  p.bias = bias; p.act = act; p.dact = dact;
  hipStream_t st = stream ? stream : S();
  // latency-bound shapes (incl. the tiny ones of a one-sample step) run on the small-GEMM kernel: it carries
  // every fused epilogue for both element types (the tiled fp64 kernel has none)
  const int64_t t64 = ((M + 63) / 64) * ((N + 63) / 64);
  if (gemm_small_applicable(p) || (t64 < 200 && gemm_small_can(p))) {
    p.rowsum = rowsum;
    if (head && gemm_small_fuses_loss(p)) {
      p.loss_rows = head->kind; p.target = head->target; p.loss_out = head->loss_out;
      if (head->tail_out && gemm_small_fuses_tail(p, head->tail_n)) {
        p.tail_w = head->tail_w; p.tail_h = head->tail_h; p.tail_out = head->tail_out; p.tail_n = head->tail_n;
        head->tail_done = true;
      }
      head->done = true;
    }
    launch_gemm_small(p, st);
    return rowsum != nullptr;
  }
  // the tiled fp64 kernel has no fused epilogue: the caller (the trainer) falls back to the generic path
  TO_CHECK(dtype == TO_F32, TO_ERR_UNSUPPORTED, "pre-fused fp64 path: a contraction is outside the small-GEMM range");
  launch_gemm_mfma(p, st);
  return false;
}

// sgd: gw/gb are the parameters themselves and the weight-gradient launches apply
// P <- P - rate * gradient in their epilogue (alpha = -rate, beta = 1, Cin = C = W; the bias through the
// accumulating row sum): the step loses its separate update launch.
static void fflayer_stack_impl(int n_layers, const to_tensor* w, const to_tensor* b, int hidden_act, int out_act,
                               int loss, to_tensor x, to_tensor y, const to_tensor* gw, const to_tensor* gb,
                               to_tensor losses, bool sgd, double rate) {
  require_init();
  NONNULL(w); NONNULL(b); NONNULL(x); NONNULL(y); NONNULL(gw); NONNULL(gb);
  TO_CHECK(n_layers >= 1, TO_ERR_ARG, "need at least one layer");
  ensure(x);
  ensure(y);
  if (losses) { ensure(losses); before_write(losses); }
  for (int l = 0; l < n_layers; ++l) {
    NONNULL(w[l]); NONNULL(b[l]); NONNULL(gw[l]); NONNULL(gb[l]);
    ensure(w[l]); ensure(b[l]); ensure(gw[l]); ensure(gb[l]);
    before_write(gw[l]);
    before_write(gb[l]);
    if (sgd) { w[l]->id = fresh_id(); b[l]->id = fresh_id(); }
  }
  TO_CHECK(hidden_act == TO_ACT_LOGISTIC, TO_ERR_UNSUPPORTED, "fused path: hidden activation must be logistic");
  const bool sm_ce = out_act == TO_ACT_SOFTMAX && loss == TO_LOSS_CROSS_ENTROPY;
  const bool lg_se = out_act == TO_ACT_LOGISTIC && loss == TO_LOSS_SQUARED_ERROR;
  TO_CHECK(sm_ce || lg_se, TO_ERR_UNSUPPORTED,
           "fused path: (softmax, crossEntropy) or (logistic, squaredError) only");
  TO_CHECK(x->rank == 1 && y->rank == 1 && x->batch > 0 && x->batch == y->batch, TO_ERR_SHAPE,
           "x and y must be batched vectors with the same batch, got " + shape_str(x) + " " + shape_str(y));
  TO_CHECK(x->contiguous() && y->contiguous(), TO_ERR_ARG, "x and y must be contiguous");
  const int dt = x->dtype;
  TO_CHECK(y->dtype == dt, TO_ERR_ARG, "x and y have different dtypes");
  const int64_t B = x->batch;
  int64_t fan_in = x->dims[0];
  for (int l = 0; l < n_layers; ++l) {
    NONNULL(w[l]); NONNULL(b[l]); NONNULL(gw[l]); NONNULL(gb[l]);
    TO_CHECK(w[l]->dtype == dt && b[l]->dtype == dt && gw[l]->dtype == dt && gb[l]->dtype == dt, TO_ERR_ARG,
             "parameters, gradients and data must share one dtype");
    TO_CHECK(w[l]->rank == 2 && w[l]->batch == 0 && w[l]->dims[1] == fan_in && w[l]->contiguous(),
             TO_ERR_SHAPE, "layer " + std::to_string(l) + ": W has shape " + shape_str(w[l]));
    TO_CHECK(b[l]->rank == 1 && b[l]->batch == 0 && b[l]->dims[0] == w[l]->dims[0] && b[l]->contiguous(),
             TO_ERR_SHAPE, "layer " + std::to_string(l) + ": b has shape " + shape_str(b[l]));
    TO_CHECK(same_shape(gw[l], w[l]) && gw[l]->contiguous() && same_shape(gb[l], b[l]) && gb[l]->contiguous(),
             TO_ERR_SHAPE, "gradient destinations must match the parameters");
    fan_in = w[l]->dims[0];
  }
  TO_CHECK(y->dims[0] == fan_in, TO_ERR_SHAPE, "y does not match the output layer");
  if (losses) TO_CHECK(losses->rank == 0 && losses->batch == B && losses->contiguous(), TO_ERR_SHAPE,
                       "losses must be a batched scalar");

  if (sgd)  // nothing may be half-updated: every weight gradient must be one small-GEMM launch
    for (int l = 0; l < n_layers; ++l) {
      GemmProblem q{};
      q.dtype = dt;
      q.M = w[l]->dims[0]; q.N = w[l]->dims[1]; q.K = B;
      q.a_sm = 1; q.a_sk = q.M; q.b_sk = q.N; q.b_sn = 1; q.c_sm = q.N;
      q.batch = 1;
      const int64_t t64 = ((q.M + 63) / 64) * ((q.N + 63) / 64);
      TO_CHECK(gemm_small_applicable(q) || (t64 < 200 && gemm_small_can(q)), TO_ERR_UNSUPPORTED,
               "fused SGD step: a weight gradient is outside the small-GEMM range");
    }
  static const int fuse_tail = [] { const char* e = ab_getenv("TOPS_STEP_FUSE_TAIL"); return e ? atoi(e) : 1; }();
  Holder tail;  // dz_{L-1} when the last layer's launch produced it
  LossHead head;
  head.kind = sm_ce ? 1 : 2;
  head.target = y->ptr;
  head.loss_out = losses ? losses->ptr : nullptr;
  // forward: a_l = logistic(a_{l-1} W_l^T + b_l) for hidden layers, z_L for the last
  std::vector<Holder> act(n_layers);  // act[l]: [B; n_l]; the last holds z_L, then is reused as dz_L
  const void* prev = x->ptr;
  int64_t prev_n = x->dims[0];
  for (int l = 0; l < n_layers; ++l) {
    const int64_t n = w[l]->dims[0];
    act[l].t = new_tensor(1, &n, B, dt);
    // C[B,n] = A[B,prev_n] . W^T : B operand element (k, j) = W[j*prev_n + k]
    // (last layer: the loss head runs in the same launch when the row fits one 16-wide tile)
    const bool last = l + 1 == n_layers;
    if (last && n_layers >= 2 && fuse_tail) {
      // the loss-head launch also produces dz_{L-1} = (dz_L . W_L) * h (1 - h) for its rows
      tail.t = new_tensor(1, &prev_n, B, dt);
      head.tail_w = w[l]->ptr;
      head.tail_h = act[l - 1].t->ptr;
      head.tail_out = tail.t->ptr;
      head.tail_n = (int)prev_n;
    }
    fused_gemm(dt, prev, prev_n, 1, w[l]->ptr, 1, prev_n, act[l].t->ptr, B, n, prev_n, b[l]->ptr,
               last ? 0 : 1, nullptr, nullptr, nullptr, last ? &head : nullptr);
    prev = act[l].t->ptr;
    prev_n = n;
  }
  // loss gradient wrt z_L, per sample row
  const int64_t nL = w[n_layers - 1]->dims[0];
  Holder cur;
  if (head.done) {
    cur.t = act[n_layers - 1].t;  // already dz_L
    act[n_layers - 1].t = nullptr;
  } else {
    cur.t = new_tensor(1, &nL, B, dt);
    launch_loss_grad_rows(dt, act[n_layers - 1].t->ptr, y->ptr, cur.t->ptr, losses ? losses->ptr : nullptr, B, nL,
                          sm_ce ? 0 : 1, S());
  }
  // The weight gradient of layer l (dz_l^T . a_in, + its row sums = the bias gradient) as a GEMM problem:
  // A element (i,k) = dz[k*n + i], B element (k,j) = a_in[k*m + j]
  auto wgrad = [&](int l, const void* dz) {
    GemmProblem p{};
    const int64_t n = w[l]->dims[0], m = w[l]->dims[1];
    p.dtype = dt;
    p.A = dz; p.B = l > 0 ? act[l - 1].t->ptr : x->ptr; p.C = gw[l]->ptr;
    p.M = n; p.N = m; p.K = B;
    p.a_sm = 1; p.a_sk = n; p.b_sk = m; p.b_sn = 1; p.c_sm = m;
    p.batch = 1;
    p.alpha = 1.0; p.beta = 0.0;
    p.rowsum = gb[l]->ptr;
    if (sgd) {
      p.alpha = -rate; p.beta = 1.0; p.Cin = w[l]->ptr;
      p.rowsum_acc = true; p.rowsum_alpha = -rate;
    }
    return p;
  };
  // backward, phase 1: every dz_l (the propagation reads W_l, which phase 2 may overwrite in place)
  // dz_{l-1}[B,m] = (dz_l[B,n] . W_l[n,m]) * h (1 - h), h = act[l-1]
  std::vector<Holder> dz(n_layers);
  dz[n_layers - 1].t = cur.take();
  for (int l = n_layers - 1; l > 0; --l) {
    const int64_t n = w[l]->dims[0], m = w[l]->dims[1];
    if (l == n_layers - 1 && head.tail_done && tail.t) {
      dz[l - 1].t = tail.take();  // came out of the loss-head launch
    } else {
      dz[l - 1].t = new_tensor(1, &m, B, dt);
      fused_gemm(dt, dz[l].t->ptr, n, 1, w[l]->ptr, m, 1, dz[l - 1].t->ptr, B, m, n, nullptr, 0, act[l - 1].t->ptr);
    }
  }
  // phase 2: the weight gradients, independent of each other.  The two last ones go out as ONE launch when
  // their shapes allow (one launch floor, ~4 us, less per step: 33.6 -> 28.0 us on config 3).
  // (Running them on a side stream instead measured slower: the fork/join events cost more than the overlap
  // buys, 0.0485 -> 0.0591 ms/step.)
  // one sample: every weight gradient is an outer product -- all layers in one launch
  static const int rank1 = [] { const char* e = ab_getenv("TOPS_STEP_RANK1"); return e ? atoi(e) : 1; }();
  if (B == 1 && rank1) {
    for (int l0 = 0; l0 < n_layers; l0 += RANK1_MAX_LAYERS) {
      const int cnt = std::min(RANK1_MAX_LAYERS, n_layers - l0);
      const void *dzp[RANK1_MAX_LAYERS], *ap[RANK1_MAX_LAYERS];
      void *wp[RANK1_MAX_LAYERS], *bp[RANK1_MAX_LAYERS];
      int64_t rows[RANK1_MAX_LAYERS], cols[RANK1_MAX_LAYERS];
      for (int q = 0; q < cnt; ++q) {
        const int l = l0 + q;
        dzp[q] = dz[l].t->ptr;
        ap[q] = l > 0 ? act[l - 1].t->ptr : x->ptr;
        wp[q] = gw[l]->ptr;
        bp[q] = gb[l]->ptr;
        rows[q] = w[l]->dims[0];
        cols[q] = w[l]->dims[1];
      }
      launch_rank1_many(dt, cnt, dzp, ap, wp, bp, rows, cols, sgd ? -rate : 1.0, sgd, S());
    }
    return;
  }
  int first = n_layers - 1;
  if (n_layers >= 2 &&
      launch_gemm_small_pair(wgrad(n_layers - 2, dz[n_layers - 2].t->ptr), wgrad(n_layers - 1, dz[n_layers - 1].t->ptr), S()))
    first = n_layers - 3;
  for (int l = first; l >= 0; --l) {
    const GemmProblem p = wgrad(l, dz[l].t->ptr);
    const int64_t t64 = ((p.M + 63) / 64) * ((p.N + 63) / 64);
    if (gemm_small_applicable(p) || (t64 < 200 && gemm_small_can(p))) {
      launch_gemm_small(p, S());
      continue;
    }
    TO_CHECK(!sgd, TO_ERR_UNSUPPORTED, "fused SGD step: a weight gradient is outside the small-GEMM range");
    TO_CHECK(dt == TO_F32, TO_ERR_UNSUPPORTED, "pre-fused fp64 path: a contraction is outside the small-GEMM range");
    GemmProblem q = p;
    q.rowsum = nullptr;
    launch_gemm_mfma(q, S());
    launch_sum_axis(dt, dz[l].t->ptr, gb[l]->ptr, 1, B, w[l]->dims[0], 0, w[l]->dims[0], 1, S());
  }
}

to_status to_fflayer_stack_grad(int n_layers, const to_tensor* w, const to_tensor* b, int hidden_act,
                                int out_act, int loss, to_tensor x, to_tensor y, const to_tensor* gw,
                                const to_tensor* gb, to_tensor losses) {
  API_BEGIN
  fflayer_stack_impl(n_layers, w, b, hidden_act, out_act, loss, x, y, gw, gb, losses, false, 0.0);
  API_END
}

to_status to_fflayer_stack_sgd(int n_layers, const to_tensor* w, const to_tensor* b, int hidden_act, int out_act,
                               int loss, to_tensor x, to_tensor y, double rate, to_tensor losses) {
  API_BEGIN
  fflayer_stack_impl(n_layers, w, b, hidden_act, out_act, loss, x, y, w, b, losses, true, rate);
  API_END
}

// `foldl' trainNetwork` over samples (app/MNIST.hs:390-396) of an ffLayer stack as ONE persistent launch.
static void online_sgd_impl(int n_layers, const to_tensor* w, const to_tensor* b, int hidden_act, int out_act, int loss,
                            to_tensor X, to_tensor Y, int64_t n_idx, const int64_t* idx, double rate) {
  require_init();
  no_capture("to_fflayer_stack_online_sgd");
  NONNULL(w); NONNULL(b); NONNULL(X); NONNULL(Y);
  TO_CHECK(n_layers >= 2 && n_layers <= 6, TO_ERR_UNSUPPORTED, "online SGD kernel: 2..6 layers");
  TO_CHECK(hidden_act == TO_ACT_LOGISTIC, TO_ERR_UNSUPPORTED, "online SGD kernel: hidden activation must be logistic");
  const bool sm_ce = out_act == TO_ACT_SOFTMAX && loss == TO_LOSS_CROSS_ENTROPY;
  const bool lg_se = out_act == TO_ACT_LOGISTIC && loss == TO_LOSS_SQUARED_ERROR;
  TO_CHECK(sm_ce || lg_se, TO_ERR_UNSUPPORTED, "online SGD kernel: (softmax, crossEntropy) or (logistic, squaredError) only");
  ensure(X);
  ensure(Y);
  TO_CHECK(X->rank == 1 && Y->rank == 1 && X->batch > 0 && X->batch == Y->batch && X->contiguous() && Y->contiguous(),
           TO_ERR_SHAPE, "X and Y must be contiguous batched vectors of one batch, got " + shape_str(X) + " " + shape_str(Y));
  const int dt = X->dtype;
  TO_CHECK(Y->dtype == dt, TO_ERR_ARG, "X and Y have different dtypes");
  TO_CHECK(n_idx >= 0, TO_ERR_ARG, "negative sample count");
  int64_t dims[8];
  dims[0] = X->dims[0];
  void *wp[6], *bp[6];
  for (int l = 0; l < n_layers; ++l) {
    NONNULL(w[l]); NONNULL(b[l]);
    ensure(w[l]);
    ensure(b[l]);
    TO_CHECK(w[l]->dtype == dt && b[l]->dtype == dt, TO_ERR_ARG, "parameters and data must share one dtype");
    TO_CHECK(w[l]->rank == 2 && w[l]->batch == 0 && w[l]->dims[1] == dims[l] && w[l]->contiguous(), TO_ERR_SHAPE,
             "layer " + std::to_string(l) + ": W has shape " + shape_str(w[l]));
    TO_CHECK(b[l]->rank == 1 && b[l]->batch == 0 && b[l]->dims[0] == w[l]->dims[0] && b[l]->contiguous(), TO_ERR_SHAPE,
             "layer " + std::to_string(l) + ": b has shape " + shape_str(b[l]));
    dims[l + 1] = w[l]->dims[0];
  }
  TO_CHECK(Y->dims[0] == dims[n_layers], TO_ERR_SHAPE, "Y does not match the output layer");
  int G = 0, rpw = 0;
  size_t lds = 0;
  TO_CHECK(online_sgd_plan(dt, n_layers, dims, &G, &rpw, &lds), TO_ERR_UNSUPPORTED,
           "online SGD kernel: the stack does not fit (input <= 2048, head <= 64 outputs, 160 KiB of LDS per workgroup)");
  for (int64_t k = 0; k < n_idx; ++k)
    TO_CHECK(!idx || (idx[k] >= 0 && idx[k] < X->batch), TO_ERR_SHAPE, "sample index out of range");
  TO_CHECK(idx || n_idx <= X->batch, TO_ERR_SHAPE, "more samples than rows");
  for (int l = 0; l < n_layers; ++l) {  // in-place writes: recorded readers of the old values first, new identities after
    before_write(w[l]);
    before_write(b[l]);
    w[l]->id = fresh_id();
    b[l]->id = fresh_id();
    wp[l] = w[l]->ptr;
    bp[l] = b[l]->ptr;
  }
  if (n_idx == 0) return;
  Holder order;
  const long long* idx_dev = nullptr;
  if (idx) {
    const int64_t nl = (n_idx * 8 + 3) / 4;
    order.t = new_tensor(1, &nl, 0);
    host_to_device(order.t->ptr, idx, (size_t)n_idx * sizeof(int64_t), S());
    idx_dev = static_cast<const long long*>(order.t->ptr);
  }
  online_sgd_reset_status();
  launch_online_sgd(dt, n_layers, dims, wp, bp, X->ptr, Y->ptr, idx_dev, n_idx, rate, sm_ce ? 1 : 2, S());
  TO_HIP(hipStreamSynchronize(S()));  // the order buffer goes back to the pool; the watchdog's verdict is read
  TO_CHECK(online_sgd_status() == 0, TO_ERR_HIP,
           "online SGD kernel: a workgroup barrier timed out at sample " + std::to_string(online_sgd_status() - 1) +
               " (the write-back is all-or-nothing: commit or abort is ONE word decided by compare-and-swap and obeyed by every workgroup; this run aborted, the parameters are unchanged)");
}

to_status to_fflayer_stack_online_sgd(int n_layers, const to_tensor* w, const to_tensor* b, int hidden_act, int out_act,
                                      int loss, to_tensor X, to_tensor Y, int64_t n_idx, const int64_t* idx_or_null,
                                      double rate) {
  API_BEGIN
  online_sgd_impl(n_layers, w, b, hidden_act, out_act, loss, X, Y, n_idx, idx_or_null, rate);
  API_END
}

// Is the captured step the trainNetwork step of an ffLayer stack on ONE sample?  Reads the launches the planner made of it:
//   forward   l = 1..L-1 : a_l = logistic(W_l a_{l-1} + b_l)              (GEMV, bias + activation in the epilogue)
//             l = L      : z_L = W_L a_{L-1} + b_L -> loss head -> dz_L   [-> tail: dz_{L-1}]
//   backward  l = ..1    : dz_l = (W_{l+1}^T dz_{l+1}) a_l (1 - a_l)
//   update               : W_l += alpha dz_l (x) a_{l-1}, b_l += alpha dz_l for every layer, in place, one launch
// with every pointer chaining into the next launch.  Returns the stack, or false.
static int64_t g_online_runs = 0, g_online_samples = 0;  // to_online_sgd_stats
struct OnlineForm {
  int dtype = TO_F32;
  int L = 0;
  int64_t dims[8] = {0};
  void* W[6] = {nullptr};
  void* b[6] = {nullptr};
  const void* x = nullptr;
  const void* y = nullptr;
  double rate = 0.0;
  int head = 0;
};
#define NOPE                                                                                              \
  do {                                                                                                    \
    if (ab_getenv("TOPS_ONLINE_DEBUG")) std::fprintf(stderr, "[online] not an ffLayer step: check at line %d\n", __LINE__); \
    return false;                                                                                         \
  } while (0)
static bool online_form_of(const to_graph_s& g, OnlineForm& f) {
  const auto& D = g.desc;
  if (D.size() < 3 || D.back().kind != 1) NOPE;
  const StepDesc& up = D.back();
  const int L = up.n;
  if (L < 2 || L > 6) NOPE;
  f.dtype = up.p.dtype;
  for (size_t i = 0; i + 1 < D.size(); ++i)
    if (D[i].kind != 0) NOPE;
  if (!g.replay_list || g.launches.size() != D.size()) NOPE;  // nothing else was captured
  if ((int)D.size() - 1 < L) NOPE;
  // forward chain
  const void* prev = nullptr;
  for (int l = 0; l < L; ++l) {
    const GemmProblem& p = D[(size_t)l].p;
    if (p.dtype != f.dtype || p.M != 1 || p.batch != 1 || p.reduce_batch || p.alpha != 1.0 || p.beta != 0.0 || p.dact ||
        p.rowsum || !p.bias || (p.a_sk != 1 && p.K != 1) || (p.b_sk != 1 && p.K != 1) || (p.b_sn != p.K && p.N != 1))
      NOPE;
    if (l == 0) f.x = p.A;
    else if (p.A != prev) NOPE;
    if (l > 0 && p.K != f.dims[l]) NOPE;
    f.dims[l] = p.K;
    f.dims[l + 1] = p.N;
    f.W[l] = const_cast<void*>(p.B);
    f.b[l] = const_cast<void*>(p.bias);
    if (l + 1 < L) {
      if (p.act != 1 || p.loss_rows) NOPE;
    } else {
      if (p.act != 0 || (p.loss_rows != 1 && p.loss_rows != 2) || !p.target) NOPE;
      f.head = p.loss_rows;
      f.y = p.target;
    }
    prev = p.C;
  }
  // backward chain: dzs[l] = cotangent of layer l's pre-activation (0-based)
  const void* dzs[6] = {nullptr};
  const void* acts[6] = {nullptr};  // acts[l] = output of layer l (input of layer l+1)
  for (int l = 0; l + 1 < L; ++l) acts[l] = D[(size_t)l].p.C;
  const GemmProblem& last = D[(size_t)L - 1].p;
  dzs[L - 1] = last.C;
  int next_l = L - 2;  // the next cotangent to find
  size_t i = (size_t)L;
  if (last.tail_out) {
    if (last.tail_w != f.W[L - 1] || last.tail_h != acts[L - 2] || last.tail_n != f.dims[L - 1]) NOPE;
    dzs[L - 2] = last.tail_out;
    next_l = L - 3;
  }
  for (; next_l >= 0; --next_l, ++i) {
    if (i + 1 >= D.size()) NOPE;
    const GemmProblem& p = D[i].p;
    // dz_l [1 x o_l] = dz_{l+1} [1 x o_{l+1}] . W_{l+1} [o_{l+1} x o_l], times a_l (1 - a_l)
    if (p.dtype != f.dtype || p.M != 1 || p.batch != 1 || p.reduce_batch || p.alpha != 1.0 || p.beta != 0.0 || p.bias ||
        p.act || p.dact_kind || p.rowsum || p.loss_rows || (p.a_sk != 1 && p.K != 1) || (p.b_sn != 1 && p.N != 1) || (p.b_sk != p.N && p.K != 1))
      NOPE;
    if (p.A != dzs[next_l + 1] || p.B != f.W[next_l + 1] || p.dact != acts[next_l] || p.N != f.dims[next_l + 1] ||
        p.K != f.dims[next_l + 2])
      NOPE;
    dzs[next_l] = p.C;
  }
  if (i + 1 != D.size()) NOPE;
  // the update launch: every layer once, in place, one rate
  bool seen[6] = {false};
  for (int k = 0; k < L; ++k) {
    int l = -1;
    for (int q = 0; q < L; ++q)
      if (up.w[k] == f.W[q]) l = q;
    if (l < 0 || seen[l]) NOPE;
    seen[l] = true;
    if (up.w_in[k] != up.w[k] || up.b[k] != f.b[l] || up.b_in[k] != up.b[k] || up.dz[k] != dzs[l] ||
        up.a[k] != (l == 0 ? f.x : acts[l - 1]) || up.rows[k] != f.dims[l + 1] || up.cols[k] != f.dims[l] ||
        up.alpha[k] != up.alpha[0])
      NOPE;
  }
  f.L = L;
  f.rate = -up.alpha[0];
  return true;
}
#undef NOPE

to_status to_graph_online_sgd(to_graph g, to_tensor x_buf, to_tensor y_buf, to_tensor X, to_tensor Y, int64_t n_idx,
                              const int64_t* idx_or_null, int* handled) {
  API_BEGIN
  require_init();
  NONNULL(g); NONNULL(x_buf); NONNULL(y_buf); NONNULL(X); NONNULL(Y); NONNULL(handled);
  no_capture("to_graph_online_sgd");
  *handled = 0;
  const char* en = getenv("TOPS_ONLINE_KERNEL");  // (read per call: a host can compare the two ways in one process)
  const int enable = en ? atoi(en) : 1;
  OnlineForm f;
  if (!enable || !online_form_of(*g, f)) return TO_OK;
  ensure(x_buf); ensure(y_buf); ensure(X); ensure(Y);
  if (f.x != x_buf->ptr || f.y != y_buf->ptr) return TO_OK;
  if (X->dtype != f.dtype || Y->dtype != f.dtype || X->rank != 1 || Y->rank != 1 || X->batch < 1 || X->batch != Y->batch ||
      !X->contiguous() || !Y->contiguous() || X->dims[0] != f.dims[0] || Y->dims[0] != f.dims[f.L] || n_idx < 0)
    return TO_OK;
  int G = 0, rpw = 0;
  size_t lds = 0;
  if (!online_sgd_plan(f.dtype, f.L, f.dims, &G, &rpw, &lds)) return TO_OK;
  if (!online_sgd_placement_ok(S())) return TO_OK;   // (the captured step is replayed per sample instead)
  for (int64_t k = 0; k < n_idx; ++k)
    TO_CHECK(!idx_or_null || (idx_or_null[k] >= 0 && idx_or_null[k] < X->batch), TO_ERR_SHAPE, "sample index out of range");
  TO_CHECK(idx_or_null || n_idx <= X->batch, TO_ERR_SHAPE, "more samples than rows");
  lazy_flush_all();  // like a replay: the parameter buffers are about to change, recorded readers come first
  if (n_idx > 0) {
    Holder order;
    const long long* idx_dev = nullptr;
    if (idx_or_null) {
      const int64_t nl = (n_idx * 8 + 3) / 4;
      order.t = new_tensor(1, &nl, 0);
      host_to_device(order.t->ptr, idx_or_null, (size_t)n_idx * sizeof(int64_t), S());
      idx_dev = static_cast<const long long*>(order.t->ptr);
    }
    online_sgd_reset_status();
    launch_online_sgd(f.dtype, f.L, f.dims, f.W, f.b, X->ptr, Y->ptr, idx_dev, n_idx, f.rate, f.head, S());
    TO_HIP(hipStreamSynchronize(S()));
    TO_CHECK(online_sgd_status() == 0, TO_ERR_HIP,
             "online SGD kernel: a workgroup barrier timed out at sample " + std::to_string(online_sgd_status() - 1) +
                 " (the write-back is all-or-nothing: commit or abort is ONE word decided by compare-and-swap and obeyed by every workgroup; this run aborted, the parameters are unchanged)");
  }
  *handled = 1;
  g_online_runs++;
  g_online_samples += n_idx;
  API_END
}

to_status to_online_sgd_stats(int64_t* runs, int64_t* samples) {
  API_BEGIN
  if (runs) *runs = g_online_runs;
  if (samples) *samples = g_online_samples;
  API_END
}

to_status to_comm_unique_id(void* out_128_bytes) {
  API_BEGIN
  NONNULL(out_128_bytes);
  comm_unique_id(out_128_bytes);
  API_END
}

to_status to_comm_init(int rank, int world, const void* id_128_bytes) {
  API_BEGIN
  require_init();
  NONNULL(id_128_bytes);
  comm_init(rank, world, id_128_bytes);
  API_END
}

to_status to_comm_allreduce_sum(to_tensor t) {
  API_BEGIN
  require_init();
  NONNULL(t);
  ensure(t);
  before_write(t);
  t->id = fresh_id();
  comm_allreduce_sum(t, S());
  API_END
}

to_status to_comm_world(int* world) {
  API_BEGIN
  NONNULL(world);
  *world = comm_world();
  API_END
}

to_status to_comm_shutdown(void) {
  API_BEGIN
  comm_shutdown();
  API_END
}

// ---- one-shot peer-to-peer all-reduce (p2p.hip) ----------------------------------------------------------------
to_status to_p2p_create(int64_t max_elems, int dtype, int world, void* out_ipc_handle_64_bytes) {
  API_BEGIN
  require_init();
  NONNULL(out_ipc_handle_64_bytes);
  check_dtype(dtype);
  no_capture("to_p2p_create");
  p2p_create(max_elems, dtype, world, out_ipc_handle_64_bytes);
  API_END
}

to_status to_p2p_connect(int rank, const void* handles_world_x_64_bytes) {
  API_BEGIN
  require_init();
  NONNULL(handles_world_x_64_bytes);
  p2p_connect(rank, handles_world_x_64_bytes);
  API_END
}

to_status to_p2p_allreduce_sum(to_tensor g) {
  API_BEGIN
  require_init();
  NONNULL(g);
  ensure(g);
  before_write(g);
  g->id = fresh_id();
  p2p_allreduce(g, nullptr, 0.0, true, S());
  API_END
}

to_status to_p2p_allreduce_sgd(to_tensor p, to_tensor g, double rate, int also_write_g) {
  API_BEGIN
  require_init();
  NONNULL(p); NONNULL(g);
  ensure(p);
  ensure(g);
  before_write(p);
  p->id = fresh_id();
  if (also_write_g) {
    before_write(g);
    g->id = fresh_id();
  }
  p2p_allreduce(g, p, rate, also_write_g != 0, S());
  API_END
}

to_status to_p2p_status(int* world, int* timed_out_code) {
  API_BEGIN
  if (world) *world = p2p_world();
  if (timed_out_code) *timed_out_code = p2p_status();
  API_END
}

to_status to_p2p_shutdown(void) {
  API_BEGIN
  p2p_shutdown();
  API_END
}

to_status to_timer_start(void) {
  API_BEGIN
  require_init();
  TO_HIP(hipEventRecord(rt().ev0, S()));
  API_END
}

to_status to_timer_stop(float* ms) {
  API_BEGIN
  require_init();
  NONNULL(ms);
  TO_HIP(hipEventRecord(rt().ev1, S()));
  TO_HIP(hipEventSynchronize(rt().ev1));
  TO_HIP(hipEventElapsedTime(ms, rt().ev0, rt().ev1));
  API_END
}

}  // extern "C"
