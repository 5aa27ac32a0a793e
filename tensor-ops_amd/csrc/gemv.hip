// matVec / vecMat / outer products / tall column sums at the HBM roofline (round 6, last).
//
// The reference's per-sample hot path is `matVec` / `vecMat` / `outer` (TOp.hs:56-94 on one sample; BLAS.hs:90-173 `gemv`, `ger`)
// and its batched form ends in a column sum (the bias gradient).  As GEMMs with N = 1, M = 1 or K = 1 they were served by the
// small-GEMM kernel (one workgroup per 32x32 tile: fine up to a few hundred thousand matrix elements, 4.5 us) and beyond it by
// whatever took slivers -- tools/ops_scan.py against torch.mv / torch.outer / torch.sum (profiles/r06_ops_scan_before.txt):
// matVec 4096 x 4096 25.7 us (torch 14.7), 10000 x 60000 9.2 ms (0.64), 100 x 60000 5.5 ms (0.026: M N < 256 fell to the
// one-thread-per-element kernel), vecMat 60000 x 10000 23 ms (1.3), outerV 10000 x 60000 1.42 ms (0.49), sumRows 60000 x 10000
// 5.7 ms (0.44).  All of these move every matrix element once and do one multiply-add on it: the bound is HBM.
//
// Two access patterns cover them.  The matrix is Mat[o * os + r * rs] (o: output index, r: reduction index), one stride is 1:
//   * rs == 1 (a row per output, contiguous along the reduction) -- `gemv_rows_kernel`: LPR lanes per row (64, or 16 for short
//     rows), 16-byte loads, a shuffle reduction;
//   * os == 1 (outputs contiguous, the reduction strides over rows) -- `gemv_cols_kernel`: a thread owns four consecutive outputs
//     and walks the rows of its share; the vector element is uniform.  With no vector at all this is the column sum of `sumRows`
//     and of the bias gradient.
// Few outputs under a long reduction: the reduction is split over blockIdx.y, partial sums go to a [splits][OUT] workspace and
// `gemv_finish_kernel` adds them in split order -- deterministic, no atomics.  alpha, beta * Cin, the bias and the activation of
// the fused layer epilogue are applied where the sum is complete.
// `outer_kernel`: C[m][n] = alpha a[m] b[n] (K = 1), four columns a thread, 16-byte stores, a few rows per workgroup; with the rank
// as a template parameter (8 or 16 rows of b in registers) also the rank-2 .. 15 update below the MFMA kernels' K of 16.
#include "common.hpp"

namespace to {

template <class S> struct V4 { typedef S type __attribute__((ext_vector_type(4))); };

struct GemvArgs {
  const void* mat;
  const void* vec;      // null: all ones (a plain sum)
  void* out;
  const void* out_in;   // beta * out_in[o * ys]
  const void* bias;
  long OUT, RED, os, rs, vs, ys;
  double alpha, beta;
  int act, bias_per_out;
  long red_per;         // reduction elements per split
  void* part;           // [gridDim.y][OUT] when the reduction is split
};

template <class S>
__device__ __forceinline__ S gemv_epilogue(const GemvArgs& g, S acc, long o) {
  S v = (S)g.alpha * acc;
  if (g.beta != 0.0) v += (S)g.beta * static_cast<const S*>(g.out_in)[o * g.ys];
  if (g.bias) v += static_cast<const S*>(g.bias)[g.bias_per_out ? o : 0];
  if (g.act == 1) v = S(1) / (S(1) + exp(-v));
  else if (g.act == 2) v = tanh(v);
  return v;
}

// rs == 1.  LPR lanes per row; VEC: 16-byte loads (rows and the vector 16-byte aligned, os % 4 == 0, vs == 1)
template <class S, int LPR, bool VEC>
__global__ __launch_bounds__(256) void gemv_rows_kernel(GemvArgs g) {
  typedef typename V4<S>::type S4;
  constexpr int RPB = 256 / LPR;   // rows per workgroup
  const int sub = threadIdx.x / LPR, l = threadIdx.x % LPR;
  const long o = (long)blockIdx.x * RPB + sub;
  if (o >= g.OUT) return;
  const long r0 = (long)blockIdx.y * g.red_per, r1 = r0 + g.red_per < g.RED ? r0 + g.red_per : g.RED;
  const S* __restrict__ row = static_cast<const S*>(g.mat) + o * g.os;
  const S* __restrict__ v = static_cast<const S*>(g.vec);
  S acc = S(0);
  if constexpr (VEC) {
    S4 a4 = {S(0), S(0), S(0), S(0)};
    long r = r0 + 4 * l;
    for (; r + 12 * LPR + 4 <= r1; r += 16 * LPR) {   // four loads in flight
      const S4 m0 = *reinterpret_cast<const S4*>(row + r), m1 = *reinterpret_cast<const S4*>(row + r + 4 * LPR);
      const S4 m2 = *reinterpret_cast<const S4*>(row + r + 8 * LPR), m3 = *reinterpret_cast<const S4*>(row + r + 12 * LPR);
      if (v) {
        a4 += m0 * *reinterpret_cast<const S4*>(v + r);
        a4 += m1 * *reinterpret_cast<const S4*>(v + r + 4 * LPR);
        a4 += m2 * *reinterpret_cast<const S4*>(v + r + 8 * LPR);
        a4 += m3 * *reinterpret_cast<const S4*>(v + r + 12 * LPR);
      } else {
        a4 += m0;
        a4 += m1;
        a4 += m2;
        a4 += m3;
      }
    }
    for (; r + 4 * LPR + 4 <= r1; r += 8 * LPR) {   // two loads in flight
      const S4 m0 = *reinterpret_cast<const S4*>(row + r), m1 = *reinterpret_cast<const S4*>(row + r + 4 * LPR);
      if (v) {
        a4 += m0 * *reinterpret_cast<const S4*>(v + r);
        a4 += m1 * *reinterpret_cast<const S4*>(v + r + 4 * LPR);
      } else {
        a4 += m0;
        a4 += m1;
      }
    }
    for (; r + 4 <= r1; r += 4 * LPR) {
      const S4 m0 = *reinterpret_cast<const S4*>(row + r);
      a4 += v ? m0 * *reinterpret_cast<const S4*>(v + r) : m0;
    }
    acc = (a4.x + a4.y) + (a4.z + a4.w);
    // (red_per is a multiple of 4 for every split but the last: the ragged end, at most three elements, on the lanes it falls to)
    const long tail = r0 + ((r1 - r0) & ~3L) + l;
    if (l < 4 && tail < r1) acc += v ? row[tail] * v[tail] : row[tail];
  } else {
    for (long r = r0 + l; r < r1; r += LPR) acc += v ? row[r] * v[r * g.vs] : row[r];
  }
#pragma unroll
  for (int d = LPR / 2; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if (l == 0) {
    if (gridDim.y > 1) static_cast<S*>(g.part)[(long)blockIdx.y * g.OUT + o] = acc;
    else static_cast<S*>(g.out)[o * g.ys] = gemv_epilogue<S>(g, acc, o);
  }
}

// os == 1.  W outputs a thread (4: 16-byte loads -- rs % 4 == 0, the matrix 16-byte aligned; 1 otherwise).  The workgroup is CT
// column-threads x 256 / CT row-lanes (CT a power of two, at most 256): with few outputs the other threads take more rows of the
// share, and the row-lanes' sums are added in lane order through LDS.
template <class S, int W>
__global__ __launch_bounds__(256) void gemv_cols_kernel(GemvArgs g, int ct_log2) {
  typedef typename V4<S>::type S4;
  __shared__ __attribute__((aligned(16))) S red[256 * W];
  const int CT = 1 << ct_log2, RL = 256 >> ct_log2;
  const int cx = threadIdx.x & (CT - 1), ry = threadIdx.x >> ct_log2;
  const long o = ((long)blockIdx.x * CT + cx) * W;
  const bool live = o < g.OUT;
  const long r0 = (long)blockIdx.y * g.red_per, r1 = r0 + g.red_per < g.RED ? r0 + g.red_per : g.RED;
  const S* __restrict__ p = static_cast<const S*>(g.mat) + (live ? o : 0);
  const S* __restrict__ v = static_cast<const S*>(g.vec);
  S acc[W];
  if constexpr (W == 4) {
    S4 a0 = {S(0), S(0), S(0), S(0)}, a1 = a0, a2 = a0, a3 = a0;
    if (live) {
      long r = r0 + ry;
      for (; r + 3 * RL < r1; r += 4 * RL) {   // four rows in flight
        const S4 m0 = *reinterpret_cast<const S4*>(p + r * g.rs), m1 = *reinterpret_cast<const S4*>(p + (r + RL) * g.rs);
        const S4 m2 = *reinterpret_cast<const S4*>(p + (r + 2 * RL) * g.rs), m3 = *reinterpret_cast<const S4*>(p + (r + 3 * RL) * g.rs);
        if (v) {
          a0 += m0 * v[r * g.vs]; a1 += m1 * v[(r + RL) * g.vs]; a2 += m2 * v[(r + 2 * RL) * g.vs]; a3 += m3 * v[(r + 3 * RL) * g.vs];
        } else {
          a0 += m0; a1 += m1; a2 += m2; a3 += m3;
        }
      }
      for (; r < r1; r += RL) {
        const S4 m0 = *reinterpret_cast<const S4*>(p + r * g.rs);
        a0 += v ? m0 * v[r * g.vs] : m0;
      }
    }
    const S4 t = (a0 + a1) + (a2 + a3);
    acc[0] = t.x; acc[1] = t.y; acc[2] = t.z; acc[3] = t.w;
  } else {
    S a0 = S(0), a1 = S(0);
    if (live) {
      long r = r0 + ry;
      for (; r + RL < r1; r += 2 * RL) {
        const S m0 = p[r * g.rs], m1 = p[(r + RL) * g.rs];
        a0 += v ? m0 * v[r * g.vs] : m0;
        a1 += v ? m1 * v[(r + RL) * g.vs] : m1;
      }
      if (r < r1) a0 += v ? p[r * g.rs] * v[r * g.vs] : p[r * g.rs];
    }
    acc[0] = a0 + a1;
  }
  if (RL > 1) {   // (uniform)
#pragma unroll
    for (int e = 0; e < W; ++e) red[(ry * CT + cx) * W + e] = acc[e];
    __syncthreads();
    if (ry != 0) return;
    for (int q = 1; q < RL; ++q)
#pragma unroll
      for (int e = 0; e < W; ++e) acc[e] += red[(q * CT + cx) * W + e];
  }
  if (!live) return;
  if (gridDim.y > 1) {
#pragma unroll
    for (int e = 0; e < W; ++e) static_cast<S*>(g.part)[(long)blockIdx.y * g.OUT + o + e] = acc[e];   // (OUT % 4 == 0 with W == 4)
  } else {
    S* out = static_cast<S*>(g.out);
#pragma unroll
    for (int e = 0; e < W; ++e) out[(o + e) * g.ys] = gemv_epilogue<S>(g, acc[e], o + e);
  }
}

// out[o] = epilogue(sum over the splits, in split order)
template <class S>
__global__ __launch_bounds__(256) void gemv_finish_kernel(GemvArgs g, int splits) {
  const long o = (long)blockIdx.x * 256 + threadIdx.x;
  if (o >= g.OUT) return;
  const S* part = static_cast<const S*>(g.part);
  S acc = S(0);
  int sp = 0;
  for (; sp + 8 <= splits; sp += 8) {   // (eight loads in flight; the order of the additions is fixed)
    S t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = part[(long)(sp + e) * g.OUT + o];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += t[e];
  }
  for (; sp < splits; ++sp) acc += part[(long)sp * g.OUT + o];
  static_cast<S*>(g.out)[o * g.ys] = gemv_epilogue<S>(g, acc, o);
}

struct OuterArgs {
  const void* a; const void* b; void* C; const void* Cin; const void* bias;
  long M, N, a_sm, a_sk, b_sk, b_sn, c_sm;
  double alpha, beta;
  int act, rows, K;   // rows per workgroup; K <= 16: the rank of the update
};
// C[m][n] = alpha sum_{k < K} a[m][k] b[k][n] (+ beta Cin + bias[n], activation), K <= 16: an outer product (K = 1) or the weight
// gradient of a minibatch of a few samples.  A thread owns W consecutive columns and keeps b[.][n .. n + W) in registers; the rows
// of its share pass by once -- the bound is the store of C.
template <class S, int W, int KB>   // KB: 1, 8 or 16 -- the rank rounded up (the b rows beyond K are zero)
__global__ __launch_bounds__(256) void outer_kernel(OuterArgs g) {
  typedef typename V4<S>::type S4;
  const long n = ((long)blockIdx.x * 256 + threadIdx.x) * W;
  if (n >= g.N) return;
  const S* a = static_cast<const S*>(g.a);
  const S* b = static_cast<const S*>(g.b);
  S* C = static_cast<S*>(g.C);
  const long m0 = (long)blockIdx.y * g.rows, m1 = m0 + g.rows < g.M ? m0 + g.rows : g.M;
  S bv[KB][W], bi[W];
#pragma unroll
  for (int k = 0; k < KB; ++k)
#pragma unroll
    for (int e = 0; e < W; ++e) bv[k][e] = k < g.K ? (S)g.alpha * b[k * g.b_sk + (n + e) * g.b_sn] : S(0);
  long ak[KB];   // (element offsets of a row's K values; beyond K: the first one again, times a zero)
#pragma unroll
  for (int k = 0; k < KB; ++k) ak[k] = (k < g.K ? k : 0) * g.a_sk;
#pragma unroll
  for (int e = 0; e < W; ++e) bi[e] = g.bias ? static_cast<const S*>(g.bias)[n + e] : S(0);
  const bool plain = g.beta == 0.0 && g.act == 0;
  for (long m = m0; m < m1; ++m) {
    S v[W];
#pragma unroll
    for (int e = 0; e < W; ++e) v[e] = bi[e];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const S am = a[m * g.a_sm + ak[k]];
#pragma unroll
      for (int e = 0; e < W; ++e) v[e] += am * bv[k][e];
    }
    if (!plain) {
#pragma unroll
      for (int e = 0; e < W; ++e) {
        if (g.beta != 0.0) v[e] += (S)g.beta * static_cast<const S*>(g.Cin)[m * g.c_sm + n + e];
        if (g.act == 1) v[e] = S(1) / (S(1) + exp(-v[e]));
        else if (g.act == 2) v[e] = tanh(v[e]);
      }
    }
    if constexpr (W == 4) {
      const S4 o = {v[0], v[1], v[2], v[3]};
      *reinterpret_cast<S4*>(C + m * g.c_sm + n) = o;   // (plain stores: nontemporal ones measured 5.3 TB/s against 5.6)
    } else {
      C[m * g.c_sm + n] = v[0];
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------

static bool gemv_enabled() {
  static const bool on = [] { const char* e = getenv("TOPS_GEMV"); return !(e && e[0] == '0'); }();   // product switch: TOPS_GEMV=0
  return on;
}

static bool gemv_plain(const GemmProblem& p) {
  return (p.dtype == TO_F32 || p.dtype == TO_F64) && p.batch == 1 && !p.reduce_batch && !p.rowsum && !p.loss_rows && !p.dact &&
         !p.tail_out && !p.a_table && p.M >= 1 && p.N >= 1 && p.K >= 1 && (p.beta == 0.0 || p.Cin) && p.act >= 0 && p.act <= 2;
}

// 0: not here; 1: matVec / vecMat; 2: outer product
int gemv_form(const GemmProblem& p, bool standalone) {
  if (!gemv_enabled() || !gemv_plain(p)) return 0;
  if (p.K == 1 && p.M > 1 && p.N > 1) return p.M * p.N >= (1 << 20) ? 2 : 0;   // (below ~1M elements the launch is the cost either way)
  // (a rank-2 .. 15 update of a large matrix -- the weight gradient of a minibatch of eight: below the wave-split kernel's K of 16,
  //  and all store: 60000 x 8 x 10000 0.80 ms on the old 64x64 body, vendor 0.53)
  if (p.K >= 2 && p.K < 16 && p.M >= 256 && p.N >= 256 && p.M * p.N >= (1 << 22)) return 2;
  if (p.N != 1 && p.M != 1) return 0;
  if (p.M == 1 && p.N == 1) return p.K >= (1 << 16) && p.a_sk == 1 ? 1 : 0;   // (a long dot product)
  const int64_t out = p.N == 1 ? p.M : p.N, os = p.N == 1 ? p.a_sm : p.b_sn, rs = p.N == 1 ? p.a_sk : p.b_sk;
  if (os != 1 && rs != 1) return 0;
  if (p.N == 1 && p.bias) return 0;   // (the bias is per column: a scalar here -- left to the small-GEMM kernel)
  // the small-GEMM kernel holds these up to ~1M matrix elements with a reduction of a few thousand (4.5 us: a launch); beyond,
  // and wherever M N < 256 would have meant one thread per output walking K alone, here
  static const long min_elems = [] { const char* e = ab_getenv("TOPS_GEMV_MIN"); return e ? atol(e) : (1L << 20); }();   // (development knob)
  // (a reduction of a hundred: the small-GEMM kernel's K split has nothing to split -- m x 100 . 100: 14 us there, 4-5 here)
  if (out * p.K >= min_elems || p.K >= 2048) return 1;
  // ... and, for a product that stands alone (run_gemm; inside a fused group the planner hands these to the small-GEMM kernel
  // with their neighbours): fewer than 256 outputs, which would have meant one thread per output walking K alone, and a
  // reduction of a hundred
  return standalone && ((out < 256 && p.K >= 32) || (p.K <= 128 && out >= 256 && out * p.K >= (1 << 14))) ? 1 : 0;
}

template <class S>
static void gemv_launch(GemvArgs g, hipStream_t s) {
  const int dtype = sizeof(S) == 8 ? TO_F64 : TO_F32;
  const bool al16 = [&](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }(g.mat);
  const long target_waves = 256L * 16;   // enough waves in flight to cover the HBM latency
  Holder tmp;
  if (g.rs == 1) {
    const bool vec = al16 && g.os % 4 == 0 && (!g.vec || (g.vs == 1 && (reinterpret_cast<uintptr_t>(g.vec) & 15u) == 0));
    const int lpr = g.RED <= 512 ? 16 : 64;
    const long rows_per_block = 256 / lpr, blocks = (g.OUT + rows_per_block - 1) / rows_per_block;
    long splits = 1;
    const long waves = (g.OUT * lpr + 63) / 64;
    if (waves < target_waves && g.RED >= 8192) {
      splits = (target_waves + waves - 1) / waves;
      if (splits > g.RED / 2048) splits = g.RED / 2048;
      if (splits > 256) splits = 256;
      if (splits < 1) splits = 1;
    }
    g.red_per = ((g.RED + splits - 1) / splits + 3) & ~3L;
    splits = (g.RED + g.red_per - 1) / g.red_per;
    if (splits > 1) {
      const int64_t pd[2] = {splits, g.OUT};
      tmp.t = new_tensor(2, pd, 0, dtype);
      g.part = tmp.t->ptr;
    }
    const dim3 grid((unsigned)blocks, (unsigned)splits);
    if (lpr == 16) {
      if (vec) launch_k((gemv_rows_kernel<S, 16, true>), grid, dim3(256), 0, s, g);
      else launch_k((gemv_rows_kernel<S, 16, false>), grid, dim3(256), 0, s, g);
    } else {
      if (vec) launch_k((gemv_rows_kernel<S, 64, true>), grid, dim3(256), 0, s, g);
      else launch_k((gemv_rows_kernel<S, 64, false>), grid, dim3(256), 0, s, g);
    }
    TO_HIP(hipGetLastError());
    count_launch();
    if (splits > 1) {
      launch_k((gemv_finish_kernel<S>), dim3((unsigned)((g.OUT + 255) / 256)), dim3(256), 0, s, g, (int)splits);
      TO_HIP(hipGetLastError());
      count_launch();
    }
    return;
  }
  // os == 1
  const bool w4 = al16 && g.rs % 4 == 0 && g.OUT % 4 == 0;
  const long threads = w4 ? (g.OUT + 3) / 4 : g.OUT;   // column-threads needed
  int ct_log2 = 8;
  while (ct_log2 > 4 && (1L << (ct_log2 - 1)) >= threads) --ct_log2;   // (16 .. 256 column-threads a workgroup)
  const long CT = 1L << ct_log2, RL = 256 >> ct_log2, blocks = (threads + CT - 1) / CT;
  long splits = 1;
  if (blocks < 1024 && g.RED >= 64 * RL) {   // (~1,024 workgroups in flight, at least sixteen rows a row-lane, at most 256 partial sums an output)
    splits = (1024 + blocks - 1) / blocks;
    if (splits > g.RED / (16 * RL)) splits = g.RED / (16 * RL);
    if (splits > 256) splits = 256;
    if (splits < 1) splits = 1;
  }
  g.red_per = (g.RED + splits - 1) / splits;
  splits = (g.RED + g.red_per - 1) / g.red_per;
  if (splits > 1) {
    const int64_t pd[2] = {splits, g.OUT};
    tmp.t = new_tensor(2, pd, 0, dtype);
    g.part = tmp.t->ptr;
  }
  const dim3 grid((unsigned)blocks, (unsigned)splits);
  if (w4) launch_k((gemv_cols_kernel<S, 4>), grid, dim3(256), 0, s, g, ct_log2);
  else launch_k((gemv_cols_kernel<S, 1>), grid, dim3(256), 0, s, g, ct_log2);
  TO_HIP(hipGetLastError());
  count_launch();
  if (splits > 1) {
    launch_k((gemv_finish_kernel<S>), dim3((unsigned)((g.OUT + 255) / 256)), dim3(256), 0, s, g, (int)splits);
    TO_HIP(hipGetLastError());
    count_launch();
  }
}

void launch_gemv(const GemmProblem& p, hipStream_t s) {
  const int form = gemv_form(p, true);
  TO_CHECK(form != 0, TO_ERR_STATE, "launch_gemv: not applicable");
  if (form == 2) {
    OuterArgs g{};
    g.a = p.A; g.b = p.B; g.C = p.C; g.Cin = p.Cin; g.bias = p.bias;
    g.M = p.M; g.N = p.N; g.a_sm = p.a_sm; g.a_sk = p.a_sk; g.b_sk = p.b_sk; g.b_sn = p.b_sn; g.c_sm = p.c_sm; g.K = (int)p.K;
    g.alpha = p.alpha; g.beta = p.beta; g.act = p.act;
    const bool w4 = p.N % 4 == 0 && p.c_sm % 4 == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15u) == 0 &&
                    (p.beta == 0.0 || (reinterpret_cast<uintptr_t>(p.Cin) & 15u) == 0);
    const long per_block = w4 ? 1024 : 256, bx = (p.N + per_block - 1) / per_block;
    long rows = p.K == 1 ? 16 : 64;
    const long min_rows = p.K == 1 ? 1 : 16;   // (a rank-K update reloads K rows of b per workgroup: at least sixteen rows of C for them)
    while (rows > min_rows && bx * ((p.M + rows - 1) / rows) < 2048) rows /= 2;
    g.rows = (int)rows;
    const dim3 grid((unsigned)bx, (unsigned)((p.M + rows - 1) / rows));
    const int kb = p.K == 1 ? 1 : p.K <= 8 ? 8 : 16;
#define OUTER_GO(S, W) do { if (kb == 1) launch_k((outer_kernel<S, W, 1>), grid, dim3(256), 0, s, g); else if (kb == 8) launch_k((outer_kernel<S, W, 8>), grid, dim3(256), 0, s, g); else launch_k((outer_kernel<S, W, 16>), grid, dim3(256), 0, s, g); } while (0)
    if (p.dtype == TO_F64) {
      if (w4) OUTER_GO(double, 4);
      else OUTER_GO(double, 1);
    } else {
      if (w4) OUTER_GO(float, 4);
      else OUTER_GO(float, 1);
    }
#undef OUTER_GO
    TO_HIP(hipGetLastError());
    count_launch();
    return;
  }
  GemvArgs g{};
  if (p.N == 1) {   // y[m] = sum_k A[m, k] b[k]
    g.mat = p.A; g.vec = p.B; g.OUT = p.M; g.RED = p.K; g.os = p.a_sm; g.rs = p.a_sk; g.vs = p.b_sk; g.ys = p.c_sm;
    g.bias_per_out = 0;
  } else {          // y[n] = sum_k a[k] B[k, n]
    g.mat = p.B; g.vec = p.A; g.OUT = p.N; g.RED = p.K; g.os = p.b_sn; g.rs = p.b_sk; g.vs = p.a_sk; g.ys = 1;
    g.bias_per_out = 1;
  }
  g.out = p.C; g.out_in = p.Cin; g.bias = p.bias;
  g.alpha = p.alpha; g.beta = p.beta; g.act = p.act;
  if (p.dtype == TO_F64) gemv_launch<double>(g, s);
  else gemv_launch<float>(g, s);
}

// out[j] = sum_i x[i * si + j], one tall matrix (sumRows, the bias gradient of a batch): the column kernel with no vector
bool launch_column_sum(int dtype, const void* x, void* out, int64_t R, int64_t J, int64_t si, hipStream_t s) {
  if (!gemv_enabled() || (dtype != TO_F32 && dtype != TO_F64) || R * J < (1 << 18)) return false;   // (any width: ten columns under a million rows -- the bias gradient of a narrow last layer -- 1.04 ms -> 12 us)
  GemvArgs g{};
  g.mat = x; g.vec = nullptr; g.out = out; g.OUT = J; g.RED = R; g.os = 1; g.rs = si; g.vs = 0; g.ys = 1;
  g.alpha = 1.0; g.beta = 0.0;
  if (dtype == TO_F64) gemv_launch<double>(g, s);
  else gemv_launch<float>(g, s);
  return true;
}

}  // namespace to
