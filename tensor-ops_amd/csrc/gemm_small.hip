// Latency-bound fp32 GEMMs (few output tiles, long K): one 32x32 output tile per
// workgroup, the K loop split across the workgroup's NW waves (one wave per SIMD and
// more), operands loaded from global memory STRAIGHT into the MFMA fragment layout
// (no LDS staging, no barrier in the loop -- the operands of these shapes are L2/MALL
// resident), partial accumulators reduced through LDS, epilogue fused.
//
// Why: at 1024x256x784 a 64x64-tile kernel has 64 workgroups; the matrix pipes of 192
// CUs idle and each wave owns 392 dependent-latency-bound MFMAs.  Spreading 256 tiles x
// 4..16 K-slices over the chip cuts the per-wave chain to <= 128 MFMAs.
//
// Serves the batched ffLayer step (config 3): X.W1^T, dZ^T.X, H.W2^T, dZ2^T.H, dZ2.W2.
#include "common.hpp"

namespace to {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <class S>
struct SmallArgsT {
  const S* A;
  const S* B;
  S* C;
  const S* Cin;
  int M, N, K;
  long a_sm, a_sk, b_sk, b_sn, c_sm;
  long a_sb, b_sb, c_sb;
  int a_vec, b_vec;  // 16-byte loads along k legal (k-contiguous modes only)
  unsigned a_bytes, b_bytes;  // extents for the buffer descriptors (hardware bounds check)
  int tiles_n;
  int tiles_m;       // (tile grid; with tile_order != 0 the workgroup -> tile map below is XCD-aware)
  int tile_order;    // 0: bid row-major.  1 / 2: XCD x (= bid % 8, where the hardware puts the workgroup) owns a
                     // contiguous run of the row-major (1) / column-major (2) tile sequence, so its private L2
                     // pulls a few A panels + all of B (1) or a few B panels + all of A (2) instead of everything
  int kper;          // k extent per wave (multiple of the chunk)
  S alpha, beta;
  const S* bias;
  const S* dact;
  int act;
  int dact_kind;
  S* rowsum;  // optional [batch][M]: sum_k A[m,k] (the bias gradient next to dW = dZ^T.X)
  const S* rowsum_in;  // rowsum_acc: rowsum = rowsum_in + rowsum_alpha * sum (rowsum itself when updating in place)
  S rowsum_alpha;  // rowsum_acc: rowsum += rowsum_alpha * sum
  int rowsum_acc;
  int loss_rows;  // TS == 16 only: the whole output row sits in 16 lanes of one wave (GemmProblem::loss_rows)
  const S* target;
  S* loss_out;
  const S* tail_w;  // GemmProblem::tail_* (behind the loss head)
  const S* tail_h;
  S* tail_out;
  int tail_n;
#ifdef TOPS_AB_KNOBS
  long long* dbg;   // development build, TOPS_SMALL_STAMPS=1|2|3: phase stamps of tile 0 (wall_clock64, 10 ns) -- see small_stamps()
#endif
};

#ifdef TOPS_AB_KNOBS
#define SM_STAMP(i) do { if (stamp_on) g.dbg[i] = wall_clock64(); } while (0)
#else
#define SM_STAMP(i) do { } while (0)
#endif


// workgroup -> tile.  The 8 XCDs each have their own L2 and workgroup b lands on XCD b % 8: with the plain
// row-major map every XCD touches every panel of both operands and the fabric carries 8 copies of them
// (config 3 forward, 1024x784x256: 26.6 MB fetched for 4 MB of operands, rocprofv3 FETCH_SIZE).
template <class G>
__device__ __forceinline__ void tile_of(const G& g, int bid, int& tile_m, int& tile_n) {
  if (g.tile_order) {
    const int T = g.tiles_m * g.tiles_n, x = bid & 7, q = T >> 3, r = T & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);   // bijective for any T
  }
  if (g.tile_order == 2) {
    tile_n = bid / g.tiles_m;
    tile_m = bid - tile_n * g.tiles_m;
  } else {
    tile_m = bid / g.tiles_n;
    tile_n = bid - tile_m * g.tiles_n;
  }
}

__device__ __forceinline__ float tanh_s(float x) { return tanhf(x); }
__device__ __forceinline__ double tanh_s(double x) { return tanh(x); }
__device__ __forceinline__ float exp_s(float x) { return expf(x); }
__device__ __forceinline__ double exp_s(double x) { return exp(x); }
__device__ __forceinline__ float log_s(float x) { return logf(x); }
__device__ __forceinline__ double log_s(double x) { return log(x); }
__device__ __forceinline__ float max_s(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double max_s(double a, double b) { return fmax(a, b); }

// AMODE 0: A k-contiguous (a_sk == 1)   1: A m-contiguous / general strides
// BMODE 0: B n-contiguous / general     1: B k-contiguous (b_sk == 1)
// Within a chunk of 8 k the MFMA j of half-wave `half` consumes k = k0 + 4*half + j, for A and B alike.
// TS = 32: v_mfma_f32_32x32x2_f32 (lane: row/col l&31, k-group l>>5 of 2)
// TS = 16: v_mfma_f32_16x16x4_f32 (lane: row/col l&15, k-group l>>4 of 4) -- for skinny outputs
//          (min(M,N) <= 16): 4x more tiles, 4x shorter MFMA chains, less padding waste
// ONESHOT = n: the wave's whole K slice (<= n chunks) is fetched by ONE batch of loads -- a single
// memory round trip instead of a chain of pipeline stages, which is what bounds these kernels
// (1024x784x256: four dependent stages of ~1.2 us each at NW = 8).
// S = double (the fp64 instance): TS = 16 only, v_mfma_f64_16x16x4_f64 -- same A/B lane mapping, but the
// accumulator register r of lane l is row (l>>4) + 4*r (fp32: 4*(l>>4) + r).
// EXT: the reduction buffers live in caller-provided LDS (the chained kernel runs several bodies in one launch
// and they share one dynamic allocation: their static arrays together would not fit)
// tid: the thread's index within the NW waves that run this body (threadIdx.x, unless a larger workgroup runs several
// bodies side by side: the seam kernel's two 8-wave head tiles inside a 16-wave workgroup)
template <class S, int AMODE, int BMODE, int NW, int TS, int ONESHOT = 0, bool EXT = false>   // ONESHOT: 0, or the chunks one batch holds
__device__ __forceinline__ void gemm_small_body(const SmallArgsT<S>& g, const int bid, const long bz, S* ext_lds = nullptr,
                                                const int tid_in = -1) {
  constexpr int ES = (int)sizeof(S);
  static_assert(ES == 4 || TS == 16, "the fp64 matrix instruction is 16x16x4");
  constexpr int KG = (TS == 32) ? 2 : 4;      // k-groups per MFMA
  constexpr int NR = (TS == 32) ? 16 : 4;     // accumulator registers
  constexpr int CK = 4 * KG;                  // k per chunk (4 MFMAs)
  typedef S accv __attribute__((ext_vector_type(NR)));
  S (*red)[NR][64];
  S (*rsum)[64];
  S* dzs;  // the tile's loss gradient, for the fused tail
  if constexpr (EXT) {
    red = reinterpret_cast<S (*)[NR][64]>(ext_lds);
    rsum = reinterpret_cast<S (*)[64]>(ext_lds + NW * NR * 64);
    dzs = ext_lds + NW * NR * 64 + NW * 64;
  } else {
    __shared__ S red_s[NW][NR][64];
    __shared__ S rsum_s[NW][64];
    __shared__ S dzs_s[TS == 16 ? 16 * 17 : 1];
    red = red_s;
    rsum = rsum_s;
    dzs = dzs_s;
  }
  const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & (TS - 1), half = lane / TS;   // half = k-group of this lane
#ifdef TOPS_AB_KNOBS
  const bool stamp_on = g.dbg && bid == 0 && tid == 0;
#endif
  SM_STAMP(0);
  int tile_m, tile_n;
  tile_of(g, bid, tile_m, tile_n);
  const long m = (long)tile_m * TS + l31, n = (long)tile_n * TS + l31;
  const bool mv = m < g.M, nv = n < g.N;

  accv acc;
#pragma unroll
  for (int r = 0; r < NR; ++r) acc[r] = S(0);
  S asum = S(0);  // sum over k of this lane's A elements (fused row sums of A = bias gradients)

  const int kbeg = wave * g.kper;
  int kend = kbeg + g.kper;
  if (kend > g.K) kend = g.K;

  // Software pipeline, two register stages of ST chunks (8 k each): the loads of stage s+1
  // are in flight while the 4*ST MFMAs of stage s run -- one wave per SIMD has no other
  // way to hide the L2/MALL latency.  (Named ping/pong buffers: a runtime-indexed register
  // array would go to scratch.)
  constexpr int ST = ONESHOT ? ONESHOT : 4;
  S a0[ST][4], b0[ST][4];
  // Buffer loads with hardware bounds checking: an out-of-range element gets the byte
  // offset 0x7fffffff (>= num_records) and the hardware returns 0 -- no branch, no select on
  // the loaded value, so the compiler cannot turn the guard into a waited conditional load.
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<S*>(g.A), 0, g.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<S*>(g.B), 0, g.b_bytes, 0x00020000);
  const int a_base = (int)((bz * g.a_sb + (mv ? m : 0) * g.a_sm) * ES);
  const int b_base = (int)((bz * g.b_sb + (nv ? n : 0) * g.b_sn) * ES);
  const int a_sk4 = (int)g.a_sk * ES, b_sk4 = (int)g.b_sk * ES;
  // arithmetic select (no short-circuit &&, no ?:) so that no control flow is generated
  auto sel = [](bool ok, int off) { const int msk = -(int)ok; return (off & msk) | (0x7fffffff & ~msk); };
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  // one element / one quad of 4 consecutive k through the bounds-checked descriptor
  auto ld1 = [&](__amdgpu_buffer_rsrc_t r, int off) -> S {
    if constexpr (ES == 4) {
      return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
    } else {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
      return __hiloint2double((int)v.y, (int)v.x);
    }
  };
  auto ld4 = [&](__amdgpu_buffer_rsrc_t r, bool ok, int off, S (&d)[4]) {
    if constexpr (ES == 4) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, sel(ok, off), 0, 0);
      d[0] = __uint_as_float(v.x); d[1] = __uint_as_float(v.y);
      d[2] = __uint_as_float(v.z); d[3] = __uint_as_float(v.w);
    } else {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, sel(ok, off), 0, 0);
      const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(r, sel(ok, off + 16), 0, 0);
      d[0] = __hiloint2double((int)v.y, (int)v.x); d[1] = __hiloint2double((int)v.w, (int)v.z);
      d[2] = __hiloint2double((int)w.y, (int)w.x); d[3] = __hiloint2double((int)w.w, (int)w.z);
    }
  };
  auto load_stage = [&](S (&a)[ST][4], S (&b)[ST][4], int k0) {
#pragma unroll
    for (int c = 0; c < ST; ++c) {
      const int kb = k0 + CK * c + 4 * half;
      if (AMODE == 0 && g.a_vec) {  // K % 4 == 0: a quad is entirely in or out of range
        ld4(ra, mv & (kb < kend), a_base + kb * ES, a[c]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[c][j] = ld1(ra, sel(mv & (kb + j < kend), a_base + (kb + j) * a_sk4));
      }
      if (BMODE == 1 && g.b_vec) {
        ld4(rb, nv & (kb < kend), b_base + kb * ES, b[c]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[c][j] = ld1(rb, sel(nv & (kb + j < kend), b_base + (kb + j) * b_sk4));
      }
    }
  };
  auto mma_stage = [&](const S (&a)[ST][4], const S (&b)[ST][4]) {
#pragma unroll
    for (int c = 0; c < ST; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (ES == 8) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c][j], b[c][j], acc, 0, 0, 0);
        else if constexpr (TS == 32) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][j], b[c][j], acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], b[c][j], acc, 0, 0, 0);
        asum += a[c][j];
      }
  };
  // row of accumulator register r of this lane's k-group within the 16x16 tile (see the header comment)
  auto row16 = [&](int kg, int r) { return ES == 8 ? kg + 4 * r : 4 * kg + r; };
  // Epilogue operands (Cin, the activation derivative's h, the bias) do not depend on the product: their
  // loads go out FIRST and are long back when the cross-wave reduction ends, instead of a dependent load there
  // (measured neutral on config 3's in-place SGD epilogue, Cin = W: 26.5 us per step either way -- the
  // reduction's barrier hides it -- kept because it never costs).
  constexpr int NRW = (NR + NW - 1) / NW;  // registers r = wave, wave + NW, ... finished by this wave
  S* Cb = g.C + bz * g.c_sb;
  const S* Ci = g.Cin ? g.Cin + bz * g.c_sb : nullptr;
  const S* Hd = g.dact ? g.dact + bz * g.c_sb : nullptr;
  S pf_ci[NRW], pf_hd[NRW], pf_bias = S(0);
  {
    const long col = (long)tile_n * TS + l31;
    if (g.bias && col < g.N) pf_bias = g.bias[col];
#pragma unroll
    for (int i = 0; i < NRW; ++i) {
      const int r = wave + i * NW;
      const int lrow = (TS == 32) ? (r & 3) + 8 * (r >> 2) + 4 * half : row16(half, r);
      const long row = (long)tile_m * TS + lrow;
      const bool ok = r < NR && row < g.M && col < g.N;
      pf_ci[i] = (Ci && ok) ? Ci[row * g.c_sm + col] : S(0);
      pf_hd[i] = (Hd && ok) ? Hd[row * g.c_sm + col] : S(0);
    }
  }
  constexpr int SK = CK * ST;
  SM_STAMP(1);   // (set-up done, the epilogue's operands asked for)
  if constexpr (ONESHOT) {
    if (kbeg < kend) {  // kper <= SK by construction (launch_gemm_small)
      load_stage(a0, b0, kbeg);
#ifdef TOPS_AB_KNOBS
      if (stamp_on) { SM_STAMP(2); __builtin_amdgcn_s_waitcnt(0x0F70); SM_STAMP(3); }   // (the batch issued; all of it landed -- stamped wave only)
#endif
      mma_stage(a0, b0);
    }
  } else {
    S a1[ST][4], b1[ST][4];
    if (kbeg < kend) load_stage(a0, b0, kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 2 * SK) {
      if (k0 + SK < kend) load_stage(a1, b1, k0 + SK);
      mma_stage(a0, b0);
      if (k0 + SK < kend) {
        if (k0 + 2 * SK < kend) load_stage(a0, b0, k0 + 2 * SK);
        mma_stage(a1, b1);
      }
    }
  }

  // operands of the fused tail (GemmProblem::tail_*): independent of this kernel's own result, so their
  // loads are issued now and land during the reduction and the loss head
  S tl_bw[2][4], tl_hv[2][4];
  if constexpr (TS == 16) {
    if (g.loss_rows && g.tail_out) {
      const int l15 = lane & 15, kg = lane >> 4;
      const int ntiles = (g.tail_n + 15) / 16;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = wave + u * NW;
        const long col = (long)t * 16 + l15;
        const bool cv = t < ntiles && col < g.tail_n;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          const int k = 4 * st + kg;
          tl_bw[u][st] = (cv && k < g.N) ? g.tail_w[(long)k * g.tail_n + col] : S(0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long row = (long)tile_m * 16 + row16(kg, r);
          tl_hv[u][r] = (cv && row < g.M) ? g.tail_h[row * g.tail_n + col] : S(0);
        }
      }
    }
  }
  // cross-wave reduction through LDS, then the fused epilogue: wave w finishes regs r = w, w+NW, ...
#pragma unroll
  for (int r = 0; r < NR; ++r) red[wave][r][lane] = acc[r];
  rsum[wave][lane] = asum;
  SM_STAMP(4);   // (MFMAs done, partial tile written to LDS)
  __syncthreads();
  SM_STAMP(5);   // (every wave's partial is there)
  if (g.rowsum && tile_n == 0 && wave == 0 && lane < TS) {
    // rowsum[m] = sum_k A[m,k]: add the k-groups (lanes l, l+TS, ...) of every wave
    S v = S(0);
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
      for (int q = 0; q < KG; ++q) v += rsum[w][lane + q * TS];
    const long row = (long)tile_m * TS + lane;
    if (row < g.M) g.rowsum[bz * g.M + row] = g.rowsum_acc ? g.rowsum_in[bz * g.M + row] + g.rowsum_alpha * v : v;
  }
#pragma unroll
  for (int i = 0; i < NRW; ++i) {
    const int r = wave + i * NW;
    if (r >= NR) break;
    S v = S(0);
#pragma unroll
    for (int w = 0; w < NW; ++w) v += red[w][r][lane];
    const int lrow = (TS == 32) ? (r & 3) + 8 * (r >> 2) + 4 * half : row16(half, r);
    const long row = (long)tile_m * TS + lrow;
    const long col = (long)tile_n * TS + l31;
    if constexpr (TS == 16) {
      if (g.loss_rows) {
        // loss head on the finished row: the 16 lanes of a k-group hold columns 0..15 of one row
        const bool valid = row < g.M && col < g.N;
        v = v * g.alpha + pf_bias;
        const S t = valid ? g.target[row * g.c_sm + col] : S(0);
        auto sum16 = [](S x) {
#pragma unroll
          for (int off = 8; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
          return x;
        };
        S out, l;
        if (g.loss_rows == 1) {
          S mx = valid ? v : S(-INFINITY);
#pragma unroll
          for (int off = 8; off > 0; off >>= 1) mx = max_s(mx, __shfl_xor(mx, off, 64));
          const S e = valid ? exp_s(v - mx) : S(0);
          const S se = sum16(e), sy = sum16(t);
          const S pr = e / se;
          out = pr * sy - t;
          l = valid ? -t * log_s(pr) : S(0);
        } else {
          const S sg = S(1) / (S(1) + exp_s(-v));
          const S e = t - sg;
          out = S(-2) * e * sg * (S(1) - sg);
          l = valid ? e * e : S(0);
        }
        if (valid) Cb[row * g.c_sm + col] = out;
        if (g.tail_out) dzs[lrow * 17 + l31] = valid ? out : S(0);
        if (g.loss_out) {
          l = sum16(l);
          if (l31 == 0 && row < g.M) g.loss_out[row] = l;
        }
        continue;
      }
    }
    if (row < g.M && col < g.N) {
      v *= g.alpha;
      if (Ci) v += g.beta * pf_ci[i];
      v += pf_bias;
      if (g.act == 1) v = S(1) / (S(1) + exp_s(-v));
      else if (g.act == 2) v = tanh_s(v);
      if (Hd) {
        const S h = pf_hd[i];
        v *= g.dact_kind ? S(1) - h * h : h * (S(1) - h);
      }
      Cb[row * g.c_sm + col] = v;
    }
  }
  SM_STAMP(6);   // (reduction, epilogue / loss head, stores issued)
  if constexpr (TS == 16) {
    if (g.loss_rows && g.tail_out) {
      // fused tail: tail_out[16 rows][tail_n] = (dz[16][N] . W[N][tail_n]) * h(1-h); the waves share the
      // 16-column tiles of the output, K = N <= 16 is at most four 16x16x4 MFMA steps
      __syncthreads();
      typedef S acc4 __attribute__((ext_vector_type(4)));
      const int l15 = lane & 15, kg = lane >> 4;
      const int ntiles = (g.tail_n + 15) / 16;
      // one pass of two tiles per wave (tail_n <= 32 * NW, checked by the launcher); the W columns and h
      // were fetched before the cross-wave reduction (tl_bw / tl_hv), so the only dependent work left
      // here is 2 x 3 MFMAs and the stores
      acc4 acc2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc2[u][r] = S(0);
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        if (4 * st < g.N) {  // (uniform)
          const S a = dzs[l15 * 17 + 4 * st + kg];  // zero beyond N
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if constexpr (ES == 8) acc2[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, tl_bw[u][st], acc2[u], 0, 0, 0);
            else acc2[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, tl_bw[u][st], acc2[u], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = wave + u * NW;
        const long col = (long)t * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long row = (long)tile_m * 16 + row16(kg, r);
          if (t < ntiles && row < g.M && col < g.tail_n)
            g.tail_out[row * g.tail_n + col] = acc2[u][r] * tl_hv[u][r] * (S(1) - tl_hv[u][r]);
        }
      }
      SM_STAMP(7);   // (tail stores issued)
    }
  }
#ifdef TOPS_AB_KNOBS
  if (stamp_on) { __builtin_amdgcn_s_waitcnt(0); SM_STAMP(8); }   // (this wave's stores acknowledged)
#endif
}

template <class S, int AMODE, int BMODE, int NW, int TS, int ONESHOT = 0>
__global__ __launch_bounds__(NW * 64) void gemm_small_kernel(SmallArgsT<S> g) {
  gemm_small_body<S, AMODE, BMODE, NW, TS, ONESHOT>(g, (int)blockIdx.x, (long)blockIdx.z);
}

// Two independent latency-bound GEMMs in ONE launch (the weight gradients of two layers once both
// cotangents exist): workgroups [0, n1) run the first problem, the rest the second.  Each launch costs
// ~4 us of dispatch and cache maintenance whatever its size, so grouping is worth one such floor.
// The block is sized for the larger configuration; the surplus waves of the smaller one exit at once
// (a finished wave no longer counts at s_barrier).
template <class S, int A1, int B1, int NW1, int TS1, int OS1, int A2, int B2, int NW2, int TS2, int OS2>
__global__ __launch_bounds__((NW1 > NW2 ? NW1 : NW2) * 64) void gemm_small_pair_kernel(SmallArgsT<S> g1,
                                                                                         SmallArgsT<S> g2, int n1) {
  if ((int)blockIdx.x < n1) {
    if (NW1 < NW2 && (int)(threadIdx.x >> 6) >= NW1) return;
    gemm_small_body<S, A1, B1, NW1, TS1, OS1>(g1, (int)blockIdx.x, 0);
  } else {
    if (NW2 < NW1 && (int)(threadIdx.x >> 6) >= NW2) return;
    gemm_small_body<S, A2, B2, NW2, TS2, OS2>(g2, (int)blockIdx.x - n1, 0);
  }
}

// ---- a whole batched training step in ONE launch ----------------------------------------------------------
// The config-3 step is three dependent launches (forward + activation; output layer + loss head + the hidden
// layer's cotangent; the two weight gradients with their updates), each at its latency floor, and ~2.3 us of each
// is the launch boundary itself (dispatch, ramp, drain, the kernel-boundary cache maintenance of an 8-XCD part).
// Here the three run as stages of one launch of 256 co-resident workgroups, separated by two grid barriers
// (arrive counter + device-scope release/acquire, the protocol of cooperative-groups grid sync).  The counters
// only ever grow and the epoch is read from device memory, so a replayed launch record works; a barrier that is
// not satisfied within the watchdog interval makes every workgroup give up and flags it, instead of hanging.
struct ChainSync {
  unsigned* ctr;        // [which * 256 + workgroup]: the epoch that workgroup has reached at barrier `which`; [512]: epoch
  int* status;          // host-visible: nonzero = a barrier timed out
  long long timeout;    // wall_clock64 ticks (100 MHz)
  int dev;              // development (TOPS_CHAIN_DEV): 1 = barriers are no-ops (timing of the stages alone; results invalid),
                        // 2 = no fences around the barrier
};

// A flag per workgroup instead of one arrive counter: 256 device-scope read-modify-writes of ONE address are
// serialised at the memory side (measured: ~25 us per barrier, the chained step ran at 74 us); 256 plain stores to
// 256 words and a 1 KiB poll per workgroup are not.
__device__ __forceinline__ bool chain_barrier(const ChainSync& cs, int which, unsigned epoch) {
  __shared__ int ok;
  if (cs.dev == 1) {
    __syncthreads();
    return true;
  }
  if (cs.dev != 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // every wave: its own stores are out (other XCDs read them through memory)
  if (threadIdx.x == 0) ok = 1;
  __syncthreads();
  unsigned* flags = cs.ctr + which * 256;
  if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x < gridDim.x) {
    const long long t0 = wall_clock64();
    // (relaxed polls: an acquire per poll would invalidate this CU's caches every few cycles)
    while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
      if (wall_clock64() - t0 > cs.timeout) {
        ok = 0;
        *cs.status = 1 + which;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  if (cs.dev != 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return ok != 0;
}

struct ChainArgs {
  SmallArgsT<float> g[4];   // forward, output layer + loss head (+ tail), weight gradient 1, weight gradient 2
  int nblk[2];              // workgroups of stages 0 and 1
  int n1, n2;               // ... of the two weight gradients
  ChainSync cs;
};

__global__ __launch_bounds__(1024) void gemm_small_chain_kernel(ChainArgs c) {
  extern __shared__ __attribute__((aligned(16))) float chain_lds[];
  // (every workgroup reads the epoch before it can arrive at the first barrier; workgroup 0 advances it after the
  //  second, when nobody of this launch will read it again and the next launch has not started)
  const unsigned epoch = __hip_atomic_load(c.cs.ctr + 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  const int bid = (int)blockIdx.x;
  if (bid < c.nblk[0]) gemm_small_body<float, 0, 1, 16, 32, 8, true>(c.g[0], bid, 0, chain_lds);
  if (!chain_barrier(c.cs, 0, epoch)) return;
  if (bid < c.nblk[1]) gemm_small_body<float, 0, 1, 16, 16, 8, true>(c.g[1], bid, 0, chain_lds);
  if (!chain_barrier(c.cs, 1, epoch)) return;
  if (bid == 0 && threadIdx.x == 0) __hip_atomic_store(c.cs.ctr + 512, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (bid < c.n1) gemm_small_body<float, 1, 0, 16, 32, 8, true>(c.g[2], bid, 0, chain_lds);
  else if (bid < c.n1 + c.n2) gemm_small_body<float, 1, 0, 16, 16, 8, true>(c.g[3], bid - c.n1, 0, chain_lds);
}

// ---- forward layer + loss head in ONE launch, joined by an intra-XCD seam (round 4) ------------------------------------
// The step's first two launches are a layer `H = logistic(X W1^T + b1)` on 32x32 tiles and, behind a kernel boundary,
// the output layer with its loss head and the hidden layer's cotangent for 16-row tiles of the SAME rows: everything the
// second needs of the first is the 32 rows of H its rows belong to -- the tiles_n column tiles of one row block.  A grid
// barrier is the wrong tool on an 8-XCD part (gemm_small_chain_kernel: 181 us with the fences a correct one needs); a
// seam inside ONE XCD needs none of that:
//  * the XCD-aware tile order (tile_order 1) gives every XCD whole row blocks: the writers of a row block's H and its
//    reader share one L2, so plain stores (complete -- vmcnt -- when the L2 has them) and plain loads meet there; no
//    write-back, no invalidate (the reader's L1 cannot hold those lines: nothing in this launch read them before);
//  * the workgroup that finishes a tile bumps the row block's counter (one device-scope atomic, counters only grow:
//    a replayed launch record works, nothing to reset); the one that sees the LAST tile arrive carries on as the head of
//    those 32 rows -- two 16-row head tiles side by side on its two halves of 8 waves -- and the others exit.  Nobody
//    waits for anybody: no co-residency requirement, no watchdog.
// What it buys: the second launch's boundary (dispatch, ramp, drain, the 8-L2 cache maintenance) against one vmcnt wait
// and one atomic round trip (profiles/README.md, round 4, has the stamps).
struct SeamArgs {
  SmallArgsT<float> fwd;    // H = act(X W^T + b): 16 waves, 32x32 tiles, one-shot
  SmallArgsT<float> head;   // the loss-head problem on 16-row tiles (A = H), with its fused tail
  unsigned* ctr;            // [row block][32]: word 0 = tiles of the row block that have been stored, ever; word 1 = what
                            // word 0 stood at when the previous launch had finished with the row block (128 bytes per
                            // row block: the atomics of different row blocks do not queue up behind one cache line)
  int mode;                 // 1: the last arriver carries on as the head.  2: the workgroup of the row block's LAST tile
                            //    (highest workgroup index: dispatched after its peers) is the head and waits for them
  long long* dbg;           // development: time stamps of row block 0's head workgroup (TOPS_SEAM_STAMPS, A/B builds)
  int* status;              // mode 2: host-visible, nonzero = the wait for the peers timed out
};

__global__ __launch_bounds__(1024) void gemm_small_seam_kernel(SeamArgs c) {
  extern __shared__ __attribute__((aligned(16))) float seam_lds[];
  __shared__ int seam_go;
  const int bid = (int)blockIdx.x;
  int tile_m, tile_n;
  tile_of(c.fwd, bid, tile_m, tile_n);
  unsigned* ctr = c.ctr + tile_m * 32;
  const bool stamp = c.dbg && tile_m == 0 && threadIdx.x == 0;
  long long t_start = 0, t_fwd = 0, t_stored = 0;
  if (stamp) t_start = wall_clock64();
  // (what the counter stood at when the previous launch was done with this row block: written by that launch's head,
  //  visible across the kernel boundary; read before anything of this launch can have changed it -- only a head writes it)
  const unsigned base = ctr[1];
  gemm_small_body<float, 0, 1, 16, 32, 8, true>(c.fwd, bid, 0, seam_lds);
  if (stamp) t_fwd = wall_clock64();
  // every wave's stores of its part of the tile are in the L2 ...
  __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) expcnt(0) lgkmcnt(0)
  __syncthreads();
  if (stamp) t_stored = wall_clock64();
  const unsigned ntile = (unsigned)c.fwd.tiles_n;
  if (c.mode == 2) {
    if (tile_n != (int)ntile - 1) {   // a peer: one atomic nobody waits for, and out
      if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (threadIdx.x == 0) {
      // the peers carry lower workgroup indices: they were dispatched before this workgroup, nothing it holds keeps them
      // from finishing.  (A watchdog all the same: a wrong answer must not look like a hang.)
      const long long t0 = wall_clock64();
      int ok = 1;
      while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (base + ntile - 1u)) < 0) {
        if (wall_clock64() - t0 > 200000000LL) {   // 2 s
          ok = 0;
          if (c.status) *c.status = 1 + tile_m;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      seam_go = ok;
      if (ok) ctr[1] = base + ntile - 1u;
    }
  } else {
    if (threadIdx.x == 0) {
      const unsigned seen = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      seam_go = (seen + 1u - base) == ntile;
      if (seam_go) ctr[1] = seen + 1u;
    }
  }
  __syncthreads();
  if (!seam_go) return;
  long long t_go = 0;
  if (stamp) t_go = wall_clock64();
  // ... and every tile of the row block has arrived: rows [32 tile_m, 32 tile_m + 32) of H are complete in this XCD's L2.
  // Two head tiles of 16 rows, one per half of the workgroup (the head body's 8-wave form; its barriers are the whole
  // workgroup's, both halves run the same sequence); LDS: the forward body's reduction buffers are free again.
  const int hw = (int)threadIdx.x >> 9;
  constexpr int HEAD_LDS = 8 * 4 * 64 + 8 * 64 + 16 * 17 + 16;   // red + rsum + dzs (+ pad) of an <8 waves, 16x16> body
  gemm_small_body<float, 0, 1, 8, 16, 2, true>(c.head, 2 * tile_m + hw, 0, seam_lds + hw * HEAD_LDS, (int)threadIdx.x & 511);
  if (stamp) {
    c.dbg[0] = t_start; c.dbg[1] = t_fwd; c.dbg[2] = t_stored; c.dbg[3] = t_go; c.dbg[4] = wall_clock64();
  }
}

// ---- fp64, 32x32 output tile as 2x2 blocks of v_mfma_f64_16x16x4_f64 ---------------------------------
// The 16x16 fp64 tile of the kernel above pulls (16 + 16) rows of K doubles through the CU's L1 per 256
// outputs -- four times the bytes per output of the fp32 32x32 tile, and the L1 (64 B/clk) is what bounds
// these kernels once K is long (config 3 in fp64: X.W1^T 34.5 us, dZ1^T.X 23.8 us).  Here a lane holds the
// A values of two row blocks and the B values of two column blocks and feeds four MFMAs per k-step: half the
// operand bytes per flop.  Same K split over the waves, same cross-wave reduction and fused epilogue
// (alpha/beta/Cin, bias, logistic, logistic', row sums); no loss head (that needs N <= 16).
// A ring of D register stages of ST 16-k chunks each (D = 2, ST = 1 measured best at 8 waves).
template <int AMODE, int BMODE, int NW, int ST, int D = 2>
__device__ __forceinline__ void gemm_small_f64_t32_body(const SmallArgsT<double>& g, const int bid, const long bz) {
  typedef double S;
  typedef double acc4 __attribute__((ext_vector_type(4)));
  constexpr int CK = 16, SK = ST * CK, NQ = 16;  // NQ: accumulator values per lane
  __shared__ S red[NW][NQ][64];
  __shared__ S rsum[NW][2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kg = lane >> 4;
  int tile_m, tile_n;
  tile_of(g, bid, tile_m, tile_n);
  long m[2], n[2];
  bool mv[2], nv[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    m[h] = (long)tile_m * 32 + 16 * h + l15;
    n[h] = (long)tile_n * 32 + 16 * h + l15;
    mv[h] = m[h] < g.M;
    nv[h] = n[h] < g.N;
  }
  acc4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = acc4{0.0, 0.0, 0.0, 0.0};
  S asum[2] = {0.0, 0.0};
  const int kbeg = wave * g.kper;
  int kend = kbeg + g.kper;
  if (kend > g.K) kend = g.K;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<S*>(g.A), 0, g.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<S*>(g.B), 0, g.b_bytes, 0x00020000);
  int a_base[2], b_base[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    a_base[h] = (int)((bz * g.a_sb + (mv[h] ? m[h] : 0) * g.a_sm) * 8);
    b_base[h] = (int)((bz * g.b_sb + (nv[h] ? n[h] : 0) * g.b_sn) * 8);
  }
  const int a_sk8 = (int)g.a_sk * 8, b_sk8 = (int)g.b_sk * 8;
  auto sel = [](bool ok, int off) { const int msk = -(int)ok; return (off & msk) | (0x7fffffff & ~msk); };
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  auto ld1 = [&](__amdgpu_buffer_rsrc_t r, int off) -> S {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    return __hiloint2double((int)v.y, (int)v.x);
  };
  auto ld4 = [&](__amdgpu_buffer_rsrc_t r, bool ok, int off, S (&d)[4]) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, sel(ok, off), 0, 0);
    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(r, sel(ok, off + 16), 0, 0);
    d[0] = __hiloint2double((int)v.y, (int)v.x); d[1] = __hiloint2double((int)v.w, (int)v.z);
    d[2] = __hiloint2double((int)w.y, (int)w.x); d[3] = __hiloint2double((int)w.w, (int)w.z);
  };
  auto load_stage = [&](S (&a)[2][ST][4], S (&b)[2][ST][4], int k0) {
#pragma unroll
    for (int c = 0; c < ST; ++c) {
      const int kb = k0 + CK * c + 4 * kg;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (AMODE == 0 && g.a_vec) {
          ld4(ra, mv[h] & (kb < kend), a_base[h] + kb * 8, a[h][c]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[h][c][j] = ld1(ra, sel(mv[h] & (kb + j < kend), a_base[h] + (kb + j) * a_sk8));
        }
        if (BMODE == 1 && g.b_vec) {
          ld4(rb, nv[h] & (kb < kend), b_base[h] + kb * 8, b[h][c]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) b[h][c][j] = ld1(rb, sel(nv[h] & (kb + j < kend), b_base[h] + (kb + j) * b_sk8));
        }
      }
    }
  };
  auto mma_stage = [&](const S (&a)[2][ST][4], const S (&b)[2][ST][4]) {
#pragma unroll
    for (int c = 0; c < ST; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int q = 0; q < 2; ++q) acc[i][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][c][j], b[q][c][j], acc[i][q], 0, 0, 0);
          asum[i] += a[i][c][j];
        }
      }
  };
  // epilogue operands first (see gemm_small_body): value q = (i*2 + j)*4 + r of this lane is
  // row 16 i + kg + 4 r, column 16 j + l15 of the tile; wave w finishes q = w, w + NW, ...
  constexpr int NQW = (NQ + NW - 1) / NW;
  S* Cb = g.C + bz * g.c_sb;
  const S* Ci = g.Cin ? g.Cin + bz * g.c_sb : nullptr;
  const S* Hd = g.dact ? g.dact + bz * g.c_sb : nullptr;
  S pf_ci[NQW], pf_hd[NQW], pf_bias[NQW];
#pragma unroll
  for (int u = 0; u < NQW; ++u) {
    const int q = wave + u * NW;
    const long row = (long)tile_m * 32 + 16 * (q >> 3) + kg + 4 * (q & 3);
    const long col = (long)tile_n * 32 + 16 * ((q >> 2) & 1) + l15;
    const bool ok = q < NQ && row < g.M && col < g.N;
    pf_ci[u] = (Ci && ok) ? Ci[row * g.c_sm + col] : 0.0;
    pf_hd[u] = (Hd && ok) ? Hd[row * g.c_sm + col] : 0.0;
    pf_bias[u] = (g.bias && ok) ? g.bias[col] : 0.0;
  }
  {
    // ring of D register stages: D - 1 batches of loads in flight under the MFMAs of the oldest
    // (all indices are compile-time after unrolling)
    S a[D][2][ST][4], b[D][2][ST][4];
    const int nst = (kend - kbeg + SK - 1) / SK;
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
      if (d < nst) load_stage(a[d], b[d], kbeg + d * SK);
    for (int s0 = 0; s0 < nst; s0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int st = s0 + d;
        if (st < nst) {
          if (st + D - 1 < nst) load_stage(a[(d + D - 1) % D], b[(d + D - 1) % D], kbeg + (st + D - 1) * SK);
          mma_stage(a[d], b[d]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][(i * 2 + j) * 4 + r][lane] = acc[i][j][r];
  rsum[wave][0][lane] = asum[0];
  rsum[wave][1][lane] = asum[1];
  __syncthreads();
  if (g.rowsum && tile_n == 0 && wave == 0 && lane < 32) {
    const int h = lane >> 4;
    S v = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
      for (int q = 0; q < 4; ++q) v += rsum[w][h][l15 + 16 * q];
    const long row = (long)tile_m * 32 + lane;
    if (row < g.M) g.rowsum[bz * g.M + row] = g.rowsum_acc ? g.rowsum_in[bz * g.M + row] + g.rowsum_alpha * v : v;
  }
#pragma unroll
  for (int u = 0; u < NQW; ++u) {
    const int q = wave + u * NW;
    if (q >= NQ) break;
    S v = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += red[w][q][lane];
    const long row = (long)tile_m * 32 + 16 * (q >> 3) + kg + 4 * (q & 3);
    const long col = (long)tile_n * 32 + 16 * ((q >> 2) & 1) + l15;
    if (row < g.M && col < g.N) {
      v *= g.alpha;
      if (Ci) v += g.beta * pf_ci[u];
      v += pf_bias[u];
      if (g.act == 1) v = 1.0 / (1.0 + exp(-v));
      else if (g.act == 2) v = tanh(v);
      if (Hd) v *= g.dact_kind ? 1.0 - pf_hd[u] * pf_hd[u] : pf_hd[u] * (1.0 - pf_hd[u]);
      Cb[row * g.c_sm + col] = v;
    }
  }
}

template <int AMODE, int BMODE, int NW, int ST, int D = 2>
__global__ __launch_bounds__(NW * 64) void gemm_small_f64_t32_kernel(SmallArgsT<double> g) {
  gemm_small_f64_t32_body<AMODE, BMODE, NW, ST, D>(g, (int)blockIdx.x, (long)blockIdx.z);
}

// the fp64 form of gemm_small_pair_kernel: a 32x32-tile problem (8 waves, one-chunk stages) and a 16x16-tile
// one-shot problem (8 waves) in one launch
template <int A1, int B1, int A2, int B2>
__global__ __launch_bounds__(512) void gemm_small_pair_f64_kernel(SmallArgsT<double> g1, SmallArgsT<double> g2, int n1) {
  if ((int)blockIdx.x < n1) gemm_small_f64_t32_body<A1, B1, 8, 1>(g1, (int)blockIdx.x, 0);
  else gemm_small_body<double, A2, B2, 8, 16, 8>(g2, (int)blockIdx.x - n1, 0);
}

// the loss head needs the whole output row inside one 16x16 tile and a single batch entry
bool gemm_small_fuses_loss(const GemmProblem& p) {
  // (any row count down to the single row of an online-SGD step: rows beyond M are masked)
  const int64_t tiles64 = ((p.M + 63) / 64) * ((p.N + 63) / 64) * p.batch;
  return gemm_small_can(p) && tiles64 < 200 && p.N <= 16 && p.batch == 1 && p.beta == 0.0 && !p.dact && p.act == 0;
}

// ... and the fused tail one pass of two 16-column tiles per wave at 8 waves
bool gemm_small_fuses_tail(const GemmProblem& p, int64_t tail_n) {
  return gemm_small_fuses_loss(p) && tail_n <= 256 && (p.K + 15) / 16 >= 8;
}

// what the kernel CAN run (hard constraints) ...
bool gemm_small_can(const GemmProblem& p) {
  if (p.dtype != TO_F32 && p.dtype != TO_F64) return false;
  if (p.reduce_batch) return false;          // the planner folds the batch into K whenever it can
  if (p.batch > 65535) return false;
  if (p.M < 1 || p.N < 1 || p.K < 1) return false;
  const int64_t es = p.dtype == TO_F64 ? 8 : 4;
  auto span = [es](int64_t nb, int64_t sb, int64_t n0, int64_t s0, int64_t n1, int64_t s1) {
    return ((nb - 1) * sb + (n0 - 1) * s0 + (n1 - 1) * s1 + 1) * es;
  };
  // 32-bit buffer offsets: operands must span < 2 GiB (always true for these latency-bound shapes)
  if (span(p.batch, p.a_sb, p.M, p.a_sm, p.K, p.a_sk) >= (1LL << 31) ||
      span(p.batch, p.b_sb, p.N, p.b_sn, p.K, p.b_sk) >= (1LL << 31))
    return false;
  if (p.a_sm < 0 || p.a_sk < 0 || p.b_sk < 0 || p.b_sn < 0) return false;
  const int64_t tiles16 = ((p.M + 15) / 16) * ((p.N + 15) / 16) * p.batch;
  return tiles16 <= 65535;
}

// ... and where it is the kernel of choice: few output tiles (latency-bound), not a trivially small output
bool gemm_small_applicable(const GemmProblem& p) {
  if (!gemm_small_can(p)) return false;
  const int64_t tiles64 = ((p.M + 63) / 64) * ((p.N + 63) / 64) * p.batch;
  // (K < 8: outer products of the one-sample step -- every load is bounds-checked, a partial chunk is fine;
  //  784->300->100->10 online SGD: 77.4 -> 68.2 us per sample)
  return tiles64 < 200 && p.M * p.N >= 256;
}

// Which XCD-aware order pulls fewer operand bytes into each L2: a run of T/8 tiles in row-major order touches
// ceil(run / tiles_n) A panels (+1 when it straddles) and all of B; in column-major order the mirror image.
static int pick_tile_order(const GemmProblem& p, int ts, int tiles_m, int tiles_n) {
  static const int enable = [] { const char* e = ab_getenv("TOPS_SMALL_XCD"); return e ? atoi(e) : 1; }();
  const long T = (long)tiles_m * tiles_n;
  if (!enable || T < 16) return 0;
  if (enable == 2 || enable == 3) return enable - 1;  // (forced, for A/B runs)
  const long run = (T + 7) / 8;
  const double a_panel = (double)ts * p.K, b_panel = (double)ts * p.K;
  const double row_cost = (double)((run + tiles_n - 1) / tiles_n + 1) * a_panel + (double)(run < tiles_n ? run + 1 : tiles_n) * b_panel;
  const double col_cost = (double)((run + tiles_m - 1) / tiles_m + 1) * b_panel + (double)(run < tiles_m ? run + 1 : tiles_m) * a_panel;
  return col_cost < row_cost ? 2 : 1;
}

template <class S, int NW, int TS, int ONESHOT = 0>
static void launch_nw(SmallArgsT<S>& g, const GemmProblem& p, int amode, int bmode, hipStream_t s) {
  constexpr int CK = (TS == 32) ? 8 : 16;
  const int chunks = (int)((p.K + CK - 1) / CK);
  g.kper = ((chunks + NW - 1) / NW) * CK;
  const int tiles_m = (int)((p.M + TS - 1) / TS);
  g.tiles_n = (int)((p.N + TS - 1) / TS);
  g.tiles_m = tiles_m;
  g.tile_order = pick_tile_order(p, TS, g.tiles_m, g.tiles_n);
  dim3 grid(tiles_m * g.tiles_n, 1, (unsigned)p.batch), block(NW * 64);
  switch (amode * 2 + bmode) {
    case 0: launch_k((gemm_small_kernel<S, 0, 0, NW, TS, ONESHOT>), grid, block, 0, s, g); break;
    case 1: launch_k((gemm_small_kernel<S, 0, 1, NW, TS, ONESHOT>), grid, block, 0, s, g); break;
    case 2: launch_k((gemm_small_kernel<S, 1, 0, NW, TS, ONESHOT>), grid, block, 0, s, g); break;
    default: launch_k((gemm_small_kernel<S, 1, 1, NW, TS, ONESHOT>), grid, block, 0, s, g); break;
  }
}

#ifdef TOPS_AB_KNOBS
// TOPS_SMALL_STAMPS = 1: the weight-gradient launch's big problem (dW1: 32x32 tiles, 16 waves), 2: the loss-head launch,
// 3: the pair's small problem (dW2).  One host-mapped record, printed at exit: the LAST stamped launch of the process.
static long long* small_stamps(int which) {
  static const int want = [] { const char* e = ab_getenv("TOPS_SMALL_STAMPS"); return e ? atoi(e) : 0; }();
  if (!want || want != which) return nullptr;
  static long long* buf = [] {
    long long* p = nullptr;
    if (hipHostMalloc(&p, 16 * sizeof(long long), hipHostMallocMapped) != hipSuccess) return (long long*)nullptr;
    for (int i = 0; i < 16; ++i) p[i] = 0;
    static long long* keep = p;
    atexit([] {
      static const char* names[9] = {"entry", "set-up done", "operand batch issued", "operand batch landed", "MFMAs done, partial in LDS",
                                     "all partials there (barrier)", "reduced + epilogue / loss head, stores issued", "tail done", "stores acknowledged"};
      std::fprintf(stderr, "[small stamps %d] tile 0, thread 0, us since its workgroup began:", want);
      for (int i = 1; i < 9; ++i)
        if (keep[i]) std::fprintf(stderr, "\n  %6.2f  %s", (keep[i] - keep[0]) * 0.01, names[i]);
      std::fprintf(stderr, "\n");
    });
    return p;
  }();
  long long* dev = nullptr;
  if (buf && hipHostGetDevicePointer(reinterpret_cast<void**>(&dev), buf, 0) != hipSuccess) dev = nullptr;
  return dev;
}
#endif

struct SmallPlan {
  int ts, nw, os;  // tile size, waves, one-shot stage size (0 = two-stage pipeline)
  int amode, bmode;
  bool f64_t32;    // fp64: the 2x2-blocked 32x32 kernel (ts = 32, os = 0)
};

// kernel arguments + the configuration the heuristics pick for one problem
template <class S>
static SmallPlan plan_small(const GemmProblem& p, SmallArgsT<S>& g) {
  constexpr bool F64 = sizeof(S) == 8;
  g = SmallArgsT<S>{};
  g.A = (const S*)p.A; g.B = (const S*)p.B; g.C = (S*)p.C;
  g.Cin = (p.beta != 0.0) ? (const S*)p.Cin : nullptr;
  g.M = (int)p.M; g.N = (int)p.N; g.K = (int)p.K;
  g.a_sm = p.a_sm; g.a_sk = p.a_sk; g.b_sk = p.b_sk; g.b_sn = p.b_sn; g.c_sm = p.c_sm;
  g.a_sb = p.a_sb; g.b_sb = p.b_sb; g.c_sb = p.c_sb;
  g.alpha = (S)p.alpha; g.beta = (S)p.beta;
  g.bias = (const S*)p.bias; g.dact = (const S*)p.dact; g.act = p.act; g.dact_kind = p.dact_kind;
  g.rowsum = (S*)p.rowsum;
  g.rowsum_acc = p.rowsum_acc ? 1 : 0; g.rowsum_alpha = (S)p.rowsum_alpha;
  g.rowsum_in = p.rowsum_in ? (const S*)p.rowsum_in : (const S*)p.rowsum;
  g.loss_rows = p.loss_rows; g.target = (const S*)p.target; g.loss_out = (S*)p.loss_out;
  g.tail_w = (const S*)p.tail_w; g.tail_h = (const S*)p.tail_h;
  g.tail_out = p.loss_rows ? (S*)p.tail_out : nullptr; g.tail_n = p.tail_n;
  TO_CHECK(!g.tail_out || p.tail_n <= 256, TO_ERR_ARG, "fused tail: at most 256 columns (gemm_small_fuses_tail)");
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
  auto eff = [](int64_t stride, int64_t extent) { return extent == 1 ? (int64_t)0 : stride; };
  SmallPlan c{};
  c.amode = (p.a_sk == 1) ? 0 : 1;
  c.bmode = (p.b_sk == 1 && p.b_sn != 1) ? 1 : 0;
  constexpr int64_t VE = 16 / (int64_t)sizeof(S);  // elements per 16 bytes: quad loads need 16-byte aligned rows
  g.a_vec = c.amode == 0 && p.K % 4 == 0 && al16(p.A) && eff(p.a_sm, p.M) % VE == 0 && eff(p.a_sb, p.batch) % VE == 0;
  g.b_vec = c.bmode == 1 && p.K % 4 == 0 && al16(p.B) && eff(p.b_sn, p.N) % VE == 0 && eff(p.b_sb, p.batch) % VE == 0;
  g.a_bytes = (unsigned)(((p.batch - 1) * eff(p.a_sb, p.batch) + (p.M - 1) * eff(p.a_sm, p.M) +
                          (p.K - 1) * eff(p.a_sk, p.K) + 1) * (int64_t)sizeof(S));
  g.b_bytes = (unsigned)(((p.batch - 1) * eff(p.b_sb, p.batch) + (p.N - 1) * eff(p.b_sn, p.N) +
                          (p.K - 1) * eff(p.b_sk, p.K) + 1) * (int64_t)sizeof(S));
  // tile size: 16x16 MFMAs for skinny outputs (and always in fp64: its matrix instruction is 16x16x4);
  // waves per tile: ~4 per SIMD over the chip (TLP hides what the 2-stage pipeline does not) with at
  // least one pipeline stage each.
  // (16 waves = 1024 threads would cap the kernel at 128 VGPRs and spill the pipeline stages.)
  static const int force_nw = [] { const char* e = ab_getenv("TOPS_SMALL_NW"); return e ? atoi(e) : 0; }();
  // (short K too: nothing to pipeline, so more, smaller tiles = more memory parallelism)
  const bool t16 = F64 || (p.M <= 16 || p.N <= 16 || p.K <= 32);
  const int ts = t16 ? 16 : 32, ck = t16 ? 16 : 8;
  const int64_t tiles = ((p.M + ts - 1) / ts) * ((p.N + ts - 1) / ts) * p.batch;
  const int64_t chunks = (p.K + ck - 1) / ck;
  int nw = 8;  // (16 waves only through the one-shot variants below: the two-stage pipeline spills at 1024 threads)
  while (nw > 1 && (tiles * nw > 4096 || chunks / nw < 4)) nw >>= 1;
  if (force_nw) nw = force_nw;
  // a fused tail (GemmProblem::tail_*) is one pass of two 16-column tiles per wave over EIGHT waves
  // (gemm_small_fuses_tail guarantees the shape allows it): 0.0355 -> 0.0345 ms per step against four waves
  if (g.tail_out) {
    TO_CHECK(t16 && chunks >= 8, TO_ERR_ARG, "fused tail: shape not eligible (gemm_small_fuses_tail)");
    nw = 8;
  }
  c.ts = ts;
  c.nw = nw;
  c.os = 0;
  // big latency-bound shapes: 16 waves, each fetching its whole K slice at once
  static const int oneshot = [] { const char* e = ab_getenv("TOPS_SMALL_ONESHOT"); return e ? atoi(e) : 1; }();
  // (config 3: 0.0409 -> 0.0371 ms per step; the same for the 16x16-tile shapes measured slower, 0.0390)
  // (1024 x K x 256, us: K = 392: pipelined 7.6 / one-shot 9.3; 512: 8.2 / 9.9; 648: 10.1 / 10.5; 784: 12.3 / 10.6;
  //  1024: 13.7 / 12.8 -- the 16-wave reduction costs ~2 us, the extra pipeline stages more beyond K ~ 700;
  //  8 waves x 16 chunks measured slower: 0.0354 vs 0.0335 ms/step)
  if (!F64 && oneshot && !force_nw && !t16 && tiles * 16 <= 4096 && chunks > 88 && chunks <= 128) {
    c.nw = 16;
    c.os = 8;
  }
  // 16x16-tile shapes whose K slice per wave is 5..8 chunks: one batch of loads instead of two stages
  static const int oneshot8 = [] { const char* e = ab_getenv("TOPS_SMALL_ONESHOT8"); return e ? atoi(e) : 1; }();
  if (oneshot8 && !force_nw && t16 && nw == 8 && chunks > 32 && chunks <= 64 && !g.tail_out) c.os = 8;
  // fp64 with both output extents beyond one 16-wide block and a K worth pipelining: 32x32 tiles of 2x2 MFMA
  // blocks (half the operand bytes per flop through the L1)
  static const int f64_t32 = [] { const char* e = ab_getenv("TOPS_SMALL_F64_T32"); return e ? atoi(e) : 1; }();
  if (F64 && f64_t32 && !force_nw && !g.loss_rows && p.M > 16 && p.N > 16 && p.K >= 128) {
    const int64_t t32 = ((p.M + 31) / 32) * ((p.N + 31) / 32) * p.batch;
    int w = 8;
    while (w > 2 && (t32 * w > 4096 || chunks / w < 4)) w >>= 1;
    c.f64_t32 = true;
    c.ts = 32;
    c.nw = w;
    // (here os = chunks per register stage.  Config 3 in fp64, us per step: 8 waves x two-chunk stages 54.3,
    //  x one-chunk stages 48.9 -- and a deeper ring of one-chunk stages is SLOWER, 3 deep 51.5, 4 deep 53.8:
    //  more loads in flight do not help, fewer registers do; 16 waves spill at 128 registers, 56.2)
    c.os = w == 8 ? 1 : 2;
    g.kper = (int)(((chunks + w - 1) / w) * 16);
    g.tiles_n = (int)((p.N + 31) / 32);
    g.tiles_m = (int)((p.M + 31) / 32);
    g.tile_order = pick_tile_order(p, 32, g.tiles_m, g.tiles_n);
    return c;
  }
  g.kper = (int)(((chunks + c.nw - 1) / c.nw) * ck);
  g.tiles_n = (int)((p.N + c.ts - 1) / c.ts);
  g.tiles_m = (int)((p.M + c.ts - 1) / c.ts);
  g.tile_order = pick_tile_order(p, c.ts, g.tiles_m, g.tiles_n);
  return c;
}

template <class S>
static void launch_small_t(const GemmProblem& p, hipStream_t s) {
  constexpr bool F64 = sizeof(S) == 8;
  SmallArgsT<S> g;
  const SmallPlan c = plan_small<S>(p, g);
#ifdef TOPS_AB_KNOBS
  if (g.loss_rows) g.dbg = small_stamps(2);
#endif
  const int amode = c.amode, bmode = c.bmode;
  if constexpr (F64) {
    if (c.f64_t32) {
      const int tiles_m = (int)((p.M + 31) / 32);
      dim3 grid(tiles_m * g.tiles_n, 1, (unsigned)p.batch), block(c.nw * 64);
#define TOPS_F64_T32(AM, BM)                                                                              \
  switch (c.nw * 10 + c.os) {                                                                             \
    case 22: launch_k((gemm_small_f64_t32_kernel<AM, BM, 2, 2>), grid, block, 0, s, g); break;  \
    case 42: launch_k((gemm_small_f64_t32_kernel<AM, BM, 4, 2>), grid, block, 0, s, g); break;  \
    default: launch_k((gemm_small_f64_t32_kernel<AM, BM, 8, 1>), grid, block, 0, s, g); break;  \
  }
      switch (amode * 2 + bmode) {
        case 0: TOPS_F64_T32(0, 0) break;
        case 1: TOPS_F64_T32(0, 1) break;
        case 2: TOPS_F64_T32(1, 0) break;
        default: TOPS_F64_T32(1, 1) break;
      }
#undef TOPS_F64_T32
      TO_HIP(hipGetLastError());
      count_launch();
      return;
    }
  }
  if (c.os == 8 && c.nw == 16) {
    if constexpr (!F64) launch_nw<S, 16, 32, 8>(g, p, amode, bmode, s);
  } else if (c.os == 8) {
    launch_nw<S, 8, 16, 8>(g, p, amode, bmode, s);
  } else if (c.ts == 16) {
    switch (c.nw) {
      case 1: launch_nw<S, 1, 16>(g, p, amode, bmode, s); break;
      case 2: launch_nw<S, 2, 16>(g, p, amode, bmode, s); break;
      case 4: launch_nw<S, 4, 16>(g, p, amode, bmode, s); break;
      default: launch_nw<S, 8, 16>(g, p, amode, bmode, s); break;
    }
  } else {
    if constexpr (!F64) {
      switch (c.nw) {
        case 1: launch_nw<S, 1, 32>(g, p, amode, bmode, s); break;
        case 2: launch_nw<S, 2, 32>(g, p, amode, bmode, s); break;
        case 4: launch_nw<S, 4, 32>(g, p, amode, bmode, s); break;
        default: launch_nw<S, 8, 32>(g, p, amode, bmode, s); break;
      }
    }
  }
  TO_HIP(hipGetLastError());
  count_launch();
}

static int* g_chain_status = nullptr;

// The three launches of a batched step as one (see gemm_small_chain_kernel).  Returns false -- nothing launched --
// unless the four problems pick exactly the configurations the chained kernel is built from.
bool launch_gemm_small_chain(const GemmProblem& pa, const GemmProblem& pb, const GemmProblem& pc1, const GemmProblem& pc2,
                             hipStream_t s) {
  // OFF by default: measured on MI355X (config 3, 1024 rows), the three launches take 26.6 us; the chained launch
  // 28.4 us with its barriers stubbed out (the 16-wave forms of the 16x16-tile stages are slower than their 8-wave
  // forms), 44 us with flag barriers and no cache maintenance, 74 us with an arrive counter (256 device-scope
  // read-modify-writes of one address serialise at the memory side) and 181 us with the release/acquire fences a
  // correct barrier needs (every workgroup writes back and invalidates its XCD's L2).  A launch boundary costs
  // ~2.3 us.  On an 8-XCD part a grid barrier is an order of magnitude dearer than the boundary it would replace.
  static const int enable = [] { const char* e = ab_getenv("TOPS_STEP_CHAIN"); return e ? atoi(e) : 0; }();
  if (!enable) return false;
  const GemmProblem* ps[4] = {&pa, &pb, &pc1, &pc2};
  for (const GemmProblem* p : ps)
    if (p->dtype != TO_F32 || p->batch != 1 || !gemm_small_can(*p)) return false;
  ChainArgs c{};
  const SmallPlan ca = plan_small<float>(pa, c.g[0]);
  if (!(ca.ts == 32 && ca.nw == 16 && ca.os == 8 && ca.amode == 0 && ca.bmode == 1) || c.g[0].loss_rows) return false;
  const SmallPlan cb = plan_small<float>(pb, c.g[1]);
  if (!(cb.ts == 16 && cb.amode == 0 && cb.bmode == 1 && !cb.f64_t32) || !c.g[1].loss_rows) return false;
  const SmallPlan c1 = plan_small<float>(pc1, c.g[2]);
  if (!(c1.ts == 32 && c1.nw == 16 && c1.os == 8 && c1.amode == 1 && c1.bmode == 0) || c.g[2].loss_rows) return false;
  const SmallPlan c2 = plan_small<float>(pc2, c.g[3]);
  if (!(c2.ts == 16 && c2.amode == 1 && c2.bmode == 0) || c.g[3].loss_rows) return false;
  // the 16x16-tile stages run on all 16 waves here: one-shot slices of at most 8 chunks of 16
  auto sixteen = [](const GemmProblem& p, SmallArgsT<float>& g) {
    const int64_t chunks = (p.K + 15) / 16;
    if (chunks > 128) return false;
    g.kper = (int)(((chunks + 15) / 16) * 16);
    return true;
  };
  if (!sixteen(pb, c.g[1]) || !sixteen(pc2, c.g[3])) return false;
  if (c.g[1].tail_out && c.g[1].tail_n > 256) return false;
  c.nblk[0] = (int)((pa.M + 31) / 32) * c.g[0].tiles_n;
  c.nblk[1] = (int)((pb.M + 15) / 16) * c.g[1].tiles_n;
  c.n1 = (int)((pc1.M + 31) / 32) * c.g[2].tiles_n;
  c.n2 = (int)((pc2.M + 15) / 16) * c.g[3].tiles_n;
  const int grid = std::max(std::max(c.nblk[0], c.nblk[1]), c.n1 + c.n2);
  if (grid > 256 || grid < 1) return false;  // one workgroup per CU: all of them must be resident at once
  constexpr size_t lds = (16 * 16 * 64 + 16 * 64 + 16 * 17 + 16) * sizeof(float);
  static unsigned* ctr = nullptr;
  static int *status = nullptr, *status_dev = nullptr;
  static int resident = -1;
  if (resident < 0) {
    int per_cu = 0;
    hipDeviceProp_t prop;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_small_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_small_chain_kernel, 1024, lds) != hipSuccess ||
        hipGetDeviceProperties(&prop, rt().device) != hipSuccess) {
      (void)hipGetLastError();
      resident = 0;
    } else {
      resident = per_cu * prop.multiProcessorCount;
    }
    if (resident >= 256) {
      TO_HIP(hipMalloc(&ctr, 4096));
      TO_HIP(hipMemset(ctr, 0, 4096));
      TO_HIP(hipHostMalloc(&status, sizeof(int), hipHostMallocMapped));
      *status = 0;
      TO_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&status_dev), status, 0));
    }
  }
  if (resident < grid || !ctr) return false;
  if (*status != 0) return false;  // a barrier timed out once: never again in this process (chain_timed_out reports it)
  c.cs.ctr = ctr;
  c.cs.status = status_dev;
  static const double timeout_s = [] { const char* e = ab_getenv("TOPS_CHAIN_TIMEOUT_S"); return e ? atof(e) : 1.0; }();
  c.cs.timeout = (long long)(timeout_s * 100e6);
  g_chain_status = status;
  static const int dev = [] { const char* e = ab_getenv("TOPS_CHAIN_DEV"); return e ? atoi(e) : 0; }();
  c.cs.dev = dev;
  launch_k(gemm_small_chain_kernel, dim3(grid), dim3(1024), lds, s, c);
  TO_HIP(hipGetLastError());
  count_launch();
  return true;
}

int gemm_small_chain_status() { return g_chain_status ? *g_chain_status : 0; }

static unsigned* g_seam_ctr = nullptr;
static int *g_seam_status = nullptr, *g_seam_status_dev = nullptr;
static bool g_seam_disabled = false;
int gemm_small_seam_status() { return g_seam_status ? *g_seam_status : 0; }
int gemm_small_seam_take_failure() {
  const int st = gemm_small_seam_status();
  if (st) {
    g_seam_disabled = true;   // a launch that timed out once is not trusted again
    *g_seam_status = 0;
  }
  return st;
}

// the seam's row-block counters: allocated and zeroed once, at to_init (a first use inside a stream capture could not)
void gemm_small_seam_init() {
  if (g_seam_ctr) return;
  // plain stores and plain loads of H meet in ONE XCD's L2: every XCD must own whole row blocks, i.e. workgroup b must run
  // on XCD b % 8.  Under a CU mask or another partition mode the probe fails and the seam never launches.
  if (!xcd_placement_probe()) return;
  if (hipMalloc(&g_seam_ctr, 4096 * sizeof(unsigned)) != hipSuccess) {
    g_seam_ctr = nullptr;
    (void)hipGetLastError();
    return;
  }
  if (hipHostMalloc(&g_seam_status, sizeof(int), hipHostMallocMapped) == hipSuccess) {
    *g_seam_status = 0;
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&g_seam_status_dev), g_seam_status, 0) != hipSuccess) g_seam_status_dev = nullptr;
  }
  (void)hipGetLastError();
  if (hipMemset(g_seam_ctr, 0, 4096 * sizeof(unsigned)) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_small_seam_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)((16 * 16 * 64 + 16 * 64 + 16 * 17 + 16) * sizeof(float))) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(g_seam_ctr);
    g_seam_ctr = nullptr;
  }
}

// A forward layer and the loss-head launch that consumes its output as ONE launch (gemm_small_seam_kernel).  Returns
// false -- nothing launched -- unless the two problems are of the form that kernel is built from: the forward problem on
// 16-wave one-shot 32x32 tiles with every XCD owning whole row blocks, the head problem reading the forward's output as
// its A operand, row for row, on 16-row tiles with a K that one batch of two chunks per wave covers.
bool launch_gemm_small_seam(const GemmProblem& pf, const GemmProblem& ph, hipStream_t s) {
  // OFF by default: measured on MI355X (config 3, 1024 rows; profiles/README.md, round 4) the joined launch takes 19.0 us
  // against 11.1 + 6.7 us for the two it replaces -- the head of a row block is a 5.3 us dependent chain (operands that
  // miss the L2 after the kernel boundary, 16-wave reduction, loss head, tail) that now starts behind the SLOWEST tile of
  // its row block and the launch's own 2.4 us dispatch ramp, instead of overlapping the next launch's ramp.
  // TOPS_STEP_SEAM=1: the last arriver is the head; =2: the workgroup of the row block's last tile is, and waits.
  // round 5: the joined form on four-wave tiles, every workgroup of a row block taking four rows of the head (gemm_t32.hip; TOPS_STEP_SEAM=3)
  if (launch_gemm_t32_head(pf, ph, s)) return true;
  static const int enable = [] { const char* e = getenv("TOPS_STEP_SEAM"); const int v = e ? atoi(e) : 0; return v == 1 || v == 2 ? v : 0; }();
  if (!enable || g_seam_disabled || gemm_small_seam_status() != 0) return false;
  if (pf.dtype != TO_F32 || ph.dtype != TO_F32 || pf.batch != 1 || ph.batch != 1) return false;
  if (!gemm_small_can(pf) || !gemm_small_can(ph) || !ph.loss_rows || pf.loss_rows) return false;
  // the head reads exactly what the forward writes: same buffer, same rows, row-major
  if (ph.A != pf.C || ph.M != pf.M || ph.K != pf.N || ph.a_sk != 1 || ph.a_sm != pf.c_sm) return false;
  if (pf.rowsum || pf.dact || pf.beta != 0.0 || ph.beta != 0.0) return false;
  SeamArgs c{};
  const SmallPlan cf = plan_small<float>(pf, c.fwd);
  if (!(cf.ts == 32 && cf.nw == 16 && cf.os == 8 && cf.amode == 0 && cf.bmode == 1)) return false;
  const SmallPlan ch = plan_small<float>(ph, c.head);
  if (!(ch.ts == 16 && ch.nw == 8 && ch.amode == 0 && ch.bmode == 1 && !ch.f64_t32) || c.head.tiles_n != 1) return false;
  if (ph.K > 8 * 2 * 16) return false;            // one batch of two 16-k chunks per wave
  c.head.kper = (int)((((ph.K + 15) / 16 + 7) / 8) * 16);
  c.head.tile_order = 0;                          // head tile = 2 * row block + half
  const int T = c.fwd.tiles_m * c.fwd.tiles_n;
  // every XCD must own whole row blocks: runs of T / 8 tiles of the row-major sequence, no remainder, tiles_n | run
  if (T % 8 != 0 || (T / 8) % c.fwd.tiles_n != 0 || c.fwd.tiles_m > 128) return false;
  c.fwd.tile_order = 1;
  // (a row block whose second 16-row half lies beyond M: every row of that head tile is masked, every load bounds-checked)
  if (!g_seam_ctr) return false;   // (allocated by to_init: never inside a stream capture)
  c.ctr = g_seam_ctr;
  c.mode = enable == 2 ? 2 : 1;
  c.status = g_seam_status_dev;
  static long long* dbg = [] {
    long long* p = nullptr;
    if (ab_getenv("TOPS_SEAM_STAMPS") && hipHostMalloc(&p, 8 * sizeof(long long), hipHostMallocMapped) == hipSuccess) {
      for (int i = 0; i < 8; ++i) p[i] = 0;
      static long long* keep = p;
      atexit([] {
        std::fprintf(stderr, "[seam] row block 0's head (us since its workgroup began): forward done %.2f, stores in L2 %.2f, "
                             "all tiles arrived %.2f, head done %.2f\n",
                     (keep[1] - keep[0]) * 0.01, (keep[2] - keep[0]) * 0.01, (keep[3] - keep[0]) * 0.01, (keep[4] - keep[0]) * 0.01);
      });
      return p;
    }
    return (long long*)nullptr;
  }();
  c.dbg = dbg;
  constexpr size_t lds = (16 * 16 * 64 + 16 * 64 + 16 * 17 + 16) * sizeof(float);
  launch_k(gemm_small_seam_kernel, dim3(T), dim3(1024), lds, s, c);
  TO_HIP(hipGetLastError());
  count_launch();
  return true;
}

void launch_gemm_small(const GemmProblem& p, hipStream_t s) {
  // about one round of 32x32 tiles with a long K (the step's forward layer): four DMA-fed waves per tile (gemm_t32.hip)
  if (gemm_t32_applicable(p)) {
    launch_gemm_t32(p, s);
    return;
  }
  if (p.dtype == TO_F64) launch_small_t<double>(p, s);
  else launch_small_t<float>(p, s);
}

// Two weight-gradient GEMMs (dZ^T . A: A m-contiguous view of dZ, B n-contiguous) in one launch when the
// heuristics pick the (16-wave one-shot 32x32, 8-wave 16x16) pair of configurations -- the shapes of a
// wide hidden layer next to a narrow output layer.  Returns false when the pair is not of that form.
bool launch_gemm_small_pair(const GemmProblem& p1, const GemmProblem& p2, hipStream_t s) {
  static const int enable = [] { const char* e = ab_getenv("TOPS_SMALL_PAIR"); return e ? atoi(e) : 1; }();
  if (!enable || p1.dtype != p2.dtype || p1.batch != 1 || p2.batch != 1) return false;
  if (launch_gemm_t32_pair(p1, p2, s)) return true;   // four DMA-fed waves per 32x32 tile (gemm_t32.hip)
  if (!gemm_small_can(p1) || !gemm_small_can(p2)) return false;
  if (p1.dtype == TO_F64) {
    SmallArgsT<double> g1, g2;
    const SmallPlan c1 = plan_small<double>(p1, g1), c2 = plan_small<double>(p2, g2);
    if (!(c1.f64_t32 && c1.nw == 8 && c1.os == 1 && c1.amode == 1 && c1.bmode == 0)) return false;
    if (!(!c2.f64_t32 && c2.ts == 16 && c2.nw == 8 && c2.os == 8 && c2.amode == 1 && c2.bmode == 0)) return false;
    if (g1.loss_rows || g2.loss_rows) return false;
    const int n1 = (int)((p1.M + 31) / 32) * g1.tiles_n, n2 = (int)((p2.M + 15) / 16) * g2.tiles_n;
    launch_k((gemm_small_pair_f64_kernel<1, 0, 1, 0>), dim3(n1 + n2), dim3(512), 0, s, g1, g2, n1);
    TO_HIP(hipGetLastError());
    count_launch();
    return true;
  }
  SmallArgsT<float> g1, g2;
  const SmallPlan c1 = plan_small<float>(p1, g1), c2 = plan_small<float>(p2, g2);
#ifdef TOPS_AB_KNOBS
  g1.dbg = small_stamps(1);
  g2.dbg = small_stamps(3);
#endif
  if (!(c1.ts == 32 && c1.nw == 16 && c1.os == 8 && c1.amode == 1 && c1.bmode == 0)) return false;
  if (!(c2.ts == 16 && c2.nw == 8 && c2.os == 8 && c2.amode == 1 && c2.bmode == 0)) return false;
  if (g1.loss_rows || g2.loss_rows) return false;
  const int n1 = (int)((p1.M + 31) / 32) * g1.tiles_n, n2 = (int)((p2.M + 15) / 16) * g2.tiles_n;
  dim3 grid(n1 + n2), block(1024);
  // (both one-shot: the two-stage pipeline of the 16x16 body does not fit the 128 registers of a 1024-thread
  //  workgroup; the same K -- the batch -- puts both problems in the one-shot range together anyway)
  launch_k((gemm_small_pair_kernel<float, 1, 0, 16, 32, 8, 1, 0, 8, 16, 8>), grid, block, 0, s, g1, g2, n1);
  TO_HIP(hipGetLastError());
  count_launch();
  return true;
}

}  // namespace to
