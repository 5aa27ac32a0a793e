// fp32 GEMM for the sizes between the latency-bound and the full-chip regime (a few hundred 64x64 output tiles:
// 768^3 .. 1536^3, the Learn layers' shapes at a few thousand rows): C = alpha * A.B (+ bias, activation)
//
// Serves `gmul` (src/TensorOps/Types.hs:60-66) and `gemm` of `class BLAS` (src/TensorOps/BLAS.hs:108-123) where
// gemm_f32_mfma.hip's 128x128 tiles need a K split over workgroups to fill the chip and pay for it with partial
// products in HBM and a second pass (1024^3: 7 of 33 us).
//
// Shape of the kernel: one workgroup per 64x64 output tile, and the K loop split over its WAVES.  Every wave computes
// the whole 64x64 tile for its own run of k-tiles:
//  * a wave's operands are its own: each wave DMAs its 64x16 slab of A and 16x64 slab of B straight into its private
//    LDS images (global_load_lds_dwordx4) and waits on nothing but its own vmcnt -- there is no barrier in the K loop;
//  * the MFMA stream per wave is the one of the pinned 128x128 body (four 32x32 accumulators in AccVGPRs, fragments
//    for a half k-tile read as ONE ds_read_b128 / b64 per operand tile, one non-MFMA instruction pinned behind each
//    MFMA);
//  * the partial tiles meet in LDS (the images are dead by then), are added in wave order -- deterministic -- and
//    leave through 16-byte row stores with bias / activation applied;
//  * the last k-tile may be ragged (K % 16 != 0): the last wave adds it from global memory directly, two k per MFMA,
//    so a K tail costs no second launch (1000^3).
// Edge tiles (M or N no multiple of 64): a lane's row / column is fixed for the whole K loop, so lanes beyond the
// extent re-read the last valid one (clamped once in the pointer set-up) and their outputs are never stored.
#include <cstdio>
#include <type_traits>

#include "common.hpp"

namespace to {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct KwArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  long a_sm, a_sk, b_sk, b_sn, c_sm;
  int tiles_m, tiles_n;
  float alpha;
  const float* bias;
  const float* dact;
  int act, dact_kind;
  int wide;  // 16-byte stores legal (C aligned, c_sm % 4 == 0, N % 4 == 0)
};

// AMODE: 0 = A k-contiguous (a_sk == 1), 1 = A m-contiguous (a_sm == 1)
// BMODE: 0 = B n-contiguous (b_sn == 1), 1 = B k-contiguous (b_sk == 1)
// NW waves split the K loop; NI LDS images per wave and operand
template <int AMODE, int BMODE, int NW, int NI>
__global__ __launch_bounds__(NW * 64) void gemm_kw_kernel(KwArgs g) {
  constexpr int BM = 64, BN = 64, BK = 16, TM = 2, TN = 2, GA = 4, GB = 4;
  constexpr int IMG = BM * BK;                // floats per image (A and B alike)
  constexpr int WAVE_FLOATS = 2 * NI * IMG;   // a wave's LDS: [NI] A images, [NI] B images
  static_assert(WAVE_FLOATS >= BM * BN, "the partial tile reuses the wave's images");
  __shared__ __attribute__((aligned(16))) float smem[NW * WAVE_FLOATS];

  // XCD-aware tile order (as gemm_mfma_kernel): block b runs on XCD b % 8; each XCD gets a contiguous run of the
  // tile sequence, which walks the tile grid in bands of R tile-rows, column-major inside a band
  const int nblk = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tile_m, tile_n;
  {
    constexpr int R = 4;
    const int band = bid / (R * g.tiles_n);
    const int rows = (g.tiles_m - band * R) < R ? (g.tiles_m - band * R) : R;
    const int in = bid - band * R * g.tiles_n;
    tile_n = in / rows;
    tile_m = band * R + in % rows;
  }
  const long m0 = (long)tile_m * BM, n0 = (long)tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform: loop bounds and LDS bases stay scalar)
  const int l31 = lane & 31, half = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // this wave's run of whole k-tiles
  const int KT = g.K / BK;
  const int per = (KT + NW - 1) / NW;
  const int t_begin = wave * per < KT ? wave * per : KT;
  const int t_end = t_begin + per < KT ? t_begin + per : KT;
  const int nT = t_end - t_begin;

  float* Ag = smem + wave * WAVE_FLOATS;  // [NI][IMG]
  float* Bg = Ag + NI * IMG;              // [NI][IMG]
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  // LDS images (gemm_f32_mfma.hip, PF == 5): a wave instruction fills 1 KiB linearly (lane * 16 B), which element
  // a lane fetches shapes the image.  k-contiguous operand: [x][4 slots of 4 k], k-chunk c of row x in slot
  // c ^ ((x >> 1) & 3); m-/n-contiguous operand: [k][64].
  constexpr int RA = AMODE == 1 ? 4 : TM, RB = BMODE == 0 ? 4 : TN;  // LDS reads per half k-tile
  static_assert(RA + RB + GA + GB <= 4 * TM * TN, "a slot behind every MFMA of the second half");
  const float* pa[GA];
  const float* pb[GB];
#pragma unroll
  for (int q = 0; q < GA; ++q) {
    const int f = q * 256 + lane * 4;
    if constexpr (AMODE == 1) {
      long m = m0 + f % BM;  // four consecutive rows (M % 4 == 0: a quad is in or out)
      if (m + 4 > g.M) m = g.M - 4;
      pa[q] = g.A + (long)(f / BM) * g.a_sk + m;
    } else {
      long m = m0 + f / BK;
      if (m >= g.M) m = g.M - 1;
      pa[q] = g.A + m * g.a_sm + 4 * (((f % BK) / 4) ^ (((f / BK) >> 1) & 3));
    }
  }
#pragma unroll
  for (int q = 0; q < GB; ++q) {
    const int f = q * 256 + lane * 4;
    if constexpr (BMODE == 0) {
      long n = n0 + f % BN;
      if (n + 4 > g.N) n = g.N - 4;
      pb[q] = g.B + (long)(f / BN) * g.b_sk + n;
    } else {
      long n = n0 + f / BK;
      if (n >= g.N) n = g.N - 1;
      pb[q] = g.B + n * g.b_sn + 4 * (((f % BK) / 4) ^ (((f / BK) >> 1) & 3));
    }
  }
  const long step_a = AMODE == 1 ? (long)BK * g.a_sk : BK, step_b = BMODE == 0 ? (long)BK * g.b_sk : BK;
  // (the instruction offset advances BOTH addresses: the pieces of an operand share one M0 value, their global
  //  pointers are pre-biased by -1 KiB per piece)
#pragma unroll
  for (int q = 0; q < GA; ++q) pa[q] += (long)t_begin * step_a - q * 256;
#pragma unroll
  for (int q = 0; q < GB; ++q) pb[q] += (long)t_begin * step_b - q * 256;
  auto dma = [&](int u, int buf) {
    if (u < GA) {
      float* dst = Ag + buf * IMG;
      if (u == 0) __builtin_amdgcn_global_load_lds((gptr_t)pa[0], (lptr_t)dst, 16, 0, 0);
      if (u == 1) __builtin_amdgcn_global_load_lds((gptr_t)pa[1], (lptr_t)dst, 16, 1024, 0);
      if (u == 2) __builtin_amdgcn_global_load_lds((gptr_t)pa[2], (lptr_t)dst, 16, 2048, 0);
      if (u == 3) __builtin_amdgcn_global_load_lds((gptr_t)pa[3], (lptr_t)dst, 16, 3072, 0);
    } else {
      float* dst = Bg + buf * IMG;
      const int v = u - GA;
      if (v == 0) __builtin_amdgcn_global_load_lds((gptr_t)pb[0], (lptr_t)dst, 16, 0, 0);
      if (v == 1) __builtin_amdgcn_global_load_lds((gptr_t)pb[1], (lptr_t)dst, 16, 1024, 0);
      if (v == 2) __builtin_amdgcn_global_load_lds((gptr_t)pb[2], (lptr_t)dst, 16, 2048, 0);
      if (v == 3) __builtin_amdgcn_global_load_lds((gptr_t)pb[3], (lptr_t)dst, 16, 3072, 0);
    }
  };
  auto bump = [&](int u) {
    if (u < GA) pa[u] += step_a;
    else pb[u - GA] += step_b;
  };

  float a[2][4][TM], b[2][4][TN];  // [slot][k-step][tile]
  // LDS reads as inline asm (the compiler would order every LDS read it can see behind ALL outstanding LDS DMA).
  // The compiler does not know these reads are asynchronous: it may copy a result register right behind the read,
  // before the data is there (it did, where the two tile loops rotate the fragment registers).  So a read lands in
  // a temporary that has ONE consumer, the wait itself ("+v": the wait hands the value on) -- whatever the compiler
  // does with the value, it does behind the wait.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v ta4[TM], tb4[TN];  // k-contiguous operand: one b128 per 32-row tile (four k-steps)
  f32x2 ta2[4], tb2[4];     // m-/n-contiguous operand: one b64 per k-step (the lane's TM / TN owned rows / columns)
  const unsigned lds_a = (unsigned)(unsigned long)(lptr_t)Ag, lds_b = (unsigned)(unsigned long)(lptr_t)Bg;
  // lane (x, half) of half-tile h uses k = 4 (2 h + half) + ss for MFMA step ss (A and B agree on it).
  // An m-contiguous A / n-contiguous B is read row-/column-OWNING: lane l31 holds rows TM*l31 .. TM*l31+TM-1.
  auto frag = [&](int buf, int h, int r) {
    if (r < RA) {
      const unsigned base = lds_a + buf * IMG * 4;
      if constexpr (AMODE == 1) {
        const unsigned addr = base + ((4 * (2 * h + half) + r) * BM + TM * l31) * 4;  // r = k-step
        asm volatile("ds_read_b64 %0, %1" : "=v"(ta2[r]) : "v"(addr));
      } else {
        const int x = r * 32 + l31;
        const unsigned addr = base + (x * 4 + ((2 * h + half) ^ ((x >> 1) & 3))) * 16;
        asm volatile("ds_read_b128 %0, %1" : "=v"(ta4[r]) : "v"(addr));
      }
    } else {
      const int rr = r - RA;
      const unsigned base = lds_b + buf * IMG * 4;
      if constexpr (BMODE == 0) {
        const unsigned addr = base + ((4 * (2 * h + half) + rr) * BN + TN * l31) * 4;
        asm volatile("ds_read_b64 %0, %1" : "=v"(tb2[rr]) : "v"(addr));
      } else {
        const int x = rr * 32 + l31;
        const unsigned addr = base + (x * 4 + ((2 * h + half) ^ ((x >> 1) & 3))) * 16;
        asm volatile("ds_read_b128 %0, %1" : "=v"(tb4[rr]) : "v"(addr));
      }
    }
  };
  // the reads issued since the last landing are complete: hand them to fragment slot `slot`
  auto land = [&](int slot) {
    if constexpr (AMODE == 0 && BMODE == 0)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta4[0]), "+v"(ta4[1]), "+v"(tb2[0]), "+v"(tb2[1]), "+v"(tb2[2]), "+v"(tb2[3])::"memory");
    else if constexpr (AMODE == 0 && BMODE == 1)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta4[0]), "+v"(ta4[1]), "+v"(tb4[0]), "+v"(tb4[1])::"memory");
    else if constexpr (AMODE == 1 && BMODE == 0)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta2[0]), "+v"(ta2[1]), "+v"(ta2[2]), "+v"(ta2[3]), "+v"(tb2[0]), "+v"(tb2[1]), "+v"(tb2[2]), "+v"(tb2[3])::"memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta2[0]), "+v"(ta2[1]), "+v"(ta2[2]), "+v"(ta2[3]), "+v"(tb4[0]), "+v"(tb4[1])::"memory");
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) a[slot][ss][i] = AMODE == 1 ? ta2[ss][i] : ta4[i][ss];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) b[slot][ss][j] = BMODE == 0 ? tb2[ss][j] : tb4[j][ss];
  };

  // one k-tile: two halves of 16 MFMAs; behind each MFMA one pinned other instruction: the next half's fragments
  // and (second half, DMA) the fetch of tile t + NI into the image this tile just left
  auto tile = [&](auto dma_on, int buf, int bnext) {
    constexpr bool DMA = decltype(dma_on)::value;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int cur = h, nxt = h ^ 1;
      if (h == 1) {  // the next tile's image has landed (the wave's own DMA: its vmcnt is all the ordering needed)
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NI - 2) * (GA + GB)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < 4 * TM * TN; ++n) {
        const int ss = n / (TM * TN), i = (n % (TM * TN)) / TN, jn = n % TN;
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i][jn]) : "v"(a[cur][ss][i]), "v"(b[cur][ss][jn]));
        if (n < RA + RB) {
          if (h == 0) frag(buf, 1, n);
          else frag(bnext, 0, n);
        } else if (DMA && h == 1 && n < RA + RB + GA + GB) {
          const int u = n - (RA + RB);
          dma(u, buf);
          bump(u);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      land(nxt);  // (the next half's fragments, issued under these MFMAs)
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if (nT > 0) {
    // prologue: up to NI tiles in flight
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (i < nT) {
#pragma unroll
        for (int u = 0; u < GA + GB; ++u) {
          dma(u, i);
          bump(u);
        }
      }
    if (nT >= NI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NI - 1) * (GA + GB)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < RA + RB; ++r) frag(0, 0, r);
    land(0);
    __builtin_amdgcn_sched_barrier(0);
    int buf = 0, t = 0;
    for (; t + NI < nT; ++t) {
      const int bnext = buf + 1 == NI ? 0 : buf + 1;
      tile(std::true_type{}, buf, bnext);
      buf = bnext;
    }
    for (; t < nT; ++t) {
      const int bnext = buf + 1 == NI ? 0 : buf + 1;
      tile(std::false_type{}, buf, bnext);
      buf = bnext;
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // the ragged end of K (fewer than 16): the last wave, operands straight from global memory, two k per MFMA
  if (g.K % BK != 0 && wave == NW - 1) {
    // (compiler-scheduled MFMAs here: it knows their hazards, not those of the inline-asm stream before them)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    long ra[TM], cb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      long m = m0 + (AMODE == 1 ? TM * l31 + i : i * 32 + l31);
      ra[i] = (m < g.M ? m : g.M - 1) * g.a_sm;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      long n = n0 + (BMODE == 0 ? TN * l31 + j : j * 32 + l31);
      cb[j] = (n < g.N ? n : g.N - 1) * g.b_sn;
    }
    for (int kk = KT * BK; kk < g.K; kk += 2) {
      const int k = kk + half;
      const bool ok = k < g.K;
      const long kc = ok ? k : g.K - 1;
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float v = g.A[ra[i] + kc * g.a_sk];
        av[i] = ok ? v : 0.f;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float v = g.B[kc * g.b_sk + cb[j]];
        bv[j] = ok ? v : 0.f;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs retire before the AccVGPRs are read

  // partial tiles -> LDS (each wave into its own, now dead, images), summed in wave order
  // D reg r lane l -> row (r&3) + 8*(r>>2) + 4*half, col l31 of the MFMA tile
  {
    float* P = smem + wave * WAVE_FLOATS;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = BMODE == 0 ? TN * l31 + j : j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int tr = (r & 3) + 8 * (r >> 2) + 4 * half;
          const int row = AMODE == 1 ? TM * tr + i : i * 32 + tr;
          P[row * BN + col] = acc[i][j][r];
        }
      }
  }
  __syncthreads();
  typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int it = 0; it < BM * BN / 4 / (NW * 64); ++it) {
    const int q = it * NW * 64 + tid;
    const int row = q / (BN / 4), c4 = (q % (BN / 4)) * 4;
    f32x4 s = *reinterpret_cast<const f32x4*>(smem + row * BN + c4);
#pragma unroll
    for (int w = 1; w < NW; ++w) s += *reinterpret_cast<const f32x4*>(smem + w * WAVE_FLOATS + row * BN + c4);
    const long gr = m0 + row, gc = n0 + c4;
    if (gr >= g.M || gc >= g.N) continue;
    float v[4] = {s.x, s.y, s.z, s.w};
    float* dst = g.C + gr * g.c_sm + gc;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (gc + e >= g.N) break;
      float x = g.alpha * v[e];
      if (g.bias) x += g.bias[gc + e];
      if (g.act == 1) x = 1.0f / (1.0f + expf(-x));
      else if (g.act == 2) x = tanhf(x);
      if (g.dact) {
        const float hh = g.dact[gr * g.c_sm + gc + e];
        x *= g.dact_kind ? 1.0f - hh * hh : hh * (1.0f - hh);
      }
      v[e] = x;
    }
    if (g.wide) {
      f32x4 o = {v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(dst) = o;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (gc + e < g.N) dst[e] = v[e];
    }
  }
}

static int kw_mode() {
  static const int m = [] { const char* e = getenv("TOPS_GEMM_KW"); return e ? atoi(e) : 1; }();
  return m;
}

// Can the problem run here at all?
static bool kw_can(const GemmProblem& p) {
  if (p.dtype != TO_F32 || p.batch != 1 || p.reduce_batch || p.rowsum || p.loss_rows || p.beta != 0.0) return false;
  if (p.M < 64 || p.N < 64 || p.K < 16) return false;
  if (p.M > 2147483647LL || p.N > 2147483647LL || p.K > 2147483647LL) return false;
  const bool a_k = p.a_sk == 1, a_m = !a_k && p.a_sm == 1;
  const bool b_n = p.b_sn == 1, b_k = !b_n && p.b_sk == 1;
  if (!(a_k || a_m) || !(b_n || b_k)) return false;
  auto al4 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 3u) == 0; };
  if (!al4(p.A) || !al4(p.B)) return false;  // (global memory runs in unaligned-access mode: dword alignment is enough)
  if (a_m && p.M % 4 != 0) return false;     // an m-contiguous quad must be in or out of the matrix as a whole
  if (b_n && p.N % 4 != 0) return false;
  return true;
}

// ... and should it?  A tile count that fills the chip once or twice with 64x64 tiles, and a K long enough to split
// over a workgroup's waves.
bool gemm_kw_applicable(const GemmProblem& p) {
  const int mode = kw_mode();
  if (mode == 0 || !kw_can(p)) return false;
  if (mode >= 2) return true;
  // measured against the routes of gemm_f32_mfma.hip (tools/kw_check.py time): ahead from ~100 tiles (640^3) to ~1000
  // (1792^3 98 vs 93 TF, 2048^3 level); below, the small-GEMM kernel's in-workgroup split-K wins (512^3), above, the
  // 256x256 tiles (16384 x 256 x 4096: 110 vs 98)
  const long t64 = ((p.M + 63) / 64) * ((p.N + 63) / 64);
  return t64 >= 100 && t64 <= 1024 && p.K >= 128;
}

void launch_gemm_kw(const GemmProblem& p, hipStream_t s) {
  KwArgs g{};
  g.A = (const float*)p.A; g.B = (const float*)p.B; g.C = (float*)p.C;
  g.M = (int)p.M; g.N = (int)p.N; g.K = (int)p.K;
  g.a_sm = p.a_sm; g.a_sk = p.a_sk; g.b_sk = p.b_sk; g.b_sn = p.b_sn; g.c_sm = p.c_sm;
  g.tiles_m = (int)((p.M + 63) / 64);
  g.tiles_n = (int)((p.N + 63) / 64);
  g.alpha = (float)p.alpha;
  g.bias = (const float*)p.bias; g.dact = (const float*)p.dact; g.act = p.act; g.dact_kind = p.dact_kind;
  g.wide = (reinterpret_cast<uintptr_t>(p.C) & 15u) == 0 && p.c_sm % 4 == 0 && p.N % 4 == 0;
  const int am = p.a_sk == 1 ? 0 : 1, bm = p.b_sn == 1 ? 0 : 1;
  // Two images per wave and operand: 64 KiB per workgroup, so two workgroups share a CU and one's waits hide under the
  // other's MFMAs (against three images / one workgroup per CU: 1024^3 90 -> 92 TF, 1536^3 86 -> 94, 2048^3 117 -> 126).
  // TOPS_GEMM_KW_WAVES=8: eight waves (128 KiB), TOPS_GEMM_KW_NI=3: three images (96 KiB) -- for A/B runs.
  static const int nw8 = [] { const char* e = getenv("TOPS_GEMM_KW_WAVES"); return e ? atoi(e) == 8 : 0; }();
  static const int ni3 = [] { const char* e = getenv("TOPS_GEMM_KW_NI"); return e ? atoi(e) == 3 : 0; }();
  dim3 grid(g.tiles_m * g.tiles_n);
  if (nw8) {
    dim3 block(512);
    switch (am * 2 + bm) {
      case 0: launch_k((gemm_kw_kernel<0, 0, 8, 2>), grid, block, 0, s, g); break;
      case 1: launch_k((gemm_kw_kernel<0, 1, 8, 2>), grid, block, 0, s, g); break;
      case 2: launch_k((gemm_kw_kernel<1, 0, 8, 2>), grid, block, 0, s, g); break;
      default: launch_k((gemm_kw_kernel<1, 1, 8, 2>), grid, block, 0, s, g); break;
    }
  } else if (ni3) {
    dim3 block(256);
    switch (am * 2 + bm) {
      case 0: launch_k((gemm_kw_kernel<0, 0, 4, 3>), grid, block, 0, s, g); break;
      case 1: launch_k((gemm_kw_kernel<0, 1, 4, 3>), grid, block, 0, s, g); break;
      case 2: launch_k((gemm_kw_kernel<1, 0, 4, 3>), grid, block, 0, s, g); break;
      default: launch_k((gemm_kw_kernel<1, 1, 4, 3>), grid, block, 0, s, g); break;
    }
  } else {
    dim3 block(256);
    switch (am * 2 + bm) {
      case 0: launch_k((gemm_kw_kernel<0, 0, 4, 2>), grid, block, 0, s, g); break;
      case 1: launch_k((gemm_kw_kernel<0, 1, 4, 2>), grid, block, 0, s, g); break;
      case 2: launch_k((gemm_kw_kernel<1, 0, 4, 2>), grid, block, 0, s, g); break;
      default: launch_k((gemm_kw_kernel<1, 1, 4, 2>), grid, block, 0, s, g); break;
    }
  }
  TO_HIP(hipGetLastError());
  count_launch();
}

}  // namespace to
