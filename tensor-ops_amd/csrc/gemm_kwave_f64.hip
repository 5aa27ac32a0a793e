// fp64 GEMM between the latency-bound and the full-chip regime -- gemm_kwave.hip's design in the reference's own
// element type (`HMat Double`, src/TensorOps/BLAS/HMat.hs:35): C = alpha * A.B (+ bias, activation, act')
//
// One workgroup of four waves per 64x64 output tile, the K loop split over the WAVES: every wave computes the whole
// tile for its own run of 16-deep k-tiles from wave-private LDS images it fills itself by DMA (no barrier in the
// loop), the four partial tiles are added in LDS in wave order, a ragged last k-tile is added by the last wave
// straight from global memory.  The tiled fp64 kernel (gemm_f64.hip) needs 256x128 tiles to keep the L2 behind it and
// fills the chip only from ~2048^2 outputs on (1024^3: 25 TF of 78.6); its mid-size routes split K over workgroups
// with partial products in HBM.
//
// v_mfma_f64_16x16x4_f64: A operand lane l = A[i = l&15][k = l>>4], B operand lane l = B[k = l>>4][j = l&15],
// D: 4 f64 per lane, col = l&15, row = (l>>4) + 4 r.  A wave's 64x64 tile is 4x4 such tiles (128 accumulator
// registers).  A k-tile is consumed in two halves of two MFMA k-steps; lane (x, kg) of half-tile h, step e uses
// k = 8 h + 2 kg + e (A and B agree), so a k-contiguous operand's fragment for a half-tile is ONE ds_read_b128 (its
// image: [x][8 chunks of 2 k], chunk c of row x in slot c ^ (x & 7)), and an m-/n-contiguous operand is read
// row-/column-OWNING from its [k][64] image: lane l15 holds rows 4 l15 .. 4 l15 + 3, two 16-byte reads per k-step.
#include <cstdio>
#include <type_traits>

#include "common.hpp"

namespace to {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <int I, int N, class F>
__device__ __forceinline__ void kw64_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    kw64_static_for<I + 1, N>(f);
  }
}

struct Kw64Args {
  const double* A;
  const double* B;
  double* C;
  int M, N, K;
  long a_sm, a_sk, b_sk, b_sn, c_sm;
  int tiles_m, tiles_n;
  double alpha;
  const double* bias;
  const double* dact;
  const double* cin;   // beta * cin[m * c_sm + n] joins the sum (the layout of C; may BE C: every element is read, then written, by one thread)
  double beta;
  int act, dact_kind;
  int wide;  // 16-byte stores legal (C 16-byte aligned, c_sm and N even)
};

// AMODE: 0 = A k-contiguous (a_sk == 1), 1 = A m-contiguous (a_sm == 1)
// BMODE: 0 = B n-contiguous (b_sn == 1), 1 = B k-contiguous (b_sk == 1)
template <int AMODE, int BMODE, int NW, int NI>
__global__ __launch_bounds__(NW * 64) void gemm_kw64_kernel(Kw64Args g) {
  constexpr int BM = 64, BN = 64, BK = 16, TM = 4, TN = 4, GA = 8, GB = 8;  // GA/GB: 1-KiB DMA pieces per image
  constexpr int IMG = BM * BK;                 // doubles per image (A and B alike)
  constexpr int WAVE_DOUBLES = 2 * NI * IMG;   // a wave's LDS: [NI] A images, [NI] B images
  static_assert(WAVE_DOUBLES >= BM * BN, "the partial tile reuses the wave's images");
  __shared__ __attribute__((aligned(16))) double smem[NW * WAVE_DOUBLES];

  // XCD-aware tile order (as gemm_kw_kernel)
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
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kg = lane >> 4;

  f64x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

  const int KT = g.K / BK;
  const int per = (KT + NW - 1) / NW;
  const int t_begin = wave * per < KT ? wave * per : KT;
  const int t_end = t_begin + per < KT ? t_begin + per : KT;
  const int nT = t_end - t_begin;

  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned lds_a = (unsigned)(unsigned long)(lptr_t)(smem + wave * WAVE_DOUBLES);  // [NI][IMG]
  const unsigned lds_b = lds_a + NI * IMG * 8;                                           // [NI][IMG]

  constexpr int RA = 4, RB = 4;  // LDS reads per half k-tile and operand (k-contiguous: one per tile; owning: 2 steps x 2 pairs)
  static_assert(RA + RB + GA + GB <= 2 * TM * TN, "a slot behind every MFMA of the second half");
  // DMA: scalar base + per-lane 32-bit byte offset (gemm_kwave.hip): piece q's offset biased by -(q % 4) KiB, the base
  // by -3 KiB.  A piece is 1 KiB = 128 doubles of the image; lane l fills doubles 128 q + 2 l, + 1.
  unsigned oa[GA], ob[GB];
#pragma unroll
  for (int q = 0; q < GA; ++q) {
    const int f = q * 128 + lane * 2;
    long e;
    if constexpr (AMODE == 1) {
      long m = m0 + f % BM;  // two consecutive rows (M even: a pair is in or out)
      if (m + 2 > g.M) m = g.M - 2;
      e = (long)(f / BM) * g.a_sk + m;
    } else {
      long m = m0 + f / BK;
      const int x = f / BK;
      if (m >= g.M) m = g.M - 1;
      e = m * g.a_sm + 2 * (((f % BK) / 2) ^ (x & 7));
    }
    oa[q] = (unsigned)(e * 8 + 3072 - (q % 4) * 1024);
  }
#pragma unroll
  for (int q = 0; q < GB; ++q) {
    const int f = q * 128 + lane * 2;
    long e;
    if constexpr (BMODE == 0) {
      long n = n0 + f % BN;
      if (n + 2 > g.N) n = g.N - 2;
      e = (long)(f / BN) * g.b_sk + n;
    } else {
      long n = n0 + f / BK;
      const int x = f / BK;
      if (n >= g.N) n = g.N - 1;
      e = n * g.b_sn + 2 * (((f % BK) / 2) ^ (x & 7));
    }
    ob[q] = (unsigned)(e * 8 + 3072 - (q % 4) * 1024);
  }
  const long step_a = (AMODE == 1 ? (long)BK * g.a_sk : BK) * 8, step_b = (BMODE == 0 ? (long)BK * g.b_sk : BK) * 8;  // bytes
  const char* sa = reinterpret_cast<const char*>(g.A) - 3072 + (long)t_begin * step_a;
  const char* sb = reinterpret_cast<const char*>(g.B) - 3072 + (long)t_begin * step_b;
  // (`; @dma K` / `; @rd K` / `; @images` / `; @advance`: which k-tile's image, relative to the loop's current tile t, an
  //  access touches -- comments for tools/asm_inflight_check.py, see gemm_kwave.hip)
#define KW_DMA(OFF, BASE, IMM, TAG) asm volatile("global_load_lds_dwordx4 %0, %1 offset:" #IMM " ; @dma %2" ::"v"(OFF), "s"(BASE), "n"(TAG) : "memory")
  auto dma = [&](int u, int buf, auto tagc) {  // the tile this DMA fetches is t + tagc
    constexpr int TAG = decltype(tagc)::value;
    const bool isa = u < GA;
    const int q = isa ? u : u - GA;
    if (q % 4 == 0) {
      const unsigned m0v = (isa ? lds_a : lds_b) + buf * IMG * 8 + (q / 4) * 4096;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(m0v) : "memory");
    }
    const unsigned off = isa ? oa[q] : ob[q];
    const char* base = isa ? sa : sb;
    if (q % 4 == 0) KW_DMA(off, base, 0, TAG);
    if (q % 4 == 1) KW_DMA(off, base, 1024, TAG);
    if (q % 4 == 2) KW_DMA(off, base, 2048, TAG);
    if (q % 4 == 3) KW_DMA(off, base, 3072, TAG);
  };
#undef KW_DMA

  // Fragment reads (round 4, as gemm_kwave.hip): a read's address = one per-lane base for (operand, half-tile), + the image's
  // offset (one add per half), + a constant per read in the instruction's offset field -- no address arithmetic per read;
  // two sets of landing registers, one per half-tile parity, and the MFMAs take their operands straight from the set the
  // reads landed in -- no unpacking moves (the loop carried ~80 VALU instructions per k-tile next to its 64 MFMAs).
  // (reads land in registers whose first consumer is the wait: see gemm_kwave.hip)
  f64x2 ta[2][RA], tb[2][RB];
  unsigned a_lane[2], b_lane[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    a_lane[h] = lds_a + (AMODE == 1 ? ((8 * h + 2 * kg) * BM + TM * l15) * 8 : (l15 * BK + 2 * ((4 * h + kg) ^ (l15 & 7))) * 8);
    b_lane[h] = lds_b + (BMODE == 0 ? ((8 * h + 2 * kg) * BN + TN * l15) * 8 : (l15 * BK + 2 * ((4 * h + kg) ^ (l15 & 7))) * 8);
  }
  auto rd = [](f64x2& dst, unsigned addr, auto off, auto tagc) {
    asm volatile("ds_read_b128 %0, %1 offset:%2 ; @rd %3" : "=v"(dst) : "v"(addr), "n"(decltype(off)::value), "n"(decltype(tagc)::value));
  };
  // read r of the half whose lane bases (+ image offset) are abase / bbase, into set `slot`; tagc: the image of tile t + tagc
  auto frag = [&](int slot, unsigned abase, unsigned bbase, auto ri, auto tagc) {
    constexpr int r = decltype(ri)::value;
    if constexpr (r < RA) {
      // owning (m-contiguous): r = (k-step e of the half, row pair q): rows TM l15 + 2q, +1; k-contiguous: r = tile, row 16 r + l15
      constexpr int off = AMODE == 1 ? ((r / (TM / 2)) * BM + 2 * (r % (TM / 2))) * 8 : r * 16 * BK * 8;
      rd(ta[slot][r], abase, std::integral_constant<int, off>{}, tagc);
    } else {
      constexpr int rr = r - RA;
      constexpr int off = BMODE == 0 ? ((rr / (TN / 2)) * BN + 2 * (rr % (TN / 2))) * 8 : rr * 16 * BK * 8;
      rd(tb[slot][rr], bbase, std::integral_constant<int, off>{}, tagc);
    }
  };
  // the reads issued since the last landing are complete: set `slot` is valid from here on
  auto land = [&](int slot) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta[slot][0]), "+v"(ta[slot][1]), "+v"(ta[slot][2]), "+v"(ta[slot][3])::"memory");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tb[slot][0]), "+v"(tb[slot][1]), "+v"(tb[slot][2]), "+v"(tb[slot][3])::"memory");
  };

#define KW64_PIN_ACC asm volatile("" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]), \
    "+a"(acc[1][2]), "+a"(acc[1][3]), "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]), "+a"(acc[3][0]), "+a"(acc[3][1]), \
    "+a"(acc[3][2]), "+a"(acc[3][3]))
  auto tile = [&](auto dma_on, int buf, int bnext) {
    constexpr bool DMA = decltype(dma_on)::value;
    kw64_static_for<0, 2>([&](auto hi) {
      constexpr int h = decltype(hi)::value, cur = h, nxt = h ^ 1;
      if constexpr (h == 1) {
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NI - 2) * (GA + GB)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      // (the reads of this half fetch the NEXT half: image buf, second half -- or the next tile's image, first half)
      const unsigned abase = a_lane[h ^ 1] + (h == 0 ? buf : bnext) * IMG * 8, bbase = b_lane[h ^ 1] + (h == 0 ? buf : bnext) * IMG * 8;
      kw64_static_for<0, 2 * TM * TN>([&](auto ni) {
        constexpr int n = decltype(ni)::value;
        constexpr int e = n / (TM * TN), i = (n % (TM * TN)) / TN, jn = n % TN;
        double av, bv;
        if constexpr (AMODE == 1) av = ta[cur][e * (TM / 2) + i / 2][i % 2]; else av = ta[cur][i][e];
        if constexpr (BMODE == 0) bv = tb[cur][e * (TN / 2) + jn / 2][jn % 2]; else bv = tb[cur][jn][e];
        // (inline asm with the accumulator pinned to AccVGPRs: with the builtin the compiler carried the 128 accumulator
        // registers across the loop's back edge in VGPRs and moved them into AccVGPRs and back out every k-tile -- 256
        // moves next to 64 MFMAs)
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i][jn]) : "v"(av), "v"(bv));
        if constexpr (n < RA + RB) {
          frag(nxt, abase, bbase, ni, std::integral_constant<int, h>{});   // (h == 0: this tile's image, h == 1: the next tile's)
        } else if constexpr (DMA && h == 1 && n < RA + RB + GA + GB) {
          dma(n - (RA + RB), buf, std::integral_constant<int, NI>{});
          if constexpr (n == RA + RB + GA + GB - 1) {
            sa += step_a;
            sb += step_b;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      land(nxt);
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("; @advance");   // (for the checker: the loop's t becomes t + 1)
    // (the accumulators cross the loop's back edge IN AccVGPRs: without this the register allocator carried half of them
    // in VGPRs for two of the four operand layouts and copied them in and out around the MFMAs of every k-tile)
    KW64_PIN_ACC;
  };

  KW64_PIN_ACC;
  if (nT > 0) {
    asm volatile("; @images %0 private" ::"n"(NI));
    // (two written-out paths, each with the wait that matches what it issued: the hazard checker is not path-sensitive)
    static_assert(NI == 2, "the prologue is written out for two images");
    if (nT >= NI) {
#pragma unroll
      for (int u = 0; u < GA + GB; ++u) dma(u, 0, std::integral_constant<int, 0>{});
      sa += step_a;
      sb += step_b;
#pragma unroll
      for (int u = 0; u < GA + GB; ++u) dma(u, 1, std::integral_constant<int, 1>{});
      sa += step_a;
      sb += step_b;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NI - 1) * (GA + GB)) : "memory");
    } else {
#pragma unroll
      for (int u = 0; u < GA + GB; ++u) dma(u, 0, std::integral_constant<int, 0>{});
      sa += step_a;
      sb += step_b;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    kw64_static_for<0, RA + RB>([&](auto ri) { frag(0, a_lane[0], b_lane[0], ri, std::integral_constant<int, 0>{}); });
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
  // The last MFMAs retire before anything but another MFMA touches the AccVGPRs: the accumulators are read-write operands
  // of the statement that holds the wait states (the compiler pads the hazards of the MFMAs it issues itself, not those of
  // an asm string; tools/asm_inflight_check.py rule 6; gemm_kwave.hip).
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15"
               : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]),
                 "+a"(acc[1][3]), "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]), "+a"(acc[3][0]), "+a"(acc[3][1]),
                 "+a"(acc[3][2]), "+a"(acc[3][3])::"memory");

  // the ragged end of K (fewer than 16): the last wave, operands straight from global memory, four k per MFMA
  if (g.K % BK != 0 && wave == NW - 1) {
    long ra[TM], cb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      long m = m0 + (AMODE == 1 ? TM * l15 + i : i * 16 + l15);
      ra[i] = (m < g.M ? m : g.M - 1) * g.a_sm;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      long n = n0 + (BMODE == 0 ? TN * l15 + j : j * 16 + l15);
      cb[j] = (n < g.N ? n : g.N - 1) * g.b_sn;
    }
    for (int kk = KT * BK; kk < g.K; kk += 4) {
      const int k = kk + kg;
      const bool ok = k < g.K;
      const long kc = ok ? k : g.K - 1;
      double av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const double v = g.A[ra[i] + kc * g.a_sk];
        av[i] = ok ? v : 0.0;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const double v = g.B[kc * g.b_sk + cb[j]];
        bv[j] = ok ? v : 0.0;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  }

  // partial tiles -> LDS (each wave into its own, now dead, images), summed in wave order
  // D reg r lane l -> row (l>>4) + 4 r, col l&15 of the MFMA tile
  {
    double* P = smem + wave * WAVE_DOUBLES;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = BMODE == 0 ? TN * l15 + j : j * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int tr = kg + 4 * r;
          const int row = AMODE == 1 ? TM * tr + i : i * 16 + tr;
          P[row * BN + col] = acc[i][j][r];
        }
      }
  }
  __syncthreads();
  // (two instantiations of the way out: a plain product has no per-element branches on bias / activation / act')
  auto finish = [&](auto plainc) {
    constexpr bool PLAIN = decltype(plainc)::value;
    for (int q = tid; q < BM * BN / 2; q += NW * 64) {
      const int row = q / (BN / 2), c2 = (q % (BN / 2)) * 2;
      f64x2 s = *reinterpret_cast<const f64x2*>(smem + row * BN + c2);
#pragma unroll
      for (int w = 1; w < NW; ++w) s += *reinterpret_cast<const f64x2*>(smem + w * WAVE_DOUBLES + row * BN + c2);
      const long gr = m0 + row, gc = n0 + c2;
      if (gr >= g.M || gc >= g.N) continue;
      double* dst = g.C + gr * g.c_sm + gc;
      if constexpr (PLAIN) {  // (wide: N % 2 == 0, a pair is in or out)
        *reinterpret_cast<f64x2*>(dst) = g.alpha * s;
      } else {
        double v[2] = {s.x, s.y};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if (gc + e >= g.N) break;
          double x = g.alpha * v[e];
          if (g.cin) x += g.beta * g.cin[gr * g.c_sm + gc + e];
          if (g.bias) x += g.bias[gc + e];
          if (g.act == 1) x = 1.0 / (1.0 + exp(-x));
          else if (g.act == 2) x = tanh(x);
          if (g.dact) {
            const double hh = g.dact[gr * g.c_sm + gc + e];
            x *= g.dact_kind ? 1.0 - hh * hh : hh * (1.0 - hh);
          }
          v[e] = x;
        }
        if (g.wide) {
          f64x2 o = {v[0], v[1]};
          *reinterpret_cast<f64x2*>(dst) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 2; ++e)
            if (gc + e < g.N) dst[e] = v[e];
        }
      }
    }
  };
  if (g.wide && !g.bias && g.act == 0 && !g.dact && !g.cin) finish(std::true_type{});
  else finish(std::false_type{});
}

static int kw64_mode() {
  static const int m = [] { const char* e = ab_getenv("TOPS_GEMM64_KW"); return e ? atoi(e) : 1; }();
  return m;
}

static bool kw64_can(const GemmProblem& p) {
  if (p.dtype != TO_F64 || p.batch != 1 || p.reduce_batch || p.rowsum || p.loss_rows || (p.beta != 0.0 && !p.Cin)) return false;
  if (p.M < 16 || p.N < 16 || (p.M < 128 && p.N < 128) || p.K < 16) return false;   // (16 .. 127 rows or columns: the last tile is padding; loads clamp, stores are guarded)
  if (p.M > 2147483647LL || p.N > 2147483647LL || p.K > 2147483647LL) return false;
  const bool a_k = p.a_sk == 1, a_m = !a_k && p.a_sm == 1;
  const bool b_n = p.b_sn == 1, b_k = !b_n && p.b_sk == 1;
  if (!(a_k || a_m) || !(b_n || b_k)) return false;
  auto al8 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 7u) == 0; };
  if (!al8(p.A) || !al8(p.B)) return false;
  const int64_t ext_a = a_k ? p.M * p.a_sm : 16 * p.a_sk + p.M, ext_b = b_k ? p.N * p.b_sn : 16 * p.b_sk + p.N;
  if (ext_a * 8 + 8192 >= (1LL << 32) || ext_b * 8 + 8192 >= (1LL << 32) || p.a_sm < 0 || p.a_sk < 0 || p.b_sk < 0 || p.b_sn < 0) return false;
  if (a_m && p.M % 2 != 0) return false;  // an m-contiguous pair must be in or out of the matrix as a whole
  if (b_n && p.N % 2 != 0) return false;
  return true;
}

bool gemm_kw64_applicable(const GemmProblem& p) {
  const int mode = kw64_mode();
  if (mode == 0 || !kw64_can(p)) return false;
  if (mode >= 2) return true;
  // 128 KiB of LDS and ~330 registers: one workgroup per CU, so the win is where ONE round of 64x64 tiles covers the
  // output (ours before / now, TF: 768^3 16 / 25, 1000^3 20 / 44, 1024^3 25 / 43, 4096 x 784 x 256 26 / 43); from two
  // rounds on the tiled kernel is ahead again (1536^3 50 / 37, 2048^3 63 / 52)
  const long t64 = ((p.M + 63) / 64) * ((p.N + 63) / 64);
  if (t64 >= 100 && t64 <= 320 && p.K >= 128) return true;
  // Round 6 (last), tools/gemm_scan.py with SCAN_DTYPE=f64 (profiles/r06_gemm_scan_f64_*.txt): 432 of 969 shapes below 0.92 of the
  // vendor fp64 GEMM, half of them at 0.5 -- every extent the 256x128 tiles do not fit went to the compiler-scheduled kernel
  // (~23 TF).  This kernel forced onto those rows was 1.2 .. 2.6 x ahead on 181 of the 302 without a sliver (us, before / here;
  // vendor): 256 x 10000 x 10000 2180 / 848 (1089), 784 x 4096 x 2048 489 / 236 (245), 300 x 4096 x 60000 5070 / 2289 (3113),
  // 60000 x 784 x 300 803 / 519 (556), 2048 x 300 x 1024 97 / 28 (24); and on shapes the tiles DO fit up to 1,024 tiles
  // (2048 x 1024 x 1024 89 / 67 (73), 2048 x 256 x 1024 57 / 24 (20)); behind from a few thousand tiles on where they fit
  // (2048 x 256 x 60000 1045 / 1367).
  if (t64 < 100 || p.K < 64) return false;
  const long narrow = p.M < p.N ? p.M : p.N;
  if (narrow < 128 && t64 < 200) return false;   // (100 x 4096 x 4096: 97 us on the small-GEMM route, 118 here; 10000 x K x 100 twice as fast here)
  // (K <= 256 under thousands of tiles with both extents large: block + strips on the tiled kernel is ahead -- 2048 x 256 x 60000
  //  1.04 ms there, 1.36 here; 784 x 256 x 60000 0.62 / 0.54 the other way)
  if (p.K <= 256 && p.K % 16 == 0 && narrow >= 1024 && t64 > 2048) return false;
  const bool fits = p.M % 256 == 0 && p.N % 128 == 0 && p.K % 16 == 0;
  return !fits || t64 <= 1024;
}

// A tile per WAVE (the fp32 kernel's SPLIT = false): a K of a few hundred is not worth sharing among four waves.
static bool kw64_wave_per_tile(const GemmProblem& p) {
  static const int forced = [] { const char* e = ab_getenv("TOPS_GEMM64_KW_NW"); return e ? atoi(e) : 0; }();   // (development: 1 / 4)
  if (forced) return forced == 1;
  const long t64 = ((p.M + 63) / 64) * ((p.N + 63) / 64);
  // us, four waves a tile / a wave a tile: 4096 x 100 x 4096 121 / 101, 60000 x 100 x 1024 480 / 335, 2048 x 100 x 2048 33 / 21; behind from
  // K = 256 on (4096 x 256 x 4096 179 / 212, 10000 x 784 x 2048 537 / 710: one wave a SIMD has nothing to hide its waits under)
  // (... and two waves a tile are ahead of both from two tiles a CU on: 4096 x 100 x 4096 121 / 102 / 88, 4096 x 128 x 4096 122 / 120 / 96)
  return t64 >= 1024 && t64 < 2048 && p.K <= 128;
}

void launch_gemm_kw64(const GemmProblem& p, hipStream_t s) {
  Kw64Args g{};
  g.A = (const double*)p.A; g.B = (const double*)p.B; g.C = (double*)p.C;
  g.M = (int)p.M; g.N = (int)p.N; g.K = (int)p.K;
  g.a_sm = p.a_sm; g.a_sk = p.a_sk; g.b_sk = p.b_sk; g.b_sn = p.b_sn; g.c_sm = p.c_sm;
  g.tiles_m = (int)((p.M + 63) / 64);
  g.tiles_n = (int)((p.N + 63) / 64);
  g.alpha = p.alpha;
  g.bias = (const double*)p.bias; g.dact = (const double*)p.dact; g.act = p.act; g.dact_kind = p.dact_kind;
  g.cin = p.beta != 0.0 ? (const double*)p.Cin : nullptr; g.beta = (double)p.beta;
  g.wide = (reinterpret_cast<uintptr_t>(p.C) & 15u) == 0 && p.c_sm % 2 == 0 && p.N % 2 == 0;
  const int mode = (p.a_sk == 1 ? 0 : 2) + (p.b_sn == 1 ? 0 : 1);
  dim3 grid(g.tiles_m * g.tiles_n);
  // Two waves a tile (64 KiB of LDS: two workgroups a CU) from two tiles a CU on: the K loop is shared by two waves instead of four
  // and a CU works on two tiles at once -- us, four / two waves: 4096 x 256 x 4096 177 / 152, 10000 x 784 x 2048 536 / 486,
  // 2048 x 2048 x 2048 247 / 239; at one tile a CU half the SIMDs would idle (1024^3 34.5 / 62.2).
  static const int nw_forced = [] { const char* e = ab_getenv("TOPS_GEMM64_KW_NW"); return e ? atoi(e) : 0; }();
  const long t64 = (long)g.tiles_m * g.tiles_n;
  if (nw_forced == 2 || (nw_forced == 0 && t64 >= 512 && !kw64_wave_per_tile(p))) {
    const dim3 block(128);
    switch (mode) {
      case 0: launch_k((gemm_kw64_kernel<0, 0, 2, 2>), grid, block, 0, s, g); break;
      case 1: launch_k((gemm_kw64_kernel<0, 1, 2, 2>), grid, block, 0, s, g); break;
      case 2: launch_k((gemm_kw64_kernel<1, 0, 2, 2>), grid, block, 0, s, g); break;
      default: launch_k((gemm_kw64_kernel<1, 1, 2, 2>), grid, block, 0, s, g); break;
    }
    TO_HIP(hipGetLastError());
    count_launch();
    return;
  }
  if (kw64_wave_per_tile(p)) {   // (NW = 1: a workgroup IS a wave -- 32 KiB of LDS, four or five of them a CU)
    const dim3 block(64);
    switch (mode) {
      case 0: launch_k((gemm_kw64_kernel<0, 0, 1, 2>), grid, block, 0, s, g); break;
      case 1: launch_k((gemm_kw64_kernel<0, 1, 1, 2>), grid, block, 0, s, g); break;
      case 2: launch_k((gemm_kw64_kernel<1, 0, 1, 2>), grid, block, 0, s, g); break;
      default: launch_k((gemm_kw64_kernel<1, 1, 1, 2>), grid, block, 0, s, g); break;
    }
  } else {
    const dim3 block(256);
    switch (mode) {
      case 0: launch_k((gemm_kw64_kernel<0, 0, 4, 2>), grid, block, 0, s, g); break;
      case 1: launch_k((gemm_kw64_kernel<0, 1, 4, 2>), grid, block, 0, s, g); break;
      case 2: launch_k((gemm_kw64_kernel<1, 0, 4, 2>), grid, block, 0, s, g); break;
      default: launch_k((gemm_kw64_kernel<1, 1, 4, 2>), grid, block, 0, s, g); break;
    }
  }
  TO_HIP(hipGetLastError());
  count_launch();
}

}  // namespace to
