// fp64 GEMM on the gfx950 matrix cores (`ElemT = Double`, what the reference's apps
// instantiate: `HMat Double`, src/TensorOps/BLAS/HMat.hs:35).
//
// v_mfma_f64_16x16x4_f64: A operand lane l = A[i = l&15][k = l>>4], B operand lane l =
// B[k = l>>4][j = l&15] (one f64 each); D holds 4 f64 per lane with  col = l&15,
// row = (l>>4) + 4*r  -- NOT the f32 map (cdna_hip_programming.md section 3).
// Block tile BM x BN x 16 with WM x WN waves, every wave a 64x64 (or 32x32) sub-tile.
// fp64 operands double the bytes per flop, so the tile has to be large for the L2 to keep up:
// 256x128 (8 waves of 64x64: the 128 accumulator registers per wave rule out 16 waves) when
// the problem has enough such tiles, 128x128 (4 waves) for mid sizes, 64x64 for small ones.
// Same staging scheme as the fp32 kernel: global -> registers -> LDS image [k][m] / [k][n],
// two LDS buffers, one barrier per k-tile, XCD-aware band rasterization of the tile grid;
// all loads bounds-checked by clamp+select (branch-free).
#include <algorithm>
#include <type_traits>

#include "common.hpp"

namespace to {

typedef double f64x4 __attribute__((ext_vector_type(4)));

struct G64 {
  const double* A;
  const double* B;
  double* C;
  const double* Cin;
  int M, N, K;
  long a_sm, a_sk, b_sk, b_sn, c_sm, a_sb, b_sb, c_sb;
  int nb_reduce;
  int tiles_m, tiles_n;
  double alpha, beta;
};

template <int BM, int BN, int WM, int WN, bool GUARD>
__device__ __forceinline__ void gemm_f64_body(const G64& g, int tile_m, int tile_n) {
  constexpr int BK = 16, NT = WM * WN * 64, LDA = BM + 2, LDB = BN + 2;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;       // MFMA tiles per wave
  constexpr int QA = BM * BK / NT, QB = BN * BK / NT;       // staged elements per thread
  extern __shared__ double smem[];
  double (*As)[BK][LDA] = reinterpret_cast<double (*)[BK][LDA]>(smem);
  double (*Bs)[BK][LDB] = reinterpret_cast<double (*)[BK][LDB]>(smem + 2 * BK * LDA);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const long m0 = (long)tile_m * BM, n0 = (long)tile_n * BN;
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
  const bool red = g.nb_reduce > 1;
  const long bz = blockIdx.z;
  const double* Ab = g.A + (red ? 0 : bz * g.a_sb);
  const double* Bb = g.B + (red ? 0 : bz * g.b_sb);

  f64x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

  const int KT = (g.K + BK - 1) / BK, T = KT * g.nb_reduce;
  double ra[QA], rb[QB];
  const bool a_kc = g.a_sk == 1;  // walk k fastest when A is k-contiguous, else m fastest
  const bool b_nc = g.b_sn == 1;
  const bool kfull = (g.K & 15) == 0;  // guarded tiles: only M/N ragged
  auto gload = [&](int t) {
    const int bb = t / KT, kt = t - bb * KT;
    const long k0 = (long)kt * BK;
    const double* Ap = Ab + (red ? (long)bb * g.a_sb : 0);
    const double* Bp = Bb + (red ? (long)bb * g.b_sb : 0);
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int e = tid + q * NT;
      const int am = a_kc ? e / BK : e % BM, ak = a_kc ? e % BK : e / BM;
      if constexpr (GUARD) {
        if (kfull) {  // whole k-tiles: a row beyond M feeds only outputs that are never stored -- clamp, no select
          const long mm = m0 + am < g.M ? m0 + am : g.M - 1;
          ra[q] = Ap[mm * g.a_sm + (k0 + ak) * g.a_sk];
        } else {
          const bool av = (m0 + am < g.M) && (k0 + ak < g.K);
          const double x = Ap[(av ? m0 + am : 0) * g.a_sm + (av ? k0 + ak : 0) * g.a_sk];
          ra[q] = av ? x : 0.0;
        }
      } else {
        ra[q] = Ap[(m0 + am) * g.a_sm + (k0 + ak) * g.a_sk];
      }
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int e = tid + q * NT;
      const int bn = b_nc ? e % BN : e / BK, bk = b_nc ? e / BN : e % BK;
      if constexpr (GUARD) {
        if (kfull) {
          const long nn = n0 + bn < g.N ? n0 + bn : g.N - 1;
          rb[q] = Bp[(k0 + bk) * g.b_sk + nn * g.b_sn];
        } else {
          const bool bv = (n0 + bn < g.N) && (k0 + bk < g.K);
          const double y = Bp[(bv ? k0 + bk : 0) * g.b_sk + (bv ? n0 + bn : 0) * g.b_sn];
          rb[q] = bv ? y : 0.0;
        }
      } else {
        rb[q] = Bp[(k0 + bk) * g.b_sk + (n0 + bn) * g.b_sn];
      }
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int e = tid + q * NT;
      const int am = a_kc ? e / BK : e % BM, ak = a_kc ? e % BK : e / BM;
      As[buf][ak][am] = ra[q];
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int e = tid + q * NT;
      const int bn = b_nc ? e % BN : e / BK, bk = b_nc ? e / BN : e % BK;
      Bs[buf][bk][bn] = rb[q];
    }
  };
  if (T > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const int buf = t & 1;
    // (the accumulators cross the back edge IN AccVGPRs: left alone, the register allocator carried them in VGPRs and
    // moved them in and out around the MFMAs of every k-tile -- 137 v_accvgpr moves next to 64 MFMAs)
    // (four waves: 512 registers a lane.  The eight-wave form has 256, all of them busy: pinned, it spills)
    if constexpr (WM * WN <= 4) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+a"(acc[i][j]));
    }
    if (t + 1 < T) gload(t + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      double a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[buf][kk * 4 + kq][wm0 + i * 16 + l15];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[buf][kk * 4 + kq][wn0 + j * 16 + l15];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < T) lstore(buf ^ 1);
    __syncthreads();
  }
  double* Cb = g.C + (red ? 0 : bz * g.c_sb);
  const double* Ci = g.Cin ? g.Cin + (red ? 0 : bz * g.c_sb) : nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = m0 + wm0 + i * 16 + kq + 4 * r;  // f64 map: row = (lane>>4) + 4*reg
        const long col = n0 + wn0 + j * 16 + l15;
        if (!GUARD || (row < g.M && col < g.N)) {
          double v = g.alpha * acc[i][j][r];
          if (Ci) v += g.beta * Ci[row * g.c_sm + col];
          Cb[row * g.c_sm + col] = v;
        }
      }
}


// ---- full tiles, plain K loop: four waves of 128x64 on a written-out schedule ---------------------------
template <int I, int N, class F>
__device__ __forceinline__ void g64_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    g64_static_for<I + 1, N>(f);
  }
}

// The fp64 twin of the fp32 kernel's PF == 5 path (gemm_f32_mfma.hip, where the reasoning is spelled out):
// 256x128x16 tile, accumulators (32 tiles x 4 doubles = 256 registers per lane) in the AccVGPR file through
// inline-asm MFMAs, operands global -> LDS by DMA into two image pairs, fragments fetched 16 bytes = two
// doubles at a time, the instruction order pinned with sched_barrier.
//  * k-contiguous operand: image [x][8 slots of 2 k], k-pair c of row x in slot c ^ (x & 7) (a row is exactly
//    the 32 banks, so eight consecutive rows must land in eight different slots); lane (l15, g) reads slot
//    4 h + g for half-tile h and uses its two doubles for the two k-steps of that half
//    (k = 8 h + 2 g + e: A and B agree, so the assignment is legal).
//  * m-/n-contiguous operand: image [k][x]; lane l15 OWNS the TM (TN) consecutive rows (columns)
//    TM*l15 .. TM*l15+TM-1 of the wave's sub-tile, so one 16-byte read feeds two tiles; the epilogue maps the
//    permutation back.
// AMODE 0: A k-contiguous, 1: m-contiguous;  BMODE 0: B n-contiguous, 1: k-contiguous;  NWM x NWN waves
// one run: k-tiles [kb, ke) of output tile (tile_m, tile_n); out/out_sm: where the result goes (C, or a stream-K
// partial slot in tile-local coordinates)
template <int AMODE, int BMODE, int NWM, int NWN>
__device__ __forceinline__ void gemm_f64_w4_body(const G64& g, int tile_m, int tile_n, int kb, int ke, double* out,
                                                 long out_sm) {
  constexpr int BM = 256, BN = 128, BK = 16, NWAVES = NWM * NWN;
  constexpr int TM = BM / NWM / 16, TN = BN / NWN / 16;              // 16x16 MFMA tiles per wave
  constexpr int GA = BM * BK / 128 / NWAVES, GB = BN * BK / 128 / NWAVES;  // 1 KiB DMA pieces per wave
  constexpr int RA = TM, RB = TN;                                    // LDS reads per half-tile (both layouts)
  static_assert(TM % 2 == 0 && TN % 2 == 0 && GA >= 1 && GB >= 1, "16-byte fragments feed two tiles");
  static_assert(RA + RB + 2 * (GA + GB) <= 2 * TM * TN, "the last half has a slot for every instruction");
  extern __shared__ double smem[];
  double* Ag = smem;                 // [2][BM*BK]
  double* Bg = smem + 2 * BM * BK;   // [2][BN*BK]
  typedef __attribute__((address_space(3))) void* lptr_t;
  const long m0 = (long)tile_m * BM, n0 = (long)tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kg = lane >> 4;
  const int wm0 = (wave / NWN) * (BM / NWM), wn0 = (wave % NWN) * (BN / NWN);
  const long bz = blockIdx.z;
  const double* Ab = g.A + bz * g.a_sb;
  const double* Bb = g.B + bz * g.b_sb;

  f64x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

  // DMA addressing as in the fp32 body: a per-lane byte offset from the tile's first row / column (32 bits, fixed for
  // the whole K loop) on a SCALAR base that advances by a constant per k-tile; the pieces of an operand share one M0
  // value (piece q's lane offset biased by -q KiB, +3 KiB on every offset and -3 KiB on the base).
  constexpr int IMG_A = BM * BK * 8, IMG_B = BN * BK * 8;  // bytes per image
  unsigned oa[GA], ob[GB];
#pragma unroll
  for (int q = 0; q < GA; ++q) {
    const int f = (wave * GA + q) * 128 + lane * 2;  // first double of this lane's 16 bytes in the image
    long e;
    if constexpr (AMODE == 1) e = (long)(f / BM) * g.a_sk + f % BM;
    else e = (long)(f / BK) * g.a_sm + 2 * (((f % BK) / 2) ^ ((f / BK) & 7));
    oa[q] = (unsigned)(e * 8 + 3072 - q * 1024);
  }
#pragma unroll
  for (int q = 0; q < GB; ++q) {
    const int f = (wave * GB + q) * 128 + lane * 2;
    long e;
    if constexpr (BMODE == 0) e = (long)(f / BN) * g.b_sk + f % BN;
    else e = (long)(f / BK) * g.b_sn + 2 * (((f % BK) / 2) ^ ((f / BK) & 7));
    ob[q] = (unsigned)(e * 8 + 3072 - q * 1024);
  }
  const long step_a = (AMODE == 1 ? (long)BK * g.a_sk : BK) * 8, step_b = (BMODE == 0 ? (long)BK * g.b_sk : BK) * 8;  // bytes
  static_assert(GA <= 4 && GB <= 4, "piece offsets are written out up to 3 KiB");
  const unsigned lds_a = (unsigned)(unsigned long)(lptr_t)Ag, lds_b = (unsigned)(unsigned long)(lptr_t)Bg;
  // (M0 is written inside the asm: nothing else in this kernel uses it)
  const unsigned m0_a = __builtin_amdgcn_readfirstlane(lds_a + wave * GA * 1024), m0_b = __builtin_amdgcn_readfirstlane(lds_b + wave * GB * 1024);
  // (uniform by construction; the "s" constraint alone does not move a value into scalar registers)
  auto uniform64 = [](const void* q) {
    const unsigned long v = reinterpret_cast<unsigned long>(q);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const char*>(((unsigned long)hi << 32) | lo);
  };
  const char* sa = uniform64(reinterpret_cast<const char*>(Ab + m0 * g.a_sm) - 3072 + (long)kb * step_a);
  const char* sb = uniform64(reinterpret_cast<const char*>(Bb + n0 * g.b_sn) - 3072 + (long)kb * step_b);
  // (`; @dma K` / `; @rd K` / `; @images` / `; @advance`: which k-tile's image, relative to the loop's current tile t, an
  //  access touches -- comments for tools/asm_inflight_check.py, which proves the waits and barriers below on the
  //  generated code, back edge included)
#define G64_DMA(OFF, BASE, IMM, TAG) asm volatile("global_load_lds_dwordx4 %0, %1 offset:" #IMM " ; @dma %2" ::"v"(OFF), "s"(BASE), "n"(TAG) : "memory")
  auto dma = [&](auto uc, int buf, auto tagc) {  // the tile this DMA fetches is t + tagc
    constexpr int u = decltype(uc)::value, TAG = decltype(tagc)::value;
    constexpr bool isa = u < GA;
    constexpr int q = isa ? u : u - GA;
    if constexpr (q == 0) {
      const unsigned mv = isa ? m0_a + buf * IMG_A : m0_b + buf * IMG_B;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(mv) : "memory");
    }
    const unsigned off = isa ? oa[q] : ob[q];
    const char* base = isa ? sa : sb;
    if constexpr (q == 0) G64_DMA(off, base, 0, TAG);
    if constexpr (q == 1) G64_DMA(off, base, 1024, TAG);
    if constexpr (q == 2) G64_DMA(off, base, 2048, TAG);
    if constexpr (q == 3) G64_DMA(off, base, 3072, TAG);
  };
#undef G64_DMA
  double a[2][2][TM], b[2][2][TN];  // [slot][k-step of the half][tile]
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  // A fragment read's address = a per-lane base for (operand, half-tile, image) + a constant in the instruction's
  // offset field; the bases of the two images are swapped once per tile (four VALU instructions per k-tile).
  // ax[1] / bx[1]: second half of the current image; ax[0] / bx[0]: first half of the NEXT image
  // (every fragment, whatever its image, takes k = 8 h + 2 kg + e for k-step e of half-tile h: A and B agree)
  unsigned ax[2], bx[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    ax[h] = lds_a + (h == 0 ? IMG_A : 0) + (AMODE == 1 ? ((8 * h + 2 * kg) * BM + wm0 + TM * l15) * 8
                                                     : ((wm0 + l15) * BK + 2 * ((4 * h + kg) ^ (l15 & 7))) * 8);
    bx[h] = lds_b + (h == 0 ? IMG_B : 0) + (BMODE == 0 ? ((8 * h + 2 * kg) * BN + wn0 + TN * l15) * 8
                                                     : ((wn0 + l15) * BK + 2 * ((4 * h + kg) ^ (l15 & 7))) * 8);
  }
  auto rd128 = [](unsigned addr, auto off, auto tagc) { f64x2 v; asm volatile("ds_read_b128 %0, %1 offset:%2 ; @rd %3" : "=v"(v) : "v"(addr), "n"(decltype(off)::value), "n"(decltype(tagc)::value)); return v; };
  // LDS read r (0..RA+RB-1) of the half-tile whose bases are abase / bbase
  auto frag = [&](auto slotc, unsigned abase, unsigned bbase, auto rc, auto tagc) {  // reads the image of tile t + tagc
    constexpr int slot = decltype(slotc)::value, r = decltype(rc)::value;
    if constexpr (r < RA) {
      if constexpr (AMODE == 1) {  // r = (k-step e of the half, row pair q): rows TM*l15 + 2q, +1
        constexpr int e = r / (TM / 2), q = r % (TM / 2);
        const f64x2 v = rd128(abase, std::integral_constant<int, (e * BM + 2 * q) * 8>{}, tagc);
        a[slot][e][2 * q] = v.x; a[slot][e][2 * q + 1] = v.y;
      } else {                     // r = tile: row wm0 + 16 r + l15, k-pair 4h + kg
        const f64x2 v = rd128(abase, std::integral_constant<int, r * 16 * BK * 8>{}, tagc);
        a[slot][0][r] = v.x; a[slot][1][r] = v.y;
      }
    } else {
      constexpr int rr = r - RA;
      if constexpr (BMODE == 0) {  // rr = (k-step e, column pair q)
        constexpr int e = rr / (TN / 2), q = rr % (TN / 2);
        const f64x2 v = rd128(bbase, std::integral_constant<int, (e * BN + 2 * q) * 8>{}, tagc);
        b[slot][e][2 * q] = v.x; b[slot][e][2 * q + 1] = v.y;
      } else {
        const f64x2 v = rd128(bbase, std::integral_constant<int, rr * 16 * BK * 8>{}, tagc);
        b[slot][0][rr] = v.x; b[slot][1][rr] = v.y;
      }
    }
  };
  typedef std::integral_constant<int, 0> c0_t;
  const int T = ke - kb;
  typedef std::integral_constant<int, 1> c1_t;
  typedef std::integral_constant<int, 2> c2_t;
  asm volatile("; @images 2 shared");
  g64_static_for<0, GA + GB>([&](auto uc) { dma(uc, 0, c0_t{}); });
  sa += T > 1 ? step_a : 0;
  sb += T > 1 ? step_b : 0;
  g64_static_for<0, GA + GB>([&](auto uc) { dma(uc, 1, c1_t{}); });
  // (the DMA is inline asm: the compiler does not know there is anything to wait for -- tile 0 has landed)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GA + GB) : "memory");
  __syncthreads();
  g64_static_for<0, RA + RB>([&](auto rc) { frag(c0_t{}, ax[0] - IMG_A, bx[0] - IMG_B, rc, c0_t{}); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int buf = 0;
  int dimg_a = -IMG_A, dimg_b = -IMG_B;  // (what moves a base to the other image: alternates in sign)
  for (int t = 0; t < T; ++t) {
    const long da = t + 2 < T ? step_a : 0, db = t + 2 < T ? step_b : 0;
    g64_static_for<0, 2>([&](auto hc) {
      constexpr int h = decltype(hc)::value, cur = h;
      typedef std::integral_constant<int, (h ^ 1)> nxt_t;
      if constexpr (h == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      __builtin_amdgcn_sched_barrier(0);
      g64_static_for<0, 2 * TM * TN>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        constexpr int e = n / (TM * TN), i = (n % (TM * TN)) / TN, jn = n % TN;
        acc[i][jn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][e][i], b[cur][e][jn], acc[i][jn], 0, 0, 0);
        if constexpr (n < RA + RB) {
          frag(nxt_t{}, ax[h ^ 1], bx[h ^ 1], nc, hc);   // (h == 0: this tile's image, h == 1: the next tile's)
        } else if constexpr (n < RA + RB + 2) {
          // the bases this half has just used move to the other image
          if constexpr (n == RA + RB) ax[h ^ 1] += (h == 0 ? -dimg_a : dimg_a);
          else bx[h ^ 1] += (h == 0 ? -dimg_b : dimg_b);
        } else if constexpr (h == 1 && n < RA + RB + 2 + GA + GB) {
          if constexpr (n == RA + RB + 2) {  // the scalar bases on to tile t+2, its DMA into the image just released
            sa += da;
            sb += db;
          }
          dma(std::integral_constant<int, n - (RA + RB + 2)>{}, buf, c2_t{});
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      // (the next half's fragments, issued under the first MFMAs of this one, are waited for HERE, at the end of the half
      //  that issued them: nothing is in flight across the loop's back edge or its exit -- the compiler handed the
      //  registers of the last, unused prefetch to the epilogue's address arithmetic ahead of the old post-loop wait,
      //  tools/asm_inflight_check.py found `v_lshl_add_u64 v[130:131]` there)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    });
    buf ^= 1;
    dimg_a = -dimg_a;
    dimg_b = -dimg_b;
    asm volatile("; @advance");   // (for the checker: the loop's t becomes t + 1)
  }
  // (the last, unused DMA has landed before this workgroup -- or the next one on this CU -- reuses the LDS)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  __syncthreads();
  double* Cb = out + bz * g.c_sb;
  const double* Ci = g.Cin ? g.Cin + bz * g.c_sb : nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tr = kg + 4 * r;  // row within the MFMA tile
      const long row = m0 + wm0 + (AMODE == 1 ? TM * tr + i : i * 16 + tr);
      if constexpr (BMODE == 0) {  // the lane's TN tiles are TN consecutive columns: 32 contiguous bytes
        const long col = n0 + wn0 + TN * l15;
        double v[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          v[j] = g.alpha * acc[i][j][r];
          if (Ci) v[j] += g.beta * Ci[row * g.c_sm + col + j];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) Cb[row * out_sm + col + j] = v[j];
      } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const long col = n0 + wn0 + j * 16 + l15;
          double v = g.alpha * acc[i][j][r];
          if (Ci) v += g.beta * Ci[row * g.c_sm + col];
          Cb[row * out_sm + col] = v;
        }
      }
    }
}

template <int AMODE, int BMODE, int NWM, int NWN>
__global__ __launch_bounds__(NWM * NWN * 64) void gemm_f64_w4_kernel(G64 g) {
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  constexpr int R = 4;
  const int band = bid / (R * g.tiles_n);
  const int rows = (g.tiles_m - band * R) < R ? (g.tiles_m - band * R) : R;
  const int in = bid - band * R * g.tiles_n;
  gemm_f64_w4_body<AMODE, BMODE, NWM, NWN>(g, band * R + in % rows, in / rows, 0, g.K / 16, g.C, g.c_sm);
}

// stream-K (see gemm_f32_mfma.hip): one workgroup per CU, equal shares of the k-tile stream, whole-tile runs
// write C, partial runs a 256x128 partial into the workgroup's own slot, the fix-up adds them in workgroup order
// (hybrid, as in fp32: whole rounds of tiles [0, tile0) one tile per workgroup straight into C, the stream covers the rest)
struct StreamK64 {
  int T, upw, total, tile0;
  double* part;  // [2 * workgroups][256*128]
};

template <int AMODE, int BMODE>
__global__ __launch_bounds__(512) void gemm_f64_streamk_kernel(G64 g, StreamK64 sk) {
  constexpr int R = 4;
  {
    const int nblk = gridDim.x, xcd = blockIdx.x & 7, q = nblk >> 3, r = nblk & 7;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    for (int tile = bid; tile < sk.tile0; tile += nblk) {
      const int band = tile / (R * g.tiles_n);
      const int rows = (g.tiles_m - band * R) < R ? (g.tiles_m - band * R) : R;
      const int in = tile - band * R * g.tiles_n;
      gemm_f64_w4_body<AMODE, BMODE, 4, 2>(g, band * R + in % rows, in / rows, 0, sk.T, g.C, g.c_sm);
      __syncthreads();
    }
  }
  int u = blockIdx.x * sk.upw;
  const int u_end = (u + sk.upw < sk.total) ? u + sk.upw : sk.total;
  bool first = true;
  while (u < u_end) {
    const int st = u / sk.T;
    const int tile = sk.tile0 + st;
    const int kb = u - st * sk.T;
    const int ke = (sk.T - kb < u_end - u) ? sk.T : kb + (u_end - u);
    const int band = tile / (R * g.tiles_n);
    const int rows = (g.tiles_m - band * R) < R ? (g.tiles_m - band * R) : R;
    const int in = tile - band * R * g.tiles_n;
    const int tile_n = in / rows, tile_m = band * R + in % rows;
    if (kb == 0 && ke == sk.T) {
      gemm_f64_w4_body<AMODE, BMODE, 4, 2>(g, tile_m, tile_n, kb, ke, g.C, g.c_sm);
    } else {
      double* slot = sk.part + (size_t)(2 * blockIdx.x + (first ? 0 : 1)) * (256 * 128);
      gemm_f64_w4_body<AMODE, BMODE, 4, 2>(g, tile_m, tile_n, kb, ke, slot - ((long)tile_m * 256 * 128 + (long)tile_n * 128), 128);
    }
    __syncthreads();
    u += ke - kb;
    first = false;
  }
}

__global__ __launch_bounds__(256) void streamk64_fixup_kernel(double* C, long c_sm, int tiles_m, int tiles_n, StreamK64 sk) {
  const int tile = sk.tile0 + blockIdx.y;
  const int u0 = blockIdx.y * sk.T, u1 = u0 + sk.T;
  const int w_lo = u0 / sk.upw, w_hi = (u1 - 1) / sk.upw;
  if (w_lo == w_hi) return;
  constexpr int R = 4;
  const int band = tile / (R * tiles_n);
  const int rows = (tiles_m - band * R) < R ? (tiles_m - band * R) : R;
  const int in = tile - band * R * tiles_n;
  const long m0 = (long)(band * R + in % rows) * 256, n0 = (long)(in / rows) * 128;
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  for (int q = blockIdx.x * 256 + threadIdx.x; q < 256 * 64; q += gridDim.x * 256) {  // 16384 pairs per tile
    f64x2 acc = {0.0, 0.0};
    for (int w = w_lo; w <= w_hi; ++w) {
      const int which = (w * sk.upw >= u0) ? 0 : 1;
      acc += *reinterpret_cast<const f64x2*>(sk.part + (size_t)(2 * w + which) * 32768 + (size_t)q * 2);
    }
    const int r = q / 64, c2 = (q % 64) * 2;
    *reinterpret_cast<f64x2*>(C + (m0 + r) * c_sm + n0 + c2) = acc;
  }
}

// interior tiles (whole tile in range, K a multiple of 16) take the unguarded body: a guard's
// select on the loaded value makes the compiler wait for the load before the MFMAs of the
// current k-tile, which serialises HBM/L2 latency with the matrix pipe.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void gemm_f64_kernel(G64 g) {
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  constexpr int R = 4;
  const int band = bid / (R * g.tiles_n);
  const int rows = (g.tiles_m - band * R) < R ? (g.tiles_m - band * R) : R;
  const int in = bid - band * R * g.tiles_n;
  const int tile_m = band * R + in % rows, tile_n = in / rows;
  const bool interior = (tile_m + 1) * BM <= g.M && (tile_n + 1) * BN <= g.N && (g.K & 15) == 0;
  if (interior) gemm_f64_body<BM, BN, WM, WN, false>(g, tile_m, tile_n);
  else gemm_f64_body<BM, BN, WM, WN, true>(g, tile_m, tile_n);
}

template <int BM, int BN, int WM, int WN>
static void launch_cfg(G64& g, const GemmProblem& p, hipStream_t s) {
  g.tiles_m = (int)((p.M + BM - 1) / BM);
  g.tiles_n = (int)((p.N + BN - 1) / BN);
  constexpr size_t lds = (size_t)2 * 16 * ((BM + 2) + (BN + 2)) * sizeof(double);
  static bool once = [] {
    (void)hipFuncSetAttribute((const void*)gemm_f64_kernel<BM, BN, WM, WN>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return true;
  }();
  (void)once;
  dim3 grid(g.tiles_m * g.tiles_n, 1, p.reduce_batch ? 1 : (unsigned)p.batch);
  launch_k((gemm_f64_kernel<BM, BN, WM, WN>), grid, dim3(WM * WN * 64), lds, s, g);
}

// Would launch_gemm_f64 run this problem on the full-tile pinned kernel with (nearly) whole rounds of tiles?
// (run_gemm carves such a block out of a ragged problem, as for fp32.)
bool gemm_f64_w4_full_rounds(const GemmProblem& p) {
  static const int w4 = [] { const char* e = ab_getenv("TOPS_GEMM64_W4"); return e ? atoi(e) : 1; }();
  static const int variant = [] { const char* v = ab_getenv("TOPS_GEMM64_VARIANT"); return v ? atoi(v) : 0; }();
  if (!w4 || (variant != 0 && variant != 4) || p.dtype != TO_F64 || p.reduce_batch || p.batch > 65535) return false;
  if (p.M % 256 || p.N % 128 || p.K % 16 || p.K < 32) return false;
  const long tiles = (p.M / 256) * (p.N / 128) * p.batch;
  const bool plain = p.batch == 1 && p.beta == 0.0;  // stream-K's terms: rounds do not matter then
  if (tiles < 128) return false;
  if (!plain && (tiles < 256 || 100 * tiles < 94 * ((tiles + 255) / 256) * 256)) return false;
  const bool a_kc = p.a_sk == 1 && !(p.K == 1 && p.a_sm == 1), a_mc = p.a_sm == 1;
  const bool b_nc = p.b_sn == 1 && !(p.N == 1 && p.b_sk == 1), b_kc = p.b_sk == 1;
  return (a_kc || a_mc) && (b_nc || b_kc);
}

void launch_gemm_f64(const GemmProblem& p, hipStream_t s) {
  G64 g{};
  g.A = (const double*)p.A; g.B = (const double*)p.B; g.C = (double*)p.C;
  g.Cin = (p.beta != 0.0) ? (const double*)p.Cin : nullptr;
  g.M = (int)p.M; g.N = (int)p.N; g.K = (int)p.K;
  g.a_sm = p.a_sm; g.a_sk = p.a_sk; g.b_sk = p.b_sk; g.b_sn = p.b_sn; g.c_sm = p.c_sm;
  g.a_sb = p.a_sb; g.b_sb = p.b_sb; g.c_sb = p.c_sb;
  g.nb_reduce = p.reduce_batch ? (int)p.batch : 1;
  g.alpha = p.alpha; g.beta = p.beta;
  static const int variant = [] { const char* v = ab_getenv("TOPS_GEMM64_VARIANT"); return v ? atoi(v) : 0; }();
  const long nb = p.reduce_batch ? 1 : p.batch;
  const long t256 = ((p.M + 255) / 256) * ((p.N + 127) / 128) * nb;
  const long t128 = ((p.M + 127) / 128) * ((p.N + 127) / 128) * nb;
  int v = variant;
  // full tiles, plain K loop, whole rounds: the 4-wave kernel on the written-out schedule
  static const int w4 = [] { const char* e = ab_getenv("TOPS_GEMM64_W4"); return e ? atoi(e) : 1; }();
  const bool a_kc = p.a_sk == 1 && !(p.K == 1 && p.a_sm == 1), a_mc = p.a_sm == 1;
  const bool b_nc = p.b_sn == 1 && !(p.N == 1 && p.b_sk == 1), b_kc = p.b_sk == 1;
  const long tw4 = (p.M / 256) * (p.N / 128) * nb;
  const bool plain = nb == 1 && p.beta == 0.0;
  static const int streamk = [] { const char* e = ab_getenv("TOPS_GEMM64_STREAMK"); return e ? atoi(e) : 1; }();
  const bool sk_ok = streamk && plain && tw4 >= 32 && tw4 <= 65535 && tw4 * (p.K / 16) >= 256 * 12 &&
                     10 * tw4 < 9 * ((tw4 + 255) / 256) * 256;
  if ((v == 0 || v == 4) && w4 && !p.reduce_batch && p.M % 256 == 0 && p.N % 128 == 0 && p.K % 16 == 0 && p.K >= 32 &&
      (tw4 >= 256 || sk_ok) && (a_kc || a_mc) && (b_nc || b_kc) && p.batch <= 65535 &&
      // (32-bit lane offsets from the tile's origin: 256 rows or 16 k-steps times the stride, in bytes)
      std::max(std::max(p.a_sm, p.a_sk), std::max(p.b_sk, p.b_sn)) < (1LL << 20)) {
    g.tiles_m = (int)(p.M / 256);
    g.tiles_n = (int)(p.N / 128);
    constexpr size_t lds = (size_t)2 * 16 * (256 + 128) * sizeof(double);
    if (sk_ok) {  // tile count that does not fill whole rounds: equal shares of the k-tile stream
      StreamK64 sk{};
      sk.T = (int)(p.K / 16);
      static const int hybrid = [] { const char* e = ab_getenv("TOPS_GEMM_STREAMK_HYBRID"); return e ? atoi(e) : 1; }();
      long dp_rounds = hybrid ? tw4 / 256 : 0;
      if (dp_rounds > 0 && (tw4 - dp_rounds * 256) * sk.T < 256 * 8) --dp_rounds;
      sk.tile0 = (int)(dp_rounds * 256);
      sk.total = (int)((tw4 - sk.tile0) * sk.T);
      sk.upw = (sk.total + 255) / 256;
      const int64_t wd[2] = {512, 32768};
      Holder work;
      work.t = new_tensor(2, wd, 0, TO_F64);
      sk.part = static_cast<double*>(work.t->ptr);
      const int mode = (a_kc ? 0 : 1) * 2 + (b_nc ? 0 : 1);
#define TOPS_SK64(AM, BM_)                                                                                      \
  {                                                                                                             \
    static bool once = [] {                                                                                     \
      (void)hipFuncSetAttribute((const void*)gemm_f64_streamk_kernel<AM, BM_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      return true;                                                                                              \
    }();                                                                                                        \
    (void)once;                                                                                                 \
    launch_k((gemm_f64_streamk_kernel<AM, BM_>), dim3(256), dim3(512), lds, s, g, sk);                \
  }
      switch (mode) {
        case 0: TOPS_SK64(0, 0) break;
        case 1: TOPS_SK64(0, 1) break;
        case 2: TOPS_SK64(1, 0) break;
        default: TOPS_SK64(1, 1) break;
      }
#undef TOPS_SK64
      TO_HIP(hipGetLastError());
      count_launch();
      launch_k(streamk64_fixup_kernel, dim3(8, (unsigned)(tw4 - sk.tile0)), dim3(256), 0, s, g.C, (long)g.c_sm, g.tiles_m,
                         g.tiles_n, sk);
      TO_HIP(hipGetLastError());
      count_launch();
      return;
    }
    dim3 grid(g.tiles_m * g.tiles_n, 1, (unsigned)p.batch);
    const int mode = (a_kc ? 0 : 1) * 2 + (b_nc ? 0 : 1);
#define TOPS_W4_64(AM, BM_)                                                                                     \
  {                                                                                                             \
    static bool once = [] {                                                                                     \
      (void)hipFuncSetAttribute((const void*)gemm_f64_w4_kernel<AM, BM_, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      return true;                                                                                              \
    }();                                                                                                        \
    (void)once;                                                                                                 \
    launch_k((gemm_f64_w4_kernel<AM, BM_, 4, 2>), grid, dim3(512), lds, s, g);                        \
  }
    switch (mode) {
      case 0: TOPS_W4_64(0, 0) break;
      case 1: TOPS_W4_64(0, 1) break;
      case 2: TOPS_W4_64(1, 0) break;
      default: TOPS_W4_64(1, 1) break;
    }
#undef TOPS_W4_64
    TO_HIP(hipGetLastError());
    count_launch();
    return;
  }
  if (v == 0) v = t256 >= 200 ? 256 : (t128 >= 128 ? 128 : 64);
  if (v == 256) launch_cfg<256, 128, 4, 2>(g, p, s);
  else if (v == 128) launch_cfg<128, 128, 2, 2>(g, p, s);
  else launch_cfg<64, 64, 2, 2>(g, p, s);
  TO_HIP(hipGetLastError());
  count_launch();
}

}  // namespace to
