// fp32 GEMM on a MENU of output tiles -- 48x48, 48x64, 64x48, 80x80, 32x64, 64x32 -- for mid sizes whose 64x64 tile count quantises badly
// against the 256 CUs: C = alpha * A.B (+ bias, activation)
//
// Serves `gmul` (src/TensorOps/Types.hs:60-66) and `gemm` of `class BLAS` (src/TensorOps/BLAS.hs:108-123) next to
// gemm_kwave.hip, whose shape it shares: one workgroup per output tile, the K loop split over its four WAVES, every wave
// DMA-feeding its own private LDS images and waiting on nothing but its own vmcnt, the four partial tiles meeting in LDS.
//
// Why another tile shape: a launch of the wave-split kernel costs ~4 us + 0.228 us per k-tile a CU works through, and a CU
// works through WHOLE tiles.  768^3 is 144 tiles of 64x64 on 256 CUs (a three-way split over workgroups and its hand-over
// bring it to 13.7 us = 66 TF; stream-K loses: profiles/r06_kw_streamk_sweep.txt) -- and exactly 256 tiles of 48x48: one
// round, no hand-over, every CU busy.  1280^3 is 400 tiles of 64x64 (two rounds for 1.56 rounds of work) and 256 of 80x80.
// The tiles are built from `v_mfma_f32_16x16x4_f32` blocks (TM x TN of them; same 64 flop / clk / SIMD as the 32x32x2 form),
// so any multiple of 16 is a tile edge; launch_gemm_kw16 picks the shape whose rounds x tile cost is lowest and takes the
// problem only where that beats the 64x64 routes by a margin.
//
// Differences from gemm_kwave.hip's body: a k-tile (16 k) is ONE group of 4 TM TN MFMAs (a lane's 16-byte fragment of a
// k-contiguous operand is its k-group's four k: k = 4 kq + ss for MFMA step ss, A and B alike), the next k-tile's fragments
// and the DMA of the tile three ahead ride behind those MFMAs, three images per wave and operand (a DMA has two k-tiles of
// MFMA time to land: a k-tile is 1,152 cycles at 48x48, about one L2-miss round trip).  An m-/n-contiguous operand's image
// is the plain [k][rows]; its fragments are 4-byte reads (a lane's row of every block for its k).
#include <cstdio>
#include <type_traits>

#include "common.hpp"

namespace to {

typedef float f32x4k __attribute__((ext_vector_type(4)));

struct Kw16Args {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  long a_sm, a_sk, b_sk, b_sn, c_sm;
  int tiles_m, tiles_n;
  float alpha;
  const float* bias;
  const float* dact;
  const float* cin;   // beta * cin[m * c_sm + n] joins the sum (the layout of C; may BE C: every element is read, then written, by one thread)
  float beta;
  int act, dact_kind;
  int wide;  // 16-byte stores legal (C aligned, c_sm % 4 == 0, N % 4 == 0)
};

template <int I, int N, class F>
__device__ __forceinline__ void k16_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    k16_static_for<I + 1, N>(f);
  }
}

// AMODE: 0 = A k-contiguous (a_sk == 1), 1 = A m-contiguous (a_sm == 1);  BMODE: 0 = B n-contiguous, 1 = B k-contiguous
// TM x TN blocks of 16x16 per wave = the workgroup's output tile (16 TM x 16 TN); four waves split the K loop
template <int AMODE, int BMODE, int TM, int TN>
__global__ __launch_bounds__(256) void gemm_kw16_kernel(Kw16Args g) {
  constexpr int NW = 4, NI = 3, BM = 16 * TM, BN = 16 * TN, BK = 16, GA = TM, GB = TN;   // GA / GB: 1-KiB DMA pieces per image
  constexpr int IMG_A = BM * BK, IMG_B = BN * BK;      // floats per image
  constexpr int WAVE_FLOATS = NI * (IMG_A + IMG_B);    // a wave's LDS: [NI] A images, [NI] B images
  static_assert(BM * BN <= WAVE_FLOATS, "the partial tile fits the wave's (dead) images");
  static_assert(GA <= 8 && GB <= 8, "piece offsets are written out up to 8 KiB");
  __shared__ __attribute__((aligned(16))) float smem[NW * WAVE_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  // XCD-aware tile order (gemm_kwave.hip): block b runs on XCD b % 8; each XCD a contiguous run of the tile sequence,
  // which walks the tile grid in bands of four tile-rows, column-major inside a band
  const int ntiles = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = ntiles >> 3, r = ntiles & 7;
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

  f32x4k acc[TM * TN];
#pragma unroll
  for (int i = 0; i < TM * TN; ++i) acc[i] = f32x4k{0.f, 0.f, 0.f, 0.f};

  const int KT = g.K / BK;
  const int per = 2 * ((KT + 2 * NW - 1) / (2 * NW));   // (even runs: the K loop below is written out two k-tiles at a time)
  const int t_begin = wave * per < KT ? wave * per : KT;
  const int t_end = t_begin + per < KT ? t_begin + per : KT;
  const int nT = t_end - t_begin;

  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned lds_a = (unsigned)(unsigned long)(lptr_t)(smem + wave * WAVE_FLOATS);  // [NI][IMG_A]
  const unsigned lds_b = lds_a + NI * IMG_A * 4;                                        // [NI][IMG_B]

  // LDS images: a wave instruction fills 1 KiB linearly (lane * 16 B); which element a lane fetches shapes the image.
  // k-contiguous operand: [x][4 slots of 4 k], k-chunk c of row x in slot c ^ ((x >> 1) & 3) (the sixteen lanes a
  // ds_read_b128 serves together then cover the 64 banks once); m-/n-contiguous operand: [k][rows].
  constexpr int RA = AMODE == 1 ? 4 * TM : TM, RB = BMODE == 0 ? 4 * TN : TN;  // LDS reads per k-tile
  static_assert(RA + RB + GA + GB <= 4 * TM * TN, "a slot behind every MFMA");
  unsigned oa[GA], ob[GB];
#pragma unroll
  for (int q = 0; q < GA; ++q) {
    const int f = q * 256 + lane * 4;
    long e;
    if constexpr (AMODE == 1) {
      long m = m0 + f % BM;  // four consecutive rows (M % 4 == 0: a quad is in or out)
      if (m + 4 > g.M) m = g.M - 4;
      e = (long)(f / BM) * g.a_sk + m;
    } else {
      long m = m0 + f / BK;
      if (m >= g.M) m = g.M - 1;
      e = m * g.a_sm + 4 * (((f % BK) / 4) ^ (((f / BK) >> 1) & 3));
    }
    oa[q] = (unsigned)(e * 4 + 3072 - (q % 4) * 1024);
  }
#pragma unroll
  for (int q = 0; q < GB; ++q) {
    const int f = q * 256 + lane * 4;
    long e;
    if constexpr (BMODE == 0) {
      long n = n0 + f % BN;
      if (n + 4 > g.N) n = g.N - 4;
      e = (long)(f / BN) * g.b_sk + n;
    } else {
      long n = n0 + f / BK;
      if (n >= g.N) n = g.N - 1;
      e = n * g.b_sn + 4 * (((f % BK) / 4) ^ (((f / BK) >> 1) & 3));
    }
    ob[q] = (unsigned)(e * 4 + 3072 - (q % 4) * 1024);
  }
  const long step_a = (AMODE == 1 ? (long)BK * g.a_sk : BK) * 4, step_b = (BMODE == 0 ? (long)BK * g.b_sk : BK) * 4;  // bytes
  const char* sa = reinterpret_cast<const char*>(g.A) - 3072 + (long)t_begin * step_a;
  const char* sb = reinterpret_cast<const char*>(g.B) - 3072 + (long)t_begin * step_b;
  // (the `; @dma K` / `; @rd K` / `; @images` / `; @advance` comments are what tools/asm_inflight_check.py reads: which
  //  k-tile's image, relative to the loop's current tile t, an access touches.  They cost no instruction.)
#define K16_DMA(OFF, BASE, IMM, TAG) asm volatile("global_load_lds_dwordx4 %0, %1 offset:" #IMM " ; @dma %2" ::"v"(OFF), "s"(BASE), "n"(TAG) : "memory")
  auto dma = [&](int u, int buf, auto tagc) {  // tagc: the tile this DMA fetches is t + tagc
    constexpr int TAG = decltype(tagc)::value;
    const bool isa = u < GA;
    const int q = isa ? u : u - GA;
    if (q % 4 == 0) {
      const unsigned m0v = (isa ? lds_a + buf * IMG_A * 4 : lds_b + buf * IMG_B * 4) + (q / 4) * 4096;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(m0v) : "memory");
    }
    const unsigned off = isa ? oa[q] : ob[q];
    const char* base = isa ? sa : sb;
    if (q % 4 == 0) K16_DMA(off, base, 0, TAG);
    if (q % 4 == 1) K16_DMA(off, base, 1024, TAG);
    if (q % 4 == 2) K16_DMA(off, base, 2048, TAG);
    if (q % 4 == 3) K16_DMA(off, base, 3072, TAG);
  };
#undef K16_DMA

  // LDS reads as inline asm (the compiler would order every LDS read it can see behind ALL outstanding LDS DMA), landing in
  // temporaries whose first consumer is the wait itself ("+v": the wait hands the value on) -- gemm_kwave.hip has the story.
  // Two sets, one per tile parity: the MFMAs take their operands straight from the set the reads landed in.
  f32x4k ta4[2][AMODE == 0 ? TM : 1], tb4[2][BMODE == 1 ? TN : 1];   // k-contiguous: a block's four k in one 16-byte read
  float ta1[2][AMODE == 1 ? 4 * TM : 1], tb1[2][BMODE == 0 ? 4 * TN : 1];   // m-/n-contiguous: [ss][block], one dword each
  const unsigned a_lane = lds_a + (AMODE == 1 ? ((4 * kq) * BM + l15) * 4 : (l15 * 4 + (kq ^ ((l15 >> 1) & 3))) * 16);
  const unsigned b_lane = lds_b + (BMODE == 0 ? ((4 * kq) * BN + l15) * 4 : (l15 * 4 + (kq ^ ((l15 >> 1) & 3))) * 16);
  auto rd4 = [&](f32x4k& dst, unsigned addr, auto off, auto tagc) {
    asm volatile("ds_read_b128 %0, %1 offset:%2 ; @rd %3" : "=v"(dst) : "v"(addr), "n"(decltype(off)::value), "n"(decltype(tagc)::value));
  };
  auto rd1 = [&](float& dst, unsigned addr, auto off, auto tagc) {
    asm volatile("ds_read_b32 %0, %1 offset:%2 ; @rd %3" : "=v"(dst) : "v"(addr), "n"(decltype(off)::value), "n"(decltype(tagc)::value));
  };
  // read r of a k-tile's fragments (image bases abase / bbase), into set `slot`
  auto frag = [&](int slot, unsigned abase, unsigned bbase, auto ri, auto tagc) {
    constexpr int r = decltype(ri)::value;
    if constexpr (r < RA) {
      if constexpr (AMODE == 0) rd4(ta4[slot][r], abase, std::integral_constant<int, r * 16 * 64>{}, tagc);          // block r: 16 rows of 64 B on
      else rd1(ta1[slot][r], abase, std::integral_constant<int, ((r / TM) * BM + (r % TM) * 16) * 4>{}, tagc);          // [ss = r / TM][block r % TM]
    } else {
      constexpr int rr = r - RA;
      if constexpr (BMODE == 1) rd4(tb4[slot][rr], bbase, std::integral_constant<int, rr * 16 * 64>{}, tagc);
      else rd1(tb1[slot][rr], bbase, std::integral_constant<int, ((rr / TN) * BN + (rr % TN) * 16) * 4>{}, tagc);
    }
  };
  // the reads issued since the last landing are complete: set `slot` is valid from here on (every register of the set is an
  // operand of a wait, so whatever the compiler does with it, it does behind the wait)
  auto land = [&](int slot) {
    if constexpr (AMODE == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta4[slot][i])::"memory");
    } else {
#pragma unroll
      for (int i = 0; i < 4 * TM; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta1[slot][i])::"memory");
    }
    if constexpr (BMODE == 1) {
#pragma unroll
      for (int i = 0; i < TN; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tb4[slot][i])::"memory");
    } else {
#pragma unroll
      for (int i = 0; i < 4 * TN; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tb1[slot][i])::"memory");
    }
  };

  // one k-tile: 4 TM TN MFMAs; behind each one pinned other instruction: the next tile's fragments (into the other set) and
  // the fetch of tile t + NI into the image this tile's fragments came from (they were read during the previous tile).
  // EVERY tile fetches: where tile t + NI does not exist the wave's last tile is fetched again into an image nobody reads --
  // one body, one wait count, no variants for the compiler to reconcile (two variants per parity in one loop made it move
  // the accumulators between register sets behind inline-asm MFMAs: tools/asm_acc_lint.py, asm_inflight_check.py rule 6).
  int ahead = 0;   // the tile `sa` / `sb` point at, relative to t_begin
  auto tile = [&](auto curc, int buf, int bnext) {
    constexpr int cur = decltype(curc)::value, nxt = cur ^ 1;
    // the next tile's image has landed (the wave's own DMA: its vmcnt is all the ordering needed)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NI - 2) * (GA + GB)) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    const unsigned abase = a_lane + bnext * IMG_A * 4, bbase = b_lane + bnext * IMG_B * 4;
    k16_static_for<0, 4 * TM * TN>([&](auto ni) {
      constexpr int n = decltype(ni)::value;
      constexpr int ss = n / (TM * TN), i = (n % (TM * TN)) / TN, jn = n % TN;
      float av, bv;
      if constexpr (AMODE == 0) av = ta4[cur][i][ss]; else av = ta1[cur][ss * TM + i];
      if constexpr (BMODE == 1) bv = tb4[cur][jn][ss]; else bv = tb1[cur][ss * TN + jn];
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i * TN + jn]) : "v"(av), "v"(bv));
      if constexpr (n < RA + RB) {
        frag(nxt, abase, bbase, ni, std::integral_constant<int, 1>{});
      } else if constexpr (n < RA + RB + GA + GB) {
        dma(n - (RA + RB), buf, std::integral_constant<int, NI>{});
        if constexpr (n == RA + RB + GA + GB - 1) {
          const bool more = ahead + 1 < nT;   // (scalar selects: the pointers stop at the wave's last tile)
          sa += more ? step_a : 0;
          sb += more ? step_b : 0;
          ahead += more ? 1 : 0;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    land(nxt);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("; @advance");   // (for the checker: the loop's t becomes t + 1)
  };

  if (nT > 0) {
    asm volatile("; @images %0 private" ::"n"(NI));
    // prologue: NI tiles in flight (a wave with fewer fetches its last tile again: the waits below count instructions)
    k16_static_for<0, NI>([&](auto ic) {
#pragma unroll
      for (int u = 0; u < GA + GB; ++u) dma(u, decltype(ic)::value, ic);
      const bool more = ahead + 1 < nT;
      sa += more ? step_a : 0;
      sb += more ? step_b : 0;
      ahead += more ? 1 : 0;
    });
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NI - 1) * (GA + GB)) : "memory");
    k16_static_for<0, RA + RB>([&](auto ri) { frag(0, a_lane, b_lane, ri, std::integral_constant<int, 0>{}); });
    land(0);
    __builtin_amdgcn_sched_barrier(0);
    // (the fragment sets alternate by tile parity: the loop is written out two tiles at a time)
    int buf = 0, t = 0;
    auto next = [&](int b) { return b + 1 == NI ? 0 : b + 1; };
    // An odd run (only the wave that holds the end of K can have one) ends with a GHOST tile: its fragments are zeroed, its
    // MFMAs add nothing.  A tail of one real tile outside the loop would be a second instance of the body, and the compiler
    // then shuttles every accumulator through VGPRs to reconcile the two (tools/asm_acc_lint.py).
    for (; t < nT; t += 2) {
      tile(std::integral_constant<int, 0>{}, buf, next(buf));
      buf = next(buf);
      if (t + 1 >= nT) {
        if constexpr (AMODE == 0) {
#pragma unroll
          for (int i = 0; i < TM; ++i) ta4[1][i] = f32x4k{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
          for (int i = 0; i < 4 * TM; ++i) ta1[1][i] = 0.f;
        }
        if constexpr (BMODE == 1) {
#pragma unroll
          for (int i = 0; i < TN; ++i) tb4[1][i] = f32x4k{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
          for (int i = 0; i < 4 * TN; ++i) tb1[1][i] = 0.f;
        }
      }
      tile(std::integral_constant<int, 1>{}, buf, next(buf));
      buf = next(buf);
    }
  }
  // The last MFMAs retire before anything but another MFMA touches the AccVGPRs: the accumulators are read-write operands of
  // the statement that holds the wait states (tools/asm_inflight_check.py rule 6; gemm_kwave.hip).
#define K16_DRAIN "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15"
  if constexpr (TM * TN == 8) {
    asm volatile(K16_DRAIN : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7])::"memory");
  } else if constexpr (TM * TN == 9) {
    asm volatile(K16_DRAIN : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]), "+a"(acc[8])::"memory");
  } else if constexpr (TM * TN == 12) {
    asm volatile(K16_DRAIN : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]), "+a"(acc[8]),
                 "+a"(acc[9]), "+a"(acc[10]), "+a"(acc[11])::"memory");
  } else if constexpr (TM * TN == 20) {
    asm volatile(K16_DRAIN : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]), "+a"(acc[8]),
                 "+a"(acc[9]), "+a"(acc[10]), "+a"(acc[11]), "+a"(acc[12]), "+a"(acc[13]), "+a"(acc[14]), "+a"(acc[15]), "+a"(acc[16]), "+a"(acc[17]),
                 "+a"(acc[18]), "+a"(acc[19])::"memory");
  } else {
    static_assert(TM * TN == 25, "the operand lists are written out");
    asm volatile(K16_DRAIN : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]), "+a"(acc[8]),
                 "+a"(acc[9]), "+a"(acc[10]), "+a"(acc[11]), "+a"(acc[12]), "+a"(acc[13]), "+a"(acc[14]), "+a"(acc[15]), "+a"(acc[16]), "+a"(acc[17]),
                 "+a"(acc[18]), "+a"(acc[19]), "+a"(acc[20]), "+a"(acc[21]), "+a"(acc[22]), "+a"(acc[23]), "+a"(acc[24])::"memory");
  }
#undef K16_DRAIN

  // the ragged end of K (fewer than 16): the last wave, operands straight from global memory, four k per MFMA
  if (g.K % BK != 0 && wave == NW - 1) {
    long ra[TM], cb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const long m = m0 + i * 16 + l15;
      ra[i] = (m < g.M ? m : g.M - 1) * g.a_sm;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const long n = n0 + j * 16 + l15;
      cb[j] = (n < g.N ? n : g.N - 1) * g.b_sn;
    }
    for (int kk = KT * BK; kk < g.K; kk += 4) {
      const int k = kk + kq;
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
          acc[i * TN + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i * TN + j], 0, 0, 0);
    }
  }

  // partial tiles -> LDS (each wave into its own, now dead, images), summed in wave order.
  // D register r of lane (l15, kq) -> row 4 kq + r, column l15 of its 16x16 block
  float* P = smem + wave * WAVE_FLOATS;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[(i * 16 + 4 * kq + r) * BN + j * 16 + l15] = acc[i * TN + j][r];
  __syncthreads();
  auto finish = [&](auto plainc) {
    constexpr bool PLAIN = decltype(plainc)::value;
    for (int q = tid; q < BM * BN / 4; q += NW * 64) {
      const int row = q / (BN / 4), c4 = (q % (BN / 4)) * 4;
      f32x4k s = *reinterpret_cast<const f32x4k*>(smem + row * BN + c4);
#pragma unroll
      for (int w = 1; w < NW; ++w) s += *reinterpret_cast<const f32x4k*>(smem + w * WAVE_FLOATS + row * BN + c4);
      const long gr = m0 + row, gc = n0 + c4;
      if (gr >= g.M || gc >= g.N) continue;
      float* dst = g.C + gr * g.c_sm + gc;
      if constexpr (PLAIN) {  // (wide: N % 4 == 0, a quad is in or out)
        *reinterpret_cast<f32x4k*>(dst) = g.alpha * s;
      } else {
        float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (gc + e >= g.N) break;
          float x = g.alpha * v[e];
          if (g.cin) x += g.beta * g.cin[gr * g.c_sm + gc + e];
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
          f32x4k o = {v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4k*>(dst) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gc + e < g.N) dst[e] = v[e];
        }
      }
    }
  };
  if (g.wide && !g.bias && g.act == 0 && !g.dact && !g.cin) finish(std::true_type{});
  else finish(std::false_type{});
}

static int kw16_mode() {
  static const int m = [] { const char* e = ab_getenv("TOPS_GEMM_KW16"); return e ? atoi(e) : 1; }();
  return m;
}

// Can the kernel run the problem at all?  (gemm_kwave.hip's conditions)
static bool kw16_can(const GemmProblem& p) {
  if (p.dtype != TO_F32 || p.batch != 1 || p.reduce_batch || p.rowsum || p.loss_rows || (p.beta != 0.0 && !p.Cin)) return false;
  // (8 .. 95 rows or columns beside a large extent: the 32-row / 32-column tiles pad half as much as a 64x64 tile -- 8 x 4096 x 60000)
  if (p.M < 8 || p.N < 8 || (p.M < 96 && p.N < 96) || p.K < 64) return false;
  if (p.M > 2147483647LL || p.N > 2147483647LL || p.K > 2147483647LL) return false;
  const bool a_k = p.a_sk == 1, a_m = !a_k && p.a_sm == 1;
  const bool b_n = p.b_sn == 1, b_k = !b_n && p.b_sk == 1;
  if (!(a_k || a_m) || !(b_n || b_k)) return false;
  auto al4 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 3u) == 0; };
  if (!al4(p.A) || !al4(p.B)) return false;
  const int64_t ext_a = a_k ? p.M * p.a_sm : 16 * p.a_sk + p.M, ext_b = b_k ? p.N * p.b_sn : 16 * p.b_sk + p.N;
  if (ext_a * 4 + 8192 >= (1LL << 32) || ext_b * 4 + 8192 >= (1LL << 32) || p.a_sm < 0 || p.a_sk < 0 || p.b_sk < 0 || p.b_sn < 0) return false;
  if (a_m && p.M % 4 != 0) return false;     // an m-contiguous quad must be in or out of the matrix as a whole
  if (b_n && p.N % 4 != 0) return false;
  return true;
}

// The menu.  A launch costs 4.1 us + (the k-tiles the busiest CU works through) x (a k-tile's matrix time on the tile's
// shape / 0.90); workgroups per CU by LDS (three images a wave: 72 KiB at 48x48, 84 at 48x64, 108 at 64x80, 120 at 80x80).
struct Kw16Shape { int tm, tn, per_cu; };
static const Kw16Shape kw16_menu[] = {{3, 3, 2}, {3, 4, 1}, {4, 3, 1}, {5, 5, 1}, {2, 4, 2}, {4, 2, 2},   // (32x64 / 64x32: 1024 x K x 512 = 256 tiles)
                                      {4, 5, 1}, {5, 4, 1}};                                        // (64x80 / 80x64: 1088^3 = 238 tiles)
static double kw16_cost(const GemmProblem& p, const Kw16Shape& s) {
  const long T = ((p.M + 16 * s.tm - 1) / (16 * s.tm)) * ((p.N + 16 * s.tn - 1) / (16 * s.tn)), KT = p.K / 16;
  const long slots = 256L * s.per_cu;
  const long on_busiest = (T + 255) / 256;                       // tiles the busiest CU gets (they share its matrix pipes)
  if (T > 4 * slots) return 1e30;                                // (many rounds: the big tiles' territory)
  const double kt_us = 4.0 * s.tm * s.tn * 32.0 / 4.0 / 2300.0 / 0.90;   // us per k-tile per workgroup
  // (fitted, profiles/r06_kw16_sweep.txt, model / measured us: 768^3 10.8 / 10.5, 1280^3 on 80x80 35.0 / 35.5, on 48x48 39.0 /
  //  38.6, 1152^3 35.6 / 34.7, 1088^3 on 64x80 25.0 / 26.6, 896^3 19.7 / 19.0, 1024 x 512 x 1024 13.0 / 13.2, 1856^3 106.8 / 106.4; a tile beyond what a CU
  //  holds at once waits for a slot: 1.5 us each)
  const long queued = on_busiest > s.per_cu ? on_busiest - s.per_cu : 0;
  return 4.1 + (double)on_busiest * KT * kt_us + 1.5 * (double)queued;
}
// ... against the 64x64 routes of gemm_kwave.hip (kw_ksplit / kw_streamk's own fitted models, restated)
static double kw16_cost_64(const GemmProblem& p) {
  const long T = ((p.M + 63) / 64) * ((p.N + 63) / 64), KT = p.K / 16;
  // (one round of 96x96 tiles, kw_tile: 1536^3 -- 0.54 us per k-tile)
  const long t3 = ((p.M + 95) / 96) * ((p.N + 95) / 96);
  if (t3 >= 244 && t3 <= 256 && 100 * p.M * p.N >= 97 * t3 * 96 * 96) return 4.1 + 0.54 * KT;
  if (T > 256) {
    const double whole = 4.0 + 0.228 * (double)((T + 255) / 256) * KT, stream = 11.0 + (KT >= 512 ? 0.29 : 0.245) * (double)T / 256.0 * KT;
    return T <= 1024 && stream < 0.97 * whole ? stream : whole;
  }
  double best = 4.1 + 0.222 * KT;
  const long per_xcd = (T + 7) / 8;
  for (int S : {2, 3, 4, 6, 8}) {
    if (KT < 16L * S) continue;
    const long R = (S * per_xcd + 31) / 32;
    if (R > 2) continue;
    const double c = 4.1 + 0.222 * ((double)R * KT / S + 2.7 + 2.25 * S + (R > 1 ? 5.4 : 0.0));
    if (c < best) best = c;
  }
  return best;
}
static int kw16_pick(const GemmProblem& p) {
  static const int forced = [] { const char* e = ab_getenv("TOPS_GEMM_KW16_TILE"); return e ? atoi(e) : -1; }();   // menu index
  if (forced >= 0 && forced < (int)(sizeof(kw16_menu) / sizeof(kw16_menu[0]))) return forced;
  int best = -1;
  double bc = 0.96 * kw16_cost_64(p);   // (a margin: both sides are fitted models)
  for (int i = 0; i < (int)(sizeof(kw16_menu) / sizeof(kw16_menu[0])); ++i) {
    const double c = kw16_cost(p, kw16_menu[i]);
    if (c < bc) { bc = c; best = i; }
  }
  return best;
}

bool gemm_kw16_applicable(const GemmProblem& p) {
  const int mode = kw16_mode();
  if (mode == 0 || !kw16_can(p)) return false;
  if (mode >= 2) return true;
  const long T64 = ((p.M + 63) / 64) * ((p.N + 63) / 64);
  if (T64 < 80 || T64 > 1024 || p.K < 256) return false;   // (512^3 = 64 tiles: the small-GEMM / t32 routes, 43 vs 34 TF here)
  return kw16_pick(p) >= 0;
}

template <int TM, int TN>
static void kw16_launch_modes(int mode, dim3 grid, hipStream_t s, const Kw16Args& g) {
  switch (mode) {
    case 0: launch_k((gemm_kw16_kernel<0, 0, TM, TN>), grid, dim3(256), 0, s, g); break;
    case 1: launch_k((gemm_kw16_kernel<0, 1, TM, TN>), grid, dim3(256), 0, s, g); break;
    case 2: launch_k((gemm_kw16_kernel<1, 0, TM, TN>), grid, dim3(256), 0, s, g); break;
    default: launch_k((gemm_kw16_kernel<1, 1, TM, TN>), grid, dim3(256), 0, s, g); break;
  }
}

void launch_gemm_kw16(const GemmProblem& p, hipStream_t s) {
  int pick = kw16_pick(p);
  if (pick < 0) pick = 0;   // (forced route, TOPS_GEMM_KW16=2)
  const Kw16Shape sh = kw16_menu[pick];
  Kw16Args g{};
  g.A = (const float*)p.A; g.B = (const float*)p.B; g.C = (float*)p.C;
  g.M = (int)p.M; g.N = (int)p.N; g.K = (int)p.K;
  g.a_sm = p.a_sm; g.a_sk = p.a_sk; g.b_sk = p.b_sk; g.b_sn = p.b_sn; g.c_sm = p.c_sm;
  g.tiles_m = (int)((p.M + 16 * sh.tm - 1) / (16 * sh.tm));
  g.tiles_n = (int)((p.N + 16 * sh.tn - 1) / (16 * sh.tn));
  g.alpha = (float)p.alpha;
  g.bias = (const float*)p.bias; g.dact = (const float*)p.dact; g.act = p.act; g.dact_kind = p.dact_kind;
  g.cin = p.beta != 0.0 ? (const float*)p.Cin : nullptr; g.beta = (float)p.beta;
  g.wide = (reinterpret_cast<uintptr_t>(p.C) & 15u) == 0 && p.c_sm % 4 == 0 && p.N % 4 == 0;
  const int mode = (p.a_sk == 1 ? 0 : 2) + (p.b_sn == 1 ? 0 : 1);
  const dim3 grid(g.tiles_m * g.tiles_n);
  if (sh.tm == 3 && sh.tn == 3) kw16_launch_modes<3, 3>(mode, grid, s, g);
  else if (sh.tm == 3 && sh.tn == 4) kw16_launch_modes<3, 4>(mode, grid, s, g);
  else if (sh.tm == 4 && sh.tn == 3) kw16_launch_modes<4, 3>(mode, grid, s, g);
  else if (sh.tm == 2 && sh.tn == 4) kw16_launch_modes<2, 4>(mode, grid, s, g);
  else if (sh.tm == 4 && sh.tn == 2) kw16_launch_modes<4, 2>(mode, grid, s, g);
  else if (sh.tm == 4 && sh.tn == 5) kw16_launch_modes<4, 5>(mode, grid, s, g);
  else if (sh.tm == 5 && sh.tn == 4) kw16_launch_modes<5, 4>(mode, grid, s, g);
  else kw16_launch_modes<5, 5>(mode, grid, s, g);
  TO_HIP(hipGetLastError());
  count_launch();
}

}  // namespace to
