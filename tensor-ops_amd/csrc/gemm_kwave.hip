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
  const float* cin;   // beta * cin[m * c_sm + n] joins the sum (the layout of C; may BE C: every element is read, then written, by one thread)
  float beta;
  int act, dact_kind;
  int wide;  // 16-byte stores legal (C aligned, c_sm % 4 == 0, N % 4 == 0)
  // KS > 1 (gemm_kw_kernel<..., KS>): KS workgroups per output tile, each a KS-th of the K loop, all on ONE XCD
  float* pair_ws;      // [tile][KS][BM*BN]: a workgroup's summed partial tile
  unsigned* pair_ctr;  // [tile * 16]: arrivals (0 between launches: the last arriver resets it)
  // KS == 0 (stream-K): the grid's workgroups share the stream "tile 0's k-tiles, tile 1's k-tiles, ..." evenly: logical
  // workgroup w owns the units [w sk_base + min(w, sk_rem), ...) -- sk_base + 1 units for the first sk_rem, sk_base after
  int sk_base, sk_rem;
  unsigned long long* dbg_out;
  int dbg;   // TOPS_GEMM_KW_DBG=4: wave 0 of block 0 stamps its K loop (shader cycles, 100 MHz ticks): cycles per k-tile and the clock
};

// AMODE: 0 = A k-contiguous (a_sk == 1), 1 = A m-contiguous (a_sm == 1)
// BMODE: 0 = B n-contiguous (b_sn == 1), 1 = B k-contiguous (b_sk == 1)
// TM x TN 32x32 MFMA tiles per wave (the workgroup's output tile is 32 TM x 32 TN); NW waves split the K loop;
// NI LDS images per wave and operand
template <int N> struct KwVec { typedef float type __attribute__((ext_vector_type(N))); };

template <int I, int N, class F>
__device__ __forceinline__ void kw_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    kw_static_for<I + 1, N>(f);
  }
}

// SPLIT: true = the NW waves of a workgroup share ONE output tile and split its K loop (few tiles: every CU gets work,
// the partial tiles meet in LDS); false = every wave has a tile of its own and the whole K loop (many tiles, short K:
// no reduction, nothing at all shared between the waves -- a workgroup is just four tiles that are neighbours in L2)
// KS > 1 (with SPLIT, round 4): fewer tiles than the chip has CUs (640^3: 100 tiles, 768^3: 144 on 256 CUs) -- KS
// workgroups per tile, each with a KS-th of the k-tiles, 4 KS waves on the tile's K loop instead of four.  Their partial
// tiles meet without a second launch and without a grid-wide anything: all workgroups of a tile are placed on one XCD
// (block b runs on XCD b % 8: probed at start-up, gemm_kw_pair_init), each stores its summed partial (complete when that
// XCD's L2 has it), bumps the tile's counter, and the one that arrives LAST adds the partials IN K ORDER -- its own from
// LDS, the others' by L1-bypassing loads from the same L2 -- and writes C with the epilogue: the sum does not depend on
// who arrives last.  Nobody waits for anybody.
// KS == 0 (round 6): STREAM-K.  A tile count that is no multiple of the CUs (768^3: 144 tiles on 256 CUs; 1280^3: 400 on
// 512 slots) leaves a KS-way split either idle CUs or CUs with twice the work.  Here the grid is a fixed number of
// workgroups (one or two per CU) and logical workgroup w owns an equal, contiguous share of the stream "tile 0's k-tiles,
// tile 1's k-tiles, ...": it runs the K loop once per tile its share touches (one or two of them when a share is shorter
// than a tile's K loop).  A run that covers a whole tile writes C; otherwise the hand-over of KS > 1: the partial goes to
// the workgroup's own slot (2 w: its run starts inside its share's first tile, 2 w + 1: the run that ends its share),
// write-through, the tile's counter, and the last of the tile's contributors to arrive adds the parts IN K ORDER (=
// workgroup order) and writes C with the epilogue.  Nobody waits for anybody; which workgroup arrives last changes nothing.
template <int AMODE, int BMODE, int TM, int TN, int NW, int NI, bool SPLIT = true, int KS = 1>
__global__ __launch_bounds__(NW * 64) void gemm_kw_kernel(KwArgs g) {
  constexpr int BM = 32 * TM, BN = 32 * TN, BK = 16, GA = 2 * TM, GB = 2 * TN;  // GA/GB: 1-KiB DMA pieces per image
  constexpr bool PAIR = KS > 1, SK = KS == 0;
  constexpr int IMG_A = BM * BK, IMG_B = BN * BK;      // floats per image
  constexpr int WAVE_FLOATS = NI * (IMG_A + IMG_B);    // a wave's LDS: [NI] A images, [NI] B images
  constexpr int PASSES = (BM * BN + WAVE_FLOATS - 1) / WAVE_FLOATS;  // the partial tile leaves in this many row bands
  constexpr int RP = BM / PASSES;                                    // rows per band
  static_assert(BM % PASSES == 0 && RP * BN <= WAVE_FLOATS, "a band of the partial tile fits the wave's images");
  __shared__ __attribute__((aligned(16))) float smem[NW * WAVE_FLOATS];

  // XCD-aware tile order (as gemm_mfma_kernel): block b runs on XCD b % 8; each XCD gets a contiguous run of the
  // tile sequence, which walks the tile grid in bands of R tile-rows, column-major inside a band
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform: loop bounds and LDS bases stay scalar)
  const int l31 = lane & 31, half = lane >> 5;
  const int ntiles = g.tiles_m * g.tiles_n;
  const int nblk = SPLIT ? ntiles : (ntiles + NW - 1) / NW;   // == gridDim.x (KS > 1: the grid is 8 KS ceil(ntiles / 8))
  const int KT = g.K / BK;   // whole k-tiles
  // stream-K: this workgroup's share [sk_u, sk_end) of the stream.  XCD x (block b runs on XCD b % 8) works through a
  // contiguous eighth of it, so the parts of a tile mostly meet on one XCD and its L2 sees a run of neighbouring tiles.
  int sk_w = 0, sk_u = 0, sk_end = 0;
  auto sk_start = [&](int w) { return w * g.sk_base + (w < g.sk_rem ? w : g.sk_rem); };
  auto sk_owner = [&](int u) {   // the workgroup whose share holds unit u
    const int big = g.sk_rem * (g.sk_base + 1);
    return u < big ? u / (g.sk_base + 1) : g.sk_rem + (u - big) / g.sk_base;
  };
  if constexpr (SK) {
    static_assert(SPLIT, "the waves of a workgroup share a run");
    sk_w = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
    sk_u = sk_start(sk_w);
    sk_end = sk_start(sk_w + 1);
    if (sk_u >= sk_end) return;
  }
#ifdef TOPS_AB_KNOBS   // development build: phase stamps of one workgroup (TOPS_GEMM_KW_DBG=8 [+ 16 x block]), shader cycles
  const bool stamp_on = (g.dbg & 8) && (int)blockIdx.x == (g.dbg >> 4) && tid == 0;
  int stamp_n = 0;
#define KW_STAMP() do { if (stamp_on && stamp_n < 24) g.dbg_out[8 + stamp_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define KW_STAMP() do { } while (0)
#endif
  KW_STAMP();
 do {   // (one pass unless stream-K: one run per tile of the share)
  int bid = blockIdx.x;
  int ksp = 0;   // KS > 1: which part of the K loop
  int sk_kb = 0, sk_ke = 0, sk_np = 1, sk_mine = 0, sk_wlo = 0;   // stream-K: this run's k-tiles, the tile's parts, which one this is
  if constexpr (SK) {
    bid = sk_u / KT;
    sk_kb = sk_u - bid * KT;
    sk_ke = (KT - sk_kb < sk_end - sk_u) ? KT : sk_kb + (sk_end - sk_u);
    sk_wlo = sk_owner(bid * KT);
    sk_np = sk_owner(bid * KT + KT - 1) - sk_wlo + 1;
    sk_mine = sk_w - sk_wlo;
    sk_u += sk_ke - sk_kb;
  } else if constexpr (PAIR) {
    static_assert(SPLIT, "the workgroups of a group share one tile");
    // XCD x (= bid & 7) owns tiles [x * per, (x + 1) * per) of the sequence; its workgroups KS j .. KS j + KS - 1 share tile j
    const int per = (int)gridDim.x / (8 * KS), local = bid >> 3;
    ksp = local % KS;
    bid = (bid & 7) * per + local / KS;
    if (bid >= ntiles) return;
  } else {
    const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  if constexpr (!SPLIT) {
    bid = bid * NW + wave;      // (the waves of a workgroup: NW consecutive tiles of the sequence = one tile column of a band)
    if (bid >= ntiles) return;  // (no barrier anywhere on this path)
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
  unsigned long long dbg_c0 = 0, dbg_r0 = 0;
  if (g.dbg & 4) {
    dbg_c0 = __builtin_readcyclecounter();
    dbg_r0 = wall_clock64();
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // this wave's run of whole k-tiles
  // KS > 1 / stream-K: this workgroup's part [kt0, kt1) of the k-tiles, split over its waves like a whole K loop
  const int kt0 = SK ? sk_kb : PAIR ? (int)((long)KT * ksp / KS) : 0, kt1 = SK ? sk_ke : PAIR ? (int)((long)KT * (ksp + 1) / KS) : KT;
  const int per = SPLIT ? (kt1 - kt0 + NW - 1) / NW : KT;
  const int t_begin = SPLIT ? (kt0 + wave * per < kt1 ? kt0 + wave * per : kt1) : 0;
  const int t_end = SPLIT ? (t_begin + per < kt1 ? t_begin + per : kt1) : KT;
  const int nT = t_end - t_begin;

  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned lds_a = (unsigned)(unsigned long)(lptr_t)(smem + wave * WAVE_FLOATS);  // [NI][IMG_A]
  const unsigned lds_b = lds_a + NI * IMG_A * 4;                                        // [NI][IMG_B]

  // LDS images (gemm_f32_mfma.hip, PF == 5): a wave instruction fills 1 KiB linearly (lane * 16 B), which element
  // a lane fetches shapes the image.  k-contiguous operand: [x][4 slots of 4 k], k-chunk c of row x in slot
  // c ^ ((x >> 1) & 3); m-/n-contiguous operand: [k][32 TM].
  constexpr int RA = AMODE == 1 ? 4 : TM, RB = BMODE == 0 ? 4 : TN;  // LDS reads per half k-tile
  static_assert(RA + RB + GA + GB <= 4 * TM * TN, "a slot behind every MFMA of the second half");
  // The DMA: global_load_lds_dwordx4 with a SCALAR base and a per-lane 32-bit byte offset.  The lane offsets never
  // change; advancing a tile is two scalar adds per operand.  The instruction offset moves BOTH addresses, and the
  // pieces of an image that share one M0 value are 1 KiB apart in LDS: piece q's lane offset is biased by
  // -(q % 4) KiB, and the scalar base by -3 KiB so that the biased offsets stay non-negative.
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
  // (M0 is written inside the asm: nothing else in this kernel uses it)
  // (the `; @dma K` / `; @rd K` / `; @images` / `; @advance` comments in the asm strings are what tools/asm_inflight_check.py
  //  reads: which k-tile's image, relative to the loop's current tile t, an access touches.  They cost no instruction.)
#define KW_DMA(OFF, BASE, IMM, TAG) asm volatile("global_load_lds_dwordx4 %0, %1 offset:" #IMM " ; @dma %2" ::"v"(OFF), "s"(BASE), "n"(TAG) : "memory")
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
    if (q % 4 == 0) KW_DMA(off, base, 0, TAG);
    if (q % 4 == 1) KW_DMA(off, base, 1024, TAG);
    if (q % 4 == 2) KW_DMA(off, base, 2048, TAG);
    if (q % 4 == 3) KW_DMA(off, base, 3072, TAG);
  };
#undef KW_DMA

  // LDS reads as inline asm (the compiler would order every LDS read it can see behind ALL outstanding LDS DMA).
  // The compiler does not know these reads are asynchronous: it may copy a result register right behind the read,
  // before the data is there (it did, where the two tile loops rotate the fragment registers).  So a read lands in
  // a temporary whose first consumer is the wait itself ("+v": the wait hands the value on) -- whatever the compiler
  // does with the value, it does behind the wait.  (tools/asm_inflight_check.py scans the assembly for violations.)
  // k-contiguous operand: one b128 per 32-row tile (its four k-steps); m-/n-contiguous operand: one read per k-step
  // of the lane's TM / TN owned rows / columns (b64 / b96 / b128).  Two sets, one per half-tile parity: the MFMAs take
  // their operands straight from the set the reads landed in (no unpacking moves).
  constexpr int NA = AMODE == 0 ? TM : 4, NB = BMODE == 1 ? TN : 4;
  typedef typename KwVec<AMODE == 0 ? 4 : TM>::type va_t;
  typedef typename KwVec<BMODE == 1 ? 4 : TN>::type vb_t;
  va_t ta[2][NA];
  vb_t tb[2][NB];
  // lane (x, half) of half-tile h uses k = 4 (2 h + half) + ss for MFMA step ss (A and B agree on it).
  // An m-contiguous A / n-contiguous B is read row-/column-OWNING: lane l31 holds rows TM*l31 .. TM*l31+TM-1.
  // A read's address = a per-lane base for (operand, h), + the image's offset (one add per half), + a constant per read
  // (the instruction's offset field).
  unsigned a_lane[2], b_lane[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    a_lane[h] = lds_a + (AMODE == 1 ? ((4 * (2 * h + half)) * BM + TM * l31) * 4 : (l31 * 4 + ((2 * h + half) ^ ((l31 >> 1) & 3))) * 16);
    b_lane[h] = lds_b + (BMODE == 0 ? ((4 * (2 * h + half)) * BN + TN * l31) * 4 : (l31 * 4 + ((2 * h + half) ^ ((l31 >> 1) & 3))) * 16);
  }
  constexpr int STEP_A = AMODE == 1 ? BM * 4 : 32 * 64, STEP_B = BMODE == 0 ? BN * 4 : 32 * 64;  // bytes from read r to r + 1
  auto rd = [&](auto& dst, unsigned addr, auto off, auto tagc) {   // tagc: reads the image of tile t + tagc
    constexpr int n = (int)(sizeof(dst) / 4), o = decltype(off)::value, TAG = decltype(tagc)::value;
    static_assert(n == 2 || n == 4, "b64 / b128");
    if constexpr (n == 2) asm volatile("ds_read_b64 %0, %1 offset:%2 ; @rd %3" : "=v"(dst) : "v"(addr), "n"(o), "n"(TAG));
    else asm volatile("ds_read_b128 %0, %1 offset:%2 ; @rd %3" : "=v"(dst) : "v"(addr), "n"(o), "n"(TAG));
  };
  auto rd3 = [&](auto& dst, unsigned addr, auto off, auto tagc) {
    asm volatile("ds_read_b96 %0, %1 offset:%2 ; @rd %3" : "=v"(dst) : "v"(addr), "n"(decltype(off)::value), "n"(decltype(tagc)::value));
  };
  // read r of the half whose lane bases (+ image offset) are abase / bbase, into set `slot`
  auto frag = [&](int slot, unsigned abase, unsigned bbase, auto ri, auto tagc) {
    constexpr int r = decltype(ri)::value;
    if constexpr (r < RA) {
      if constexpr (AMODE == 1 && TM == 3) rd3(ta[slot][r], abase, std::integral_constant<int, r * STEP_A>{}, tagc);
      else rd(ta[slot][r], abase, std::integral_constant<int, r * STEP_A>{}, tagc);
    } else {
      constexpr int rr = r - RA;
      if constexpr (BMODE == 0 && TN == 3) rd3(tb[slot][rr], bbase, std::integral_constant<int, rr * STEP_B>{}, tagc);
      else rd(tb[slot][rr], bbase, std::integral_constant<int, rr * STEP_B>{}, tagc);
    }
  };
  // the reads issued since the last landing are complete: set `slot` is valid from here on
  auto land = [&](int slot) {
    if constexpr (NA == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta[slot][0]), "+v"(ta[slot][1])::"memory");
    else if constexpr (NA == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta[slot][0]), "+v"(ta[slot][1]), "+v"(ta[slot][2])::"memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta[slot][0]), "+v"(ta[slot][1]), "+v"(ta[slot][2]), "+v"(ta[slot][3])::"memory");
    if constexpr (NB == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tb[slot][0]), "+v"(tb[slot][1])::"memory");
    else if constexpr (NB == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tb[slot][0]), "+v"(tb[slot][1]), "+v"(tb[slot][2])::"memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tb[slot][0]), "+v"(tb[slot][1]), "+v"(tb[slot][2]), "+v"(tb[slot][3])::"memory");
  };

  // one k-tile: two halves of 4 TM TN MFMAs; behind each MFMA one pinned other instruction: the next half's fragments
  // and (second half, DMA) the fetch of tile t + NI into the image this tile just left
  auto tile = [&](auto dma_on, int buf, int bnext) {
    constexpr bool DMA = decltype(dma_on)::value;
    kw_static_for<0, 2>([&](auto hi) {
      constexpr int h = decltype(hi)::value, cur = h, nxt = h ^ 1;
      if constexpr (h == 1) {  // the next tile's image has landed (the wave's own DMA: its vmcnt is all the ordering needed)
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NI - 2) * (GA + GB)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      // (the reads of this half fetch the NEXT half: image buf, second half -- or the next tile's image, first half)
      const unsigned abase = a_lane[h ^ 1] + (h == 0 ? buf : bnext) * IMG_A * 4, bbase = b_lane[h ^ 1] + (h == 0 ? buf : bnext) * IMG_B * 4;
      kw_static_for<0, 4 * TM * TN>([&](auto ni) {
        constexpr int n = decltype(ni)::value;
        constexpr int ss = n / (TM * TN), i = (n % (TM * TN)) / TN, jn = n % TN;
        float av, bv;   // (if constexpr, not ?: -- the index of the branch not taken may lie outside the other shape)
        if constexpr (AMODE == 1) av = ta[cur][ss][i]; else av = ta[cur][i][ss];
        if constexpr (BMODE == 0) bv = tb[cur][ss][jn]; else bv = tb[cur][jn][ss];
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i][jn]) : "v"(av), "v"(bv));
        if constexpr (n < RA + RB) {
          frag(nxt, abase, bbase, ni, std::integral_constant<int, h>{});   // (h == 0: this tile's image, h == 1: the next tile's)
        } else if constexpr (DMA && n < RA + RB + GA + GB) {
          if constexpr (h == 1) {
            dma(n - (RA + RB), buf, std::integral_constant<int, NI>{});
            if constexpr (n == RA + RB + GA + GB - 1) {
              sa += step_a;
              sb += step_b;
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      land(nxt);  // (the next half's fragments, issued under these MFMAs)
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("; @advance");   // (for the checker: the loop's t becomes t + 1)
  };

  KW_STAMP();   // (set-up done)
  if (nT > 0) {
    // prologue: up to NI tiles in flight
    asm volatile("; @images %0 private" ::"n"(NI));
    // (two written-out paths, each with the wait that matches what it issued: the hazard checker is not path-sensitive)
    if (nT >= NI) {
      kw_static_for<0, NI>([&](auto ic) {
#pragma unroll
        for (int u = 0; u < GA + GB; ++u) dma(u, decltype(ic)::value, ic);
        sa += step_a;
        sb += step_b;
      });
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NI - 1) * (GA + GB)) : "memory");
    } else {
      kw_static_for<0, NI - 1>([&](auto ic) {
        if (decltype(ic)::value < nT) {
#pragma unroll
          for (int u = 0; u < GA + GB; ++u) dma(u, decltype(ic)::value, ic);
          sa += step_a;
          sb += step_b;
        }
      });
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    KW_STAMP();   // (first k-tile landed)
    kw_static_for<0, RA + RB>([&](auto ri) { frag(0, a_lane[0], b_lane[0], ri, std::integral_constant<int, 0>{}); });
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
  // of the statement that holds the wait states, so whatever the compiler does with them -- a spill, a shuffle, the K tail's
  // own MFMAs, the way out -- it does behind it (the compiler pads the hazards of the MFMAs it issues itself, not those of
  // an asm string; tools/asm_inflight_check.py rule 6).
#define KW_DRAIN "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15"
  if constexpr (TM == 2 && TN == 2) {
    asm volatile(KW_DRAIN : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[1][0]), "+a"(acc[1][1])::"memory");
  } else {
    static_assert(TM == 3 && TN == 3, "the operand lists are written out");
    asm volatile(KW_DRAIN : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]),
                 "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2])::"memory");
  }
#undef KW_DRAIN
  KW_STAMP();   // (K loop done)
  if ((g.dbg & 4) && blockIdx.x == 0 && threadIdx.x == 0) {   // shader cycles and 100 MHz ticks of the K loop -> the clock
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    g.dbg_out[0] = c1 - dbg_c0;
    g.dbg_out[1] = r1 - dbg_r0;
    g.dbg_out[2] = (unsigned long long)nT;
  }

  // the ragged end of K (fewer than 16): the last wave, operands straight from global memory, two k per MFMA
  if (g.K % BK != 0 && (!SPLIT || wave == NW - 1) && (!PAIR || ksp == KS - 1) && (!SK || sk_ke == KT)) {
    // (compiler-scheduled MFMAs here: it knows their hazards; those of the inline-asm stream were settled above)
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

  // partial tiles -> LDS (each wave into its own, now dead, images; in PASSES bands of RP rows when a whole tile does
  // not fit), summed in wave order
  // D reg r lane l -> row (r&3) + 8*(r>>2) + 4*half, col l31 of the MFMA tile
  typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
    if (SPLIT && pass > 0) __syncthreads();  // the previous band has been read
    float* P = smem + wave * WAVE_FLOATS;
    if constexpr (BMODE == 0 && TN == 2) {
      // (a lane owns two neighbouring columns: one 8-byte write per row instead of two 4-byte ones)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int tr = (r & 3) + 8 * (r >> 2) + 4 * half;
          const int row = (AMODE == 1 ? TM * tr + i : i * 32 + tr) - pass * RP;
          if (PASSES == 1 || (row >= 0 && row < RP)) {
            float2 pr;
            pr.x = acc[i][0][r];
            pr.y = acc[i][1][r];
            *reinterpret_cast<float2*>(P + row * BN + 2 * l31) = pr;
          }
        }
    } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = BMODE == 0 ? TN * l31 + j : j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int tr = (r & 3) + 8 * (r >> 2) + 4 * half;
          const int row = (AMODE == 1 ? TM * tr + i : i * 32 + tr) - pass * RP;
          if (PASSES == 1 || (row >= 0 && row < RP)) P[row * BN + col] = acc[i][j][r];
        }
      }
    }
    if constexpr (SPLIT) __syncthreads();
    KW_STAMP();   // (the waves' partial tiles are in LDS)
    // quad `s` of the finished tile (row, c4 of the band) -> C, with the epilogue
    auto emit = [&](f32x4 s, const int row, const int c4, auto plainc) {
      constexpr bool PLAIN = decltype(plainc)::value;
      const long gr = m0 + pass * RP + row, gc = n0 + c4;
      if (gr >= g.M || gc >= g.N) return;
      float* dst = g.C + gr * g.c_sm + gc;
      if constexpr (PLAIN) {  // (wide: N % 4 == 0, a quad is in or out)
        *reinterpret_cast<f32x4*>(dst) = g.alpha * s;
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
          f32x4 o = {v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(dst) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gc + e < g.N) dst[e] = v[e];
        }
      }
    };
    if constexpr (SK) {
      // stream-K: this run's partial tile, summed over the waves in wave order, stays in registers (four quads a thread)
      static_assert(PASSES == 1 && BM * BN / 4 % (NW * 64) == 0, "whole quads per thread, the tile in one band");
      constexpr int QPT = BM * BN / 4 / (NW * 64);
      typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
      f32x4 own[QPT];
#pragma unroll
      for (int u = 0; u < QPT; ++u) {
        const int q = tid + u * NW * 64;
        own[u] = *reinterpret_cast<const f32x4*>(smem + q * 4);
#pragma unroll
        for (int w = 1; w < NW; ++w) own[u] += *reinterpret_cast<const f32x4*>(smem + w * WAVE_FLOATS + q * 4);
      }
      bool fin = true;
      if (sk_np > 1) {   // (uniform) the tile has other contributors: the hand-over of KS > 1 -- write-through, counter, last arriver
        const int which = sk_start(sk_w) >= bid * KT ? 0 : 1;
        const __amdgpu_buffer_rsrc_t rmine = __builtin_amdgcn_make_buffer_rsrc(g.pair_ws + (long)(2 * sk_w + which) * (BM * BN), 0, BM * BN * 4, 0x00020000);
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
          const u32x4w v = {__float_as_uint(own[u].x), __float_as_uint(own[u].y), __float_as_uint(own[u].z), __float_as_uint(own[u].w)};
          __builtin_amdgcn_raw_buffer_store_b128(v, rmine, (tid + u * NW * 64) * 16, 0, 17);   // aux 17 = sc0 | sc1
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's stores have left for memory ...
        __syncthreads();
        KW_STAMP();   // (published)
        __shared__ int sk_last;
        if (tid == 0) {
          unsigned* ctr = g.pair_ctr + bid * 16;
          const unsigned seen = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          sk_last = seen == (unsigned)(sk_np - 1);
          if (sk_last) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
        __syncthreads();
        KW_STAMP();   // (counted)
        fin = sk_last != 0;   // ... and whoever arrives last finishes the tile
      }
      if (fin) {
        f32x4 tot[QPT];
#pragma unroll
        for (int u = 0; u < QPT; ++u) tot[u] = own[u];
        if (sk_np > 1) {
          // the other parts, four at a time, every load of a batch issued ahead of the sums; added in k order = workgroup order
          for (int j0 = 0; j0 < sk_np; j0 += 4) {
            f32x4 v[4][QPT];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const int j = j0 + jj, wj = sk_wlo + j;
              const int wh = sk_start(wj) >= bid * KT ? 0 : 1;
              const __amdgpu_buffer_rsrc_t rj = __builtin_amdgcn_make_buffer_rsrc(g.pair_ws + (long)(2 * wj + wh) * (BM * BN), 0, BM * BN * 4, 0x00020000);
#pragma unroll
              for (int u = 0; u < QPT; ++u) {
                u32x4w x = {0u, 0u, 0u, 0u};
                if (j < sk_np && j != sk_mine) x = __builtin_amdgcn_raw_buffer_load_b128(rj, (tid + u * NW * 64) * 16, 0, 17);   // sc0 | sc1
                v[jj][u] = f32x4{__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w)};
              }
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const int j = j0 + jj;
              if (j < sk_np) {
#pragma unroll
                for (int u = 0; u < QPT; ++u) {
                  const f32x4 val = j == sk_mine ? own[u] : v[jj][u];
                  tot[u] = j == 0 ? val : tot[u] + val;
                }
              }
            }
          }
          KW_STAMP();   // (the other parts are here)
        }
        const bool plain = g.wide && !g.bias && g.act == 0 && !g.dact && !g.cin;
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
          const int q = tid + u * NW * 64;
          if (plain) emit(tot[u], q / (BN / 4), (q % (BN / 4)) * 4, std::true_type{});
          else emit(tot[u], q / (BN / 4), (q % (BN / 4)) * 4, std::false_type{});
        }
      }
      KW_STAMP();   // (run done)
      if (sk_u < sk_end) __syncthreads();   // (the next run's DMA overwrites what the waves have just read)
      continue;
    }
    // (!SPLIT: a wave reads back what it wrote itself -- LDS operations of one wave complete in order)
    // (two instantiations of the way out: a plain product has no per-element branches on bias / activation / act')
    if constexpr (PAIR) {
      static_assert(PASSES == 1, "the protocol is written for a partial tile that fits the images");
      // this workgroup's partial tile (its waves' partials summed in wave order) -> its slot of the workspace
      // WRITE-THROUGH stores (sc0 sc1) here and system-scope loads (sc0 sc1) in the last arriver (round 5): the hand-over is
      // correct wherever the workgroups of a tile run.  Rounds 3-4 used plain stores and L1-bypassing loads that met in ONE
      // XCD's L2 -- correct only while "workgroup b runs on XCD b % 8" holds, which is observed behaviour, probed once at
      // start-up, and nothing a queue that is preempted and resumed beside other processes is known to keep.  The placement
      // is still asked for (the partials of a tile travel through one XCD's fabric port) but nothing depends on it.
      typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
      const __amdgpu_buffer_rsrc_t rmine = __builtin_amdgcn_make_buffer_rsrc(g.pair_ws + ((long)bid * KS + ksp) * (BM * BN), 0, BM * BN * 4, 0x00020000);
      for (int q = tid; q < BM * BN / 4; q += NW * 64) {
        f32x4 s = *reinterpret_cast<const f32x4*>(smem + q * 4);
#pragma unroll
        for (int w = 1; w < NW; ++w) s += *reinterpret_cast<const f32x4*>(smem + w * WAVE_FLOATS + q * 4);
        const u32x4w v = {__float_as_uint(s.x), __float_as_uint(s.y), __float_as_uint(s.z), __float_as_uint(s.w)};
        __builtin_amdgcn_raw_buffer_store_b128(v, rmine, q * 16, 0, 17);   // aux 17 = sc0 | sc1
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's stores have left for memory ...
      __syncthreads();
      __shared__ int pair_last;
      if (tid == 0) {
        unsigned* ctr = g.pair_ctr + bid * 16;
        const unsigned seen = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pair_last = seen == (unsigned)(KS - 1);
        if (pair_last) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      }
      __syncthreads();
      if (!pair_last) return;        // ... and whoever arrives last finishes the tile
    }
    // the other workgroups' partials: every load of this thread goes out in ONE batch, ahead of the LDS sums (buffer
    // loads with sc0 sc1: system scope, past this CU's L1 and whatever an L2 may hold of the slot from an earlier launch --
    // and, unlike atomic loads, nothing the compiler serialises: one round trip instead of one per quad)
    constexpr int QPT = PAIR ? BM * BN / 4 / (NW * 64) : 1;   // quads of the tile per thread
    static_assert(!PAIR || BM * BN / 4 % (NW * 64) == 0, "whole quads per thread");
    f32x4 others[PAIR ? KS : 1][QPT];
    if constexpr (PAIR) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc(g.pair_ws + (long)bid * KS * (BM * BN), 0, KS * BM * BN * 4, 0x00020000);
#pragma unroll
      for (int j = 0; j < KS; ++j)
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
          u32x4 v = {0u, 0u, 0u, 0u};
          if (j != ksp) v = __builtin_amdgcn_raw_buffer_load_b128(rws, (j * BM * BN + (tid + u * NW * 64) * 4) * 4, 0, 17);   // sc0 | sc1
          others[j][u] = f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
        }
    }
    auto finish = [&](auto plainc) {
      auto one = [&](const int q, auto uc) {   // quad q of the band (uc: which of this thread's quads, KS > 1)
        const int row = q / (BN / 4), c4 = (q % (BN / 4)) * 4;
        f32x4 s = *reinterpret_cast<const f32x4*>(smem + (SPLIT ? 0 : wave * WAVE_FLOATS) + row * BN + c4);
        if constexpr (SPLIT) {
#pragma unroll
          for (int w = 1; w < NW; ++w) s += *reinterpret_cast<const f32x4*>(smem + w * WAVE_FLOATS + row * BN + c4);
        }
        if constexpr (PAIR) {   // (k order: part 0 + part 1 + ..., whoever is adding)
          const f32x4 own = s;
#pragma unroll
          for (int j = 0; j < KS; ++j) {
            const f32x4 v = j != ksp ? others[j][decltype(uc)::value] : own;   // (uniform)
            s = j == 0 ? v : s + v;
          }
        }
        emit(s, row, c4, plainc);
      };
      if constexpr (PAIR) kw_static_for<0, QPT>([&](auto uc) { one(tid + decltype(uc)::value * NW * 64, uc); });
      else
        for (int q = SPLIT ? tid : lane; q < RP * BN / 4; q += SPLIT ? NW * 64 : 64) one(q, std::integral_constant<int, 0>{});
    };
    if (g.wide && !g.bias && g.act == 0 && !g.dact && !g.cin) finish(std::true_type{});
    else finish(std::false_type{});
  }
 } while (SK && sk_u < sk_end);
#undef KW_STAMP
}

static int kw_mode() {
  static const int m = [] { const char* e = ab_getenv("TOPS_GEMM_KW"); return e ? atoi(e) : 1; }();
  return m;
}

// Can the problem run here at all?
static bool kw_can(const GemmProblem& p) {
  if (p.dtype != TO_F32 || p.batch != 1 || p.reduce_batch || p.rowsum || p.loss_rows || (p.beta != 0.0 && !p.Cin)) return false;
  // (8 .. 127 rows or columns: a narrow Learn layer under a tall batch -- 8192 x 300 x 100, 10 x 60000 x 100 -- pads its last
  //  tile; the loads clamp and the stores are guarded as for any ragged extent)
  if (p.M < 8 || p.N < 8 || p.K < 16) return false;
  if (p.M > 2147483647LL || p.N > 2147483647LL || p.K > 2147483647LL) return false;
  const bool a_k = p.a_sk == 1, a_m = !a_k && p.a_sm == 1;
  const bool b_n = p.b_sn == 1, b_k = !b_n && p.b_sk == 1;
  if (!(a_k || a_m) || !(b_n || b_k)) return false;
  auto al4 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 3u) == 0; };
  if (!al4(p.A) || !al4(p.B)) return false;  // (global memory runs in unaligned-access mode: dword alignment is enough)
  // (the DMA addresses a lane as scalar base + 32-bit byte offset)
  const int64_t ext_a = a_k ? p.M * p.a_sm : 16 * p.a_sk + p.M, ext_b = b_k ? p.N * p.b_sn : 16 * p.b_sk + p.N;
  if (ext_a * 4 + 8192 >= (1LL << 32) || ext_b * 4 + 8192 >= (1LL << 32) || p.a_sm < 0 || p.a_sk < 0 || p.b_sk < 0 || p.b_sn < 0) return false;
  if (a_m && p.M % 4 != 0) return false;     // an m-contiguous quad must be in or out of the matrix as a whole
  if (b_n && p.N % 4 != 0) return false;
  return true;
}

// Thousands of tiles and a K of a few hundred (8192 x 512 x 8192, 16384 x 512 x 4096): a 256x256 tile is then a few dozen
// k-tiles between a prologue and a 256 KiB epilogue that nothing overlaps (one workgroup per CU).  Here every WAVE takes a
// 64x64 tile of its own with the whole K loop (SPLIT = false): eight independent waves per CU, one's prologue and
// stores under the others' MFMAs.  old / this, TF: 8192 x 384 x 8192 107 / 122, 16384 x 512 x 4096 116 / 125,
// 8192 x 768 x 8192 124 / 130, 8192 x 1024 x 8192 129 / 132; level at K = 256 (110 / 111) and at 4096 tiles; behind from
// K = 2048 on (138 / 136: A and B pass through L2 once per 64-wide panel).
static bool kw_many_tiles_mid_k(const GemmProblem& p) {
  const long t64 = ((p.M + 63) / 64) * ((p.N + 63) / 64);
  return t64 >= 8192 && p.K >= 320 && p.K <= 1536;
}

// More than 1,024 tiles: whose are they?  Round 6 (last) found by a grid scan against the vendor GEMM (tools/gemm_scan.py,
// profiles/r06_gemm_scan.txt) that the 256x256 / 128x128 tiles only own the shapes they FIT -- M and N multiples of 256, K of 16,
// whole rounds of the 256 CUs -- and that everything else (the reference's own 60000 x 784 x 300 and 60000 x 300 x 100, app/MNIST.hs
// batched; 10000 x 300 x 2048; 6144 x K x 4096 = one and a half rounds; any K below 512) runs 1.2 .. 3.4 x faster here, a tile per
// WAVE or per workgroup.  us, old route / a workgroup per tile / a wave per tile (vendor GEMM): 60000 x 784 x 300 363 / 244 / 228
// (283), 60000 x 300 x 100 87 / 52 / 43 (46), 10000 x 300 x 2048 285 / 125 / 103 (111), 784 x 300 x 10000 151 / 53 / 44 (55),
// 6144 x 1024 x 4096 455 / 382 / 360, 5120 x 1024 x 5120 499 / 400 / 414, 4096 x 256 x 2048 46 / 43 / 37 (38), 10000 x 2048 x 2048
// 707 / 604 / 581 (580), 2048 x 60000 x 10000 22.4 / 18.2 / 17.1 ms (18.8); where the big tiles fit the three agree within 1 %
// from K = 512 on (4096 x K x 4096, 8192 x K x 8192, 16384 x K x 4096) and a wave per tile is 2-5 % ahead below.
// Which of the two forms: a wave per tile runs 2,048 tiles at a time (eight waves a CU, two workgroups of four), and a last
// round that is mostly empty costs half a round all the same (3072 x 1024 x 3072 = 2,304 tiles: 147 us a workgroup per tile, 179 a
// wave per tile); a workgroup per tile runs rounds of 256 tiles.  With K below 512 a wave per tile is ahead regardless.
static bool kw_wave_per_tile_fits(long t64) {
  const long wg = (t64 + 3) / 4, r = wg / 512, rem = wg % 512;
  // (5,024 tiles, 2.45 rounds: 581 us against 604 a workgroup per tile; 6,400, 3.125: 815 against 766; under two rounds the end of
  //  the launch is uneven whatever the remainder: 3584^3 = 3,136 tiles 115 TF against 137, 2560^3 = 1,600 116 against 129)
  return r >= 4 || (r >= 2 && (rem == 0 || rem >= 128));
}
static bool kw_workgroup_per_tile_fits(long t64) { return ((t64 + 255) / 256) * 256 * 100 <= 108 * t64; }
static bool kw_big_tiles_fit(const GemmProblem& p) {
  return p.M % 256 == 0 && p.N % 256 == 0 && p.K % 16 == 0 && ((p.M / 256) * (p.N / 256)) % 256 == 0;
}
// 0: not here; 1: a workgroup per tile; 2: a wave per tile
static int kw_many_tiles_form(const GemmProblem& p) {
  const long t64 = ((p.M + 63) / 64) * ((p.N + 63) / 64);
  if (t64 <= 1024 || p.K < 16) return 0;
  if (p.K < 64) return (p.M < p.N ? p.M : p.N) >= 1024 ? 2 : 0;   // (a rank-32 update of a large matrix, W - r dZ^T X at 32 rows: all epilogue; a wave per tile)
  const int before = (t64 <= 3200 && p.K >= 512) ? 1 : kw_many_tiles_mid_k(p) ? 2 : 0;   // (rounds 4-5: measured on shapes the big tiles fit)
  // (gemm_w4_edge_whole: ragged, but its edge tiles fill whole rounds with little padding -- 4000^3 = 256 tiles, 143 TF there, 137 here)
  if ((kw_big_tiles_fit(p) || gemm_w4_edge_whole(p)) && p.K >= 512) return before;
  const bool wf = kw_wave_per_tile_fits(t64), gf = kw_workgroup_per_tile_fits(t64);
  if (p.K < 512) return 2;   // (a short K is not worth sharing among four waves: 10000 x 300 x 2048 125 us a workgroup per tile, 103 a wave)
  // (a narrow layer under a tall batch, two tile columns: a wave per tile even under two rounds -- 60000 x 784 x 100 94 us, 108 a workgroup per tile)
  if ((p.M < p.N ? p.M : p.N) < 512 && t64 >= 1800 && p.K <= 1536) return 2;
  return wf ? 2 : gf ? 1 : before;
}
// A K below 128 on 200 .. 1,024 tiles (the cotangent coming back through a narrow layer, 8192 x 100 x 300) was nobody's either:
// 20.5 us on the old 64x64 body, 12.8 / 11.8 here (vendor 17.7); 16384 x 64 x 256 12.0 -> 12.1 / 10.3.
static bool kw_short_k(const GemmProblem& p) {
  const long t64 = ((p.M + 63) / 64) * ((p.N + 63) / 64);
  return t64 >= 200 && t64 <= 1024 && p.K >= 64 && p.K < 128;
}

static bool kw_few_tiles_long_k(const GemmProblem& p);
static bool kw_stream_few_tiles(const GemmProblem& p);

// ... and should it?
bool gemm_kw_applicable(const GemmProblem& p) {
  const int mode = kw_mode();
  if (mode == 0 || !kw_can(p)) return false;
  if (mode >= 2) return true;
  // measured against the routes of gemm_f32_mfma.hip (tools/kw_check.py time, ours there / here, TF): ahead from ~100
  // tiles of 64x64 (640^3 30 / 35; 512^3 36 / 20: the small-GEMM kernel's territory) through 1024^3 64 / 93, 1536^3
  // 92 / 117, 2048^3 126 / 127, 2560^3 109 / 121, 3072^3 122 / 138 up to 3584^3 129 / 133; at 4096^3 the 256x256 tiles
  // are ahead (143 / 141).  Short K: level from K = 128 on (2048 x 128 x 2048 60 / 58, 1024 x 256 x 1024 25 / 59), but a
  // long stream of rows with a short K belongs to the 256x256 tiles or the streaming kernel (16384 x 256 x 4096: 110 / 95).
  const long t64 = ((p.M + 63) / 64) * ((p.N + 63) / 64);
  // exactly one round of 128x128 tiles of a plain product: since its refit (scalar-base DMA, immediate-offset fragment
  // reads, the plain way out) the pinned body on 128x128 tiles is ahead there -- here / there, TF, steady state:
  // 2048^3 138 / 142, 2048 x 512 x 2048 115 / 122, 2048 x 8192 x 2048 144.5 / 148, 4096 x 2048 x 1024 138 / 142
  if (p.M % 128 == 0 && p.N % 128 == 0 && (p.M / 128) * (p.N / 128) == 256 && p.K % 16 == 0 && p.K >= 256 &&
      p.alpha == 1.0 && p.beta == 0.0 && !p.bias && !p.dact && p.act == 0 && p.c_sm % 4 == 0 &&
      (reinterpret_cast<uintptr_t>(p.C) & 15u) == 0)
    return false;
  if (t64 >= 100 && p.K >= 128 && t64 <= 1024 && (p.M >= 128 || p.N >= 128)) return true;
  // few tiles and a long K, several workgroups per tile (kw_ksplit): 512 x 2048 x 512 18.9 -> 13.2 us, 384 x 4096 x 384
  // 32.7 -> 21.3, 256 x 4096 x 1024 31.4 -> 21.6; level at K = 1024 (512 x 1024 x 512: 10.1 / 9.8)
  if (kw_few_tiles_long_k(p) || kw_stream_few_tiles(p)) return true;
  return kw_many_tiles_form(p) != 0 || kw_short_k(p);
}

template <int TM, int TN, int NW, int NI, bool SPLIT = true, int KS = 1>
static void kw_launch_modes(int mode, dim3 grid, hipStream_t s, const KwArgs& g) {
  dim3 block(NW * 64);
  switch (mode) {
    case 0: launch_k((gemm_kw_kernel<0, 0, TM, TN, NW, NI, SPLIT, KS>), grid, block, 0, s, g); break;
    case 1: launch_k((gemm_kw_kernel<0, 1, TM, TN, NW, NI, SPLIT, KS>), grid, block, 0, s, g); break;
    case 2: launch_k((gemm_kw_kernel<1, 0, TM, TN, NW, NI, SPLIT, KS>), grid, block, 0, s, g); break;
    default: launch_k((gemm_kw_kernel<1, 1, TM, TN, NW, NI, SPLIT, KS>), grid, block, 0, s, g); break;
  }
}

// The split forms' workspace and counters: allocated once, at to_init (a first use inside a stream capture could not),
// and only if the placement they rely on holds: workgroup b of a grid on XCD b % 8 (SPX mode, every CU enabled: observed
// behaviour, not a documented contract).  Probed with a grid of the shape the kernels use; another partition mode or a CU
// mask fails the probe and every problem keeps one workgroup per tile.
constexpr int KW_KS_MAX = 8, KW_WS_SLOTS = 2048;   // (2048 partial-tile slots of 16 KiB: 256 tiles eight ways ... 682 tiles three ways)
constexpr int KW_WS_MAX_TILES = 1024;
static float* g_kw_pair_ws = nullptr;
static unsigned* g_kw_pair_ctr = nullptr;
__global__ void kw_xcc_probe_kernel(int* out) {
  if (threadIdx.x != 0) return;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  out[blockIdx.x] = (int)(xcc & 0xf);
}
// (shared with the seam launch of gemm_small.hip: xcd_placement_probe, common.hpp)
static bool kw_placement_probe_once() {
  constexpr int G = 512;
  int* host = nullptr;
  if (hipHostMalloc(&host, G * sizeof(int), hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return false; }
  int* dev = nullptr;
  bool ok = hipHostGetDevicePointer(reinterpret_cast<void**>(&dev), host, 0) == hipSuccess;
  for (int rep = 0; ok && rep < 3; ++rep) {   // (a placement that only sometimes holds is no placement)
    for (int i = 0; i < G; ++i) host[i] = -1;
    kw_xcc_probe_kernel<<<dim3(G), dim3(256), 0, nullptr>>>(dev);
    ok = hipDeviceSynchronize() == hipSuccess;
    for (int i = 0; ok && i < G; ++i) ok = host[i] >= 0 && host[i] == host[i & 7];
    for (int i = 1; ok && i < 8; ++i)
      for (int j = 0; ok && j < i; ++j) ok = host[i] != host[j];
  }
  (void)hipHostFree(host);
  (void)hipGetLastError();
  return ok;
}
bool xcd_placement_probe() {
  static const bool ok = kw_placement_probe_once();
  return ok;
}
void gemm_kw_pair_init() {
  if (g_kw_pair_ws) return;
  if (!xcd_placement_probe()) return;
  if (hipMalloc(&g_kw_pair_ws, (size_t)KW_WS_SLOTS * 64 * 64 * sizeof(float)) != hipSuccess ||
      hipMalloc(&g_kw_pair_ctr, (size_t)KW_WS_MAX_TILES * 16 * sizeof(unsigned)) != hipSuccess ||
      hipMemset(g_kw_pair_ctr, 0, (size_t)KW_WS_MAX_TILES * 16 * sizeof(unsigned)) != hipSuccess) {
    (void)hipGetLastError();
    g_kw_pair_ws = nullptr;
  }
}

// How many workgroups per tile?  XCD x owns ceil(T / 8) of the T tiles; with S workgroups per tile its 32 CUs work through
// R = ceil(S ceil(T / 8) / 32) parts of K / S k-tiles one after the other (co-resident workgroups share the matrix pipe).
// Fitted to the measurements below (a launch is 4.1 us + 0.222 us per k-tile a workgroup's four waves work through): the
// way out of a split tile costs 0.6 + 0.5 S us (partial tile to the L2, counter, the others' partials back), and 1.2 us
// more when workgroups share CUs.  In k-tile units:
//   cost(S) = R KT / S + [S > 1] (2.7 + 2.25 S) + [S > 1, R > 1] 5.4
// us, S = 1 / 2 / 3 / 4: 640^3 12.7 / 10.1 / 13.8 / 12.3; 704^3 13.6 / 11.2 / 14.0 / 12.6; 768^3 14.8 / 17.1 / 14.3 / -;
// 832^3 (169 tiles: 22 per XCD, three ways is three rounds) 15.8 / 19.1 / 22.7 / -; 1024 x 1024 x 512 18.2 / 12.9 / 17.5 /
// 14.8; 512 x 2048 x 512 32.1 / 19.6 / 15.6 / 13.2 (18.9 on the 128x128 split-K route); 384 x 4096 x 384 60.7 / 34.1 /
// 25.6 / 21.3 (32.7); 256 x 4096 x 1024 60.8 / 34.2 / 25.7 / 21.6 (31.4); 768 x 4096 x 768 61.0 / 62.5 / 45.9 / -.
static int kw_ksplit(const GemmProblem& p, int t) {
  static const int forced = [] { const char* e = ab_getenv("TOPS_GEMM_KW_PAIR"); return e ? atoi(e) : -1; }();
  // product switch: TOPS_GEMM_KW_KSPLIT=0 keeps one workgroup per tile (the split forms share ONE process-wide workspace
  // and counter array: they assume the library's single stream, like the pool allocator)
  static const bool off = [] { const char* e = getenv("TOPS_GEMM_KW_KSPLIT"); return e && e[0] == '0'; }();
  if (off || !g_kw_pair_ws || t != 2) return 1;
  const long T = ((p.M + 63) / 64) * ((p.N + 63) / 64), KT = p.K / 16, per_xcd = (T + 7) / 8;
  constexpr long SLOTS = KW_WS_SLOTS;
  if (T > KW_WS_MAX_TILES) return 1;
  if (forced >= 0) {   // (A/B runs: as asked, as far as the workspace goes)
    int S = forced <= 1 ? 1 : (forced > KW_KS_MAX ? KW_KS_MAX : forced);
    if (S == 5 || S == 7) --S;
    while (S > 1 && (S * T > SLOTS || S == 5 || S == 7)) --S;
    return S;
  }
  // More tiles than CUs: the dispatcher already balances whole-tile workgroups over the CUs as they finish, and a split only
  // adds its exchange -- measured (us, S = 1 / 2 / 3): 1280^3 40.4 / 46.0 / 50.3, 1408^3 44.4 / 50.6 / 60.2, 1152 x 512 x 1152
  // 19.1 / 23.8 / 23.1, 1792^3 106 / 104 / 104; only 1152^3 gains (36.6 / 41.4 / 34.0).  (The workspace would hold 1024 tiles;
  // forced splits of such problems remain for A/B runs.)
  if (T > 256) return 1;
  int best = 1;
  double best_cost = (double)((per_xcd + 31) / 32) * KT;   // (one workgroup per tile: ceil(per_xcd / 32) whole K loops per CU)
  for (int S : {2, 3, 4, 6, 8}) {
    if (KT < 16L * S || S * T > SLOTS) continue;   // (at least four k-tiles per wave)
    const long R = (S * per_xcd + 31) / 32;
    if (R > 2) continue;   // (a CU holds two workgroups at a time; three parts a CU ran 35 % over this model: 784 x 4096 x 784 four ways 66.8 us for 50.5)
    const double cost = (double)R * KT / S + 2.7 + 2.25 * S + (R > 1 ? 5.4 : 0.0);
    if (cost < best_cost) { best_cost = cost; best = S; }
  }
  return best;
}
// (gemm_kw_applicable: is a split worth taking a problem of few tiles away from the small-GEMM / split-K routes?)
static bool kw_few_tiles_long_k(const GemmProblem& p) {
  const long T = ((p.M + 63) / 64) * ((p.N + 63) / 64);
  // (16 .. 127 rows or columns too: 784 x 60000 x 100 440 -> 117 us; with 8 .. 15 the small-GEMM kernel is ahead: 8 x 2048 x 2048 7.7 us / 9.6)
  return T >= 16 && T < 100 && p.K >= 1536 && p.M >= 16 && p.N >= 16 && kw_ksplit(p, 2) > 1;
}

// A handful of tiles under a very long K -- a narrow layer's weight gradient over a big batch, 100 x 8192 x 300, 128 x 16384 x
// 256: the small-GEMM kernel gives a tile ONE workgroup whatever K is (7 us per 1,024 of K), the KS-way split at most eight.
// As a stream over 256 workgroups (128 below eight tiles) every CU takes an equal share of K and a tile's last contributor adds
// the others' parts in k order.  us, small-GEMM kernel / stream (profiles/r06_learn_shapes.txt): 100 x 4096 x 300 33 / 18,
// 100 x 8192 x 300 64 / 21 (vendor 30), 100 x 32768 x 300 240 / 41, 100 x 60000 x 300 444 / 53-65 (vendor 206), 128 x 8192 x 128
// 53 / 18, 64 x 8192 x 784 58 / 22; level at K = 2048 (18 / 17), behind below; from 16 tiles on the KS-way split is as good.
static bool kw_stream_few_tiles(const GemmProblem& p) {
  static const bool off = [] { const char* e = getenv("TOPS_GEMM_KW_KSPLIT"); return e && e[0] == '0'; }();
  const long T = ((p.M + 63) / 64) * ((p.N + 63) / 64);
  return !off && g_kw_pair_ws && T <= 16 && p.K >= 3072;   // (16 tiles, 1024 x 60000 x 32: 385 us -> 113 eight ways -> 75 as a stream)
}

// (api.cpp, gemm_small_route: a few tiles under a long K are this kernel's even where the small-GEMM kernel could run them)
bool gemm_kw_long_k(const GemmProblem& p) {
  if (kw_mode() == 0 || !kw_can(p)) return false;
  return kw_few_tiles_long_k(p) || kw_stream_few_tiles(p);
}

// One tile per WAVE instead of per workgroup?
static bool kw_unsplit(const GemmProblem& p) {
  static const int forced = [] { const char* e = ab_getenv("TOPS_GEMM_KW_SPLIT"); return e ? atoi(e) : -1; }();
  if (forced >= 0) return forced == 0;
  const long t64 = ((p.M + 63) / 64) * ((p.N + 63) / 64);
  // (a K of a hundred is too short to share among four waves: from two tiles a CU on)
  return kw_many_tiles_form(p) == 2 || (kw_short_k(p) && t64 >= 512);
}

// Output tile of a workgroup: 64x64 (two workgroups per CU: 64 KiB of LDS and ~130 registers per wave) or 96x96 (one per
// CU: ~300 registers, 96 KiB).  The larger tile has the better MFMA stream (fewer fragment reads and DMA instructions per
// MFMA) but only pays when its count fills the 256 CUs evenly.  (128x128 -- 256 accumulator registers, 128 KiB -- was
// built and measured no faster than 64x64 even at 2048^3 = 256 tiles: 126.7 vs 127.2 TF, and, a tile per wave, behind the
// barrier-synchronised 256x256 kernel at 4096^3: 137 vs 143; not instantiated.)
static int kw_tile(const GemmProblem& p) {
  static const int forced = [] { const char* e = ab_getenv("TOPS_GEMM_KW_TILE"); return e ? atoi(e) : 0; }();
  if (forced == 2 || forced == 3) return forced;
  // 96x96 only when its tiles fill one round of the 256 CUs almost exactly (1536^3: 256 tiles, 117 TF against 97 on
  // 64x64; 1408^3: 225 tiles, 99 against 112; 2048^3: 484 tiles = two rounds, 113 against 127)
  const long t3 = ((p.M + 95) / 96) * ((p.N + 95) / 96);
  if (t3 >= 244 && t3 <= 256 && 100 * p.M * p.N >= 97 * t3 * 96 * 96) return 3;
  return 2;
}

// Stream-K (KS == 0) instead of whole tiles / a KS-way split?  Returns the grid (0: no).
static int kw_streamk(const GemmProblem& p, int t) {
  static const int forced = [] { const char* e = ab_getenv("TOPS_GEMM_KW_SK"); return e ? atoi(e) : -1; }();
  static const bool off = [] { const char* e = getenv("TOPS_GEMM_KW_KSPLIT"); return e && e[0] == '0'; }();   // (the same workspace, the same switch)
  if (off || forced == 0 || !g_kw_pair_ws || t != 2) return 0;
  const long T = ((p.M + 63) / 64) * ((p.N + 63) / 64), KT = p.K / 16;
  if (T > KW_WS_MAX_TILES || KT < 1) return 0;
  if (forced > 0) {
    int G = forced & ~7;
    if (G < 8) G = 8;
    if (2 * G > KW_WS_SLOTS) G = KW_WS_SLOTS / 2;
    return G;
  }
  // More tiles than CUs, and a last round of tiles that leaves most of them idle.  Fitted to the sweep of round 6
  // (profiles/r06_kw_streamk_sweep.txt; 1088^3 .. 1984^3 and six rectangular shapes, both forms within 3 % of it):
  //   whole tiles, two workgroups a CU:  4 us + 0.228 us x ceil(T / 256) x KT     (a CU's share is whole K loops)
  //   stream-K on 512 workgroups:       11 us + 0.245 us x (T / 256) x KT         (two runs a workgroup, two hand-overs)
  // us, whole tiles / stream-K: 1088^3 35.0 / 29.8, 1152^3 37.0 / 32.4, 1472^3 67.2 / 57.0, 1792^3 106 / 95.7, 1152 x 2048 x 1152
  // 61.1 / 49.5, 1280 x 4096 x 1280 118 / 109; level or behind where the last round is more than half full (1280^3 40.5 /
  // 41.3, 1408^3 44.5 / 50.7, 1728^3 78.3 / 85.9) and wherever K is short (1280 x 512 x 1280 19.4 / 23.3).
  // Up to 256 tiles the KS-way split is ahead at every size measured (768^3: 13.7 us three ways, 16.7 as a stream over 256
  // workgroups, 17.9 over 512: a run's fixed costs -- ~1.2 us until its first k-tile has landed, ~3 us from its last MFMA
  // to C -- are paid twice by most workgroups and are as long as the K loops they sit between; stamps in profiles/README.md).
  if (kw_stream_few_tiles(p)) return T < 8 ? 128 : 256;
  if (T <= 256) return 0;
  // (from KT ~ 512 on a share costs more per k-tile -- tiles that would walk K in step and share their panels in the L2 no longer
  //  do: 0.25 .. 0.32 us measured at KT = 625 and 3,750)
  const double whole = 4.0 + 0.228 * (double)((T + 255) / 256) * KT, stream = 11.0 + (KT >= 512 ? 0.29 : 0.245) * (double)T / 256.0 * KT;
  // (a long K is better shared by ONE workgroup a CU -- 4096 x 10000 x 300 = 320 tiles: 259 us over 512 workgroups, 204 over 256;
  //  4096 x 60000 x 300 1652 / 1320; level at KT ~ 100: 1152^3 32.4 / 33.8, 1792^3 95.8 / 96.1)
  return stream < 0.97 * whole ? (KT >= 192 ? 256 : 512) : 0;
}

// development build: what TOPS_GEMM_KW_DBG asked the kernel to stamp
static void kw_dbg_report(const GemmProblem& p, const KwArgs& g, hipStream_t s) {
  if (!(g.dbg & 12)) return;
  static int printed = 0;
  if (printed++ % 40 != 39) return;
  unsigned long long h[32];
  TO_HIP(hipStreamSynchronize(s));
  TO_HIP(hipMemcpy(h, g.dbg_out, sizeof(h), hipMemcpyDeviceToHost));
  if (g.dbg & 4)
    fprintf(stderr, "kw dbg %ldx%ldx%ld: K loop of wave 0 of block 0: %llu shader cycles, %llu ticks of 100 MHz -> %.0f MHz; %llu k-tiles, %.0f cycles each\n",
            (long)p.M, (long)p.K, (long)p.N, h[0], h[1], h[1] ? 100.0 * h[0] / h[1] : 0.0, h[2], h[2] ? (double)h[0] / h[2] : 0.0);
  if (g.dbg & 8) {
    fprintf(stderr, "kw stamps %ldx%ldx%ld block %d (shader cycles since the workgroup began):", (long)p.M, (long)p.K, (long)p.N, g.dbg >> 4);
    for (int i = 1; i < 24 && h[8 + i] > h[8]; ++i) fprintf(stderr, " %llu", h[8 + i] - h[8]);
    fprintf(stderr, "\n");
  }
  TO_HIP(hipMemsetAsync(g.dbg_out, 0, 32 * sizeof(unsigned long long), s));
}

void launch_gemm_kw(const GemmProblem& p, hipStream_t s) {
  KwArgs g{};
  g.A = (const float*)p.A; g.B = (const float*)p.B; g.C = (float*)p.C;
  g.M = (int)p.M; g.N = (int)p.N; g.K = (int)p.K;
  g.a_sm = p.a_sm; g.a_sk = p.a_sk; g.b_sk = p.b_sk; g.b_sn = p.b_sn; g.c_sm = p.c_sm;
  const int t = kw_tile(p);
  g.tiles_m = (int)((p.M + 32 * t - 1) / (32 * t));
  g.tiles_n = (int)((p.N + 32 * t - 1) / (32 * t));
  g.alpha = (float)p.alpha;
  g.bias = (const float*)p.bias; g.dact = (const float*)p.dact; g.act = p.act; g.dact_kind = p.dact_kind;
  g.cin = p.beta != 0.0 ? (const float*)p.Cin : nullptr; g.beta = (float)p.beta;
  g.wide = (reinterpret_cast<uintptr_t>(p.C) & 15u) == 0 && p.c_sm % 4 == 0 && p.N % 4 == 0;
  g.dbg = [] { const char* e = ab_getenv("TOPS_GEMM_KW_DBG"); return e ? atoi(e) : 0; }();
  static unsigned long long* dbg_buf = nullptr;
  if ((g.dbg & 12) && !dbg_buf) {
    TO_HIP(hipMalloc(&dbg_buf, 32 * sizeof(unsigned long long)));
    TO_HIP(hipMemset(dbg_buf, 0, 32 * sizeof(unsigned long long)));
  }
  g.dbg_out = dbg_buf;
  const int mode = (p.a_sk == 1 ? 0 : 2) + (p.b_sn == 1 ? 0 : 1);
  // Two images per wave and operand on 64x64 tiles: 64 KiB per workgroup, so two workgroups share a CU and one's waits
  // hide under the other's MFMAs (against three images / one workgroup per CU: 1024^3 90 -> 92 TF, 1536^3 86 -> 94,
  // 2048^3 117 -> 126).  TOPS_GEMM_KW_NI=3: three images -- for A/B runs.
  static const int ni3 = [] { const char* e = ab_getenv("TOPS_GEMM_KW_NI"); return e ? atoi(e) == 3 : 0; }();
  dim3 grid(g.tiles_m * g.tiles_n);
  if (t == 2 && kw_unsplit(p)) {
    kw_launch_modes<2, 2, 4, 2, false>(mode, dim3((g.tiles_m * g.tiles_n + 3) / 4), s, g);
    TO_HIP(hipGetLastError());
    count_launch();
    return;
  }
  if (const int G = kw_streamk(p, t)) {
    const long total = (long)g.tiles_m * g.tiles_n * (p.K / 16);
    g.pair_ws = g_kw_pair_ws;
    g.pair_ctr = g_kw_pair_ctr;
    g.sk_base = (int)(total / G);
    g.sk_rem = (int)(total % G);
    kw_launch_modes<2, 2, 4, 2, true, 0>(mode, dim3(G), s, g);
    TO_HIP(hipGetLastError());
    count_launch();
    kw_dbg_report(p, g, s);
    return;
  }
  const int ks = kw_ksplit(p, t);
  if (ks > 1) {
    g.pair_ws = g_kw_pair_ws;
    g.pair_ctr = g_kw_pair_ctr;
    const int ntiles = g.tiles_m * g.tiles_n;
    const dim3 gk(8 * ks * ((ntiles + 7) / 8));
    if (ks == 2) kw_launch_modes<2, 2, 4, 2, true, 2>(mode, gk, s, g);
    else if (ks == 3) kw_launch_modes<2, 2, 4, 2, true, 3>(mode, gk, s, g);
    else if (ks == 4) kw_launch_modes<2, 2, 4, 2, true, 4>(mode, gk, s, g);
    else if (ks == 6) kw_launch_modes<2, 2, 4, 2, true, 6>(mode, gk, s, g);
    else kw_launch_modes<2, 2, 4, 2, true, 8>(mode, gk, s, g);
  } else if (t == 3) kw_launch_modes<3, 3, 4, 2>(mode, grid, s, g);
  else if (ni3) kw_launch_modes<2, 2, 4, 3>(mode, grid, s, g);
  else kw_launch_modes<2, 2, 4, 2>(mode, grid, s, g);
  TO_HIP(hipGetLastError());
  count_launch();
  kw_dbg_report(p, g, s);
}

}  // namespace to
