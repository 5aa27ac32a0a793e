// fp32 GEMM for the batched training step's two big contractions (BASELINE config 3: X W1^T -> 1024 x 256 over K = 784,
// dZ1^T X -> 256 x 784 over K = 1024; src/TensorOps/Learn/NeuralNet/FeedForward.hs:131-148 through TO.gmul,
// src/TensorOps/TOp.hs:56-94): a few hundred 32x32 output tiles with a K of several hundred -- one tile per CU, the
// whole chip busy for ONE round, latency everything.
//
// Why a third small-GEMM design (round 5).  gemm_small.hip's one-shot body gives every tile sixteen waves that each fetch
// their K slice straight into MFMA fragment layout: 4,096 waves per launch (the last workgroups start 2-3 us after the
// first), every load instruction touches 32 cache lines for 32 bytes each, and the tile is done 8.7 us after its
// workgroup began for 2.6 us of MFMAs (profiles/README.md, "where the seam loses").  Here a tile is FOUR waves, one per
// SIMD: 1,024 waves per launch; operands come global -> LDS by DMA in whole 128-byte lines (a wave instruction = 8 rows x
// 128 B), each wave streams its own quarter of K through private 8 KiB stages (two chunks in flight) and waits on nothing but its own vmcnt --
// no barrier until the four partial tiles meet in LDS.  The MFMA stream (16 x v_mfma_f32_32x32x2_f32 per 32-k chunk = 1,024
// cycles) hides the 8 DMA and 8 (k-contiguous) or 32 (row-contiguous) LDS reads of the next chunk.
//
// Operand forms: KC = k-contiguous (X rows, W rows: image [x][32 k], 16-byte quads XOR-swizzled by x, fragments by
// ds_read_b128 -- a lane takes four consecutive k and feeds them to four MFMA steps; A and B agree on which) and XC =
// row-contiguous (dZ^T, X as the right operand: image [k][32 x], fragments by ds_read_b32).  MFMA step s = 4 i + j of a
// chunk consumes k_local = 4 (2 i + half) + j for both operands.
// A ragged K (784 = 24.5 chunks) costs nothing extra: in a wave's last chunk the lanes beyond its run fetch zeros.
// Epilogue (after the cross-wave sum, one float4 per thread, whole rows per store): alpha, beta * Cin, bias, logistic / tanh,
// act' from stored activations; row sums of A (bias gradients) and their in-place update -- what gemm_small's epilogue
// offers the step planner (csrc/lazy.cpp), minus the loss head, which stays on gemm_small.
#include <cstdio>
#include <type_traits>

#include "common.hpp"

namespace to {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct T32Args {   // (pointers first, then 8-byte, then 4-byte members: a float wedged between pointers made the compiler keep a
                   //  slice of the argument block in scratch and read `bias` / `dact` back from memory, ~1 us each, serially)
  const float* A;
  const float* B;
  float* C;
  const float* Cin;
  const float* bias;
  const float* dact;
  float* rowsum;
  const float* rowsum_in;
  unsigned long long* dbg;   // development builds (TOPS_T32_STAMPS=1): ticks of the 100 MHz clock at six points of workgroups 0 and last
  long a_sx, b_sx;   // element stride between consecutive rows of the operand's IMAGE: KC: between x; XC: between k
  long c_sm;
  int M, N, K;
  int tiles_m, tiles_n;
  int gm, gn;        // the XCDs as a gm x gn grid over the tiles (t32_tile_of)
  int act, dact_kind;
  int rowsum_acc;
  int pd;            // chunks a wave keeps in flight (<= NS)
  float alpha, beta;
  float rowsum_alpha;
};

// what a lane without a valid source fetches instead: its 16 bytes of the image become zeros (a k beyond the wave's run
// must contribute nothing; an x beyond the extent only reaches rows / columns that are never stored)
__device__ __attribute__((aligned(16))) float g_t32_zero[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int T32_BK = 32, T32_NS = 2, T32_NW = 4;   // two 8 KiB stages a wave: 64 KiB a workgroup, two workgroups fit a CU
constexpr int T32_CTR = 4096;                              // row blocks the joined launch's counters cover
constexpr int T32_STAGE = 2 * 32 * T32_BK;                 // floats per stage: A image + B image
constexpr int T32_WAVE = T32_NS * T32_STAGE;               // floats per wave

// Which tile a workgroup takes.  The eight XCDs (block b runs on XCD b % 8: a placement asked for, never relied on -- it only
// decides which L2 a tile's operands meet in) form a gm x gn grid over the tile grid; XCD (xm, xn) owns the rectangle
// rows [tiles_m xm / gm, tiles_m (xm + 1) / gm) x columns [tiles_n xn / gn, ...), so its L2 fetches tiles_m / gm panels of
// A and tiles_n / gn of B -- dZ1^T X (8 x 25 tiles): a 2 x 4 grid pulls 10 panels of 128 KB through each XCD's fabric
// port where runs of the row-major sequence pulled 26 (the whole of X into every L2: the K loop ran at half the MFMA rate).
// The grid is 8 x the largest rectangle; workgroups beyond their XCD's rectangle leave at once.
__device__ __forceinline__ bool t32_tile_of(int tiles_m, int tiles_n, int gm, int gn, int bid, int& tm, int& tn) {
  const int x = bid & 7, local = bid >> 3, xm = x / gn, xn = x - xm * gn;
  const int r0 = tiles_m * xm / gm, r1 = tiles_m * (xm + 1) / gm, c0 = tiles_n * xn / gn, c1 = tiles_n * (xn + 1) / gn;
  const int w = c1 - c0;
  if (w <= 0 || local >= (r1 - r0) * w) return false;
  tm = r0 + local / w;
  tn = c0 + local - (local / w) * w;
  return true;
}

// AKC / BKC: the operand is k-contiguous (true) or row-contiguous (false)
// ARAG: a row-contiguous A whose extent M is no multiple of 4 or whose rows are not 16-byte aligned (the output layer's
// dZ^T: M = 10) -- its image is fetched a dword per lane (16 DMA instructions per chunk instead of 4), every lane beyond
// M or beyond the wave's k taking zeros
// WT: the tile leaves through write-through (sc0 sc1) stores -- for a consumer in the same launch (gemm_t32_head_kernel).
// Returns the tile's row block, or -1 for a workgroup without a tile.
// The XOR key of image row x in a k-contiguous image (128-byte rows of eight 16-byte quads).  ds_read_b128 is served in four
// groups of sixteen lanes over a 256-byte bank row (MI355X_MICROARCH.md, LDS: {0-3,12-15,20-27}, {4-11,16-19,28-31} and the
// same + 32): a group reads eight even and eight odd image rows, and an even row can only reach the lower eight slots of the
// bank row -- the key has to differ across the eight rows of one parity in a group.  (x >> 1) & 7 does for all four groups;
// x & 7 (the first version) gives each group's even rows four keys: every slot hit twice, 8 LDS cycles a read instead of 4.
__device__ __forceinline__ int t32_key(int x) { return (x >> 1) & 7; }

template <bool AKC, bool BKC, bool ARAG = false, bool WT = false>
__device__ __forceinline__ int gemm_t32_body(const T32Args& g, const int bid, float* smem) {
  static_assert(!ARAG || !AKC, "the dword form is for a row-contiguous A");
  constexpr int NA = ARAG ? 16 : 4, PER = NA + 4;   // DMA instructions per chunk
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  unsigned long long* const dbg = (g.dbg && tid == 0 && (bid == 0 || bid == (int)gridDim.x - 1)) ? g.dbg + (bid == 0 ? 0 : 8) : nullptr;
  auto stamp = [&](int i) { if (dbg) dbg[i] = __builtin_amdgcn_s_memrealtime(); };
  int tile_m, tile_n;
  if (!t32_tile_of(g.tiles_m, g.tiles_n, g.gm, g.gn, bid, tile_m, tile_n)) return -1;
  stamp(0);
  const int m0 = tile_m * 32, n0 = tile_n * 32;

  // the epilogue's operands do not depend on the product: their loads go out first
  const int erow = tid >> 3, ec4 = (tid & 7) * 4;
  const long grow = m0 + erow, gcol = n0 + ec4;
  const bool evalid = grow < g.M && gcol < g.N;         // (N % 4 == 0: a quad is in or out)
  f32x4 pf_ci = {0.f, 0.f, 0.f, 0.f}, pf_hd = {0.f, 0.f, 0.f, 0.f}, pf_bias = {0.f, 0.f, 0.f, 0.f};
  if (evalid) {
    if (g.Cin) pf_ci = *reinterpret_cast<const f32x4*>(g.Cin + grow * g.c_sm + gcol);
    if (g.dact) pf_hd = *reinterpret_cast<const f32x4*>(g.dact + grow * g.c_sm + gcol);
    if (g.bias) pf_bias = *reinterpret_cast<const f32x4*>(g.bias + gcol);
  }

  // this wave's run of k, in units of 16
  const int U = (g.K + 15) / 16;
  const int k0 = (int)((long)U * wave / T32_NW) * 16;
  int k1 = (int)((long)U * (wave + 1) / T32_NW) * 16;
  if (k1 > g.K) k1 = g.K;
  const int nC = k1 > k0 ? (k1 - k0 + T32_BK - 1) / T32_BK : 0;

  typedef __attribute__((address_space(3))) void* lptr_t;
  float* wsm = smem + wave * T32_WAVE;
  const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lptr_t)wsm);

  // per-lane byte offsets of the four 1-KiB pieces of an operand's image, relative to the chunk's scalar base
  // KC: piece p, lane l -> image row x = 8 p + l / 8, slot l % 8 holds quad (l % 8) ^ t32_key(x) of that row
  // XC: piece p, lane l -> image row k = 8 p + l / 8, quad l % 8 of the 32 x
  const int lr = lane >> 3, ls = lane & 7;
  unsigned oa[4], ob[4];
  bool xa_ok = true, xb_ok = true;   // XC: this lane's quad of x lies inside the extent
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if constexpr (AKC) {
      long x = m0 + 8 * p + lr;
      if (x >= g.M) x = g.M - 1;
      oa[p] = (unsigned)((x * g.a_sx + 4 * (ls ^ t32_key(8 * p + lr))) * 4);
    } else {
      oa[p] = (unsigned)(((long)(8 * p + lr) * g.a_sx + m0 + 4 * ls) * 4);
    }
    if constexpr (BKC) {
      long x = n0 + 8 * p + lr;
      if (x >= g.N) x = g.N - 1;
      ob[p] = (unsigned)((x * g.b_sx + 4 * (ls ^ t32_key(8 * p + lr))) * 4);
    } else {
      ob[p] = (unsigned)(((long)(8 * p + lr) * g.b_sx + n0 + 4 * ls) * 4);
    }
  }
  if constexpr (!AKC) xa_ok = m0 + 4 * ls + 3 < g.M;
  if constexpr (!BKC) xb_ok = n0 + 4 * ls + 3 < g.N;
  const bool interior = !ARAG && (AKC || m0 + 32 <= g.M) && (BKC || n0 + 32 <= g.N);   // (uniform)
  const long step_a = AKC ? T32_BK * 4L : T32_BK * g.a_sx * 4L, step_b = BKC ? T32_BK * 4L : T32_BK * g.b_sx * 4L;   // bytes per chunk
  const char* sa = reinterpret_cast<const char*>(g.A) + (AKC ? (long)k0 * 4 : (long)k0 * g.a_sx * 4);
  const char* sb = reinterpret_cast<const char*>(g.B) + (BKC ? (long)k0 * 4 : (long)k0 * g.b_sx * 4);

#define T32_DMA(OFF, BASE) asm volatile("global_load_lds_dwordx4 %0, %1 offset:0" ::"v"(OFF), "s"(BASE) : "memory")
#define T32_DMA_V(PTR) asm volatile("global_load_lds_dwordx4 %0, off offset:0" ::"v"(PTR) : "memory")
  // (scalars the lambda needs, by value: capturing `g` by reference would put the argument block in scratch)
  const long a_sx_l = g.a_sx;
  const int M_l = g.M;
  // the 8 DMA instructions of chunk c into stage c % NS
  auto issue = [&, a_sx_l, M_l](int c) {
    const int seq = c;
    const int kc = k0 + c * T32_BK;
    const unsigned st = lds_w + (unsigned)(seq % T32_NS) * (T32_STAGE * 4);
    const char* ba = sa + (long)c * step_a;
    const char* bb = sb + (long)c * step_b;
    const bool full = kc + T32_BK <= k1;
    if (full && interior) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(st + p * 1024) : "memory");
        T32_DMA(oa[p], ba);
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(st + 4096 + p * 1024) : "memory");
        T32_DMA(ob[p], bb);
      }
    } else {
      // a ragged last chunk or an edge tile: every lane still issues every instruction (vmcnt counts instructions), with a
      // 64-bit address of its own -- its source, or the zeros
      const char* zero = reinterpret_cast<const char*>(g_t32_zero);
      if constexpr (ARAG) {
        // instruction e fills image rows k = 2 e, 2 e + 1: lane l -> (k = 2 e + l / 32, x = l % 32)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int kl = 2 * e + half;
          const bool ok = kc + kl < k1 && m0 + l31 < M_l;
          const char* src = ok ? ba + ((long)kl * a_sx_l + m0 + l31) * 4 : zero;
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(st + e * 256) : "memory");
          asm volatile("global_load_lds_dword %0, off offset:0" ::"v"(src) : "memory");
        }
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const bool ok = AKC ? (kc + 4 * (ls ^ t32_key(8 * p + lr)) < k1) : (kc + 8 * p + lr < k1 && xa_ok);
          const char* src = ok ? ba + oa[p] : zero;
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(st + p * 1024) : "memory");
          T32_DMA_V(src);
        }
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const bool ok = BKC ? (kc + 4 * (ls ^ t32_key(8 * p + lr)) < k1) : (kc + 8 * p + lr < k1 && xb_ok);
        const char* src = ok ? bb + ob[p] : zero;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(st + 4096 + p * 1024) : "memory");
        T32_DMA_V(src);
      }
    }
  };
#undef T32_DMA
#undef T32_DMA_V

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float asum = 0.f;                      // sum over k of this lane's A elements (row sums of A: bias gradients)
  const bool want_rs = g.rowsum != nullptr && tile_n == 0;

  const int PD = g.pd;
  const int pro = nC < PD ? nC : PD;
  for (int c = 0; c < pro; ++c) issue(c);
  stamp(1);

  // chunk c has landed when at most the DMA groups of the chunks issued after it are outstanding
  auto wait_landed = [&](int c) {
    const int newer = (nC - 1 - c) < (PD - 1) ? (nC - 1 - c) : (PD - 1);
    static_assert(3 * PER <= 63, "vmcnt is six bits");
    if (newer >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");
    else if (newer == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // a half chunk (<= 16 k left of the wave's run) needs the first eight MFMA steps only
  auto steps_of = [&](int c) { return (k1 - (k0 + c * T32_BK)) > 16 ? 4 : 2; };
  auto read_frags = [&](int c, float (&fa)[16], float (&fb)[16]) {
    const float* sp = wsm + (c % T32_NS) * T32_STAGE;
    const int nI = steps_of(c);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < nI) {
        if constexpr (AKC) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(sp + l31 * 32 + 4 * ((2 * i + half) ^ t32_key(l31)));
          fa[4 * i] = v.x; fa[4 * i + 1] = v.y; fa[4 * i + 2] = v.z; fa[4 * i + 3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) fa[4 * i + j] = sp[(4 * (2 * i + half) + j) * 32 + l31];
        }
        if constexpr (BKC) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(sp + 1024 + l31 * 32 + 4 * ((2 * i + half) ^ t32_key(l31)));
          fb[4 * i] = v.x; fb[4 * i + 1] = v.y; fb[4 * i + 2] = v.z; fb[4 * i + 3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) fb[4 * i + j] = sp[1024 + (4 * (2 * i + half) + j) * 32 + l31];
        }
      }
    }
  };
  auto mma = [&](int c, const float (&fa)[16], const float (&fb)[16]) {
    const int nI = steps_of(c);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < nI) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[4 * i + j], fb[4 * i + j], acc, 0, 0, 0);
          if (want_rs) asum += fa[4 * i + j];
        }
      }
    }
  };
  // One chunk: wait for its DMA, take its fragments, hand its stage to the chunk PD ahead, sixteen dependent MFMAs.
  // (Measured, config 3's forward layer, 256 workgroups, stamps of workgroup 0 -- profiles/README.md round 5: the K loop
  // runs at 1,630 shader cycles a chunk (2.3 GHz) for the 1,024 of its MFMAs -- wait, fragment reads, DMA issue and the MFMA
  // chain in series; the next chunk's fragments under the MFMAs, two alternating accumulators, and one other instruction
  // behind each MFMA were built and measured: no change, no change, slower -- and MORE chunks in flight make it
  // slower -- four per wave put 32 MB of requests on the eight L2s in the launch's first microsecond and the first
  // chunk lands after 2.1 us instead of 1.2.  Two in flight is the default.)
  float fa[16], fb[16];
  for (int c = 0; c < nC; ++c) {
    wait_landed(c);
    if (c == 0) stamp(2);
    read_frags(c, fa, fb);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (c + PD < nC) issue(c + PD);
    __builtin_amdgcn_sched_barrier(0);
    mma(c, fa, fb);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stamp(3);

  // ---- the four partial tiles meet in LDS (each wave in its own stage memory: nothing of another wave is overwritten) ----
  // accumulator register r of lane (l31, half): row (r & 3) + 8 (r >> 2) + 4 half, column l31
#pragma unroll
  for (int r = 0; r < 16; ++r) wsm[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[r];
  if (want_rs) wsm[1024 + lane] = asum;
  __syncthreads();
  stamp(4);
  f32x4 s = *reinterpret_cast<const f32x4*>(smem + erow * 32 + ec4);
#pragma unroll
  for (int w = 1; w < T32_NW; ++w) s += *reinterpret_cast<const f32x4*>(smem + w * T32_WAVE + erow * 32 + ec4);
  if (evalid) {
    f32x4 v = s * g.alpha;
    if (g.Cin) v += pf_ci * g.beta;
    if (g.bias) v += pf_bias;
    if (g.act == 1) {
      v.x = 1.f / (1.f + __expf(-v.x)); v.y = 1.f / (1.f + __expf(-v.y));
      v.z = 1.f / (1.f + __expf(-v.z)); v.w = 1.f / (1.f + __expf(-v.w));
    } else if (g.act == 2) {
      v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w);
    }
    if (g.dact) {
      if (g.dact_kind == 0) v *= pf_hd * (1.f - pf_hd);
      else v *= 1.f - pf_hd * pf_hd;
    }
    if constexpr (WT) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
      float* dst = g.C + grow * g.c_sm + gcol;
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(u) : "memory");
    } else {
      *reinterpret_cast<f32x4*>(g.C + grow * g.c_sm + gcol) = v;
    }
  }
  if (want_rs && tid < 32) {
    const long m = m0 + tid;
    if (m < g.M) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < T32_NW; ++w) t += smem[w * T32_WAVE + 1024 + tid] + smem[w * T32_WAVE + 1024 + 32 + tid];
      if (g.rowsum_acc) g.rowsum[m] = (g.rowsum_in ? g.rowsum_in[m] : g.rowsum[m]) + g.rowsum_alpha * t;
      else g.rowsum[m] = t;
    }
  }
  stamp(5);
  return tile_m;
}

// ---- forward layer + loss head in ONE launch --------------------------------------------------------------------------
// `H = logistic(X W1^T + b1)` on the body above (256 output columns: eight tiles a row block), and behind it what the step's
// second launch did (gemm_small's 16x16 body with the loss head and the tail: 6.4 us behind a ~1.5 us boundary):
// `z = H W2^T + b2`, the loss head on each row (softmax >>> crossEntropy backward: softmax(z) sum(t) - t; or logistic >>>
// squaredError backward) and the previous layer's cotangent `dZ1 = (dz W2) (.) h (1 - h)` for the same rows.
// EVERY workgroup of a row block takes part: its tile of H goes out write-through, it bumps the block's arrival counter,
// waits until all eight tiles have arrived, and then does FOUR of the block's 32 rows -- a wave a row, a lane four
// columns: one system-scope float4 of H per lane, W2's ten rows in registers since before the wait, 40 FMAs, a wave
// reduction, the loss head on lanes 0..N2-1 (its result is wave-uniform: read back lane by lane), 40 FMAs of tail, one
// float4 store.  The two forms that let nobody wait were built and measured first (profiles/README.md round 5): round 4's
// seam on 16-wave tiles (-1.2 us: a 5.3 us head chain behind the slowest tile of a 2-3 us ramp) and "the last arriver
// does the block's 32 rows" on this kernel's four waves (54 us: 164 k MACs and thirty transcendentals a row are a long
// program for one wave per SIMD).  Waiting is safe because the launch is never larger than one round: 256 workgroups of
// 64 KiB LDS, every one resident from the start, so whoever is waited for is running.  A wait that does not end within
// two seconds (a device whose CUs are masked below the grid) gives up: the outputs of that launch are invalid, the
// host is told at the next synchronisation (TO_ERR_HIP) and the joined form stays off for the rest of the process.
// Hand-over: write-through stores and system-scope loads -- correct on whatever XCDs the eight workgroups run.
struct T32HeadArgs {
  const float* W2;       // [N2][256], rows k-contiguous, stride w_sn: the head's right operand and the tail's W
  const float* bias2;    // [N2] or null
  const float* target;   // [M][N2], row stride dz_sm
  float* dz;             // [M][N2], row stride dz_sm
  float* loss_out;       // [M] or null
  float* tail_out;       // [M][256] or null
  unsigned* ctr;         // [2][T32_CTR]: arrivals, departures (both 0 between launches: the last to leave resets them)
  int* status;           // host-mapped: nonzero = a wait gave up (row block + 1)
  long w_sn, dz_sm;
  int N2, loss_rows;
  float alpha2;
};

__global__ __launch_bounds__(256) void gemm_t32_head_kernel(T32Args g, T32HeadArgs h) {
  extern __shared__ __attribute__((aligned(1024))) float t32_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned long long t_entry = g.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
  int tile_m, tile_n;
  if (!t32_tile_of(g.tiles_m, g.tiles_n, g.gm, g.gn, (int)blockIdx.x, tile_m, tile_n)) return;
  // what the head needs besides H does not depend on this launch: W2's rows (this lane's four columns), the bias, the
  // target of lanes 0..N2-1 -- their loads go out ahead of everything
  const long grow = (long)tile_m * 32 + 4 * tile_n + wave;   // the row this wave finishes
  const bool rv = grow < g.M;
  f32x4 w4[16];
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    w4[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (n < h.N2) w4[n] = *reinterpret_cast<const f32x4*>(h.W2 + (long)n * h.w_sn + 4 * lane);
  }
  const float b_l = (lane < h.N2 && h.bias2) ? h.bias2[lane] : 0.f;
  const float t_l = (lane < h.N2 && rv) ? h.target[grow * h.dz_sm + lane] : 0.f;

  gemm_t32_body<true, true, false, true>(g, (int)blockIdx.x, t32_smem);

  unsigned long long* const hd = (g.dbg && tid == 0 && blockIdx.x == 0) ? g.dbg + 8 : nullptr;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's stores have left for memory (every wave's, behind the barrier)
  __syncthreads();
  __shared__ int gave_up;
  if (tid == 0) {
    unsigned* arrive = h.ctr + tile_m;
    __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (hd) { hd[0] = t_entry; hd[1] = __builtin_amdgcn_s_memrealtime(); hd[7] = 1; }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int bad = 0;
    while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)g.tiles_n) {
      __builtin_amdgcn_s_sleep(2);
      if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {   // two seconds of the 100 MHz clock
        bad = 1;
        break;
      }
    }
    gave_up = bad;
    if (bad && h.status) *h.status = tile_m + 1;
  }
  __syncthreads();
  if (gave_up) return;
  if (hd) hd[2] = __builtin_amdgcn_s_memrealtime();

  // ---- the wave's row: lane l holds columns 4 l .. 4 l + 3 ----
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  f32x4 h4 = {0.f, 0.f, 0.f, 0.f};
  if (rv) {
    u32x4 u;
    const float* src = g.C + grow * g.c_sm + 4 * lane;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(u) : "v"(src) : "memory");
    h4 = f32x4{__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
  }
  float z[16];
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    z[n] = 0.f;
    if (n < h.N2) {
      z[n] = h4.x * w4[n].x + h4.y * w4[n].y + h4.z * w4[n].z + h4.w * w4[n].w;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) z[n] += __shfl_xor(z[n], off, 64);
    }
  }
  // lane n < N2 finishes output n; the sums over the row's outputs are sixteen-lane butterflies
  float v = -INFINITY;
#pragma unroll
  for (int n = 0; n < 16; ++n)
    if (n < h.N2 && lane == n) v = z[n] * h.alpha2 + b_l;
  auto sum16 = [](float x) {
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
  };
  const bool lv = lane < h.N2;
  float out_l, loss_l;
  if (h.loss_rows == 1) {
    float mx = v;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const float e = lv ? expf(v - mx) : 0.f;
    const float se = sum16(e), sy = sum16(t_l);
    const float pr = e / se;
    out_l = lv ? pr * sy - t_l : 0.f;
    loss_l = lv ? -t_l * logf(pr) : 0.f;
  } else {
    const float sg = 1.f / (1.f + expf(-v));
    const float e = t_l - sg;
    out_l = lv ? -2.f * e * sg * (1.f - sg) : 0.f;
    loss_l = lv ? e * e : 0.f;
  }
  if (rv) {
    if (lv) h.dz[grow * h.dz_sm + lane] = out_l;
    if (h.loss_out) {
      const float l = sum16(loss_l);
      if (lane == 0) h.loss_out[grow] = l;
    }
  }
  if (hd) hd[3] = __builtin_amdgcn_s_memrealtime();
  if (h.tail_out) {
    // the tail: dZ1[row][j] = (sum_n dz[n] W2[n][j]) h (1 - h); dz[n] sits in lane n
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 16; ++n)
      if (n < h.N2) {
        const float d = __shfl(out_l, n, 64);
        a += w4[n] * d;
      }
    if (rv) *reinterpret_cast<f32x4*>(h.tail_out + grow * 256 + 4 * lane) = a * h4 * (1.f - h4);
  }
  if (hd) hd[4] = __builtin_amdgcn_s_memrealtime();
  // leave: the last of the block's eight workgroups to get here resets both counters for the next launch
  __syncthreads();
  if (tid == 0) {
    unsigned* leave = h.ctr + T32_CTR + tile_m;
    const unsigned seen = __hip_atomic_fetch_add(leave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seen == (unsigned)(g.tiles_n - 1)) {
      __hip_atomic_store(h.ctr + tile_m, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(leave, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_t32_kernel(T32Args g) {
  extern __shared__ __attribute__((aligned(1024))) float t32_smem[];
  gemm_t32_body<AKC, BKC>(g, (int)blockIdx.x, t32_smem);
}

// Two weight-gradient contractions (dZ^T . A: both operands row-contiguous, K = the batch) in ONE launch: the first
// problem's tiles, then the second's -- a narrow output layer's dZ^T (M = 10) through the dword form.
template <bool ARAG2>
__global__ __launch_bounds__(256) void gemm_t32_pair_kernel(T32Args g1, T32Args g2, int n1) {
  extern __shared__ __attribute__((aligned(1024))) float t32_smem[];
  if ((int)blockIdx.x < n1) gemm_t32_body<false, false, false>(g1, (int)blockIdx.x, t32_smem);
  else gemm_t32_body<false, false, ARAG2>(g2, (int)blockIdx.x - n1, t32_smem);
}

// ---- host side ------------------------------------------------------------------------------------------------------
// ragged_a: out: A is row-contiguous and needs the dword form (only looked for when allow_ragged_a)
static bool t32_fill(const GemmProblem& p, T32Args& g, bool& akc, bool& bkc, bool allow_ragged_a = false, bool* ragged_a = nullptr) {
  if (p.dtype != TO_F32 || p.batch != 1 || p.reduce_batch) return false;
  if (p.loss_rows || p.tail_out || p.loss_out) return false;
  akc = p.a_sk == 1 && p.K > 1;
  bkc = p.b_sk == 1 && p.K > 1;
  if (!akc && p.a_sm != 1) return false;
  if (!bkc && p.b_sn != 1) return false;
  if (p.K % 4 != 0 || p.N % 4 != 0) return false;            // quads along k; whole float4 stores
  const bool rag = !akc && (p.M % 4 != 0 || p.a_sk % 4 != 0 || (reinterpret_cast<uintptr_t>(p.A) & 15u));
  if (ragged_a) *ragged_a = rag;
  if (rag && !allow_ragged_a) return false;                   // quads along x of a row-contiguous operand
  if (akc && p.a_sm % 4 != 0) return false;                   // 16-byte aligned quads
  if (bkc && p.b_sn % 4 != 0) return false;
  if (!bkc && p.b_sk % 4 != 0) return false;
  if ((reinterpret_cast<uintptr_t>(p.A) & 3u) || ((akc || !rag) && (reinterpret_cast<uintptr_t>(p.A) & 15u))) return false;
  if ((reinterpret_cast<uintptr_t>(p.B) | reinterpret_cast<uintptr_t>(p.C)) & 15u) return false;
  if (p.c_sm % 4 != 0) return false;
  if (p.Cin && (reinterpret_cast<uintptr_t>(p.Cin) & 15u)) return false;
  if (p.dact && (reinterpret_cast<uintptr_t>(p.dact) & 15u)) return false;
  if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15u)) return false;
  if (p.beta != 0.0 && !p.Cin) return false;
  if (p.act > 2) return false;
  // byte offsets are 32-bit
  if ((p.M + 32) * std::max<int64_t>(p.a_sm, 1) * 4 + p.K * std::max<int64_t>(p.a_sk, 1) * 4 >= (1LL << 31)) return false;
  if ((p.N + 32) * std::max<int64_t>(p.b_sn, 1) * 4 + p.K * std::max<int64_t>(p.b_sk, 1) * 4 >= (1LL << 31)) return false;
  g.A = (const float*)p.A; g.B = (const float*)p.B; g.C = (float*)p.C;
  g.Cin = p.beta != 0.0 ? (const float*)p.Cin : nullptr;
  g.M = (int)p.M; g.N = (int)p.N; g.K = (int)p.K;
  g.a_sx = akc ? p.a_sm : p.a_sk;
  g.b_sx = bkc ? p.b_sn : p.b_sk;
  g.c_sm = p.c_sm;
  g.tiles_m = (int)((p.M + 31) / 32); g.tiles_n = (int)((p.N + 31) / 32);
  {   // the XCD grid that pulls the fewest panels into each L2
    int best = 1 << 30;
    for (int gm : {8, 4, 2, 1}) {
      const int gn = 8 / gm;
      const int cost = (g.tiles_m + gm - 1) / gm + (g.tiles_n + gn - 1) / gn;
      if (cost < best) { best = cost; g.gm = gm; g.gn = gn; }
    }
  }
  g.alpha = (float)p.alpha; g.beta = (float)p.beta;
  g.bias = (const float*)p.bias; g.dact = (const float*)p.dact; g.act = p.act; g.dact_kind = p.dact_kind;
  g.rowsum = (float*)p.rowsum; g.rowsum_in = (const float*)p.rowsum_in; g.rowsum_alpha = (float)p.rowsum_alpha;
  g.rowsum_acc = p.rowsum_acc ? 1 : 0;
  static const int pd = [] { const char* e = ab_getenv("TOPS_T32_PD"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > T32_NS ? T32_NS : v); }();
  g.pd = pd;
  static unsigned long long* dbg = [] {
    unsigned long long* p = nullptr;
    if (ab_getenv("TOPS_T32_STAMPS") && hipHostMalloc(reinterpret_cast<void**>(&p), 16 * sizeof(unsigned long long), hipHostMallocMapped) == hipSuccess) {
      for (int i = 0; i < 16; ++i) p[i] = 0;
      static unsigned long long* keep = p;
      atexit([] {
        if (keep[15]) {   // the joined launch: slots 8..15 belong to the head of row block 0
          const unsigned long long* k = keep + 8;
          std::fprintf(stderr, "[t32] joined launch, workgroup 0, us since it began: its tile stored and arrived %.2f, all eight tiles of the row block "
                               "there %.2f, loss head of its four rows done %.2f, tail stored %.2f (entered %.2f us after stamp 0)\n",
                       (k[1] - k[0]) * 0.01, (k[2] - k[0]) * 0.01, (k[3] - k[0]) * 0.01, (k[4] - k[0]) * 0.01, ((double)k[0] - (double)keep[0]) * 0.01);
        }
        for (int w = 0; w < (keep[15] ? 1 : 2); ++w) {
          const unsigned long long* k = keep + 8 * w;
          std::fprintf(stderr, "[t32] %s workgroup of the last launch, us since it began: prologue DMA issued %.2f, first chunk landed %.2f, K loop done %.2f, "
                               "partials met %.2f, done %.2f; began %.2f us after workgroup 0\n", w ? "last" : "first",
                       (k[1] - k[0]) * 0.01, (k[2] - k[0]) * 0.01, (k[3] - k[0]) * 0.01, (k[4] - k[0]) * 0.01, (k[5] - k[0]) * 0.01,
                       ((double)k[0] - (double)keep[0]) * 0.01);
        }
      });
    }
    return p;
  }();
  g.dbg = dbg;
  return true;
}

// The shapes this design is for: about one round of 32x32 tiles on the 256 CUs and a K long enough for the stream to
// matter.  (More tiles: the 64x64 wave-split kernel takes over; fewer or a short K: gemm_small's bodies.)
bool gemm_t32_applicable(const GemmProblem& p) {
  static const int enable = [] { const char* e = ab_getenv("TOPS_GEMM_T32"); return e ? atoi(e) : 1; }();
  if (!enable) return false;
  T32Args g;
  bool akc, bkc;
  if (!t32_fill(p, g, akc, bkc)) return false;
  const long tiles = (long)g.tiles_m * g.tiles_n;
  static const long max_tiles = [] { const char* e = ab_getenv("TOPS_T32_MAXTILES"); return e ? atol(e) : 512L; }();
  return tiles >= 96 && tiles <= max_tiles && p.K >= 256 && p.K <= 8192;
}

// Both problems of a weight-gradient pair on this design: together about one round of tiles, the same K.
bool launch_gemm_t32_pair(const GemmProblem& p1, const GemmProblem& p2, hipStream_t s);

// workgroups of a launch: eight times the largest rectangle of the XCD grid
static long t32_grid(const T32Args& g) {
  const long rm = (g.tiles_m + g.gm - 1) / g.gm, rn = (g.tiles_n + g.gn - 1) / g.gn;
  return 8 * rm * rn;
}

static void t32_attr(const void* k) { TO_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, T32_NW * T32_WAVE * 4)); }

void launch_gemm_t32(const GemmProblem& p, hipStream_t s) {
  T32Args g;
  bool akc, bkc;
  TO_CHECK(t32_fill(p, g, akc, bkc), TO_ERR_STATE, "launch_gemm_t32: not applicable");
  static bool attr = false;
  if (!attr) {
    t32_attr(reinterpret_cast<const void*>(gemm_t32_kernel<true, true>));
    t32_attr(reinterpret_cast<const void*>(gemm_t32_kernel<true, false>));
    t32_attr(reinterpret_cast<const void*>(gemm_t32_kernel<false, true>));
    t32_attr(reinterpret_cast<const void*>(gemm_t32_kernel<false, false>));
    attr = true;
  }
  const dim3 grid((unsigned)t32_grid(g)), block(256);
  const size_t lds = (size_t)T32_NW * T32_WAVE * 4;
  if (akc && bkc) launch_k(gemm_t32_kernel<true, true>, grid, block, lds, s, g);
  else if (akc) launch_k(gemm_t32_kernel<true, false>, grid, block, lds, s, g);
  else if (bkc) launch_k(gemm_t32_kernel<false, true>, grid, block, lds, s, g);
  else launch_k(gemm_t32_kernel<false, false>, grid, block, lds, s, g);
  TO_HIP(hipGetLastError());
  count_launch();
}

// the row-block counters of gemm_t32_head_kernel: allocated and zeroed once, at to_init (never inside a stream capture)
static unsigned* g_t32_ctr = nullptr;
static int *g_t32_status = nullptr, *g_t32_status_dev = nullptr;
static bool g_t32_head_off = false;
void gemm_t32_init() {
  if (g_t32_ctr) return;
  if (hipMalloc(&g_t32_ctr, 2 * T32_CTR * sizeof(unsigned)) != hipSuccess || hipMemset(g_t32_ctr, 0, 2 * T32_CTR * sizeof(unsigned)) != hipSuccess) {
    (void)hipGetLastError();
    g_t32_ctr = nullptr;
    return;
  }
  if (hipHostMalloc(&g_t32_status, sizeof(int), hipHostMallocMapped) == hipSuccess) {
    *g_t32_status = 0;
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&g_t32_status_dev), g_t32_status, 0) != hipSuccess) g_t32_status_dev = nullptr;
  }
  (void)hipGetLastError();
}
// nonzero ONCE after a joined launch gave up waiting (row block + 1): its outputs are invalid; the joined form is off for
// the rest of the process and the counters are cleared (checked by to_sync and by every flush of recorded ops)
int gemm_t32_take_failure() {
  const int st = g_t32_status ? *g_t32_status : 0;
  if (st) {
    g_t32_head_off = true;
    *g_t32_status = 0;
    if (g_t32_ctr) (void)hipMemset(g_t32_ctr, 0, 2 * T32_CTR * sizeof(unsigned));
  }
  return st;
}

// A forward layer and the loss-head launch that reads its output, as one launch (gemm_t32_head_kernel).  Returns false --
// nothing launched -- unless the pair has the form that kernel is written for.
bool launch_gemm_t32_head(const GemmProblem& pf, const GemmProblem& ph, hipStream_t s) {
  // OFF by default, like round 4's seam forms (TOPS_STEP_SEAM=1|2); TOPS_STEP_SEAM=3 runs it.  Measured (config 3, rocprofv3,
  // 411 launches; stamps of workgroup 0 in profiles/README.md round 5): 22.0 us against 9.6 + 6.4 us for the two launches it
  // replaces and the ~1.5 us boundary between them.  The tile is stored and has arrived at 7.6 us (write-through stores
  // drain for 1.5 us), the last of the block's eight tiles at 11.6, and the four rows' head -- one system-scope float4 of H
  // from memory, sixty cross-lane steps of reduction, the transcendentals -- takes 5.8 us more: the same dependent chain
  // the separate launch runs in 6.4 us INCLUDING its launch, now behind the slowest tile instead of beside the next
  // launch's ramp.  The same verdict as round 4's seam, for the same reason.
  static const int enable = [] { const char* e = getenv("TOPS_STEP_SEAM"); return e ? atoi(e) : 0; }();
  if (enable != 3 || !g_t32_ctr || !g_t32_status_dev || g_t32_head_off) return false;
  T32Args g;
  bool akc, bkc;
  if (!t32_fill(pf, g, akc, bkc) || !akc || !bkc) return false;
  const long tiles = (long)g.tiles_m * g.tiles_n;
  if (tiles < 96 || tiles > 512 || pf.K < 256 || pf.K > 8192 || g.tiles_m > T32_CTR) return false;
  if (pf.act != 1 || pf.dact || pf.rowsum || pf.beta != 0.0) return false;
  if (pf.N != 256 || pf.c_sm != pf.N) return false;   // eight tiles a row block: four rows a workgroup, a lane four columns
  if (ph.dtype != TO_F32 || ph.batch != 1 || ph.reduce_batch) return false;
  if (ph.A != pf.C || ph.M != pf.M || ph.K != pf.N || ph.a_sk != 1 || ph.a_sm != pf.c_sm) return false;
  if (ph.b_sk != 1 || ph.b_sn % 4 != 0 || (reinterpret_cast<uintptr_t>(ph.B) & 15u)) return false;
  if (ph.N < 1 || ph.N > 16 || (ph.loss_rows != 1 && ph.loss_rows != 2) || !ph.target) return false;
  if (ph.beta != 0.0 || ph.dact || ph.act != 0 || ph.rowsum) return false;
  if (ph.tail_out) {
    if (ph.tail_n != pf.N || ph.tail_h != pf.C || ph.tail_w != ph.B || ph.b_sn != ph.tail_n) return false;
    if (reinterpret_cast<uintptr_t>(ph.tail_out) & 15u) return false;
  }
  g.gm = 8; g.gn = 1;   // a row block's eight tiles on one XCD (for speed; the hand-over does not depend on it)
  if (g.tiles_m < 8) return false;
  // every workgroup must be resident from the start (they wait for each other): one round of the chip at most
  {
    static const int cus = [] { hipDeviceProp_t pr; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 0; }();
    if (t32_grid(g) > cus) return false;
  }
  T32HeadArgs h{};
  h.W2 = (const float*)ph.B; h.bias2 = (const float*)ph.bias; h.target = (const float*)ph.target;
  h.dz = (float*)ph.C; h.loss_out = (float*)ph.loss_out; h.tail_out = (float*)ph.tail_out;
  h.ctr = g_t32_ctr;
  h.status = g_t32_status_dev;
  h.w_sn = ph.b_sn; h.dz_sm = ph.c_sm;
  h.N2 = (int)ph.N; h.loss_rows = ph.loss_rows; h.alpha2 = (float)ph.alpha;

  static bool attr = false;
  if (!attr) {
    TO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_t32_head_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, T32_NW * T32_WAVE * 4));
    attr = true;
  }
  launch_k(gemm_t32_head_kernel, dim3((unsigned)t32_grid(g)), dim3(256), (size_t)T32_NW * T32_WAVE * 4, s, g, h);
  TO_HIP(hipGetLastError());
  count_launch();
  return true;
}

bool launch_gemm_t32_pair(const GemmProblem& p1, const GemmProblem& p2, hipStream_t s) {
  // OFF in product builds (measured, config 3, rocprofv3 over 411 launches: 13.4-14.3 us against 10.2 us for gemm_small's
  // pair launch; profiles/README.md round 5).  Both operands of dZ^T . X are row-contiguous, which is the one case where
  // gemm_small's one-shot register loads already fetch whole 128-byte lines, and it has ALL 256 KB of a tile in flight at
  // once (the register file is the landing buffer: 512 KB a CU) where four 8 KiB LDS stages a wave hold half of it: a
  // second ~3 us round trip through an L2 that 200 workgroups hit at the same instant.  TOPS_GEMM_T32_PAIR=1 in a
  // development build runs it; the body it shares with the forward kernel is what tests/test_gpu_fuzz_gemm.py covers
  // (tools/t32_check.py: all four operand layouts), and with the knob set tests/test_gpu_full_size.py walks the pair itself.
  static const int enable = [] { const char* e = ab_getenv("TOPS_GEMM_T32_PAIR"); return e ? atoi(e) : 0; }();
  if (!enable) return false;
  T32Args g1, g2;
  bool akc1, bkc1, akc2, bkc2, rag2 = false;
  if (!t32_fill(p1, g1, akc1, bkc1) || akc1 || bkc1) return false;
  if (!t32_fill(p2, g2, akc2, bkc2, true, &rag2) || akc2 || bkc2) return false;
  const long n1 = (long)g1.tiles_m * g1.tiles_n, n2 = (long)g2.tiles_m * g2.tiles_n;
  if (n1 < 96 || n1 + n2 > 512 || p1.K < 256 || p1.K > 8192 || p2.K < 256 || p2.K > 8192 || n2 > n1) return false;
  static bool attr = false;
  if (!attr) {
    t32_attr(reinterpret_cast<const void*>(gemm_t32_pair_kernel<false>));
    t32_attr(reinterpret_cast<const void*>(gemm_t32_pair_kernel<true>));
    attr = true;
  }
  const long w1 = t32_grid(g1), w2 = t32_grid(g2);   // workgroups (whole multiples of 8: the second problem's blocks keep b % 8)
  const dim3 grid((unsigned)(w1 + w2)), block(256);
  const size_t lds = (size_t)T32_NW * T32_WAVE * 4;
  if (rag2) launch_k(gemm_t32_pair_kernel<true>, grid, block, lds, s, g1, g2, (int)w1);
  else launch_k(gemm_t32_pair_kernel<false>, grid, block, lds, s, g1, g2, (int)w1);
  TO_HIP(hipGetLastError());
  count_launch();
  return true;
}

}  // namespace to
