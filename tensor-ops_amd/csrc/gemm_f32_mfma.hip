// fp32 GEMM on the gfx950 matrix cores: C = alpha * A.B + beta * Cin
//
// Serves `gmul` (src/TensorOps/Types.hs:60-66) in its flat-GEMM formulation
// (SURVEY.md Appendix A), `gemm`/`gemv`/`ger` of `class BLAS`
// (src/TensorOps/BLAS.hs:108-123) and the batch-reduced weight gradients
// dW = sum_b dz_b (x) x_b.
//
// Design (CDNA4, wave64):
//  * v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), 64 cycles per
//    instruction per SIMD, one VGPR per operand per lane:
//      A operand lane l = A[i = l&31][k = l>>5],  B operand lane l = B[k = l>>5][j = l&31],
//      D reg r lane l   = D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
//  * Block tile BM x BN x BK, WM x WN waves; each wave owns a (BM/WM) x (BN/WN)
//    sub-tile made of 32x32 MFMA tiles held in accumulators.
//  * Operands are staged global -> registers -> LDS (register staging lets one
//    kernel serve all four transpose combinations: the LDS image is always
//    [k][m] / [k][n], so the fragment reads are conflict-free ds_read_b32 with
//    consecutive lanes on consecutive banks).  Global loads are 16-byte
//    (dwordx4) along whichever dimension is contiguous in memory.
//  * Two LDS buffers, the global loads of tile t+1 are issued before the MFMAs of
//    tile t and written to LDS after them: one barrier per k-tile.
//  * blockIdx -> tile mapping is XCD-aware: block b runs on XCD b % 8, so each
//    XCD is handed a contiguous strip of tiles that share A rows / B columns in
//    its private L2.
#include <cstdio>
#include <type_traits>
#include <vector>

#include "common.hpp"

namespace to {

#ifdef TOPS_GEMM_DEV
// Development build (seconds instead of minutes): only the kernels launched through `(launch_k)(...)` -- the pinned
// 256x256 body -- are instantiated; every other route of this file aborts.  Never defined by build.py.
template <class... A>
static void gemm_dev_skip(A&&...) {
  fprintf(stderr, "TOPS_GEMM_DEV build: this route is not compiled\n");
  abort();
}
#define launch_k(K, ...) gemm_dev_skip(__VA_ARGS__)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int I, int N, class F>
__device__ __forceinline__ void g_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    g_static_for<I + 1, N>(f);
  }
}

struct GemmKArgs {
  const float* A;
  const float* B;
  float* C;
  const float* Cin;
  int M, N, K;
  long a_sm, a_sk, b_sk, b_sn, c_sm;
  long a_sb, b_sb, c_sb;
  int nb_reduce;   // batches folded into the K loop (1 when not reducing)
  int a_mode;      // 0: k contiguous (a_sk == 1), 1: m contiguous (a_sm == 1), 2: general
  int b_mode;      // 0: n contiguous (b_sn == 1), 1: k contiguous (b_sk == 1), 2: general
  int a_vec, b_vec;  // 16-byte loads legal
  int tiles_m, tiles_n;
  float alpha, beta;
  const float* bias;  // fused epilogue, see GemmProblem
  const float* dact;
  int act;
  int dact_kind;
  int wide_store;   // plain epilogue on aligned full tiles: stage through LDS, 16-byte row stores
  int nt_store;     // nontemporal hint on those stores (streaming outputs larger than the caches)
  int ksplit;       // > 1: blockIdx.y owns k-tiles [y*t_per_split, (y+1)*t_per_split) and writes its
  int t_per_split;  //      partial product to C + y*M*N (a [ksplit][M][N] workspace, summed afterwards)
  int seg_begin, seg_end;  // ksplit == -1 (stream-K segment): this call covers k-tiles [seg_begin, seg_end) of its tile
  unsigned long long* dbg;  // development: per-workgroup timestamps (TOPS_GEMM_DBG=file), else null
};

// quad = 4 consecutive elements along the "inner" tile dimension.
// element (o, i) lives at base[o * so + i * si]; o < O, i < I are the bounds.
// kmode (guarded tiles whose K extent is whole k-tiles, uniform): rows/columns beyond M/N only feed outputs that
// are never stored, so they need no zeros -- their INDEX is clamped and the loaded value used as is.
//   1: the inner (quad) dimension is k: clamp the row, one 16-byte load when the operand allows it;
//   2: the outer dimension is k: clamp each element's index along the ragged inner dimension
//      (four scalar loads; ONE 16-byte load from a clamped start plus select chains that shift the in-range
//      elements into their slots measured slower: 1000^3 0.085 against 0.073 ms).
// 0: full guard (K tail inside the tile: out-of-range k MUST read as zero).
template <bool GUARD>
__device__ __forceinline__ float4 load_quad(const float* __restrict__ base, long o, long i, long O,
                                            long I, long so, long si, int vec, int kmode = 0) {
  if constexpr (!GUARD) {
    return *reinterpret_cast<const float4*>(base + o * so + i);  // si == 1, aligned, in bounds
  } else if (kmode == 1) {
    const float* row = base + (o < O ? o : O - 1) * so;
    if (vec) return *reinterpret_cast<const float4*>(row + i);
    float4 v;
    v.x = row[(i + 0) * si]; v.y = row[(i + 1) * si]; v.z = row[(i + 2) * si]; v.w = row[(i + 3) * si];
    return v;
  } else if (kmode == 2) {
    const float* row = base + o * so;
    const long l = I - 1;
    float4 v;
    v.x = row[(i + 0 < l ? i + 0 : l) * si]; v.y = row[(i + 1 < l ? i + 1 : l) * si];
    v.z = row[(i + 2 < l ? i + 2 : l) * si]; v.w = row[(i + 3 < l ? i + 3 : l) * si];
    return v;
  } else {
    // branch-free: clamp the address, load, select -- divergent branches here made the
    // compiler serialise the loads behind s_waitcnt vmcnt(0)
    (void)vec;
    const bool ov = o < O;
    const float* row = base + (ov ? o : 0) * so;
    const bool v0 = ov && (i + 0 < I), v1 = ov && (i + 1 < I), v2 = ov && (i + 2 < I), v3 = ov && (i + 3 < I);
    const float x0 = row[(v0 ? i + 0 : 0) * si];
    const float x1 = row[(v1 ? i + 1 : 0) * si];
    const float x2 = row[(v2 ? i + 2 : 0) * si];
    const float x3 = row[(v3 ? i + 3 : 0) * si];
    float4 v;
    v.x = v0 ? x0 : 0.f;
    v.y = v1 ? x1 : 0.f;
    v.z = v2 ? x2 : 0.f;
    v.w = v3 ? x3 : 0.f;
    return v;
  }
}

// AMODE: 0 = quads along k (A k-contiguous), 1 = quads along m (A m-contiguous)
// BMODE: 0 = quads along n (B n-contiguous), 1 = quads along k (B k-contiguous)
// GUARD: bounds-checked loads/stores (edge tiles, K tails, unaligned operands)
// edge (PF == 5 only): the tile hangs over the M / N extent.  Its loads stay unguarded -- every lane's row / column
// is fixed for the whole K loop, so a lane beyond the extent simply re-reads the last valid row / column (clamped
// once, in the pointer set-up) and computes outputs nobody stores -- and its stores are guarded.
// (a template parameter, so that the full-tile instantiation is exactly the code it was; edge tiles always leave
//  through the wide-store epilogue -- the route that sends them here requires its alignment)
template <int BM, int BN, int BK, int WM, int WN, int AMODE, int BMODE, bool GUARD, int PF = 0, bool EDGE = false>
__device__ __forceinline__ void gemm_body(const GemmKArgs& g, float* smem, int tile_m, int tile_n) {
  constexpr bool edge = EDGE;
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  constexpr int LDA = BM + 4;  // floats; +4 keeps rows 16-B aligned and staggers banks
  constexpr int LDB = BN + 4;
  constexpr int QA = BM * BK / 4 / NT;  // quads per thread
  constexpr int QB = BN * BK / 4 / NT;
  static_assert(QA >= 1 && QB >= 1, "tile too small for the block");

  float* As = smem;                  // [2][BK][LDA]
  float* Bs = smem + 2 * BK * LDA;   // [2][BK][LDB]

  const long m0 = (long)tile_m * BM, n0 = (long)tile_n * BN;
  const int bz = blockIdx.z;
  const bool red = g.nb_reduce > 1;

  const float* Ab = g.A + (red ? 0 : (long)bz * g.a_sb);
  const float* Bb = g.B + (red ? 0 : (long)bz * g.b_sb);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
  const int l31 = lane & 31, half = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = (g.K + BK - 1) / BK;
  int t_begin = 0, T = KT * g.nb_reduce;
  if (g.ksplit < 0) {
    t_begin = g.seg_begin;
    T = g.seg_end;
  } else if (g.ksplit > 1) {
    t_begin = blockIdx.y * g.t_per_split;
    const int t_end = t_begin + g.t_per_split;
    T = t_end < T ? t_end : T;
  }

  float4 ra[QA], rb[QB];
  const bool kfull = g.K % BK == 0;  // guarded tiles: only M/N are ragged (see load_quad)

  auto gload = [&](int t) {
    const int bb = t / KT, kt = t - bb * KT;
    const long k0 = (long)kt * BK;
    const float* Ap = Ab + (red ? (long)bb * g.a_sb : 0);
    const float* Bp = Bb + (red ? (long)bb * g.b_sb : 0);
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int qi = tid + q * NT;
      if constexpr (AMODE == 1) {  // inner = m
        const int k = qi / (BM / 4), mq = (qi % (BM / 4)) * 4;
        ra[q] = load_quad<GUARD>(Ap, k0 + k, m0 + mq, g.K, g.M, g.a_sk, g.a_sm, g.a_vec, kfull ? 2 : 0);
      } else {                     // inner = k
        const int m = qi / (BK / 4), kq = (qi % (BK / 4)) * 4;
        ra[q] = load_quad<GUARD>(Ap, m0 + m, k0 + kq, g.M, g.K, g.a_sm, g.a_sk, g.a_vec, kfull ? 1 : 0);
      }
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int qi = tid + q * NT;
      if constexpr (BMODE == 1) {  // inner = k
        const int n = qi / (BK / 4), kq = (qi % (BK / 4)) * 4;
        rb[q] = load_quad<GUARD>(Bp, n0 + n, k0 + kq, g.N, g.K, g.b_sn, g.b_sk, g.b_vec, kfull ? 1 : 0);
      } else {                     // inner = n
        const int k = qi / (BN / 4), nq = (qi % (BN / 4)) * 4;
        rb[q] = load_quad<GUARD>(Bp, k0 + k, n0 + nq, g.K, g.N, g.b_sk, g.b_sn, g.b_vec, kfull ? 2 : 0);
      }
    }
  };

  auto lstore = [&](int buf) {
    float* Ad = As + buf * BK * LDA;
    float* Bd = Bs + buf * BK * LDB;
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int qi = tid + q * NT;
      if constexpr (AMODE == 1) {
        const int k = qi / (BM / 4), mq = (qi % (BM / 4)) * 4;
        *reinterpret_cast<float4*>(Ad + k * LDA + mq) = ra[q];
      } else {
        const int m = qi / (BK / 4), kq = (qi % (BK / 4)) * 4;
        Ad[(kq + 0) * LDA + m] = ra[q].x;
        Ad[(kq + 1) * LDA + m] = ra[q].y;
        Ad[(kq + 2) * LDA + m] = ra[q].z;
        Ad[(kq + 3) * LDA + m] = ra[q].w;
      }
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int qi = tid + q * NT;
      if constexpr (BMODE == 1) {
        const int n = qi / (BK / 4), kq = (qi % (BK / 4)) * 4;
        Bd[(kq + 0) * LDB + n] = rb[q].x;
        Bd[(kq + 1) * LDB + n] = rb[q].y;
        Bd[(kq + 2) * LDB + n] = rb[q].z;
        Bd[(kq + 3) * LDB + n] = rb[q].w;
      } else {
        const int k = qi / (BN / 4), nq = (qi % (BN / 4)) * 4;
        *reinterpret_cast<float4*>(Bd + k * LDB + nq) = rb[q];
      }
    }
  };

  // ---- PF == 3: direct global -> LDS staging (global_load_lds_dwordx4), no VGPR round trip ------
  // A wave instruction fills 1 KiB of LDS linearly (wave-uniform base + lane * 16 B); the per-lane
  // GLOBAL address is free, so the LDS image is shaped by which element each lane fetches:
  //   operand contiguous along m/n: image [k][x] unpadded (the two 32-lane groups of a fragment
  //                                 read are serviced separately, each 32 consecutive floats);
  //   operand contiguous along k  : image [x][4 chunks of 4 k], chunk j of row x stored at slot
  //                                 j ^ ((x >> 2) & 3); a lane reads 8 bytes = two k of one row.
  // Both forms use the k-slot assignment  k(s, half) = 4*(s>>1) + 2*half + (s&1)  for MFMA step s,
  // which is legal because A and B agree on it (the MFMA sums over its k slots).
  constexpr int GA = BM * BK / 256 / (NT / 64);  // wave instructions per wave and operand
  constexpr int GB = BN * BK / 256 / (NT / 64);
  // LDS images of the pinned body.  (Three images on 128x128 tiles -- a k-tile is 2048 MFMA cycles there, a DMA issued
  // one tile ahead has ~0.85 us to land -- measured no faster: 2048^3 110.9 -> 112.5 TF.  What the small tile pays is
  // the per-tile fixed work under the MFMAs: pointer bumps, DMA issue, the wait + barrier.)
  constexpr int NI = 2;
  float* Ag = smem;                  // [NI][BM*BK]
  float* Bg = smem + NI * BM * BK;   // [NI][BN*BK]
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto gl_issue = [&](int t) {
    const int bb = t / KT, kt = t - bb * KT;
    const long k0 = (long)kt * BK;
    const float* Ap = Ab + (red ? (long)bb * g.a_sb : 0);
    const float* Bp = Bb + (red ? (long)bb * g.b_sb : 0);
    float* Ad = Ag + (t & 1) * BM * BK;
    float* Bd = Bg + (t & 1) * BN * BK;
#pragma unroll
    for (int q = 0; q < GA; ++q) {
      const int inst = wave * GA + q, f = inst * 256 + lane * 4;  // first float of this lane's 16 bytes
      const float* src;
      if constexpr (AMODE == 1) {
        src = Ap + (k0 + f / BM) * g.a_sk + (m0 + f % BM);
      } else {
        const int x = f / BK, jj = (f % BK) / 4, j = jj ^ ((x >> 2) & 3);
        src = Ap + (m0 + x) * g.a_sm + (k0 + 4 * j);
      }
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Ad + inst * 256), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < GB; ++q) {
      const int inst = wave * GB + q, f = inst * 256 + lane * 4;
      const float* src;
      if constexpr (BMODE == 0) {
        src = Bp + (k0 + f / BN) * g.b_sk + (n0 + f % BN);
      } else {
        const int x = f / BK, jj = (f % BK) / 4, j = jj ^ ((x >> 2) & 3);
        src = Bp + (n0 + x) * g.b_sn + (k0 + 4 * j);
      }
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Bd + inst * 256), 16, 0, 0);
    }
  };

  if constexpr (PF == 3) {
    static_assert(BK == 16, "the direct-to-LDS image is laid out for BK = 16");
    static_assert(GA >= 1 && GB >= 1, "tile too small for one wave instruction per wave");
    if (t_begin < T) gl_issue(t_begin);
    __syncthreads();  // (carries the vmcnt(0) that retires the LDS DMA)
    for (int t = t_begin; t < T; ++t) {
      if (t + 1 < T) gl_issue(t + 1);
      const float* Ar = Ag + (t & 1) * BM * BK;
      const float* Br = Bg + (t & 1) * BN * BK;
#pragma unroll
      for (int j = 0; j < BK / 4; ++j) {
        float a[2][TM], b[2][TN];  // [s & 1][tile]
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int x = wm0 + i * 32 + l31;
          if constexpr (AMODE == 1) {
            a[0][i] = Ar[(4 * j + 2 * half) * BM + x];
            a[1][i] = Ar[(4 * j + 2 * half + 1) * BM + x];
          } else {
            const float2 v = *reinterpret_cast<const float2*>(Ar + (x * 4 + (j ^ ((x >> 2) & 3))) * 4 + 2 * half);
            a[0][i] = v.x;
            a[1][i] = v.y;
          }
        }
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
          const int x = wn0 + jn * 32 + l31;
          if constexpr (BMODE == 0) {
            b[0][jn] = Br[(4 * j + 2 * half) * BN + x];
            b[1][jn] = Br[(4 * j + 2 * half + 1) * BN + x];
          } else {
            const float2 v = *reinterpret_cast<const float2*>(Br + (x * 4 + (j ^ ((x >> 2) & 3))) * 4 + 2 * half);
            b[0][jn] = v.x;
            b[1][jn] = v.y;
          }
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
              acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ss][i], b[ss][jn], acc[i][jn], 0, 0, 0);
      }
      __syncthreads();
    }
  } else if constexpr (PF == 5 && !GUARD) {
    // One wave per SIMD: 4 waves x 128x128 sub-tiles (half the fragment reads per MFMA of the 16-wave shape,
    // accumulators in the AccVGPR file).  Nothing but the wave's own instruction stream hides latency, so the
    // order is written out and pinned (sched_barrier after every MFMA + one other instruction):
    //  * operands go global -> LDS directly (the PF == 3 images): no staging registers, no LDS writes;
    //  * fragments are read one k-step PAIR ahead (32 MFMAs cover the LDS latency);
    //  * the barrier sits before the LAST pair's MFMAs; right behind it come the next tile's first fragments
    //    (other image) and the DMA of the tile after that (into the image just released), so a DMA always
    //    has a whole tile of MFMA time to land;
    //  * per-lane source pointers are bumped by a constant per tile: no address arithmetic in the loop.
    // The k-tile is consumed in two halves of four MFMA k-steps; lane (x, half) of half-tile h uses
    // k = 4 (2 h + half) + ss for step ss -- legal because A and B agree (the MFMA sums over its k slots) -- so a
    // k-contiguous operand's fragment for a whole half-tile is ONE ds_read_b128 (the vendor kernel's LRVW4).
    // Its image is [x][4 slots of 4 k], k-chunk c of row x in slot c ^ ((x >> 1) & 3): eight consecutive rows then
    // cover all 32 banks in a 16-byte read.
    // An n-contiguous B has no k-contiguous image to offer (a DMA lane's 16 bytes are four columns of one k),
    // but the MFMA does not care WHICH column a lane feeds it: lane l31 owns the TN consecutive columns
    // TN*l31 .. TN*l31+TN-1 of the wave's sub-tile (fragment jn = column TN*l31 + jn, the epilogue maps back), so
    // the fragments of all TN tiles for one k are ONE 16-byte (TN = 4) or 8-byte (TN = 2) read.
    static_assert(TN == 4 || TN == 2, "column-owning B fragments are read as b128 / b64");
    // (the same for an m-contiguous A: lane l31 owns rows TM*l31 .. TM*l31+TM-1)
    static_assert(TM == 4 || TM == 2, "row-owning A fragments are read as b128 / b64");
    constexpr int RA = AMODE == 1 ? 4 : TM, RB = BMODE == 0 ? 4 : TN;  // LDS reads per half-tile
    static_assert(BK == 16 && RA + RB + 2 * (GA + GB) <= 4 * TM * TN, "the last half has a slot for every instruction");
    static_assert(NI == 2, "the K loop below is written out for two images (compile-time image offsets)");
    constexpr int IMG_A = BM * BK * 4, IMG_B = BN * BK * 4;  // bytes per image
    static_assert(IMG_A + 11 * BM * 4 < 65536 && IMG_B + 11 * BN * 4 < 65536, "fragment reads address their image and k-step through the 16-bit offset field");
    // DMA addressing: a per-lane byte offset from the tile's first row / column (fixed for the whole K loop, 32 bits:
    // make_args sends operands with a stride of 2^22 elements or more elsewhere) on a SCALAR base that advances by a
    // constant per k-tile -- no vector address arithmetic in the loop.  The pieces of an operand share one M0 value:
    // the instruction offset advances the global and the LDS address alike, so piece q's lane offset is biased by
    // -q KiB (+3 KiB on every lane offset, -3 KiB on the base, keeps it non-negative).
    unsigned oa[GA], ob[GB];
#pragma unroll
    for (int q = 0; q < GA; ++q) {
      const int f = (wave * GA + q) * 256 + lane * 4;
      long e;
      if constexpr (AMODE == 1) {
        long m = m0 + f % BM;               // four consecutive rows (M % 4 == 0 on edge tiles: a quad is in or out)
        if (edge && m + 4 > g.M) m = g.M - 4;
        e = (long)(f / BM) * g.a_sk + (m - m0);
      } else {
        long m = m0 + f / BK;
        if (edge && m >= g.M) m = g.M - 1;
        e = (m - m0) * g.a_sm + 4 * (((f % BK) / 4) ^ (((f / BK) >> 1) & 3));
      }
      oa[q] = (unsigned)(e * 4 + 3072 - q * 1024);
    }
#pragma unroll
    for (int q = 0; q < GB; ++q) {
      const int f = (wave * GB + q) * 256 + lane * 4;
      long e;
      if constexpr (BMODE == 0) {
        long n = n0 + f % BN;
        if (edge && n + 4 > g.N) n = g.N - 4;
        e = (long)(f / BN) * g.b_sk + (n - n0);
      } else {
        long n = n0 + f / BK;
        if (edge && n >= g.N) n = g.N - 1;
        e = (n - n0) * g.b_sn + 4 * (((f % BK) / 4) ^ (((f / BK) >> 1) & 3));
      }
      ob[q] = (unsigned)(e * 4 + 3072 - q * 1024);
    }
    const long step_a = (AMODE == 1 ? (long)BK * g.a_sk : BK) * 4, step_b = (BMODE == 0 ? (long)BK * g.b_sk : BK) * 4;  // bytes
    static_assert(GA <= 4 && GB <= 4, "piece offsets are written out up to 3 KiB");
    const unsigned lds_a = (unsigned)(unsigned long)(lptr_t)Ag, lds_b = (unsigned)(unsigned long)(lptr_t)Bg;
    // (M0 is written inside the asm: nothing else in this instantiation uses it)
    const unsigned m0_a = __builtin_amdgcn_readfirstlane(lds_a + wave * GA * 1024), m0_b = __builtin_amdgcn_readfirstlane(lds_b + wave * GB * 1024);
    // (uniform by construction -- tile indices come from blockIdx -- but where the compiler cannot see that, the "s"
    //  constraint alone does not move a value into scalar registers)
    auto uniform64 = [](const void* q) {
      const unsigned long v = reinterpret_cast<unsigned long>(q);
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
      return reinterpret_cast<const char*>(((unsigned long)hi << 32) | lo);
    };
    const char* sa = uniform64(reinterpret_cast<const char*>(Ab + m0 * g.a_sm) - 3072 + (long)t_begin * step_a);
    const char* sb = uniform64(reinterpret_cast<const char*>(Bb + n0 * g.b_sn) - 3072 + (long)t_begin * step_b);
    // (`; @dma K` / `; @rd K` / `; @images` / `; @advance`: which k-tile's image, relative to the loop's current tile t, an
    //  access touches -- comments for tools/asm_inflight_check.py, which proves the waits and barriers below on the
    //  generated code of every instantiation, back edge included; they cost no instruction)
#define G5_DMA(OFF, BASE, IMM, TAG) asm volatile("global_load_lds_dwordx4 %0, %1 offset:" #IMM " ; @dma %2" ::"v"(OFF), "s"(BASE), "n"(TAG) : "memory")
    auto dma = [&](auto uc, int buf, auto tagc) {  // unit u of the wave instructions that fill one image pair; the tile fetched is t + tagc
      constexpr int u = decltype(uc)::value, TAG = decltype(tagc)::value;
      constexpr bool isa = u < GA;
      constexpr int q = isa ? u : u - GA;
      if constexpr (q == 0) {
        const unsigned mv = isa ? m0_a + buf * IMG_A : m0_b + buf * IMG_B;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(mv) : "memory");
      }
      const unsigned off = isa ? oa[q] : ob[q];
      const char* base = isa ? sa : sb;
      if constexpr (q == 0) G5_DMA(off, base, 0, TAG);
      if constexpr (q == 1) G5_DMA(off, base, 1024, TAG);
      if constexpr (q == 2) G5_DMA(off, base, 2048, TAG);
      if constexpr (q == 3) G5_DMA(off, base, 3072, TAG);
    };
#undef G5_DMA
    float a[2][4][TM], b[2][4][TN];  // [slot][ss][tile]
    // (LDS reads as inline asm: the compiler orders every LDS read it can see behind ALL outstanding LDS DMA
    //  with s_waitcnt vmcnt(0), which would stall each tile on the DMA issued a few MFMAs earlier; the
    //  lgkmcnt waits for these reads are written out at the half-tile boundaries below)
    // A read's address = a per-lane base for (operand, half-tile h, image) + a constant in the instruction's offset
    // field (read r).  The bases of the two images are swapped once per tile: four VALU instructions per k-tile are
    // all the address arithmetic the loop has.
    // ax[1] / bx[1]: second half of the current image; ax[0] / bx[0]: first half of the NEXT image
    unsigned ax[2], bx[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      ax[h] = lds_a + (h == 0 ? IMG_A : 0) + (AMODE == 1 ? ((4 * (2 * h + half)) * BM + wm0 + TM * l31) * 4
                                                       : ((wm0 + l31) * 4 + ((2 * h + half) ^ ((l31 >> 1) & 3))) * 16);
      bx[h] = lds_b + (h == 0 ? IMG_B : 0) + (BMODE == 0 ? ((4 * (2 * h + half)) * BN + wn0 + TN * l31) * 4
                                                       : ((wn0 + l31) * 4 + ((2 * h + half) ^ ((l31 >> 1) & 3))) * 16);
    }
    auto rd64 = [](unsigned addr, auto off, auto tagc) { float2 v; asm volatile("ds_read_b64 %0, %1 offset:%2 ; @rd %3" : "=v"(v) : "v"(addr), "n"(decltype(off)::value), "n"(decltype(tagc)::value)); return v; };
    auto rd128 = [](unsigned addr, auto off, auto tagc) { float4 v; asm volatile("ds_read_b128 %0, %1 offset:%2 ; @rd %3" : "=v"(v) : "v"(addr), "n"(decltype(off)::value), "n"(decltype(tagc)::value)); return v; };
    // LDS read r (0..RA+RB-1) of the half-tile whose bases are abase / bbase
    auto frag = [&](auto slotc, unsigned abase, unsigned bbase, auto rc, auto tagc) {  // reads the image of tile t + tagc
      constexpr int slot = decltype(slotc)::value, r = decltype(rc)::value;
      if constexpr (r < RA) {
        if constexpr (AMODE == 1) {  // r = k-step
          if constexpr (TM == 4) {
            const float4 v = rd128(abase, std::integral_constant<int, r * BM * 4>{}, tagc);
            a[slot][r][0] = v.x; a[slot][r][1] = v.y; a[slot][r][2] = v.z; a[slot][r][3] = v.w;
          } else {
            const float2 v = rd64(abase, std::integral_constant<int, r * BM * 4>{}, tagc);
            a[slot][r][0] = v.x; a[slot][r][1] = v.y;
          }
        } else {                     // r = 32-row tile
          const float4 v = rd128(abase, std::integral_constant<int, r * 2048>{}, tagc);
          a[slot][0][r] = v.x; a[slot][1][r] = v.y; a[slot][2][r] = v.z; a[slot][3][r] = v.w;
        }
      } else {
        constexpr int rr = r - RA;
        if constexpr (BMODE == 0) {
          if constexpr (TN == 4) {
            const float4 v = rd128(bbase, std::integral_constant<int, rr * BN * 4>{}, tagc);
            b[slot][rr][0] = v.x; b[slot][rr][1] = v.y; b[slot][rr][2] = v.z; b[slot][rr][3] = v.w;
          } else {
            const float2 v = rd64(bbase, std::integral_constant<int, rr * BN * 4>{}, tagc);
            b[slot][rr][0] = v.x; b[slot][rr][1] = v.y;
          }
        } else {
          const float4 v = rd128(bbase, std::integral_constant<int, rr * 2048>{}, tagc);
          b[slot][0][rr] = v.x; b[slot][1][rr] = v.y; b[slot][2][rr] = v.z; b[slot][3][rr] = v.w;
        }
      }
    };
    typedef std::integral_constant<int, 0> c0_t;
    // split-K (blockIdx.y): this workgroup's k-tiles are [t_begin, T); an empty split still writes its zero partial
    const int nT = __builtin_amdgcn_readfirstlane(T - t_begin);
    if (nT > 0) {
    // prologue: tiles 0 .. NI-1 in flight, first fragments once tile 0 has landed
    asm volatile("; @images %0 shared" ::"n"(NI));
    g_static_for<0, NI>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      g_static_for<0, GA + GB>([&](auto uc) { dma(uc, i, ic); });
      if constexpr (i + 1 < NI) {
        sa += nT > i + 1 ? step_a : 0;
        sb += nT > i + 1 ? step_b : 0;
      }
    });
    // (waits and barriers written out: __syncthreads would drain ALL the DMA; a wave's own counted vmcnt followed by a
    //  barrier the reader has passed is what orders an LDS DMA before a ds_read)
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NI - 1) * (GA + GB)) : "memory");
    g_static_for<0, RA + RB>([&](auto rc) { frag(c0_t{}, ax[0] - IMG_A, bx[0] - IMG_B, rc, c0_t{}); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int buf = 0;
    int dimg_a = -IMG_A, dimg_b = -IMG_B;  // (what moves a base to the other image: alternates in sign)
    for (int t = 0; t < nT; ++t) {
      const long da = t + NI < nT ? step_a : 0, db = t + NI < nT ? step_b : 0;  // (the last passes re-fetch the last tile)
      g_static_for<0, 2>([&](auto hc) {
        constexpr int h = decltype(hc)::value, cur = h;
        typedef std::integral_constant<int, (h ^ 1)> nxt_t;
        if constexpr (h == 1) {  // every wave is done with image `buf`; the DMA of tile t+1 (NI-1 tiles ago) has landed
          asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NI - 2) * (GA + GB)) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        g_static_for<0, 4 * TM * TN>([&](auto nc) {
          constexpr int n = decltype(nc)::value;
          constexpr int ss = n / (TM * TN), i = (n % (TM * TN)) / TN, jn = n % TN;
          asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i][jn]) : "v"(a[cur][ss][i]), "v"(b[cur][ss][jn]));
          if constexpr (n < RA + RB) {  // the next half's fragments (the next tile's first half behind the barrier)
            frag(nxt_t{}, ax[h ^ 1], bx[h ^ 1], nc, hc);   // (h == 0: this tile's image, h == 1: the next tile's)
          } else if constexpr (n < RA + RB + 2) {
            // the bases this half has just used move to the other image
            if constexpr (n == RA + RB) ax[h ^ 1] += (h == 0 ? -dimg_a : dimg_a);
            else bx[h ^ 1] += (h == 0 ? -dimg_b : dimg_b);
          } else if constexpr (h == 1 && n < RA + RB + 2 + GA + GB) {
            // the scalar bases on to tile t+NI, its DMA into the image just released
            if constexpr (n == RA + RB + 2) {
              sa += da;
              sb += db;
            }
            dma(std::integral_constant<int, n - (RA + RB + 2)>{}, buf, std::integral_constant<int, NI>{});
          }
          __builtin_amdgcn_sched_barrier(0);  // pin: one MFMA, one other instruction
        });
        // the next half's fragments were issued under the first MFMAs of this one: long back.  (The wait sits at the end
        // of the half that issued the reads, not at the start of the half that uses them: the same place in the
        // instruction stream, but nothing is in flight across the loop's back edge or its exit.)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      });
      buf ^= 1;
      dimg_a = -dimg_a;
      dimg_b = -dimg_b;
      asm volatile("; @advance");   // (for the checker: the loop's t becomes t + 1)
    }
    }  // nT > 0
    // The last MFMAs retire before the epilogue reads AccVGPRs, and the last, unused DMA lands before the epilogue
    // reuses the LDS (inline asm: the compiler's barrier knows nothing of it).  The statement OWNS what it waits for:
    // every accumulator tile is a read-write operand of the statement that holds the wait states -- otherwise the
    // compiler is free to put its own code between the loop's exit and the wait, and it did (round 4, product build:
    // accumulator spills and AccVGPR shuffles `v_accvgpr_read v2, a248` a handful of scalar instructions behind the last
    // MFMA, through VGPRs whose ds_read -- the last, unused prefetch -- was still in flight; tools/asm_inflight_check.py
    // rules 1 and 6).  The fragment reads themselves are waited for at the END of the half that issued them (below), so
    // nothing is in flight when the loop is left.
#define G5_DRAIN "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15"
    if constexpr (TM == 4 && TN == 4) {
      asm volatile(G5_DRAIN : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[1][3])::"memory");
      // (an asm statement takes 30 operands: the other eight tiles change hands in a statement of their own, behind the wait states)
      asm volatile("" : "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]), "+a"(acc[3][0]), "+a"(acc[3][1]), "+a"(acc[3][2]), "+a"(acc[3][3])::"memory");
    } else if constexpr (TM == 4 && TN == 2) {
      asm volatile(G5_DRAIN : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[3][0]), "+a"(acc[3][1])::"memory");
    } else {
      static_assert(TM == 2 && TN == 2, "the operand lists are written out");
      asm volatile(G5_DRAIN : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[1][0]), "+a"(acc[1][1])::"memory");
    }
#undef G5_DRAIN
    __syncthreads();
  } else {
  if (t_begin < T) {  // (an empty split still writes its zero partial below)
    gload(t_begin);
    lstore(t_begin & 1);
  }
  __syncthreads();
  if (g.dbg && threadIdx.x == 0) g.dbg[blockIdx.x * 8 + 1] = wall_clock64();

  for (int t = t_begin; t < T; ++t) {
    const int buf = t & 1;
    if (g.dbg && threadIdx.x == 0 && (t == T / 4 || t == T / 2)) g.dbg[blockIdx.x * 8 + (t == T / 4 ? 6 : 7)] = wall_clock64();
    if (t + 1 < T) gload(t + 1);
    const float* Ar = As + buf * BK * LDA + wm0 + l31;
    const float* Br = Bs + buf * BK * LDB + wn0 + l31;
    if constexpr (PF == 1) {
      // fragment software pipeline: the LDS reads of k-step kk+1 are in flight under the MFMAs of kk
      float a[2][TM], b[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[0][i] = Ar[half * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[0][j] = Br[half * LDB + j * 32];
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < BK / 2) {
#pragma unroll
          for (int i = 0; i < TM; ++i) a[nxt][i] = Ar[((kk + 1) * 2 + half) * LDA + i * 32];
#pragma unroll
          for (int j = 0; j < TN; ++j) b[nxt][j] = Br[((kk + 1) * 2 + half) * LDB + j * 32];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      if constexpr (PF == 2) {
        // the next tile's LDS image is written in the MIDDLE of this tile's MFMA sequence: the other
        // buffer is free since the last barrier, and the stores' latency (and their wait for the global
        // loads) no longer sits between the last MFMA and the barrier
        if (kk == BK / 4 && t + 1 < T) lstore(buf ^ 1);
      }
      if constexpr (PF == 4) {
        // ... and the four waves of a SIMD (w, w+4, w+8, w+12) do it at four different k-steps, so that
        // three of them keep the matrix pipe fed while the fourth writes LDS
        if (kk == 1 + 2 * (wave >> 2) && t + 1 < T) lstore(buf ^ 1);
      }
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = Ar[(kk * 2 + half) * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Br[(kk * 2 + half) * LDB + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    }
    if (PF != 2 && PF != 4 && t + 1 < T) lstore(buf ^ 1);
    __syncthreads();
  }

  }  // PF != 3
  if (g.dbg && threadIdx.x == 0) g.dbg[blockIdx.x * 8 + 2] = wall_clock64();
  // epilogue: D reg r lane l -> row (r&3) + 8*(r>>2) + 4*half, col l31
  // (PF 5 with an n-contiguous B: tile j, lane l31 is column TN*l31 + j of the wave's sub-tile, see above)
  constexpr bool COLOWN = PF == 5 && !GUARD && BMODE == 0, ROWOWN = PF == 5 && !GUARD && AMODE == 1;
  auto wcol = [&](int j) { return COLOWN ? TN * l31 + j : j * 32 + l31; };
  auto wrow = [&](int i, int tr) { return ROWOWN ? TM * tr + i : i * 32 + tr; };  // tr: row within the MFMA tile
  float* Cb = g.C + (red ? 0 : (long)bz * g.c_sb) + (g.ksplit > 1 ? (long)blockIdx.y * g.M * g.N : 0);
  const float* Ci = g.Cin ? g.Cin + (red ? 0 : (long)bz * g.c_sb) : nullptr;
  if constexpr (!GUARD) {
    if (g.wide_store) {
      // The MFMA layout gives each store instruction two 128-byte row pieces.  Transpose
      // 16-row bands through a wave-private LDS strip instead and store whole rows of the
      // wave's sub-tile with dwordx4: 4x fewer store instructions, 1 KiB contiguous each.
      constexpr int LDW = TN * 32 + 4;
      float* Ws = smem + wave * (16 * LDW);  // the staging buffers are dead after the last barrier
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      // (one instantiation per activation: no per-element branches on bias / activation in the way out; ACT < 0 is the
      //  plain gmul, which does not even add a bias)
      float bj[TN];  // the lane's bias values, one per tile column (edge tiles: a column beyond N is never stored)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const long col = n0 + wn0 + wcol(j);
        bj[j] = (g.bias && (!edge || col < g.N)) ? g.bias[col] : 0.f;
      }
      auto leave = [&](auto actc) {
        constexpr int ACT = decltype(actc)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int band = 0; band < 2; ++band) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int rr = 0; rr < 8; ++rr) {
                const int r = band * 8 + rr;
                const int lrow = (r & 3) + 8 * ((r >> 2) & 1) + 4 * half;
                float v = g.alpha * acc[i][j][r];
                if constexpr (ACT >= 0) {
                  // bias and activation ride along (the fused `map logistic (gmul ...)` of config 5 stores once)
                  v += bj[j];
                  if constexpr (ACT == 1) v = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
                  else if constexpr (ACT == 2) v = tanhf(v);
                }
                Ws[lrow * LDW + wcol(j)] = v;
              }
            // the band's rows back as 16-byte pieces: all the reads, then all the stores
            f32x4 piece[TN * 2];
#pragma unroll
            for (int it = 0; it < TN * 2; ++it) {
              const int idx = it * 64 + lane;
              const int lrow = idx / (TN * 8), c4 = (idx % (TN * 8)) * 4;
              piece[it] = *reinterpret_cast<const f32x4*>(Ws + lrow * LDW + c4);
            }
#pragma unroll
            for (int it = 0; it < TN * 2; ++it) {
              const int idx = it * 64 + lane;
              const int lrow = idx / (TN * 8), c4 = (idx % (TN * 8)) * 4;
              const long row = m0 + wm0 + wrow(i, band * 16 + lrow);
              if (edge && (row >= g.M || n0 + wn0 + c4 >= g.N)) continue;  // (N % 4 == 0: a quad is in or out)
              f32x4* dst = reinterpret_cast<f32x4*>(Cb + row * g.c_sm + n0 + wn0 + c4);
              if (g.nt_store) __builtin_nontemporal_store(piece[it], dst);
              else *dst = piece[it];
            }
          }
      };
      if (!g.bias && g.act == 0) leave(std::integral_constant<int, -1>{});
      else if (g.act == 1) leave(std::integral_constant<int, 1>{});
      else if (g.act == 2) leave(std::integral_constant<int, 2>{});
      else leave(std::integral_constant<int, 0>{});
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const long col = n0 + wn0 + wcol(j);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long row = m0 + wm0 + wrow(i, (r & 3) + 8 * (r >> 2) + 4 * half);
        if (!GUARD || (row < g.M && col < g.N)) {
          float v = g.alpha * acc[i][j][r];
          if (Ci) v += g.beta * Ci[row * g.c_sm + col];
          if (g.bias) v += g.bias[col];
          if (g.act == 1) v = 1.0f / (1.0f + expf(-v));
          else if (g.act == 2) v = tanhf(v);
          if (g.dact) {
            const float h = g.dact[(red ? 0 : (long)bz * g.c_sb) + row * g.c_sm + col];
            v *= g.dact_kind ? 1.0f - h * h : h * (1.0f - h);
          }
          Cb[row * g.c_sm + col] = v;
        }
      }
    }
}

template <int BM, int BN, int BK, int WM, int WN, int AMODE, int BMODE, int PF = 0>
__global__ __launch_bounds__(WM* WN * 64) void gemm_mfma_kernel(GemmKArgs g) {
  constexpr int NI = 2;   // LDS images (see gemm_body)
  constexpr int STAGE_FLOATS = NI * BK * (BM + 4 + BN + 4);
  constexpr int STORE_FLOATS = WM * WN * 16 * (BN / WN + 4);  // wide-store epilogue strips
  __shared__ __attribute__((aligned(16))) float smem[STAGE_FLOATS > STORE_FLOATS ? STAGE_FLOATS : STORE_FLOATS];
  // XCD-aware tile order: hardware places block b on XCD b % 8; give each XCD a contiguous
  // run of the tile sequence (bijective for any grid size), and walk the tile grid in bands of
  // 4 tile-rows (column-major inside a band) so that one XCD's run is a squarish 4 x n block:
  // its private L2 then streams 4 A-panels + n B-panels instead of 2 A-panels + ALL B-panels
  // (PMC at 4096^3: FETCH_SIZE 604 MB with row-major runs).
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
    const int rows = (g.tiles_m - band * R) < R ? (g.tiles_m - band * R) : R;  // last band may be short
    const int in = bid - band * R * g.tiles_n;
    tile_n = in / rows;
    tile_m = band * R + in % rows;
  }
  const bool full = (long)(tile_m + 1) * BM <= g.M && (long)(tile_n + 1) * BN <= g.N &&
                    (g.K % BK) == 0 && g.a_vec && g.b_vec;
  if (g.dbg && threadIdx.x == 0) {
    g.dbg[blockIdx.x * 8 + 0] = wall_clock64();
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    g.dbg[blockIdx.x * 8 + 4] = ((unsigned long long)xcc << 32) | hwid;
    g.dbg[blockIdx.x * 8 + 5] = ((unsigned long long)tile_m << 32) | (unsigned)tile_n;
  }
  // an edge tile of a problem whose K is whole k-tiles stays on the pinned body (clamped loads, guarded stores)
  const bool edge_ok = PF == 5 && (g.K % BK) == 0 && g.a_vec && g.b_vec && (AMODE == 0 || g.M % 4 == 0) &&
                       (BMODE == 1 || g.N % 4 == 0) && g.M >= 4 && g.N >= 4 && g.N % 4 == 0 && g.wide_store;
  if (full)
    gemm_body<BM, BN, BK, WM, WN, AMODE, BMODE, false, PF>(g, smem, tile_m, tile_n);
  else if (edge_ok)
    gemm_body<BM, BN, BK, WM, WN, AMODE, BMODE, false, PF, true>(g, smem, tile_m, tile_n);
  else
    gemm_body<BM, BN, BK, WM, WN, AMODE, BMODE, true>(g, smem, tile_m, tile_n);
  if (g.dbg && threadIdx.x == 0) g.dbg[blockIdx.x * 8 + 3] = wall_clock64();
}

// ---- stream-K: equal shares of the k-tile stream for every workgroup ------------------------------
// A tile count that is no multiple of the 256 CUs leaves most of the chip idle in the last round of 256x256
// tiles (3072^3 = 144 tiles: 56 % of the CUs for the whole launch).  Here the grid is one workgroup per CU and
// workgroup w owns the units [w*upw, (w+1)*upw) of the stream "tile 0's k-tiles, tile 1's k-tiles, ...": it runs
// the PF = 5 body once per tile it touches.  A run that covers a whole tile writes C directly; a partial run
// writes a 256x256 partial to its own slot (2w: its first run, 2w+1: its last) and streamk_fixup_kernel adds the
// partials of every split tile in workgroup order -- deterministic, no atomics.
// Hybrid form (round 3): with more than one round of tiles only the LAST, ragged round needs splitting.  The first
// `tile0` tiles (whole rounds of 256) are done one whole tile per workgroup, straight into C, in the plain kernel's
// XCD-aware order; the stream covers tiles [tile0, tiles).  6144^3 (576 tiles): 64 split tiles instead of ~512, the
// fix-up reads 67 MB instead of ~540 (145 -> TF see DESIGN).
struct StreamK {
  int T;        // k-tiles per output tile
  int upw;      // units per workgroup
  int total;    // (tiles - tile0) * T
  int tile0;    // first tile of the stream (a multiple of the grid size)
  float* part;  // [2 * workgroups][256*256]
};

template <int AMODE, int BMODE>
__global__ __launch_bounds__(256) void gemm_mfma_streamk_kernel(GemmKArgs g, StreamK sk) {
  constexpr int BM = 256, BN = 256, BK = 16;
  constexpr int STAGE2 = 2 * BK * (BM + 4 + BN + 4), STORE_FLOATS = 4 * 16 * (BN / 2 + 4);
  __shared__ __attribute__((aligned(16))) float smem[STAGE2 > STORE_FLOATS ? STAGE2 : STORE_FLOATS];
  constexpr int R = 4;
#ifdef TOPS_AB_KNOBS
  // development stamps (TOPS_GEMM_DBG): [begin, after each whole tile and each run of the stream ...] per workgroup
  unsigned long long* const sk_dbg = g.dbg ? g.dbg + 65536 * 8 + blockIdx.x * 8 : nullptr;
  int sk_di = 0;
#define SK_STAMP() do { if (sk_dbg && threadIdx.x == 0 && sk_di < 8) sk_dbg[sk_di++] = wall_clock64(); } while (0)
  g.dbg = nullptr;   // (the body's own stamps are the plain kernel's)
#else
#define SK_STAMP() do { } while (0)
#endif
  SK_STAMP();
  // whole rounds first: workgroup b (XCD b % 8) takes the tile the plain kernel would give it in each round
  {
    const int nblk = gridDim.x, xcd = blockIdx.x & 7, q = nblk >> 3, r = nblk & 7;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    for (int tile = bid; tile < sk.tile0; tile += nblk) {
      const int band = tile / (R * g.tiles_n);
      const int rows = (g.tiles_m - band * R) < R ? (g.tiles_m - band * R) : R;
      const int in = tile - band * R * g.tiles_n;
      gemm_body<BM, BN, BK, 2, 2, AMODE, BMODE, false, 5>(g, smem, band * R + in % rows, in / rows);
      __syncthreads();  // the epilogue's LDS strips overlap the next run's images
      SK_STAMP();
    }
  }
  int u = blockIdx.x * sk.upw;
  const int u_end = (u + sk.upw < sk.total) ? u + sk.upw : sk.total;
  bool first = true;
  while (u < u_end) {
    const int st = u / sk.T;  // tile of the stream
    const int tile = sk.tile0 + st;
    const int kb = u - st * sk.T;
    const int ke = (sk.T - kb < u_end - u) ? sk.T : kb + (u_end - u);
    // the tile order of the plain kernel (XCD-aware bands) is a property of blockIdx there; here consecutive
    // workgroups simply walk consecutive tiles of a row-major band order
    const int band = tile / (R * g.tiles_n);
    const int rows = (g.tiles_m - band * R) < R ? (g.tiles_m - band * R) : R;
    const int in = tile - band * R * g.tiles_n;
    const int tile_n = in / rows, tile_m = band * R + in % rows;
    GemmKArgs h = g;
    h.ksplit = -1;
    h.seg_begin = kb;
    h.seg_end = ke;
    if (!(kb == 0 && ke == sk.T)) {  // partial: into this workgroup's slot, in tile-local coordinates
      float* slot = sk.part + (size_t)(2 * blockIdx.x + (first ? 0 : 1)) * (BM * BN);
      h.C = slot - ((long)tile_m * BM * BN + (long)tile_n * BN);
      h.c_sm = BN;
      h.wide_store = 1;
      h.nt_store = 0;
    }
    gemm_body<BM, BN, BK, 2, 2, AMODE, BMODE, false, 5>(h, smem, tile_m, tile_n);
    __syncthreads();  // the epilogue's LDS strips overlap the next run's images
    SK_STAMP();
    u += ke - kb;
    first = false;
  }
#undef SK_STAMP
}

// C tile <- sum of the partial runs of every tile that no single workgroup covered (workgroup order)
__global__ __launch_bounds__(256) void streamk_fixup_kernel(float* C, long c_sm, int tiles_m, int tiles_n, StreamK sk) {
  const int tile = sk.tile0 + blockIdx.y;
  const int u0 = blockIdx.y * sk.T, u1 = u0 + sk.T;
  const int w_lo = u0 / sk.upw, w_hi = (u1 - 1) / sk.upw;
  if (w_lo == w_hi) return;  // one workgroup did the whole tile, straight into C
  constexpr int R = 4;
  const int band = tile / (R * tiles_n);
  const int rows = (tiles_m - band * R) < R ? (tiles_m - band * R) : R;
  const int in = tile - band * R * tiles_n;
  const long m0 = (long)(band * R + in % rows) * 256, n0 = (long)(in / rows) * 256;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  for (int q = blockIdx.x * 256 + threadIdx.x; q < 256 * 64; q += gridDim.x * 256) {  // 16384 quads per tile
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int w = w_lo; w <= w_hi; ++w) {
      // workgroup w's run on this tile is its first run when w's range starts inside the tile, else its last
      const int wu0 = w * sk.upw;
      const int which = (wu0 >= u0) ? 0 : 1;
      // ... except that a range starting exactly at the tile start and covering it fully never gets here
      const f32x4 v = *reinterpret_cast<const f32x4*>(sk.part + (size_t)(2 * w + which) * 65536 + (size_t)q * 4);
      acc += v;
    }
    const int r = q / 64, c4 = (q % 64) * 4;
    *reinterpret_cast<f32x4*>(C + (m0 + r) * c_sm + n0 + c4) = acc;
  }
}

// ---- persistent variant: every tile full, plain epilogue ---------------------------------------
// One workgroup per CU slot walks tiles b, b+G, b+2G, ...  The k-loop is ONE software pipeline
// across tile boundaries: the global loads of the next tile's first k-tile are issued before the
// last MFMAs of the current tile, and the C stores go out through a separate LDS strip behind an
// LDS-only barrier.  Matters when K is short: at 262144x64x512 (BASELINE config 5) a tile is 4
// k-tiles of MFMA work followed by a 256 KiB store.  Measured there: 0.266 -> 0.228 ms.  The K
// sweep (t = 0.08 ms + K * 2.3 us) shows store time and compute time still ADD: VMEM operations
// retire in issue order per wave, so the loads issued after a tile's stores wait for their drain.
// Fetching a tile's whole K extent before its predecessor's stores removed that wait but spilled
// (1024 threads = 128 VGPRs) and was slower (0.277 ms).
// Timeline of config 5a (K = 64, TOPS_GEMM_DBG stamps): tile period 24.6 us = 4.5 us of epilogue
// issue + 20 us for the four k-tiles (14.4 us of MFMA).  Tried and measured equal (0.24-0.25 ms):
//  * de-phasing the workgroups by quarter-tile start delays (the store burst is NOT a chip-wide
//    HBM-bound burst: each CU drains its 256 KB at ~25 GB/s whatever the others do);
//  * a two-deep register prefetch with hand-placed `s_waitcnt vmcnt(N)` (untracked inline-asm
//    loads) so that no wait covers the C stores: the loads issued AFTER the stores still queue
//    behind them in the CU's in-order vector-memory pipe.
//  * a 4-stage direct-to-LDS ring (global_load_lds, loads three k-tiles ahead of the stores, waits by
//    hand): validated on all layouts, same 0.24 ms.  Its stamps and two ablations located the time:
//    per tile 16.1 us of MFMA phase (4 k-tiles), 4.9 us of epilogue -- identical WITHOUT the global
//    stores, i.e. the LDS transposition of 64 values per lane (64 ds_write_b32 + 16 ds_read_b128 per
//    lane, two lgkmcnt waits per pass, 16 waves at once) -- and 2.4 us of wait + barrier; without any C
//    store the kernel takes 0.184 ms, without MFMAs 0.117 ms (5.2 TB/s).  Storing straight from the
//    MFMA layout (64 narrow stores per lane) is issue-bound at 4.1 us and waits longer afterwards.
//  * swapping the MFMA operands so that four consecutive accumulator registers are four consecutive
//    columns (dwordx4 stores straight from registers, no LDS): correct, but every store instruction then
//    writes 32-byte pieces of 32 different rows, and those partial-line writes are far slower than
//    whole 128-byte lines (config 5a 0.24 -> 0.275 ms, 4096^3 131 -> 113 TF).
//  * two 128x256 workgroups per CU (8-row strips, 67 KB of LDS each) so that one's epilogue overlaps the
//    other's MFMAs: 0.194-0.198 ms against 0.191 ms in the warmed-up steady state -- no gain either.
//  * transposing the accumulators IN REGISTERS instead of through LDS (two butterfly stages of quad-permute
//    DPP moves + selects turn four registers x four lanes around, so every lane owns 16 contiguous bytes of
//    one row; with v_permlane32_swap pairing the wave's two 32-column blocks first, a store instruction writes
//    4 rows x 256 B): bit-identical results, no LDS strip, no waits -- and 0.223-0.225 ms against 0.208 ms in
//    the same run.  Sixteen dwordx4 stores issued back to back stall the wave longer than the LDS round trip
//    that spaces them out.
// (All variants agree within 3 %: the shape sits at 0.19 ms warm / 0.235 ms after three launches.)
template <int BM, int BN, int BK, int WM, int WN, int AMODE, int BMODE>
__global__ __launch_bounds__(WM* WN * 64) void gemm_mfma_persistent_kernel(GemmKArgs g, int ntiles) {
  constexpr int PF = 0;  // (the fragment-prefetch experiment lives in the non-persistent kernel)
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  constexpr int LDA = BM + 4;
  constexpr int LDB = BN + 4;
  constexpr int QA = BM * BK / 4 / NT;
  constexpr int QB = BN * BK / 4 / NT;
  constexpr int LDW = TN * 32 + 4;
  constexpr int STAGE_FLOATS = 2 * BK * (LDA + LDB);
  __shared__ __attribute__((aligned(16))) float smem[STAGE_FLOATS + WM * WN * 16 * LDW];
  float* As = smem;
  float* Bs = smem + 2 * BK * LDA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* Ws = smem + STAGE_FLOATS + wave * (16 * LDW);  // epilogue strip, disjoint from the staging buffers
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
  const int l31 = lane & 31, half = lane >> 5;
  const long bz = blockIdx.z;
  const float* Ab = g.A + bz * g.a_sb;
  const float* Bb = g.B + bz * g.b_sb;
  float* Cb = g.C + bz * g.c_sb;

  const int T = g.K / BK;  // k-tiles per output tile (K % BK == 0 here)
  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

  auto tile_origin = [&](int seq, long& m0, long& n0) {
    int bid = blockIdx.x + seq * gridDim.x;
    const int xcd = bid & 7, q = ntiles >> 3, r = ntiles & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    constexpr int R = 4;
    const int band = bid / (R * g.tiles_n);
    const int rows = (g.tiles_m - band * R) < R ? (g.tiles_m - band * R) : R;
    const int in = bid - band * R * g.tiles_n;
    n0 = (long)(in / rows) * BN;
    m0 = (long)(band * R + in % rows) * BM;
  };

  float4 ra[QA], rb[QB];
  auto gload = [&](int it) {
    const int seq = it / T, kt = it - seq * T;
    long m0, n0;
    tile_origin(seq, m0, n0);
    const long k0 = (long)kt * BK;
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int qi = tid + q * NT;
      if constexpr (AMODE == 1) {
        const int k = qi / (BM / 4), mq = (qi % (BM / 4)) * 4;
        ra[q] = load_quad<false>(Ab, k0 + k, m0 + mq, g.K, g.M, g.a_sk, g.a_sm, 1);
      } else {
        const int m = qi / (BK / 4), kq = (qi % (BK / 4)) * 4;
        ra[q] = load_quad<false>(Ab, m0 + m, k0 + kq, g.M, g.K, g.a_sm, g.a_sk, 1);
      }
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int qi = tid + q * NT;
      if constexpr (BMODE == 1) {
        const int n = qi / (BK / 4), kq = (qi % (BK / 4)) * 4;
        rb[q] = load_quad<false>(Bb, n0 + n, k0 + kq, g.N, g.K, g.b_sn, g.b_sk, 1);
      } else {
        const int k = qi / (BN / 4), nq = (qi % (BN / 4)) * 4;
        rb[q] = load_quad<false>(Bb, k0 + k, n0 + nq, g.K, g.N, g.b_sk, g.b_sn, 1);
      }
    }
  };
  auto lstore = [&](int buf) {
    float* Ad = As + buf * BK * LDA;
    float* Bd = Bs + buf * BK * LDB;
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int qi = tid + q * NT;
      if constexpr (AMODE == 1) {
        const int k = qi / (BM / 4), mq = (qi % (BM / 4)) * 4;
        *reinterpret_cast<float4*>(Ad + k * LDA + mq) = ra[q];
      } else {
        const int m = qi / (BK / 4), kq = (qi % (BK / 4)) * 4;
        Ad[(kq + 0) * LDA + m] = ra[q].x;
        Ad[(kq + 1) * LDA + m] = ra[q].y;
        Ad[(kq + 2) * LDA + m] = ra[q].z;
        Ad[(kq + 3) * LDA + m] = ra[q].w;
      }
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int qi = tid + q * NT;
      if constexpr (BMODE == 1) {
        const int n = qi / (BK / 4), kq = (qi % (BK / 4)) * 4;
        Bd[(kq + 0) * LDB + n] = rb[q].x;
        Bd[(kq + 1) * LDB + n] = rb[q].y;
        Bd[(kq + 2) * LDB + n] = rb[q].z;
        Bd[(kq + 3) * LDB + n] = rb[q].w;
      } else {
        const int k = qi / (BN / 4), nq = (qi % (BN / 4)) * 4;
        *reinterpret_cast<float4*>(Bd + k * LDB + nq) = rb[q];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int total = my_tiles * T;
  if (total > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();

  int kt = 0, seq = 0;
  for (int it = 0; it < total; ++it) {
    const int buf = it & 1;
    if (it + 1 < total) gload(it + 1);
    const float* Ar = As + buf * BK * LDA + wm0 + l31;
    const float* Br = Bs + buf * BK * LDB + wn0 + l31;
    if constexpr (PF == 1) {
      // fragment software pipeline: the LDS reads of k-step kk+1 are in flight under the MFMAs of kk
      float a[2][TM], b[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[0][i] = Ar[half * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[0][j] = Br[half * LDB + j * 32];
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < BK / 2) {
#pragma unroll
          for (int i = 0; i < TM; ++i) a[nxt][i] = Ar[((kk + 1) * 2 + half) * LDA + i * 32];
#pragma unroll
          for (int j = 0; j < TN; ++j) b[nxt][j] = Br[((kk + 1) * 2 + half) * LDB + j * 32];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      // LDS stores of the next k-tile inside the MFMA sequence, staggered over the SIMD's four waves
      if (kk == 1 + 2 * (wave >> 2) && it + 1 < total) lstore(buf ^ 1);
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = Ar[(kk * 2 + half) * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Br[(kk * 2 + half) * LDB + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    }
    if (++kt == T) {
      // tile finished: 16-row bands through the wave-private strip, whole-row dwordx4 stores
      long m0, n0;
      tile_origin(seq, m0, n0);
      if (g.dbg && tid == 0 && blockIdx.x < 64 && seq < 16) g.dbg[blockIdx.x * 64 + seq * 4 + 0] = wall_clock64();
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int band = 0; band < 2; ++band) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
              const int r = band * 8 + rr;
              const int lrow = (r & 3) + 8 * ((r >> 2) & 1) + 4 * half;
              float v = g.alpha * acc[i][j][r];
              if (g.bias) v += g.bias[n0 + wn0 + j * 32 + l31];
              if (g.act == 1) v = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
              else if (g.act == 2) v = tanhf(v);
              Ws[lrow * LDW + j * 32 + l31] = v;
            }
#pragma unroll
          for (int s4 = 0; s4 < TN * 2; ++s4) {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const int idx = s4 * 64 + lane;
            const int lrow = idx / (TN * 8), c4 = (idx % (TN * 8)) * 4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(Ws + lrow * LDW + c4);
            const long row = m0 + wm0 + i * 32 + band * 16 + lrow;
            f32x4* dst = reinterpret_cast<f32x4*>(Cb + row * g.c_sm + n0 + wn0 + c4);
            if (g.nt_store) __builtin_nontemporal_store(v, dst);
            else *dst = v;
          }
        }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      if (g.dbg && tid == 0 && blockIdx.x < 64 && seq < 16) g.dbg[blockIdx.x * 64 + seq * 4 + 1] = wall_clock64();
      kt = 0;
      ++seq;
      if (g.dbg && tid == 0 && blockIdx.x < 64 && seq <= 16) g.dbg[blockIdx.x * 64 + (seq - 1) * 4 + 2] = wall_clock64();
    }
    // LDS-only barrier: __syncthreads() would also wait vmcnt(0), i.e. for the C stores just
    // issued to drain
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
}

// ---- fallback: one thread per output element, any strides, float or double -----------------
// (degenerate contractions: K = 1 outer products, per-sample dots, tiny matrices)
template <class S>
struct NaiveArgs {
  const S* A;
  const S* B;
  S* C;
  const S* Cin;
  int M, N, K;
  long a_sm, a_sk, b_sk, b_sn, c_sm, a_sb, b_sb, c_sb;
  int nb_reduce;
  S alpha, beta;
  // the fused elementwise epilogue of GemmProblem (a border strip or the K tail of a split problem ends up here and
  // must finish its elements like every other kernel): + bias[n], logistic, * h(1-h)
  const S* bias;
  const S* dact;
  int act;
  int dact_kind;
};

template <class S>
__global__ void gemm_naive_kernel(NaiveArgs<S> g, long total) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long mn = (long)g.M * g.N;
  const long bz = g.nb_reduce > 1 ? 0 : idx / mn;
  const long rem = idx - bz * mn;
  const long m = rem / g.N, n = rem - m * g.N;
  S acc = S(0);
  for (int bb = 0; bb < g.nb_reduce; ++bb) {
    const S* Ap = g.A + (g.nb_reduce > 1 ? bb : bz) * g.a_sb + m * g.a_sm;
    const S* Bp = g.B + (g.nb_reduce > 1 ? bb : bz) * g.b_sb + n * g.b_sn;
    for (int k = 0; k < g.K; ++k) acc = fma(Ap[k * g.a_sk], Bp[k * g.b_sk], acc);
  }
  S v = g.alpha * acc;
  const long off = bz * g.c_sb + m * g.c_sm + n;
  if (g.Cin) v += g.beta * g.Cin[off];
  if (g.bias) v += g.bias[n];
  if (g.act == 1) v = S(1) / (S(1) + exp(-v));
  else if (g.act == 2) v = tanh(v);
  if (g.dact) {
    const S h = g.dact[off];
    v *= g.dact_kind ? S(1) - h * h : h * (S(1) - h);
  }
  g.C[off] = v;
}

static GemmKArgs make_args(const GemmProblem& p) {
  GemmKArgs g{};
  g.A = (const float*)p.A; g.B = (const float*)p.B; g.C = (float*)p.C;
  g.Cin = (p.beta != 0.0) ? (const float*)p.Cin : nullptr;
  g.M = (int)p.M; g.N = (int)p.N; g.K = (int)p.K;
  g.a_sm = p.a_sm; g.a_sk = p.a_sk; g.b_sk = p.b_sk; g.b_sn = p.b_sn; g.c_sm = p.c_sm;
  g.a_sb = p.a_sb; g.b_sb = p.b_sb; g.c_sb = p.c_sb;
  g.nb_reduce = p.reduce_batch ? (int)p.batch : 1;
  g.alpha = (float)p.alpha; g.beta = (float)p.beta;
  g.ksplit = 1; g.t_per_split = 0;
  g.bias = (const float*)p.bias; g.dact = (const float*)p.dact; g.act = p.act; g.dact_kind = p.dact_kind;
  // Global memory on gfx9 under HSA runs in unaligned-access mode: a 16-byte load (and the global side of an LDS
  // DMA) needs dword alignment only.  So operand rows that are not 16-byte aligned (4097 columns ...) still take
  // the vector paths: 4097x4096x4097 3.25 -> 1.05 ms, bit-exact on all four layouts (tools/unaligned_check.py).
  // (The wide C stores keep their alignment condition.)
  static const int unal = [] { const char* e = ab_getenv("TOPS_GEMM_UNALIGNED"); return e ? atoi(e) : 1; }();
  auto al16c = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
  auto al16 = [&](const void* q) { return unal || al16c(q); };
  auto eff = [&](int64_t stride, int64_t extent) { return (extent == 1 || unal) ? (int64_t)0 : stride; };
  static const int wide_env = [] { const char* e = ab_getenv("TOPS_GEMM_WIDE_STORE"); return e ? atoi(e) : 1; }();
  static const int nt_env = [] { const char* e = ab_getenv("TOPS_GEMM_NT_STORE"); return e ? atoi(e) : -1; }();
  g.wide_store = wide_env && !g.Cin && !p.dact && p.act <= 1 && al16c(p.C) && p.c_sm % 4 == 0 &&
                 (p.batch == 1 || p.c_sb % 4 == 0);
  // streaming output (larger than the 256 MiB Infinity Cache): do not let it evict the operands
  g.nt_store = nt_env >= 0 ? nt_env : (p.M * p.N * 4 * (p.reduce_batch ? 1 : p.batch) > (256LL << 20));
  const int64_t nb = p.batch;
  // A: element (m,k) at A[m*a_sm + k*a_sk].  mode 0 = quads along k, 1 = quads along m.
  if (p.a_sk == 1 && !(p.K == 1 && p.a_sm == 1)) {
    g.a_mode = 0;
    g.a_vec = al16(p.A) && eff(p.a_sm, p.M) % 4 == 0 && eff(p.a_sb, nb) % 4 == 0;
  } else if (p.a_sm == 1) {
    g.a_mode = 1;
    g.a_vec = al16(p.A) && eff(p.a_sk, p.K) % 4 == 0 && eff(p.a_sb, nb) % 4 == 0;
  } else {
    g.a_mode = 0;
    g.a_vec = 0;
  }
  // B: element (k,n) at B[k*b_sk + n*b_sn].  mode 0 = quads along n, 1 = quads along k.
  if (p.b_sn == 1 && !(p.N == 1 && p.b_sk == 1)) {
    g.b_mode = 0;
    g.b_vec = al16(p.B) && eff(p.b_sk, p.K) % 4 == 0 && eff(p.b_sb, nb) % 4 == 0;
  } else if (p.b_sk == 1) {
    g.b_mode = 1;
    g.b_vec = al16(p.B) && eff(p.b_sn, p.N) % 4 == 0 && eff(p.b_sb, nb) % 4 == 0;
  } else {
    g.b_mode = 0;
    g.b_vec = 0;
  }
  // (the pinned body reaches a tile's rows / columns through 32-bit byte offsets from the tile's origin: 256 rows or
  //  16 k-steps times the stride; an operand with a stride of 2^21 elements or more takes the guarded path)
  auto small = [](int64_t st) { return st > -(1LL << 21) && st < (1LL << 21); };
  if (!small(p.a_sm) || !small(p.a_sk)) g.a_vec = 0;
  if (!small(p.b_sk) || !small(p.b_sn)) g.b_vec = 0;
  return g;
}

// Would launch_gemm_mfma run this problem on the full-tile 4-wave kernel (PF = 5) with (nearly) whole rounds of tiles
// -- or, with a plain epilogue, through stream-K, which does not care about rounds?
// (run_gemm uses it to carve such a block out of a ragged problem.)
bool gemm_w4_full_rounds(const GemmProblem& p) {
  static const int w4 = [] { const char* e = ab_getenv("TOPS_GEMM_W4"); return e ? atoi(e) : 1; }();
  static const int variant = [] { const char* e = ab_getenv("TOPS_GEMM_VARIANT"); return e ? atoi(e) : 0; }();
  if (!w4 || variant != 0 || p.dtype != TO_F32 || p.reduce_batch) return false;
  if (p.M % 256 || p.N % 256 || p.K % 16) return false;
  const long tiles = (p.M / 256) * (p.N / 256) * p.batch;
  const bool plain = p.batch == 1 && p.beta == 0.0 && p.alpha == 1.0 && !p.bias && !p.dact && p.act == 0;  // stream-K's terms
  if (tiles < 128) return false;
  if (!plain && (tiles < 256 || 100 * tiles < 94 * ((tiles + 255) / 256) * 256)) return false;  // last round >= 94 % full
  if (p.K / 16 <= 16) return false;  // short K: the persistent kernel's territory
  const GemmKArgs g = make_args(p);
  return g.a_vec && g.b_vec && g.nb_reduce == 1;
}

// A large problem whose extents are multiples of 4 but not of 256, and whose 256x256 tiles (edge tiles included)
// fill whole rounds of the 256 CUs well: it runs WHOLE on the pinned kernel (edge tiles: clamped loads, guarded
// stores) instead of being carved into a block of full tiles plus border strips (4000^3: 256 tiles = one round).
bool gemm_w4_edge_whole(const GemmProblem& p) {
  static const int enable = [] { const char* e = ab_getenv("TOPS_GEMM_W4_EDGE"); return e ? atoi(e) : 1; }();
  static const int w4 = [] { const char* e = ab_getenv("TOPS_GEMM_W4"); return e ? atoi(e) : 1; }();
  static const int variant = [] { const char* e = ab_getenv("TOPS_GEMM_VARIANT"); return e ? atoi(e) : 0; }();
  if (!enable || !w4 || variant != 0 || p.dtype != TO_F32 || p.reduce_batch || p.batch != 1) return false;
  if (p.M % 4 || p.N % 4 || p.K % 16 || p.M < 256 || p.N < 256) return false;
  if (p.M % 256 == 0 && p.N % 256 == 0) return false;  // nothing ragged about it
  if (p.beta != 0.0 || p.dact || p.act > 1 || p.rowsum || p.loss_rows) return false;
  if (p.c_sm % 4 || (reinterpret_cast<uintptr_t>(p.C) & 15u)) return false;
  if (p.K / 16 <= 16) return false;
  const long tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  if (tiles < 200) return false;
  const long rounds = (tiles + 255) / 256;
  if (100 * tiles < 88 * rounds * 256) return false;                       // the last round at least ~88 % full
  if (100 * p.M * p.N < 88 * tiles * 65536) return false;                  // little padding inside the edge tiles
  const GemmKArgs g = make_args(p);
  return g.a_vec && g.b_vec && g.wide_store;
}

bool gemm_mfma_worthwhile(const GemmProblem& p) {
  const int64_t kk = p.K * (p.reduce_batch ? p.batch : 1);
  return p.M >= 8 && p.N >= 8 && kk >= 8 && p.M * p.N >= 1024;
}

template <int BM, int BN, int BK, int WM, int WN>
static bool launch_persistent(GemmKArgs& g, const GemmProblem& p, int nbz, hipStream_t s) {
  static const int enable = [] { const char* e = ab_getenv("TOPS_GEMM_PERSISTENT"); return e ? atoi(e) : 1; }();
  if (!enable || !g.wide_store || !g.a_vec || !g.b_vec || g.ksplit > 1 || g.nb_reduce > 1) return false;
  if (p.M % BM || p.N % BN || p.K % BK) return false;
  g.tiles_m = (int)(p.M / BM);
  g.tiles_n = (int)(p.N / BN);
  const int ntiles = g.tiles_m * g.tiles_n;
  // resident workgroups per CU by LDS (160 KiB) and threads; registers allow at least these
  constexpr int lds = (2 * BK * (BM + 4 + BN + 4) + WM * WN * 16 * (BN / WN + 4)) * 4;
  int per_cu = 163840 / lds;
  const int by_threads = 2048 / (WM * WN * 64);
  if (per_cu > by_threads) per_cu = by_threads;
  if (per_cu > 2) per_cu = 2;
  if (per_cu < 1) return false;
  const int slots = 256 * per_cu;
  if (ntiles <= slots) return false;  // a single round of tiles: nothing to overlap
  if (p.K / BK > 16) return false;    // short-K only: with a long K loop prologue/epilogue are <2% and
                                      // the fused loop schedules slightly worse (8192^3: 130 -> 120 TF)
  dim3 grid(slots, 1, nbz), block(WM * WN * 64);
  switch (g.a_mode * 2 + g.b_mode) {
    case 0: launch_k((gemm_mfma_persistent_kernel<BM, BN, BK, WM, WN, 0, 0>), grid, block, 0, s, g, ntiles); break;
    case 1: launch_k((gemm_mfma_persistent_kernel<BM, BN, BK, WM, WN, 0, 1>), grid, block, 0, s, g, ntiles); break;
    case 2: launch_k((gemm_mfma_persistent_kernel<BM, BN, BK, WM, WN, 1, 0>), grid, block, 0, s, g, ntiles); break;
    default: launch_k((gemm_mfma_persistent_kernel<BM, BN, BK, WM, WN, 1, 1>), grid, block, 0, s, g, ntiles); break;
  }
  return true;
}

template <int BM, int BN, int BK, int WM, int WN, int PF = 0>
static void launch_cfg(GemmKArgs& g, const GemmProblem& p, int nbz, hipStream_t s) {
  g.tiles_m = (int)((p.M + BM - 1) / BM);
  g.tiles_n = (int)((p.N + BN - 1) / BN);
  dim3 grid(g.tiles_m * g.tiles_n, g.ksplit > 1 ? g.ksplit : 1, nbz), block(WM * WN * 64);
  const int mode = g.a_mode * 2 + g.b_mode;
#ifdef TOPS_GEMM_DEV
  if constexpr (BM == 256 && BN == 256 && WM == 2 && WN == 2 && PF == 5) {
    switch (mode) {
      case 0: (launch_k)((gemm_mfma_kernel<BM, BN, BK, WM, WN, 0, 0, PF>), grid, block, 0, s, g); return;
#if TOPS_GEMM_DEV > 1
      case 1: (launch_k)((gemm_mfma_kernel<BM, BN, BK, WM, WN, 0, 1, PF>), grid, block, 0, s, g); return;
      case 2: (launch_k)((gemm_mfma_kernel<BM, BN, BK, WM, WN, 1, 0, PF>), grid, block, 0, s, g); return;
      case 3: (launch_k)((gemm_mfma_kernel<BM, BN, BK, WM, WN, 1, 1, PF>), grid, block, 0, s, g); return;
#endif
    }
  }
#endif
  switch (mode) {
    case 0: launch_k((gemm_mfma_kernel<BM, BN, BK, WM, WN, 0, 0, PF>), grid, block, 0, s, g); break;
    case 1: launch_k((gemm_mfma_kernel<BM, BN, BK, WM, WN, 0, 1, PF>), grid, block, 0, s, g); break;
    case 2: launch_k((gemm_mfma_kernel<BM, BN, BK, WM, WN, 1, 0, PF>), grid, block, 0, s, g); break;
    default: launch_k((gemm_mfma_kernel<BM, BN, BK, WM, WN, 1, 1, PF>), grid, block, 0, s, g); break;
  }
}

void launch_gemm_mfma(const GemmProblem& p, hipStream_t s) {
  GemmKArgs g = make_args(p);
  static const char* dbg_path = ab_getenv("TOPS_GEMM_DBG");
  static unsigned long long* dbg_buf = nullptr;
  if (dbg_path && !dbg_buf) {
    TO_HIP(hipMalloc(&dbg_buf, (65536 * 8 + 256 * 8) * sizeof(unsigned long long)));
    TO_HIP(hipMemset(dbg_buf, 0, (65536 * 8 + 256 * 8) * sizeof(unsigned long long)));
  }
  g.dbg = dbg_path ? dbg_buf : nullptr;
  const int nbz = p.reduce_batch ? 1 : (int)p.batch;
  static const int variant = [] {
    const char* e = ab_getenv("TOPS_GEMM_VARIANT");  // development knob: force a tile shape
    return e ? atoi(e) : 0;
  }();
  int v = variant;
  if (v == 0) {
    // largest tile that still fills the 256 CUs at least once (measured at 4096^3 fp32:
    // 256x256 120 TF, 128x128 111 TF), else 64x64 tiles
    const long t256 = (p.M / 256) * (p.N / 256) * nbz;
    const long t128 = ((p.M + 127) / 128) * ((p.N + 127) / 128) * nbz;
    v = (t256 >= 256) ? 5 : ((t128 >= 256 && p.M >= 128 && p.N >= 128) ? 1 : 9);
    if (gemm_w4_edge_whole(p)) v = 5;
  }
  Holder work;
  // Mid sizes (16..256 tiles of 128x128): the pinned 4-wave body on 128x128 tiles -- four waves of 64x64, 8-byte
  // row-/column-owning fragments -- with the K loop split over blockIdx.y so that a few hundred workgroups are in
  // flight (two per CU fit).  Measured against the routes below (256x256 tiles under split-K / stream-K):
  // 1024^3 51 -> 60 TF (4 splits), 1536^3 64 -> 77 (3 splits), 2048^3 100 -> 111 (256 tiles, no split); from ~300
  // tiles on the big tiles under stream-K win again (3072^3 120 vs 100).  TOPS_GEMM_W4_128=0 switches it off,
  // a value > 1 sets the workgroup count aimed at.
  static const int w4_128 = [] { const char* e = ab_getenv("TOPS_GEMM_W4_128"); return e ? atoi(e) : 1; }();
  if (variant == 0 && w4_128 > 0 && nbz == 1 && !p.reduce_batch && p.beta == 0.0 && p.alpha == 1.0 && !p.bias && !p.dact &&
      p.act == 0 && g.a_vec && g.b_vec && p.M % 4 == 0 && p.N % 4 == 0 && p.M >= 128 && p.N >= 128 && p.K % 16 == 0 &&
      p.c_sm % 4 == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15u) == 0) {
    // (extents that are no multiple of 128: the edge tiles stay on the pinned body -- clamped loads, guarded stores)
    const long t128 = ((p.M + 127) / 128) * ((p.N + 127) / 128), KT = p.K / 16;
    long ks = 1;
    if (w4_128 > 1) ks = w4_128 / (t128 > 0 ? t128 : 1);
    else if (t128 < 224) ks = std::min<long>(4, 512 / (t128 > 0 ? t128 : 1));
    if (ks > KT / 8) ks = KT / 8;  // at least 8 k-tiles per split
    if (ks < 1) ks = 1;
    if (t128 >= 16 && t128 <= 288 && KT >= 8) {
      if (ks >= 2) {
        g.t_per_split = (int)((KT + ks - 1) / ks);
        g.ksplit = (int)((KT + g.t_per_split - 1) / g.t_per_split);
        const int64_t wd[3] = {g.ksplit, p.M, p.N};
        work.t = new_tensor(3, wd, 0);
        g.C = work.t->f32();
        g.c_sm = p.N;
        g.wide_store = 1;
      }
      launch_cfg<128, 128, 16, 2, 2, 5>(g, p, nbz, s);
      TO_HIP(hipGetLastError());
      count_launch();
      if (g.dbg) {  // development: per-workgroup timestamps (of split 0 .. the last split to finish overwrites)
        TO_HIP(hipStreamSynchronize(s));
        const int nb = g.tiles_m * g.tiles_n;
        std::vector<unsigned long long> h((size_t)nb * 8);
        TO_HIP(hipMemcpy(h.data(), dbg_buf, h.size() * 8, hipMemcpyDeviceToHost));
        if (FILE* f = fopen(dbg_path, "w")) {
          for (int b = 0; b < nb; ++b)
            fprintf(f, "%d %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu\n", b, h[b * 8], h[b * 8 + 1], h[b * 8 + 2], h[b * 8 + 3],
                    h[b * 8 + 4] >> 32, h[b * 8 + 4] & 0xffffffffull, h[b * 8 + 5] >> 32, h[b * 8 + 5] & 0xffffffffull,
                    h[b * 8 + 6], h[b * 8 + 7]);
          fclose(f);
        }
      }
      if (g.ksplit > 1) {
        launch_sum_splits_strided(work.t->ptr, p.C, g.ksplit, p.M, p.N, p.c_sm, s);  // (16-byte loads and stores; C may be a block of a larger matrix)
      }
      return;
    }
  }
  // Mid-size problems (16..255 full 256x256 tiles): the 4-wave kernel with the K loop split over blockIdx.y so that
  // ~256 workgroups run; the partial products go to a [ksplit][M][N] workspace and are summed by a second,
  // deterministic pass.
  static const int w4split = [] { const char* e = ab_getenv("TOPS_GEMM_W4_SPLITK"); return e ? atoi(e) : 1; }();
  if (variant == 0 && w4split && nbz == 1 && !p.reduce_batch && p.beta == 0.0 && p.alpha == 1.0 && !p.bias && !p.dact &&
      p.act == 0 && g.a_vec && g.b_vec && p.M % 256 == 0 && p.N % 256 == 0 && p.K % 16 == 0) {
    const long t256 = (p.M / 256) * (p.N / 256), KT = p.K / 16;
    // stream-K when the tile count leaves the last round of tiles mostly empty and is no divisor of 256
    static const int streamk = [] { const char* e = ab_getenv("TOPS_GEMM_STREAMK"); return e ? atoi(e) : 1; }();
    const long rounds = (t256 + 255) / 256;
    if (streamk && t256 >= 16 && t256 <= 65535 && t256 * KT < (1L << 30) && (streamk == 2 || 10 * t256 < 9 * rounds * 256) && (streamk == 3 || !(t256 < 32 && 256 % t256 == 0 && KT / (256 / t256) >= 16)) &&  // (few tiles: plain split-K sums fewer partials)
        t256 * KT >= 256 * 12) {  // at least a dozen k-tiles per workgroup
      g.tiles_m = (int)(p.M / 256);
      g.tiles_n = (int)(p.N / 256);
      StreamK sk{};
      sk.T = (int)KT;
      // whole rounds straight into C, the ragged remainder as the stream (one round more when the remainder
      // alone would leave a workgroup fewer than eight k-tiles)
      static const int hybrid = [] { const char* e = ab_getenv("TOPS_GEMM_STREAMK_HYBRID"); return e ? atoi(e) : 1; }();
      long dp_rounds = hybrid ? t256 / 256 : 0;
      if (dp_rounds > 0 && (t256 - dp_rounds * 256) * KT < 256 * 8) --dp_rounds;
      sk.tile0 = (int)(dp_rounds * 256);
      sk.total = (int)((t256 - sk.tile0) * KT);
      sk.upw = (sk.total + 255) / 256;
      const int64_t wd[2] = {512, 65536};
      work.t = new_tensor(2, wd, 0);
      sk.part = work.t->f32();
      dim3 grid(256), block(256);
      switch (g.a_mode * 2 + g.b_mode) {
        case 0: launch_k((gemm_mfma_streamk_kernel<0, 0>), grid, block, 0, s, g, sk); break;
        case 1: launch_k((gemm_mfma_streamk_kernel<0, 1>), grid, block, 0, s, g, sk); break;
        case 2: launch_k((gemm_mfma_streamk_kernel<1, 0>), grid, block, 0, s, g, sk); break;
        default: launch_k((gemm_mfma_streamk_kernel<1, 1>), grid, block, 0, s, g, sk); break;
      }
      TO_HIP(hipGetLastError());
      count_launch();
#ifdef TOPS_AB_KNOBS
      if (g.dbg) {  // development: every workgroup's stamps, microseconds since the first workgroup began
        TO_HIP(hipStreamSynchronize(s));
        std::vector<unsigned long long> h(256 * 8);
        TO_HIP(hipMemcpy(h.data(), dbg_buf + 65536 * 8, h.size() * 8, hipMemcpyDeviceToHost));
        TO_HIP(hipMemset(dbg_buf + 65536 * 8, 0, h.size() * 8));
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < 256; ++b) if (h[b * 8] && h[b * 8] < t0) t0 = h[b * 8];
        if (FILE* f = fopen(dbg_path, "w")) {
          for (int b = 0; b < 256; ++b) {
            fprintf(f, "%d", b);
            for (int i = 0; i < 8 && h[b * 8 + i]; ++i) fprintf(f, " %.2f", (double)(h[b * 8 + i] - t0) * 0.01);
            fprintf(f, "\n");
          }
          fclose(f);
        }
      }
#endif
      launch_k(streamk_fixup_kernel, dim3(8, (unsigned)(t256 - sk.tile0)), dim3(256), 0, s, (float*)p.C, (long)p.c_sm, g.tiles_m,
                         g.tiles_n, sk);
      TO_HIP(hipGetLastError());
      count_launch();
      return;
    }
    long ks = 256 / (t256 > 0 ? t256 : 1);
    if (ks > KT / 16) ks = KT / 16;  // at least 16 k-tiles per split
    // (only when the split fills most of the chip: 1024^3 = 16 tiles x 4 splits ran at 27 TF against 52 TF on
    //  64x64 tiles, 1024x512x1024 at 14 against 47)
    if (t256 >= 16 && t256 < 256 && ks >= 2 && t256 * ks >= 192 && p.c_sm == p.N) {
      g.t_per_split = (int)((KT + ks - 1) / ks);
      g.ksplit = (int)((KT + g.t_per_split - 1) / g.t_per_split);
      const int64_t wd[3] = {g.ksplit, p.M, p.N};
      work.t = new_tensor(3, wd, 0);
      g.C = work.t->f32();
      g.c_sm = p.N;
      g.wide_store = 1;
      launch_cfg<256, 256, 16, 2, 2, 5>(g, p, nbz, s);
      TO_HIP(hipGetLastError());
      count_launch();
      if ((reinterpret_cast<uintptr_t>(p.C) & 15u) == 0) launch_sum_splits_strided(work.t->ptr, p.C, g.ksplit, p.M, p.N, p.c_sm, s);
      else launch_sum_axis(TO_F32, work.t->ptr, p.C, 1, g.ksplit, p.M * p.N, 0, p.M * p.N, 1, s);
      return;
    }
  }
  // split-K for latency-bound shapes: too few 64x64 tiles to fill 256 CUs but a long K loop.
  // Partials go to a [ksplit][M][N] workspace and are summed by a second, deterministic pass.
  if (v == 9 && nbz == 1 && p.beta == 0.0 && p.alpha == 1.0 && !p.bias && !p.dact && p.act == 0) {
    const long tiles = ((p.M + 63) / 64) * ((p.N + 63) / 64);
    const long T = ((p.K + 15) / 16) * (p.reduce_batch ? p.batch : 1);
    long ks = 512 / tiles;                 // aim at ~2 workgroups per CU
    if (ks > T / 2) ks = T / 2;            // at least two k-tiles per split
    if (ks > 64) ks = 64;
    if (tiles <= 128 && ks >= 2) {
      g.t_per_split = (int)((T + ks - 1) / ks);
      g.ksplit = (int)((T + g.t_per_split - 1) / g.t_per_split);
      const int64_t wd[3] = {g.ksplit, p.M, p.N};
      work.t = new_tensor(3, wd, 0);
      g.C = work.t->f32();
      g.c_sm = p.N;
    }
  }
  switch (v) {
    case 1:  // (mid-tile LDS stores measured equal here: 93.8 vs 94.3 TF at 2048^3)
      if (!launch_persistent<128, 128, 16, 2, 2>(g, p, nbz, s)) launch_cfg<128, 128, 16, 2, 2>(g, p, nbz, s);
      break;
#ifdef TOPS_GEMM_AB_VARIANTS  // (tile shapes / staging variants of the A/B runs quoted in the comments: a third of this
                              //  file's compile time, built only with -DTOPS_GEMM_AB_VARIANTS)
    case 2: launch_cfg<128, 128, 32, 2, 2>(g, p, nbz, s); break;
    case 3: launch_cfg<256, 128, 16, 4, 2>(g, p, nbz, s); break;
    case 4: launch_cfg<128, 256, 16, 2, 4>(g, p, nbz, s); break;
#endif
    case 5:  // LDS stores inside the MFMA sequence, staggered over the four waves of a SIMD (PF = 4):
             // end of tile 128.3 TF, mid-tile (PF = 2) 130.9-131.5, staggered k-steps 1/3/5/7 134.5-135.1
             // at 4096^3 (k-steps 4..7: 132.8, 2..5: 133.1; loads a whole tile ahead: 129.1)
      // ... and, where every tile is full and the K loop is plain, four waves of 128x128 on the written-out
      // schedule of PF = 5 (16-byte fragment reads, DMA-fed images, AccVGPR accumulators): 134.9 -> 143-145 TF
      // at 4096^3 on every operand layout, 144 at 8192^3
      // (round 3: with its way out at 5.6 us per tile the pinned body also wins the short-K row streams the
      //  persistent 16-wave kernel was written for -- 16384x256x4096 110 -> 128 TF, 16384x128x4096 98 -> 109 --
      //  so it goes first; the persistent kernel remains behind TOPS_GEMM_W4=0)
      {
        static const int w4 = [] { const char* e = ab_getenv("TOPS_GEMM_W4"); return e ? atoi(e) : 1; }();
        if (w4 && g.nb_reduce == 1 && g.ksplit <= 1 && g.a_vec && g.b_vec && p.K % 16 == 0 &&
            ((p.M % 256 == 0 && p.N % 256 == 0) || gemm_w4_edge_whole(p)))
          launch_cfg<256, 256, 16, 2, 2, 5>(g, p, nbz, s);
        else if (!launch_persistent<256, 256, 16, 4, 4>(g, p, nbz, s))
          launch_cfg<256, 256, 16, 4, 4, 4>(g, p, nbz, s);
      }
      break;
#ifdef TOPS_GEMM_AB_VARIANTS
    case 17: launch_cfg<256, 256, 16, 4, 4>(g, p, nbz, s); break;  // (end-of-tile LDS stores, for A/B runs)
    case 18: launch_cfg<256, 256, 16, 4, 4, 3>(g, p, nbz, s); break;  // direct global->LDS staging
    case 20: launch_cfg<256, 256, 16, 4, 4, 2>(g, p, nbz, s); break;  // (un-staggered mid-tile LDS stores, for A/B runs)
#endif
    case 34:  // (forced, whatever the tile count)
      if (g.nb_reduce == 1 && g.ksplit <= 1) launch_cfg<256, 256, 16, 2, 2, 5>(g, p, nbz, s);
      else launch_cfg<256, 256, 16, 4, 4, 4>(g, p, nbz, s);
      break;
    // (Also measured on this schedule, 4096^3: depth-32 k-tiles -- half the barriers -- 141.5 TF; three image
    //  pairs with the DMA pieces spread over the first half-tile, six MFMAs apart, 142.9: neither beats 144.4.)
    // (The same schedule on 128x128 tiles -- four waves of 64x64, 8-byte row-owning fragments -- measured with one
    //  workgroup per CU: 2048^3 110 TF against 98 on the split-K route, 4096^3 133, 2304^3 82 against 103, 1024^3 31
    //  against 52: it needs two workgroups per CU and its own stream-K to pay; not instantiated.)
#ifdef TOPS_GEMM_AB_VARIANTS
    case 35:  // 8 waves x 128x64 on the same schedule: 142.2-142.8 TF (4 waves: 143-145).  History of PF = 5 at
              // 4096^3: b32/b64 fragment reads 131.0 (4 waves) / 134.95 (8) / 134.6 (16) -- no better than the
              // compiler-scheduled default; b128 reads for k-contiguous operands 137.5 (ta0 tb0) / 141.4 (ta0 tb1);
              // row/column-owning b128 fragments for the m-/n-contiguous ones: 143-145 on all four layouts.
              // (The vendor GEMM reaches 151 TF on the same box, tools/vendor_gemm.py.)
      if (g.nb_reduce == 1 && g.ksplit <= 1) launch_cfg<256, 256, 16, 2, 4, 5>(g, p, nbz, s);
      else launch_cfg<256, 256, 16, 4, 4, 4>(g, p, nbz, s);
      break;
    case 6: launch_cfg<256, 128, 16, 2, 2>(g, p, nbz, s); break;
    case 7: launch_cfg<128, 128, 8, 2, 2>(g, p, nbz, s); break;
#else
    case 2: case 3: case 4: case 6: case 7: case 17: case 18: case 20: case 35: {
      static const bool told = [] {
        fprintf(stderr, "tensorops: TOPS_GEMM_VARIANT names an A/B variant this build does not contain "
                        "(-DTOPS_GEMM_AB_VARIANTS); running 64x64 tiles\n");
        return true;
      }();
      (void)told;
      launch_cfg<64, 64, 16, 2, 2>(g, p, nbz, s);
      break;
    }
#endif
    default: launch_cfg<64, 64, 16, 2, 2>(g, p, nbz, s); break;
  }
  TO_HIP(hipGetLastError());
  count_launch();
  if (g.dbg && ab_getenv("TOPS_GEMM_DBG_P")) {  // persistent kernel: [64 workgroups][16 tiles][4 stamps]
    TO_HIP(hipStreamSynchronize(s));
    std::vector<unsigned long long> h(64 * 64);
    TO_HIP(hipMemcpy(h.data(), dbg_buf, h.size() * 8, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(dbg_path, "w")) {
      for (int b = 0; b < 64; ++b) {
        for (int k = 0; k < 64; ++k) fprintf(f, "%llu ", h[b * 64 + k]);
        fprintf(f, "\n");
      }
      fclose(f);
    }
  } else if (g.dbg) {  // development: dump the last launch's per-workgroup timestamps
    TO_HIP(hipStreamSynchronize(s));
    const int nb = g.tiles_m * g.tiles_n;
    std::vector<unsigned long long> h((size_t)nb * 8);
    TO_HIP(hipMemcpy(h.data(), dbg_buf, h.size() * 8, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(dbg_path, "w")) {
      for (int b = 0; b < nb; ++b)
        fprintf(f, "%d %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu\n", b, h[b * 8], h[b * 8 + 1], h[b * 8 + 2], h[b * 8 + 3],
                h[b * 8 + 4] >> 32, h[b * 8 + 4] & 0xffffffffull, h[b * 8 + 5] >> 32, h[b * 8 + 5] & 0xffffffffull,
                h[b * 8 + 6], h[b * 8 + 7]);
      fclose(f);
    }
  }
  if (g.ksplit > 1) {
    // C[m,n] = sum_split P[split][m][n]  (rows of C may be strided: c_sm)
    if (p.c_sm == p.N) {
      launch_sum_axis(TO_F32, work.t->ptr, p.C, 1, g.ksplit, p.M * p.N, 0, p.M * p.N, 1, s);
    } else {
      for (int64_t m = 0; m < p.M; ++m)  // never hit by the planner (it always asks for packed C)
        launch_sum_axis(TO_F32, work.t->f32() + m * p.N, (float*)p.C + m * p.c_sm, 1, g.ksplit, p.N, 0,
                        p.M * p.N, 1, s);
    }
  }
}

template <class S>
static void naive_t(const GemmProblem& p, hipStream_t s) {
  NaiveArgs<S> g{};
  g.A = (const S*)p.A; g.B = (const S*)p.B; g.C = (S*)p.C;
  g.Cin = (p.beta != 0.0) ? (const S*)p.Cin : nullptr;
  g.M = (int)p.M; g.N = (int)p.N; g.K = (int)p.K;
  g.a_sm = p.a_sm; g.a_sk = p.a_sk; g.b_sk = p.b_sk; g.b_sn = p.b_sn; g.c_sm = p.c_sm;
  g.a_sb = p.a_sb; g.b_sb = p.b_sb; g.c_sb = p.c_sb;
  g.nb_reduce = p.reduce_batch ? (int)p.batch : 1;
  g.alpha = (S)p.alpha; g.beta = (S)p.beta;
  g.bias = (const S*)p.bias; g.dact = (const S*)p.dact; g.act = p.act; g.dact_kind = p.dact_kind;
  TO_CHECK(!p.rowsum && !p.loss_rows, TO_ERR_STATE, "internal: row sums / loss head routed to the fallback kernel");
  const long total = (long)p.M * p.N * (p.reduce_batch ? 1 : p.batch);
  if (total == 0) return;
  launch_k(gemm_naive_kernel<S>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, total);
  TO_HIP(hipGetLastError());
  count_launch();
}

void launch_gemm_naive(const GemmProblem& p, hipStream_t s) {
  if (p.dtype == TO_F64) naive_t<double>(p, s);
  else naive_t<float>(p, s);
}

}  // namespace to
