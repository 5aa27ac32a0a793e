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
        const bool av = (m0 + am < g.M) && (k0 + ak < g.K);
        const double x = Ap[(av ? m0 + am : 0) * g.a_sm + (av ? k0 + ak : 0) * g.a_sk];
        ra[q] = av ? x : 0.0;
      } else {
        ra[q] = Ap[(m0 + am) * g.a_sm + (k0 + ak) * g.a_sk];
      }
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int e = tid + q * NT;
      const int bn = b_nc ? e % BN : e / BK, bk = b_nc ? e / BN : e % BK;
      if constexpr (GUARD) {
        const bool bv = (n0 + bn < g.N) && (k0 + bk < g.K);
        const double y = Bp[(bv ? k0 + bk : 0) * g.b_sk + (bv ? n0 + bn : 0) * g.b_sn];
        rb[q] = bv ? y : 0.0;
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
  hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, WM, WN>), grid, dim3(WM * WN * 64), lds, s, g);
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
  static const int variant = [] { const char* v = getenv("TOPS_GEMM64_VARIANT"); return v ? atoi(v) : 0; }();
  const long nb = p.reduce_batch ? 1 : p.batch;
  const long t256 = ((p.M + 255) / 256) * ((p.N + 127) / 128) * nb;
  const long t128 = ((p.M + 127) / 128) * ((p.N + 127) / 128) * nb;
  int v = variant;
  if (v == 0) v = t256 >= 200 ? 256 : (t128 >= 128 ? 128 : 64);
  if (v == 256) launch_cfg<256, 128, 4, 2>(g, p, s);
  else if (v == 128) launch_cfg<128, 128, 2, 2>(g, p, s);
  else launch_cfg<64, 64, 2, 2>(g, p, s);
  TO_HIP(hipGetLastError());
  count_launch();
}

}  // namespace to
