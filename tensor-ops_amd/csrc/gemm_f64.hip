// fp64 GEMM on the gfx950 matrix cores (`ElemT = Double`, what the reference's apps
// instantiate: `HMat Double`, src/TensorOps/BLAS/HMat.hs:35).
//
// v_mfma_f64_16x16x4_f64: A operand lane l = A[i = l&15][k = l>>4], B operand lane l =
// B[k = l>>4][j = l&15] (one f64 each); D holds 4 f64 per lane with  col = l&15,
// row = (l>>4) + 4*r  -- NOT the f32 map (cdna_hip_programming.md section 3).
// Block tile 64x64x16, 4 waves (2x2), each wave a 32x32 sub-tile = 2x2 MFMA tiles.
// Same staging scheme as the fp32 kernel: global -> registers -> LDS image [k][m] / [k][n],
// two LDS buffers, one barrier per k-tile; all loads bounds-checked by clamp+select (this is
// the correctness-first fp64 path; the fp32 kernels carry the tuned fast paths).
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
  int tiles_n;
  double alpha, beta;
};

__global__ __launch_bounds__(256) void gemm_f64_kernel(G64 g) {
  constexpr int BM = 64, BN = 64, BK = 16, LDA = BM + 2, LDB = BN + 2;
  __shared__ double As[2][BK][LDA];
  __shared__ double Bs[2][BK][LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int tile_m = blockIdx.x / g.tiles_n, tile_n = blockIdx.x % g.tiles_n;
  const long m0 = (long)tile_m * BM, n0 = (long)tile_n * BN;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
  const bool red = g.nb_reduce > 1;
  const long bz = blockIdx.z;
  const double* Ab = g.A + (red ? 0 : bz * g.a_sb);
  const double* Bb = g.B + (red ? 0 : bz * g.b_sb);

  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

  const int KT = (g.K + BK - 1) / BK, T = KT * g.nb_reduce;
  // 64x16 = 1024 elements per operand tile, 4 per thread; thread -> (row r, k chunk)
  double ra[4], rb[4];
  const bool a_kc = g.a_sk == 1;  // walk k fastest when A is k-contiguous, else m fastest
  const bool b_nc = g.b_sn == 1;
  auto gload = [&](int t) {
    const int bb = t / KT, kt = t - bb * KT;
    const long k0 = (long)kt * BK;
    const double* Ap = Ab + (red ? (long)bb * g.a_sb : 0);
    const double* Bp = Bb + (red ? (long)bb * g.b_sb : 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + q * 256;
      const int am = a_kc ? e / BK : e % BM, ak = a_kc ? e % BK : e / BM;
      const bool av = (m0 + am < g.M) && (k0 + ak < g.K);
      const double x = Ap[(av ? m0 + am : 0) * g.a_sm + (av ? k0 + ak : 0) * g.a_sk];
      ra[q] = av ? x : 0.0;
      const int bn = b_nc ? e % BN : e / BK, bk = b_nc ? e / BN : e % BK;
      const bool bv = (n0 + bn < g.N) && (k0 + bk < g.K);
      const double y = Bp[(bv ? k0 + bk : 0) * g.b_sk + (bv ? n0 + bn : 0) * g.b_sn];
      rb[q] = bv ? y : 0.0;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + q * 256;
      const int am = a_kc ? e / BK : e % BM, ak = a_kc ? e % BK : e / BM;
      As[buf][ak][am] = ra[q];
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
      double a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[buf][kk * 4 + kq][wm0 + i * 16 + l15];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[buf][kk * 4 + kq][wn0 + j * 16 + l15];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < T) lstore(buf ^ 1);
    __syncthreads();
  }
  double* Cb = g.C + (red ? 0 : bz * g.c_sb);
  const double* Ci = g.Cin ? g.Cin + (red ? 0 : bz * g.c_sb) : nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = m0 + wm0 + i * 16 + kq + 4 * r;  // f64 map: row = (lane>>4) + 4*reg
        const long col = n0 + wn0 + j * 16 + l15;
        if (row < g.M && col < g.N) {
          double v = g.alpha * acc[i][j][r];
          if (Ci) v += g.beta * Ci[row * g.c_sm + col];
          Cb[row * g.c_sm + col] = v;
        }
      }
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
  const int tiles_m = (int)((p.M + 63) / 64);
  g.tiles_n = (int)((p.N + 63) / 64);
  dim3 grid(tiles_m * g.tiles_n, 1, p.reduce_batch ? 1 : (unsigned)p.batch);
  hipLaunchKernelGGL(gemm_f64_kernel, grid, dim3(256), 0, s, g);
  TO_HIP(hipGetLastError());
  count_launch();
}

}  // namespace to
