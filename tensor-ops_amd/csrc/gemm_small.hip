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

struct SmallArgs {
  const float* A;
  const float* B;
  float* C;
  const float* Cin;
  int M, N, K;
  long a_sm, a_sk, b_sk, b_sn, c_sm;
  long a_sb, b_sb, c_sb;
  int a_vec, b_vec;  // 16-byte loads along k legal (k-contiguous modes only)
  int tiles_n;
  int kper;          // k extent per wave (multiple of 8)
  float alpha, beta;
  const float* bias;
  const float* dact;
  int act;
};

// AMODE 0: A k-contiguous (a_sk == 1)   1: A m-contiguous / general strides
// BMODE 0: B n-contiguous / general     1: B k-contiguous (b_sk == 1)
// Within a chunk of 8 k the MFMA j of half-wave `half` consumes k = k0 + 4*half + j, for A and B alike.
template <int AMODE, int BMODE, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_small_kernel(SmallArgs g) {
  __shared__ float red[NW][16][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int tile_m = blockIdx.x / g.tiles_n, tile_n = blockIdx.x % g.tiles_n;
  const long m = (long)tile_m * 32 + l31, n = (long)tile_n * 32 + l31;
  const long bz = blockIdx.z;
  const float* A = g.A + bz * g.a_sb;
  const float* B = g.B + bz * g.b_sb;
  const bool mv = m < g.M, nv = n < g.N;
  const float* Arow = A + (mv ? m : 0) * g.a_sm;  // + k * a_sk
  const float* Bcol = B + (nv ? n : 0) * g.b_sn;  // + k * b_sk

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int kbeg = wave * g.kper;
  int kend = kbeg + g.kper;
  if (kend > g.K) kend = g.K;

#pragma unroll 2
  for (int k0 = kbeg; k0 < kend; k0 += 8) {
    const int kb = k0 + 4 * half;
    float a[4], b[4];
    if (AMODE == 0 && g.a_vec) {
      const bool ok = mv && kb + 3 < kend;
      const float4 v = *reinterpret_cast<const float4*>(Arow + (ok ? kb : 0));
      a[0] = ok ? v.x : 0.f; a[1] = ok ? v.y : 0.f; a[2] = ok ? v.z : 0.f; a[3] = ok ? v.w : 0.f;
      if (!ok && mv) {  // ragged tail of this wave's slice (K or kper not a multiple of 4)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = (kb + j < kend) ? Arow[kb + j] : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = mv && kb + j < kend;
        const float x = Arow[(long)(ok ? kb + j : 0) * g.a_sk];
        a[j] = ok ? x : 0.f;
      }
    }
    if (BMODE == 1 && g.b_vec) {
      const bool ok = nv && kb + 3 < kend;
      const float4 v = *reinterpret_cast<const float4*>(Bcol + (ok ? kb : 0));
      b[0] = ok ? v.x : 0.f; b[1] = ok ? v.y : 0.f; b[2] = ok ? v.z : 0.f; b[3] = ok ? v.w : 0.f;
      if (!ok && nv) {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = (kb + j < kend) ? Bcol[kb + j] : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = nv && kb + j < kend;
        const float x = Bcol[(long)(ok ? kb + j : 0) * g.b_sk];
        b[j] = ok ? x : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
  }

  // cross-wave reduction through LDS, then the fused epilogue: wave w finishes regs r = w, w+NW, ...
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  float* Cb = g.C + bz * g.c_sb;
  const float* Ci = g.Cin ? g.Cin + bz * g.c_sb : nullptr;
  const float* Hd = g.dact ? g.dact + bz * g.c_sb : nullptr;
  for (int r = wave; r < 16; r += NW) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += red[w][r][lane];
    const long row = (long)tile_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    const long col = (long)tile_n * 32 + l31;
    if (row < g.M && col < g.N) {
      v *= g.alpha;
      if (Ci) v += g.beta * Ci[row * g.c_sm + col];
      if (g.bias) v += g.bias[col];
      if (g.act == 1) v = 1.0f / (1.0f + expf(-v));
      if (Hd) {
        const float h = Hd[row * g.c_sm + col];
        v *= h * (1.0f - h);
      }
      Cb[row * g.c_sm + col] = v;
    }
  }
}

bool gemm_small_applicable(const GemmProblem& p) {
  if (p.reduce_batch) return false;          // the planner folds the batch into K whenever it can
  if (p.batch > 65535) return false;
  const int64_t tiles64 = ((p.M + 63) / 64) * ((p.N + 63) / 64) * p.batch;
  return tiles64 < 200 && p.K >= 8 && p.M * p.N >= 256;
}

template <int NW>
static void launch_nw(SmallArgs& g, const GemmProblem& p, int amode, int bmode, hipStream_t s) {
  const int chunks = (int)((p.K + 7) / 8);
  g.kper = ((chunks + NW - 1) / NW) * 8;
  const int tiles_m = (int)((p.M + 31) / 32);
  g.tiles_n = (int)((p.N + 31) / 32);
  dim3 grid(tiles_m * g.tiles_n, 1, (unsigned)p.batch), block(NW * 64);
  switch (amode * 2 + bmode) {
    case 0: hipLaunchKernelGGL((gemm_small_kernel<0, 0, NW>), grid, block, 0, s, g); break;
    case 1: hipLaunchKernelGGL((gemm_small_kernel<0, 1, NW>), grid, block, 0, s, g); break;
    case 2: hipLaunchKernelGGL((gemm_small_kernel<1, 0, NW>), grid, block, 0, s, g); break;
    default: hipLaunchKernelGGL((gemm_small_kernel<1, 1, NW>), grid, block, 0, s, g); break;
  }
}

void launch_gemm_small(const GemmProblem& p, hipStream_t s) {
  SmallArgs g{};
  g.A = p.A; g.B = p.B; g.C = p.C; g.Cin = (p.beta != 0.f) ? p.Cin : nullptr;
  g.M = (int)p.M; g.N = (int)p.N; g.K = (int)p.K;
  g.a_sm = p.a_sm; g.a_sk = p.a_sk; g.b_sk = p.b_sk; g.b_sn = p.b_sn; g.c_sm = p.c_sm;
  g.a_sb = p.a_sb; g.b_sb = p.b_sb; g.c_sb = p.c_sb;
  g.alpha = p.alpha; g.beta = p.beta;
  g.bias = p.bias; g.dact = p.dact; g.act = p.act;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
  auto eff = [](int64_t stride, int64_t extent) { return extent == 1 ? (int64_t)0 : stride; };
  const int amode = (p.a_sk == 1) ? 0 : 1;
  const int bmode = (p.b_sk == 1 && p.b_sn != 1) ? 1 : 0;
  g.a_vec = amode == 0 && al16(p.A) && eff(p.a_sm, p.M) % 4 == 0 && eff(p.a_sb, p.batch) % 4 == 0;
  g.b_vec = bmode == 1 && al16(p.B) && eff(p.b_sn, p.N) % 4 == 0 && eff(p.b_sb, p.batch) % 4 == 0;
  // waves per tile: enough K-slices to give every SIMD of the chip a wave, capped by the K extent
  const int64_t tiles = ((p.M + 31) / 32) * ((p.N + 31) / 32) * p.batch;
  const int64_t chunks = (p.K + 7) / 8;
  int nw = 4;
  if (tiles * 4 < 1024 && chunks >= 32) nw = 8;
  if (tiles * 8 < 1024 && chunks >= 64) nw = 16;
  if (chunks < 8) nw = chunks >= 2 ? 2 : 1;
  switch (nw) {
    case 1: launch_nw<1>(g, p, amode, bmode, s); break;
    case 2: launch_nw<2>(g, p, amode, bmode, s); break;
    case 4: launch_nw<4>(g, p, amode, bmode, s); break;
    case 8: launch_nw<8>(g, p, amode, bmode, s); break;
    default: launch_nw<16>(g, p, amode, bmode, s); break;
  }
  TO_HIP(hipGetLastError());
  count_launch();
}

}  // namespace to
