// Short-K streaming GEMM for gfx950: C[M,N] = act(alpha * A[M,K] . B[K,N] + bias), K in {16, 32, 64}, N a multiple
// of 256, M in the hundreds of thousands -- BASELINE config 5, `gmul '[512,512,64] x '[64,512]` (+ mapped
// logistic): M = 262144, K = 64, N = 512, 17.18 GFLOP against 604 MB (537 MB of it the store of C).  The
// reference maps 512 boxed per-slice GEMMs (src/TensorOps/Backend/BTensor.hs:695-713) and then `cmap`s the
// closure over the result (src/TensorOps/Learn/NeuralNet.hs:42-44).
//
// Why a kernel of its own.  With K = 64 a 256x256 tile is four k-steps of MFMAs and then 256 KB of stores; the
// tiled kernels alternate the two phases behind workgroup barriers and the matrix pipe idles while C drains
// (0.19-0.20 ms, 57 % of the MFMA bound).  Here nothing is shared between waves after the prologue, so nothing
// has to be waited for collectively:
//   * a workgroup owns one 256-column panel: its slice of B (<= 64 KiB) is transposed into LDS once ([n][k],
//     16-byte groups XOR-swizzled) and stays;
//   * every wave owns a stream of 32-row blocks of that panel: its A rows come straight from global memory into
//     the MFMA fragment layout (eight 16-byte loads per lane per block, issued a block ahead and BEFORE the
//     previous block's stores, so the in-order memory counter returns them first), B fragments by ds_read_b128;
//   * 256 v_mfma_f32_32x32x2_f32 per block into 128 accumulators;
//   * the block leaves through a WAVE-PRIVATE 8-row LDS strip, four passes: the memory system wants whole
//     1 KiB rows per store instruction (measured on this shape: dword stores straight from the MFMA layout, two
//     128-byte lines per instruction, 2.3 TB/s; 16-byte stores of 32-byte row pieces worse; whole rows 5.5 TB/s);
//   * no barrier in the loop: the eight waves of a workgroup (two per SIMD) run out of phase, so one wave's
//     stores -- and the `logistic` of a fused map, VALU/transcendental work -- sit under the other's MFMAs.
// Bound: max(MFMA 109 us, HBM 76 us at spec).
#include "common.hpp"

namespace to {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct SkinnyArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  long M;
  int N, K;
  long a_sm;        // A row stride (elements); A is k-contiguous
  long b_sk, b_sn;  // B element strides
  long c_sm;
  float alpha;
  int npanels;      // N / 256
  int nrb;          // M / 32
  int stagger;
};

constexpr int SK_ROW = 264;  // strip row stride in floats: the two half-waves land on disjoint bank halves

// KQ = K / 8 (k-slots come in groups of 8: four for each half-wave); ACT, NT compile-time: the epilogue is
// straight-line code
template <int KQ, int ACT, int NT>
__global__ __launch_bounds__(512) void gemm_skinnyk_kernel(SkinnyArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int K = KQ * 8, GROUPS = KQ * 2;
  float* Bs = smem;                        // [256][K], 16-byte group q of column n at slot q ^ (n & (GROUPS-1))
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* strip = smem + 256 * K + wave * (8 * SK_ROW);
  const int l31 = lane & 31, half = lane >> 5;
  const int panel = blockIdx.x % g.npanels, wg_in_panel = blockIdx.x / g.npanels;
  const int wgs_per_panel = gridDim.x / g.npanels;
  const int n0 = panel * 256;
  // ---- prologue: this panel of B -> LDS, transposed ---------------------------------------------------------------
  for (int e = tid; e < 256 * K; e += 512) {
    int n, k;
    if (g.b_sn == 1) { k = e >> 8; n = e & 255; }   // walk B the way it is contiguous
    else { n = e / K; k = e - n * K; }
    const float v = g.B[(long)k * g.b_sk + (long)(n0 + n) * g.b_sn];
    const int grp = (k >> 2) ^ (n & (GROUPS - 1));
    Bs[n * K + grp * 4 + (k & 3)] = v;
  }
  __syncthreads();
  // ---- the wave's stream of 32-row blocks --------------------------------------------------------------------------
  const int stride = wgs_per_panel * 8;
  int rb = wg_in_panel * 8 + wave;
  if (wg_in_panel >= wgs_per_panel || rb >= g.nrb) return;
  // the two waves of a SIMD (w and w + 4) start together and do identical work: put them in antiphase
  if (wave >= 4 && g.stagger) __builtin_amdgcn_s_sleep(127);
  auto a_ptr = [&](int b) { return g.A + ((long)b * 32 + l31) * g.a_sm + 4 * half; };
  f32x4 a_cur[KQ], a_nxt[KQ];
  {
    const float* ap = a_ptr(rb);
#pragma unroll
    for (int q = 0; q < KQ; ++q) a_cur[q] = *reinterpret_cast<const f32x4*>(ap + 8 * q);
  }
  float bj[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bj[j] = g.bias ? g.bias[n0 + j * 32 + l31] : 0.f;
  while (true) {
    const int nxt = rb + stride;
    const bool more = nxt < g.nrb;
    if (more) {
      const float* ap = a_ptr(nxt);
#pragma unroll
      for (int q = 0; q < KQ; ++q) a_nxt[q] = *reinterpret_cast<const f32x4*>(ap + 8 * q);
    }
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // B fragments one pair of column tiles ahead of the MFMAs that consume them (the LDS round trip of a pair
    // hides behind the eight MFMAs of the pair before it)
    auto b_frag = [&](int q, int j) {
      const int n = j * 32 + l31;
      const int grp = (2 * q + half) ^ (n & (GROUPS - 1));
      return *reinterpret_cast<const f32x4*>(Bs + n * K + grp * 4);
    };
    // four column tiles per group: four independent accumulator chains keep the matrix pipe issuing back to back
    // (two chains leave it a few passes idle between dependent MFMAs)
    f32x4 bc[4], bn[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bc[t] = b_frag(0, t);
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
#pragma unroll
      for (int jg = 0; jg < 2; ++jg) {
        const int nq = (jg == 1) ? q + 1 : q, nj = (jg == 1) ? 0 : 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) bn[t] = (nq < KQ) ? b_frag(nq, nj + t) : bc[t];
        __builtin_amdgcn_sched_barrier(0);  // (the scheduler would sink the reads to just before their use)
#pragma unroll
        for (int ss = 0; ss < 4; ++ss)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc[4 * jg + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q][ss], bc[t][ss], acc[4 * jg + t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) bc[t] = bn[t];
      }
    }
    // register r of lane (l31, half) is row (r&3) + 8*(r>>2) + 4*half, column j*32 + l31 of the block.
    // Pass p moves rows 8p .. 8p+7 through the strip and out as eight whole 1 KiB rows.
    float* crow = g.C + ((long)rb * 32) * g.c_sm + n0 + 4 * lane;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          float v = g.alpha * acc[j][4 * p + rr] + bj[j];
          if (ACT == 1) v = __frcp_rn(1.0f + __expf(-v));
          strip[(rr + 4 * half) * SK_ROW + j * 32 + l31] = v;
        }
#pragma unroll
      for (int row = 0; row < 8; ++row) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(strip + row * SK_ROW + 4 * lane);
        f32x4* dst = reinterpret_cast<f32x4*>(crow + (long)(8 * p + row) * g.c_sm);
        if (NT) __builtin_nontemporal_store(v, dst);
        else *dst = v;
      }
    }
    if (!more) break;
    rb = nxt;
#pragma unroll
    for (int q = 0; q < KQ; ++q) a_cur[q] = a_nxt[q];
  }
}

bool gemm_skinnyk_applicable(const GemmProblem& p) {
  static const int enable = [] { const char* e = getenv("TOPS_GEMM_SKINNYK"); return e ? atoi(e) : 1; }();
  if (!enable || p.dtype != TO_F32 || p.batch != 1 || p.reduce_batch) return false;
  if (p.K != 64 && p.K != 32 && p.K != 16) return false;
  if (p.N % 256 != 0 || p.N < 256 || p.N > 256 * 64) return false;
  if (p.M % 32 != 0 || p.M * p.N < (1LL << 24) || p.M / 32 < 2048) return false;  // a long stream of rows
  if (p.a_sk != 1 || p.a_sm % 4 != 0 || (reinterpret_cast<uintptr_t>(p.A) & 15u)) return false;
  if (p.c_sm < p.N || p.c_sm % 4 != 0 || (reinterpret_cast<uintptr_t>(p.C) & 15u)) return false;
  if (p.beta != 0.0 || p.dact || p.rowsum || p.loss_rows || p.act > 1) return false;
  if (p.M / 32 > 2147483647LL) return false;
  return true;
}

void launch_gemm_skinnyk(const GemmProblem& p, hipStream_t s) {
  SkinnyArgs g{};
  g.A = (const float*)p.A; g.B = (const float*)p.B; g.C = (float*)p.C;
  g.bias = (const float*)p.bias;
  g.M = p.M; g.N = (int)p.N; g.K = (int)p.K;
  g.a_sm = p.a_sm; g.b_sk = p.b_sk; g.b_sn = p.b_sn; g.c_sm = p.c_sm;
  g.alpha = (float)p.alpha;
  g.npanels = (int)(p.N / 256);
  g.nrb = (int)(p.M / 32);
  static const int stagger = [] { const char* e = getenv("TOPS_SKINNYK_STAGGER"); return e ? atoi(e) : 1; }();
  g.stagger = stagger;
  bool nt = p.M * p.N * 4 > (256LL << 20);
  static const int nt_env = [] { const char* e = getenv("TOPS_SKINNYK_NT"); return e ? atoi(e) : -1; }();
  if (nt_env >= 0) nt = nt_env != 0;
  const size_t lds = ((size_t)256 * p.K + 8 * 8 * SK_ROW) * 4;
  const int grid = 256 / g.npanels * g.npanels;  // whole panels' worth of workgroups, one per CU
  static bool attr_set[12] = {false};
  auto launch = [&](auto kern, int which) {
    if (!attr_set[which]) {
      TO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set[which] = true;
    }
    launch_k(kern, dim3(grid), dim3(512), lds, s, g);
  };
  const int v = (p.act ? 2 : 0) + (nt ? 1 : 0);
#define TOPS_SKINNY(KQ, base)                                                  \
  switch (v) {                                                                 \
    case 0: launch(gemm_skinnyk_kernel<KQ, 0, 0>, base + 0); break;            \
    case 1: launch(gemm_skinnyk_kernel<KQ, 0, 1>, base + 1); break;            \
    case 2: launch(gemm_skinnyk_kernel<KQ, 1, 0>, base + 2); break;            \
    default: launch(gemm_skinnyk_kernel<KQ, 1, 1>, base + 3); break;           \
  }
  switch (p.K) {
    case 64: TOPS_SKINNY(8, 0) break;
    case 32: TOPS_SKINNY(4, 4) break;
    default: TOPS_SKINNY(2, 8) break;
  }
#undef TOPS_SKINNY
  TO_HIP(hipGetLastError());
  count_launch();
}

}  // namespace to
