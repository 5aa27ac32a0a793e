// Short-K streaming GEMM in fp64: C[M,N] = act(alpha * A[M,K] . B[K,N] + bias), K = 64, N a multiple of 128,
// M in the hundreds of thousands -- BASELINE config 5 (`gmul '[512,512,64] x '[64,512]`, + mapped logistic) in the
// reference's own element type (`HMat Double`, src/TensorOps/BLAS/HMat.hs:35): 17.18 GFLOP against 1.21 GB, 1.07 GB of
// it the store of C.  Bound: max(fp64 MFMA 218 us, HBM 151 us).  The tiled fp64 kernel (gemm_f64.hip) runs it as 256x128
// tiles of four k-steps each between a prologue and a 256 KiB epilogue: 0.47 ms.
//
// The fp32 kernel's plan (gemm_skinnyk.hip) on v_mfma_f64_16x16x4_f64:
//  * a workgroup owns one 128-column panel: its slice of B (K x 128 doubles <= 64 KiB) is transposed into LDS once
//    ([n][k], 16-byte chunks XOR-swizzled by n) and stays;
//  * every wave owns a stream of 16-row blocks of that panel.  A rows come straight from global memory into the MFMA
//    fragment layout: lane (row l15, kg) takes the 16-byte chunks kg, kg + 4, kg + 8 ... of its row (k = 8 q + 2 kg, +1
//    for MFMA steps 2 q, 2 q + 1), so the four lanes of a row read 64 consecutive bytes per instruction; loads are issued
//    a block ahead; B fragments by ds_read_b128 (two steps per read);
//  * K/4 x 8 MFMAs per block into 8 accumulator tiles (64 registers);
//  * the block leaves through a WAVE-PRIVATE 4-row LDS strip, four passes (D register r holds rows 4 r + kg): whole
//    1 KiB rows per store instruction;
//  * no barrier after the prologue.  Two waves per SIMD do NOT hide the drain (they fall into lockstep on the shared
//    matrix pipe: both compute, then both drain -- 0.30 ms of MFMA + LDS time for 0.22 ms of MFMAs); so a wave drains
//    block i - 1 from a second accumulator set UNDER the MFMAs of block i, one wave per SIMD.
// The workgroups that stream the SAME rows through different panels sit on one XCD (the later readers of A hit its L2).
#include <type_traits>

#include "common.hpp"

namespace to {

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

struct Skinny64Args {
  const double* A;
  const double* B;
  double* C;
  const double* bias;
  long M;
  int N, K;
  long a_sm;        // A row stride (elements); A is k-contiguous
  long b_sk, b_sn;  // B element strides
  long c_sm;
  double alpha;
  int npanels;      // N / 128
  long nrb;         // ceil(M / 16)
  int act;
  int nt;           // nontemporal stores (an output larger than the caches)
};

template <int I, int N, class F>
__device__ __forceinline__ void sk64_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sk64_static_for<I + 1, N>(f);
  }
}

// logistic in fp64 at 20 instructions an element (round 5; libm's exp + an IEEE division were ~30 and made the fused
// launch 46 % slower than the plain product: an fp64 MFMA runs at the vector unit's own rate, nothing the VALU does
// overlaps with it, so every epilogue instruction is 4 cycles added to a block's 8,192).  exp(-v) = 2^(k/1024) e^r with
// k = rint(-v 1024 / ln 2): 2^((k mod 1024)/1024) from a 1,024-entry table in LDS (reads are free under the MFMAs), 2^(k div
// 1024) by v_ldexp_f64, e^r (|r| <= 3.4e-4) a cubic; the reciprocal from v_rcp_f32 of the rounded denominator and one
// second-order correction step in fp64 (y0 (1 + e + e^2), e = 1 - d y0: e^3 ~ 2e-21 left).  Relative error <= 1e-15 over |v| <= 40 (tools/c5_f64_probe.py, tests/test_gpu_f64.py).
// The denominator is clamped at 1e38: logistic(v) for -745 < v < -87 returns 1e-38 instead of its true (smaller) value.
// Beyond what exp covers (|v| >= 745, the infinities, NaN) the reduction below is meaningless (inf - inf; a NaN that fmin
// would swallow): those arguments get the limits of 1 / (1 + exp(-v)) -- 1, 0, and the NaN itself -- by one compare and a
// select on the way out (a diverged run has to stay visible).
__device__ __forceinline__ double logistic64_tab(double v, const double* tab) {
  const double v_in = v;
  const bool wild = !(__builtin_fabs(v) < 745.0);   // (true for NaN)
  v = wild ? 0.0 : v;
  const double k = __builtin_rint(v * -1477.3197218702985);             // -v * 1024 / ln 2
  double r = __builtin_fma(k, -0x1.62e42fefa0000p-11, -v);              // ln 2 / 1024, its leading 36 bits (k C_hi is exact)
  r = __builtin_fma(k, -1.6079802420132516e-15, r);                     // ... and the rest
  const int ki = (int)k;
  const double tj = tab[ki & 1023];
  double p = __builtin_fma(r, 1.0 / 6.0, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  const double d = __builtin_fmin(ldexp(tj * p, ki >> 10) + 1.0, 1e38);
  const double y0 = (double)__builtin_amdgcn_rcpf((float)d);
  const double e = __builtin_fma(-d, y0, 1.0);             // 1 - d y0, |e| <= 1.2e-7 (the rounding of d to fp32 and v_rcp_f32's ulp)
  const double y = __builtin_fma(y0, __builtin_fma(e, e, e), y0);    // y0 (1 + e + e^2): e^3 left
  return wild ? (v_in > 0.0 ? 1.0 : (v_in < 0.0 ? 0.0 : v_in)) : y;
}

// KS = K / 4: MFMA k-steps per block; ACT, BIAS, NT compile-time: the way out is straight-line code that can be
// pinned between the MFMAs
template <int KS, int ACT, bool BIAS, bool NT>
__global__ __launch_bounds__(256) void gemm_skinnyk64_kernel(Skinny64Args g) {
  constexpr int K = KS * 4, CH = K / 2;          // 16-byte chunks per column of the panel image
  constexpr int NW = 4, STRIP = 4 * 128;         // doubles per wave-private strip: 4 rows x 128 columns
  extern __shared__ __attribute__((aligned(16))) double smem64[];
  double* Bs = smem64;                            // [128][K], chunk q of column n at slot q ^ (n & (CH - 1))
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* strip = smem64 + 128 * K + wave * STRIP;
  const double* tab = smem64 + 128 * K + NW * STRIP;   // ACT == 1: 2^(j / 1024), j < 1024
  if constexpr (ACT == 1) {
    double* tw = smem64 + 128 * K + NW * STRIP;
#pragma unroll
    for (int u = 0; u < 1024 / (NW * 64); ++u) tw[u * (NW * 64) + tid] = exp2((double)(u * (NW * 64) + tid) * (1.0 / 1024.0));
  }
  const int l15 = lane & 15, kg = lane >> 4;
  // workgroup -> (panel, stream): the npanels workgroups of one row stream on ONE XCD (block b runs on XCD b % 8)
  const int nwg = gridDim.x;
  int panel, wg_in_panel;
  const int wgs_per_panel = nwg / g.npanels;
  if (nwg % (8 * g.npanels) == 0) {
    panel = (blockIdx.x >> 3) % g.npanels;
    wg_in_panel = (blockIdx.x / (8 * g.npanels)) * 8 + (blockIdx.x & 7);
  } else {
    panel = blockIdx.x % g.npanels;
    wg_in_panel = blockIdx.x / g.npanels;
  }
  const int n0 = panel * 128;

  // ---- prologue: this panel of B -> LDS, transposed.  One unit = two consecutive k of one column -------------------
  {
    constexpr int UNITS = 128 * CH, PER = UNITS / (NW * 64);
    f64x2 v[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int unit = u * (NW * 64) + tid, n = unit % 128, q = unit / 128;  // (lanes walk n: coalesced when B is n-contiguous)
      const double* src = g.B + (long)(2 * q) * g.b_sk + (long)(n0 + n) * g.b_sn;
      v[u].x = src[0];
      v[u].y = src[g.b_sk];
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int unit = u * (NW * 64) + tid, n = unit % 128, q = unit / 128;
      *reinterpret_cast<f64x2*>(Bs + n * K + 2 * (q ^ (n & (CH - 1)))) = v[u];
    }
  }
  __syncthreads();

  const long stream = (long)wg_in_panel * NW + wave, nstreams = (long)wgs_per_panel * NW;
  constexpr int AL = KS / 2;  // 16-byte loads per lane and block
  // load q of lane (row, kg): the 16-byte chunk 4 q + kg of the row -- the four lanes of a row read 64 consecutive
  // bytes per instruction (chunk 8 kg + q, a lane's own 128-byte line in eight instructions, ran the A loads alone at
  // 3 TB/s: sixty-four different lines per instruction)
  auto load_a = [&](long rb, f64x2* dst) {
    const f64x2* src = reinterpret_cast<const f64x2*>(g.A + (rb * 16 + l15) * g.a_sm) + kg;
#pragma unroll
    for (int q = 0; q < AL; ++q) dst[q] = src[4 * q];
  };
  f64x2 bias2 = {0.0, 0.0};
  if constexpr (BIAS) bias2 = *reinterpret_cast<const f64x2*>(g.bias + n0 + 2 * lane);
  const double alpha = g.alpha;
  double* cbase = g.C + n0 + 2 * lane;

  // One block: 2 KS x 8 MFMAs of block `rb` into `ac`, with the previous block (`ap`, rows from rbp * 16) leaving UNDER
  // them -- two waves per SIMD do not hide the way out (they fall into lockstep on the shared matrix pipe, and a wave
  // that streams MFMAs starves its neighbour's other instructions).  D register r of tile j = row 4 r + kg, column
  // 16 j + l15.  Pass r: eight ds_write_b64 put rows 4 r .. 4 r + 3 into the wave's strip, four ds_read_b128 take them
  // back as whole 1 KiB rows, four stores.  Pinned: one MFMA, at most one other instruction.
  // MFMA steps 2q, 2q + 1: lane (x, kg) uses k = 8 q + 2 kg (+1) -> chunk 4 q + kg of column n (A and B agree); the
  // eight column tiles take each step in turn, so that consecutive MFMAs never share an accumulator; B fragments a step
  // pair ahead.
  auto block = [&](auto drain_on, f64x4 (&ac)[8], const f64x4 (&ap)[8], const f64x2 (&a)[AL], long rbp) {
    constexpr bool DRAIN = decltype(drain_on)::value;
    constexpr int NM = 2 * AL * 8;           // MFMAs per block
    constexpr int PER_PASS = NM / 4;         // MFMA slots per pass of the way out (K = 64: 32)
    static_assert(!DRAIN || PER_PASS >= 16, "a pass needs 16 slots (K >= 32)");
    f64x2 bq[2][8];
    f64x2 rv[4];
    auto load_b = [&](int q, int j, f64x2* dst) {
      const int n = j * 16 + l15;
      dst[j] = *reinterpret_cast<const f64x2*>(Bs + n * K + 2 * ((4 * q + kg) ^ (n & (CH - 1))));
    };
#pragma unroll
    for (int j = 0; j < 8; ++j) load_b(0, j, bq[0]);
    double* crow = cbase + rbp * 16 * g.c_sm;
    sk64_static_for<0, NM>([&](auto mi) {   // (a guaranteed full unroll: every register-array index is a constant)
      constexpr int m = decltype(mi)::value;
      constexpr int q = m / 16, e = (m / 8) & 1, j = m % 8;
      if constexpr (q == 0 && e == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ac[j][r] = 0.0;
      }
      ac[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(e ? a[q].y : a[q].x, e ? bq[q & 1][j].y : bq[q & 1][j].x, ac[j], 0, 0, 0);
      // the next step pair's B fragments: behind the first eight MFMAs of this pair
      if constexpr (e == 0 && q + 1 < AL) load_b(q + 1, j, bq[(q + 1) & 1]);
      if constexpr (DRAIN) {
        if constexpr (e == 1 || q + 1 == AL) {   // (slots without a B load; the last pair has none at all)
          // slot number among the drain slots of this block
          constexpr int ds = (q + 1 == AL) ? (AL - 1) * 8 + (m - (AL - 1) * 16) : q * 8 + j;
          constexpr int NDS = (AL - 1) * 8 + 16;          // drain slots per block (K = 64: 72)
          constexpr int SPP = NDS / 4;                    // ... per pass (18)
          static_assert(SPP >= 16, "a pass needs 16 slots");
          constexpr int r = ds / SPP, u = ds % SPP;
          if constexpr (r < 4) {
            if constexpr (u < 8) strip[kg * 128 + u * 16 + l15] = ap[u][r];
            else if constexpr (u < 12) rv[u - 8] = *reinterpret_cast<const f64x2*>(strip + (u - 8) * 128 + 2 * lane);
            else if constexpr (u < 16) {
              f64x2 v = rv[u - 12] * alpha;
              if constexpr (BIAS) v += bias2;
              if constexpr (ACT == 1) {
                v.x = logistic64_tab(v.x, tab);
                v.y = logistic64_tab(v.y, tab);
              }
              f64x2* dst = reinterpret_cast<f64x2*>(crow + (4 * r + (u - 12)) * g.c_sm);
              if constexpr (NT) __builtin_nontemporal_store(v, dst);
              else *dst = v;
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // the way out on its own (the last block of a stream)
  auto drain = [&](const f64x4 (&ap)[8], long rbp) {
    double* crow = cbase + rbp * 16 * g.c_sm;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int j = 0; j < 8; ++j) strip[kg * 128 + j * 16 + l15] = ap[j][r];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f64x2 v = *reinterpret_cast<const f64x2*>(strip + q * 128 + 2 * lane) * alpha;
        if constexpr (BIAS) v += bias2;
        if constexpr (ACT == 1) {
          v.x = logistic64_tab(v.x, tab);
          v.y = logistic64_tab(v.y, tab);
        }
        f64x2* dst = reinterpret_cast<f64x2*>(crow + (4 * r + q) * g.c_sm);
        if constexpr (NT) __builtin_nontemporal_store(v, dst);
        else *dst = v;
      }
    }
  };

  // ping-pong: block i into acc0 while acc1 (block i - 1) leaves, then the other way round; A a block ahead
  f64x4 acc0[8], acc1[8];
  f64x2 a0[AL], a1[AL];
  long rb = stream;
  if (rb >= g.nrb) return;
  load_a(rb, a0);
  if (rb + nstreams < g.nrb) load_a(rb + nstreams, a1);
  block(std::false_type{}, acc0, acc1, a0, 0);
  long rbp = rb;
  rb += nstreams;
  while (true) {
    if (rb >= g.nrb) { drain(acc0, rbp); break; }
    if (rb + nstreams < g.nrb) load_a(rb + nstreams, a0);
    block(std::true_type{}, acc1, acc0, a1, rbp);
    rbp = rb;
    rb += nstreams;
    if (rb >= g.nrb) { drain(acc1, rbp); break; }
    if (rb + nstreams < g.nrb) load_a(rb + nstreams, a1);
    block(std::true_type{}, acc0, acc1, a0, rbp);
    rbp = rb;
    rb += nstreams;
  }
}

bool gemm_skinnyk64_applicable(const GemmProblem& p) {
  static const int enable = [] { const char* e = ab_getenv("TOPS_GEMM64_SKINNYK"); return e ? atoi(e) : 1; }();
  if (!enable || p.dtype != TO_F64 || p.batch != 1 || p.reduce_batch) return false;
  if (p.K != 64) return false;   // (K = 32 leaves 40 slots for the 64 instructions of the way out: not built)
  if (p.N % 128 != 0 || p.N < 128 || p.N > 128 * 64) return false;
  if (p.M % 16 != 0 || p.M * p.N < (1LL << 24) || p.M / 16 < 4096) return false;  // a long stream of whole 16-row blocks
  if (p.a_sk != 1 || p.a_sm % 2 != 0 || (reinterpret_cast<uintptr_t>(p.A) & 15u)) return false;
  if (p.c_sm < p.N || p.c_sm % 2 != 0 || (reinterpret_cast<uintptr_t>(p.C) & 15u)) return false;
  if (p.beta != 0.0 || p.dact || p.rowsum || p.loss_rows || p.act > 1) return false;
  return true;
}

void launch_gemm_skinnyk64(const GemmProblem& p, hipStream_t s) {
  Skinny64Args g{};
  g.A = (const double*)p.A; g.B = (const double*)p.B; g.C = (double*)p.C;
  g.bias = (const double*)p.bias;
  g.M = p.M; g.N = (int)p.N; g.K = (int)p.K;
  g.a_sm = p.a_sm; g.b_sk = p.b_sk; g.b_sn = p.b_sn; g.c_sm = p.c_sm;
  g.alpha = p.alpha;
  g.npanels = (int)(p.N / 128);
  g.nrb = p.M / 16;
  g.act = p.act;
  bool nt = p.M * p.N * 8 > (256LL << 20);   // an output larger than the caches
  if (const char* e = ab_getenv("TOPS_SK64_NT")) nt = atoi(e) != 0;
  g.nt = nt;
  const size_t lds = ((size_t)128 * p.K + 4 * 4 * 128 + (p.act ? 1024 : 0)) * 8;
  const int grid = 256 / g.npanels * g.npanels;  // whole panels' worth of workgroups, one per CU
  static bool attr_set[32] = {false};
  auto launch = [&](auto kern, int which) {
    if (!attr_set[which]) {
      TO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set[which] = true;
    }
    launch_k(kern, dim3(grid), dim3(256), lds, s, g);
  };
  const bool bias = p.bias != nullptr;
  const int which = (p.act ? 4 : 0) + (bias ? 2 : 0) + (nt ? 1 : 0);
#define SK64(KS_, ACT_, BIAS_, NT_) launch(gemm_skinnyk64_kernel<KS_, ACT_, BIAS_, NT_>, which)
#define SK64_K(KS_)                                                                              \
  do {                                                                                           \
    if (p.act) { if (bias) { if (nt) SK64(KS_, 1, true, true); else SK64(KS_, 1, true, false); }  \
                 else      { if (nt) SK64(KS_, 1, false, true); else SK64(KS_, 1, false, false); } } \
    else       { if (bias) { if (nt) SK64(KS_, 0, true, true); else SK64(KS_, 0, true, false); }  \
                 else      { if (nt) SK64(KS_, 0, false, true); else SK64(KS_, 0, false, false); } } \
  } while (0)
  SK64_K(16);
#undef SK64_K
#undef SK64
  TO_HIP(hipGetLastError());
  count_launch();
}

}  // namespace to
