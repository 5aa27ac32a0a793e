// Elementwise kernels: `liftT`/`liftB` (src/TensorOps/Types.hs:56-59,
// src/TensorOps/BLAS.hs:92-96), `scaleT`, `sumT`, `axpy`, the SGD step
// (src/TensorOps/Learn/NeuralNet/FeedForward.hs:141-147).
//
// HBM-bound: 16-byte (dwordx4) loads/stores, grid-stride over at most 2048
// workgroups (256 CUs x 8), every input either full-size or a shorter period
// broadcast over the hidden batch dimension.  Known closures (classified in
// expr.cpp) run as pre-fused functors; anything else runs on a small SSA
// bytecode VM whose value slots live in LDS (dynamic register indexing would
// go to scratch memory on gfx950).
#include "common.hpp"

namespace to {

struct FAffine {
  float a[4];
  float c;
  int n;
  __device__ __forceinline__ float operator()(const float* x) const {
    float r = c;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < n) r = fmaf(a[i], x[i], r);
    return r;
  }
};
struct FMul { __device__ __forceinline__ float operator()(const float* x) const { return x[0] * x[1]; } };
struct FDiv { __device__ __forceinline__ float operator()(const float* x) const { return x[0] / x[1]; } };
struct FExp { __device__ __forceinline__ float operator()(const float* x) const { return expf(x[0]); } };
struct FLog { __device__ __forceinline__ float operator()(const float* x) const { return logf(x[0]); } };
struct FRecip { __device__ __forceinline__ float operator()(const float* x) const { return 1.0f / x[0]; } };
struct FTanh { __device__ __forceinline__ float operator()(const float* x) const { return tanhf(x[0]); } };
struct FSqrt { __device__ __forceinline__ float operator()(const float* x) const { return sqrtf(x[0]); } };
// logistic x = 1 / (1 + exp (-x))   (src/TensorOps/Learn/NeuralNet.hs:42-44)
struct FLogistic {
  __device__ __forceinline__ float operator()(const float* x) const { return 1.0f / (1.0f + expf(-x[0])); }
};
// d * logistic'(x), logistic' x = s (1 - s)   (NeuralNet.hs:46-50; the `gradLift`
// form `\(d :* x) -> d * f' x`, src/TensorOps/Tensor.hs:127)
struct FMulDLogistic {
  __device__ __forceinline__ float operator()(const float* x) const {
    const float s = 1.0f / (1.0f + expf(-x[1]));
    return x[0] * (s * (1.0f - s));
  }
};
struct FConst {
  float c;
  __device__ __forceinline__ float operator()(const float*) const { return c; }
};

struct EwPtrs {
  const float* x[4];
  long period[4];
};

template <int N, class F>
__global__ __launch_bounds__(256) void ew_vec4_kernel(EwPtrs p, float* __restrict__ out, long total4,
                                                      long total, F f) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += stride) {
    const long e = q * 4;
    float4 v[N > 0 ? N : 1];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const long ei = (p.period[i] == total) ? e : (e % p.period[i]);
      v[i] = *reinterpret_cast<const float4*>(p.x[i] + ei);
    }
    float xin[4][N > 0 ? N : 1];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      xin[0][i] = v[i].x; xin[1][i] = v[i].y; xin[2][i] = v[i].z; xin[3][i] = v[i].w;
    }
    float4 r;
    r.x = f(xin[0]); r.y = f(xin[1]); r.z = f(xin[2]); r.w = f(xin[3]);
    *reinterpret_cast<float4*>(out + e) = r;
  }
}

// same, two independent quads per thread per trip (more bytes in flight per wave) and optional
// nontemporal accesses for streams larger than the caches
template <int N, class F, bool NT>
__global__ __launch_bounds__(256) void ew_vec4x2_kernel(EwPtrs p, float* __restrict__ out, long total4,
                                                        long total, F f) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const long stride = (long)gridDim.x * blockDim.x;
  const long start = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long q = start; q < total4; q += 2 * stride) {
    const long q2 = q + stride;
    const bool two = q2 < total4;
    f32x4 v[2][N > 0 ? N : 1];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long e = (u == 0 ? q : (two ? q2 : q)) * 4;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const long ei = (p.period[i] == total) ? e : (e % p.period[i]);
        const f32x4* src = reinterpret_cast<const f32x4*>(p.x[i] + ei);
        v[u][i] = NT ? __builtin_nontemporal_load(src) : *src;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
      float xin[4][N > 0 ? N : 1];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        xin[0][i] = v[u][i].x; xin[1][i] = v[u][i].y; xin[2][i] = v[u][i].z; xin[3][i] = v[u][i].w;
      }
      f32x4 r;
      r.x = f(xin[0]); r.y = f(xin[1]); r.z = f(xin[2]); r.w = f(xin[3]);
      f32x4* dst = reinterpret_cast<f32x4*>(out + (u == 0 ? q : q2) * 4);
      if (NT) __builtin_nontemporal_store(r, dst);
      else *dst = r;
    }
  }
}

template <int N, class F>
__global__ __launch_bounds__(256) void ew_scalar_kernel(EwPtrs p, float* __restrict__ out, long total,
                                                        F f) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    float xin[N > 0 ? N : 1];
#pragma unroll
    for (int i = 0; i < N; ++i) xin[i] = p.x[i][(p.period[i] == total) ? e : (e % p.period[i])];
    out[e] = f(xin);
  }
}

// ---- SSA bytecode VM: value slots in LDS, slot s of thread t at lds[s*256 + t] ----------
struct VmIO {
  const float* x[8];
  long period[8];
};

__global__ __launch_bounds__(256) void ew_vm_kernel(VmIO io, int n_in,
                                                    const int32_t* __restrict__ code,
                                                    const float* __restrict__ consts, int n_instr,
                                                    int result_slot, float* __restrict__ out,
                                                    long total) {
  extern __shared__ __attribute__((aligned(16))) float slots[];
  const int t = threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long e = (long)blockIdx.x * blockDim.x + t; e < total; e += stride) {
    for (int i = 0; i < n_in; ++i) {
      const long pi = io.period[i];
      slots[i * 256 + t] = io.x[i][pi == total ? e : e % pi];
    }
    for (int i = 0; i < n_instr; ++i) {
      const int op = code[4 * i], d = code[4 * i + 1], ia = code[4 * i + 2], ib = code[4 * i + 3];
      float r;
      if (op == TO_X_CONST) {
        r = consts[ia];
      } else {
        const float a = slots[ia * 256 + t];
        const float b = slots[ib * 256 + t];
        switch (op) {
          case TO_X_ADD: r = a + b; break;
          case TO_X_SUB: r = a - b; break;
          case TO_X_MUL: r = a * b; break;
          case TO_X_DIV: r = a / b; break;
          case TO_X_NEG: r = -a; break;
          case TO_X_RECIP: r = 1.0f / a; break;
          case TO_X_EXP: r = expf(a); break;
          case TO_X_LOG: r = logf(a); break;
          case TO_X_SQRT: r = sqrtf(a); break;
          case TO_X_ABS: r = fabsf(a); break;
          case TO_X_SIGNUM: r = (a > 0.f) ? 1.f : ((a < 0.f) ? -1.f : a); break;
          case TO_X_SIN: r = sinf(a); break;
          case TO_X_COS: r = cosf(a); break;
          case TO_X_TANH: r = tanhf(a); break;
          case TO_X_POW: r = powf(a, b); break;
          case TO_X_MAX: r = fmaxf(a, b); break;
          case TO_X_MIN: r = fminf(a, b); break;
          default: r = __builtin_nanf(""); break;
        }
      }
      slots[d * 256 + t] = r;
    }
    out[e] = slots[result_slot * 256 + t];
  }
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <int N, class F>
static void run(const EwArgs& a, F f, hipStream_t s) {
  if (a.total == 0) return;
  EwPtrs p{};
  bool vec = (a.total % 4 == 0) && al16(a.out);
  for (int i = 0; i < N; ++i) {
    p.x[i] = a.x[i];
    p.period[i] = a.period[i];
    vec = vec && al16(a.x[i]) && (a.period[i] % 4 == 0);
  }
  // Streams larger than the caches (> 64 MiB per operand): two quads in flight per thread and
  // nontemporal accesses (measured on `map logistic` over 512^3: 5.27 -> 5.99 TB/s); everything
  // else: plain loads, at most 2048 workgroups.  TOPS_EW_MODE / TOPS_EW_BLOCKS override (tuning).
  static const int ew_mode_env = [] { const char* e = getenv("TOPS_EW_MODE"); return e ? atoi(e) : -1; }();
  static const int ew_blocks_env = [] { const char* e = getenv("TOPS_EW_BLOCKS"); return e ? atoi(e) : 0; }();
  if (vec) {
    const long total4 = a.total / 4;
    const bool streaming = a.total >= (16L << 20);
    const int ew_mode = ew_mode_env >= 0 ? ew_mode_env : (streaming ? 2 : 0);
    const long ew_blocks = ew_blocks_env > 0 ? ew_blocks_env : (streaming ? 16384 : 2048);
    long blocks = (total4 + 255) / 256;
    if (blocks > ew_blocks) blocks = ew_blocks;
    if (ew_mode == 1 && total4 >= (1 << 20))
      hipLaunchKernelGGL((ew_vec4x2_kernel<N, F, false>), dim3((unsigned)blocks), dim3(256), 0, s, p, a.out,
                         total4, (long)a.total, f);
    else if (ew_mode == 2 && total4 >= (1 << 20))
      hipLaunchKernelGGL((ew_vec4x2_kernel<N, F, true>), dim3((unsigned)blocks), dim3(256), 0, s, p, a.out,
                         total4, (long)a.total, f);
    else
      hipLaunchKernelGGL((ew_vec4_kernel<N, F>), dim3((unsigned)blocks), dim3(256), 0, s, p, a.out,
                         total4, (long)a.total, f);
  } else {
    long blocks = (a.total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((ew_scalar_kernel<N, F>), dim3((unsigned)blocks), dim3(256), 0, s, p, a.out,
                       (long)a.total, f);
  }
  TO_HIP(hipGetLastError());
  count_launch();
}

void launch_ewise(const EwArgs& a, hipStream_t s) {
  switch (a.kind) {
    case EW_AFFINE: {
      FAffine f{};
      for (int i = 0; i < 4; ++i) f.a[i] = a.coef[i];
      f.c = a.c0;
      f.n = a.n;
      switch (a.n) {
        case 1: run<1>(a, f, s); break;
        case 2: run<2>(a, f, s); break;
        case 3: run<3>(a, f, s); break;
        case 4: run<4>(a, f, s); break;
        default: fail(TO_ERR_ARG, "affine arity");
      }
      return;
    }
    case EW_CONST: run<0>(a, FConst{a.c0}, s); return;
    case EW_MUL: run<2>(a, FMul{}, s); return;
    case EW_DIV: run<2>(a, FDiv{}, s); return;
    case EW_EXP: run<1>(a, FExp{}, s); return;
    case EW_LOG: run<1>(a, FLog{}, s); return;
    case EW_RECIP: run<1>(a, FRecip{}, s); return;
    case EW_TANH: run<1>(a, FTanh{}, s); return;
    case EW_SQRT: run<1>(a, FSqrt{}, s); return;
    case EW_LOGISTIC: run<1>(a, FLogistic{}, s); return;
    case EW_MUL_DLOGISTIC: run<2>(a, FMulDLogistic{}, s); return;
    case EW_VM: break;
    default: fail(TO_ERR_ARG, "unknown elementwise kind");
  }
  if (a.total == 0) return;
  if (a.jit) {  // the program was specialised at run time (expr_jit.cpp)
    jit_launch(a.jit, a, s);
    return;
  }
  // ---- VM ----
  TO_CHECK(a.n <= 8, TO_ERR_UNSUPPORTED, "VM arity > 8");
  TO_CHECK(a.n_slots <= 96, TO_ERR_UNSUPPORTED, "expression needs more than 96 live values");
  VmIO io{};
  for (int i = 0; i < a.n; ++i) {
    io.x[i] = a.x[i];
    io.period[i] = a.period[i];
  }
  long blocks = (a.total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const size_t lds = (size_t)a.n_slots * 256 * sizeof(float);
  hipLaunchKernelGGL(ew_vm_kernel, dim3((unsigned)blocks), dim3(256), lds, s, io, a.n, a.d_code,
                     a.d_consts, a.n_instr, a.result_slot, a.out, (long)a.total);
  TO_HIP(hipGetLastError());
  count_launch();
}

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float r, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    p[i] = p[i] - r * g[i];
}

void launch_sgd(float* p, const float* g, float r, int64_t n, hipStream_t s) {
  if (n == 0) return;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, g, r, (long)n);
  TO_HIP(hipGetLastError());
  count_launch();
}

}  // namespace to
