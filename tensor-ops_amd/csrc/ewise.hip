// Elementwise kernels: `liftT`/`liftB` (src/TensorOps/Types.hs:56-59,
// src/TensorOps/BLAS.hs:92-96), `scaleT`, `sumT`, `axpy`, the SGD step
// (src/TensorOps/Learn/NeuralNet/FeedForward.hs:141-147); float and double.
//
// HBM-bound: 16-byte loads/stores (4 floats or 2 doubles), grid-stride, every input either
// full-size or a shorter period broadcast over the hidden batch dimension.  Streams larger
// than the caches get two 16-byte pieces in flight per thread and nontemporal accesses
// (measured on `map logistic` over 512^3 fp32: 5.27 -> 5.9 TB/s).  Known closures
// (classified in expr.cpp) run as pre-fused functors; anything else is specialised at run
// time (expr_jit.cpp) or, failing that, runs on a small SSA bytecode VM whose value slots
// live in LDS (dynamic register indexing would go to scratch memory on gfx950).
#include <algorithm>

#include "common.hpp"

namespace to {

__device__ __forceinline__ float dexp(float x) { return expf(x); }
__device__ __forceinline__ double dexp(double x) { return exp(x); }
__device__ __forceinline__ float dlog(float x) { return logf(x); }
__device__ __forceinline__ double dlog(double x) { return log(x); }
__device__ __forceinline__ float dtanh(float x) { return tanhf(x); }
__device__ __forceinline__ double dtanh(double x) { return tanh(x); }
__device__ __forceinline__ float dsqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double dsqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float dfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double dfma(double a, double b, double c) { return fma(a, b, c); }

template <class S>
struct FAffine {
  S a[4];
  S c;
  int n;
  __device__ __forceinline__ S operator()(const S* x) const {
    S r = c;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < n) r = dfma(a[i], x[i], r);
    return r;
  }
};
template <class S> struct FMul { __device__ __forceinline__ S operator()(const S* x) const { return x[0] * x[1]; } };
template <class S> struct FDiv { __device__ __forceinline__ S operator()(const S* x) const { return x[0] / x[1]; } };
template <class S> struct FExp { __device__ __forceinline__ S operator()(const S* x) const { return dexp(x[0]); } };
template <class S> struct FLog { __device__ __forceinline__ S operator()(const S* x) const { return dlog(x[0]); } };
template <class S> struct FRecip { __device__ __forceinline__ S operator()(const S* x) const { return S(1) / x[0]; } };
template <class S> struct FTanh { __device__ __forceinline__ S operator()(const S* x) const { return dtanh(x[0]); } };
template <class S> struct FSqrt { __device__ __forceinline__ S operator()(const S* x) const { return dsqrt(x[0]); } };
// logistic x = 1 / (1 + exp (-x))   (src/TensorOps/Learn/NeuralNet.hs:42-44)
template <class S> struct FLogistic {
  __device__ __forceinline__ S operator()(const S* x) const { return S(1) / (S(1) + dexp(-x[0])); }
};
// d * logistic'(x), logistic' x = s (1 - s)   (NeuralNet.hs:46-50; the `gradLift`
// form `\(d :* x) -> d * f' x`, src/TensorOps/Tensor.hs:127)
template <class S> struct FMulDLogistic {
  __device__ __forceinline__ S operator()(const S* x) const {
    const S s = S(1) / (S(1) + dexp(-x[1]));
    return x[0] * (s * (S(1) - s));
  }
};
// d * h (1 - h) with h = logistic(x) already computed: what `d * logistic'(x)` becomes once the
// planner (lazy.cpp) has found the forward value h in the recorded graph
template <class S> struct FMulHOneMinusH {
  __device__ __forceinline__ S operator()(const S* x) const { return x[0] * (x[1] * (S(1) - x[1])); }
};
// the same pair for tanh: d * tanh'(x) = d (1 - tanh(x)^2), and on h = tanh(x) already computed
template <class S> struct FMulDTanh {
  __device__ __forceinline__ S operator()(const S* x) const {
    const S t = dtanh(x[1]);
    return x[0] * (S(1) - t * t);
  }
};
template <class S> struct FMulOneMinusH2 {
  __device__ __forceinline__ S operator()(const S* x) const { return x[0] * (S(1) - x[1] * x[1]); }
};
template <class S> struct FConst {
  S c;
  __device__ __forceinline__ S operator()(const S*) const { return c; }
};

template <class S>
struct EwPtrs {
  const S* x[4];
  long period[4];
};

// V = elements per 16-byte piece (4 floats / 2 doubles); totalv = total / V
template <class S, int N, class F>
__global__ __launch_bounds__(256) void ew_vec_kernel(EwPtrs<S> p, S* __restrict__ out, long totalv,
                                                     long total, F f) {
  constexpr int V = 16 / sizeof(S);
  typedef S vec __attribute__((ext_vector_type(V)));
  const long stride = (long)gridDim.x * blockDim.x;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < totalv; q += stride) {
    const long e = q * V;
    vec v[N > 0 ? N : 1];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const long ei = (p.period[i] == total) ? e : (e % p.period[i]);
      v[i] = *reinterpret_cast<const vec*>(p.x[i] + ei);
    }
    vec r;
#pragma unroll
    for (int c = 0; c < V; ++c) {
      S xin[N > 0 ? N : 1];
#pragma unroll
      for (int i = 0; i < N; ++i) xin[i] = v[i][c];
      r[c] = f(xin);
    }
    *reinterpret_cast<vec*>(out + e) = r;
  }
}

// two independent pieces per thread per trip + nontemporal accesses: streaming sizes
template <class S, int N, class F>
__global__ __launch_bounds__(256) void ew_stream_kernel(EwPtrs<S> p, S* __restrict__ out, long totalv,
                                                        long total, F f) {
  constexpr int V = 16 / sizeof(S);
  typedef S vec __attribute__((ext_vector_type(V)));
  const long stride = (long)gridDim.x * blockDim.x;
  const long start = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long q = start; q < totalv; q += 2 * stride) {
    const long q2 = q + stride;
    const bool two = q2 < totalv;
    vec v[2][N > 0 ? N : 1];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long e = (u == 0 ? q : (two ? q2 : q)) * V;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const long ei = (p.period[i] == total) ? e : (e % p.period[i]);
        v[u][i] = __builtin_nontemporal_load(reinterpret_cast<const vec*>(p.x[i] + ei));
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
      vec r;
#pragma unroll
      for (int c = 0; c < V; ++c) {
        S xin[N > 0 ? N : 1];
#pragma unroll
        for (int i = 0; i < N; ++i) xin[i] = v[u][i][c];
        r[c] = f(xin);
      }
      __builtin_nontemporal_store(r, reinterpret_cast<vec*>(out + (u == 0 ? q : q2) * V));
    }
  }
}

template <class S, int N, class F>
__global__ __launch_bounds__(256) void ew_scalar_kernel(EwPtrs<S> p, S* __restrict__ out, long total, F f) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    S xin[N > 0 ? N : 1];
#pragma unroll
    for (int i = 0; i < N; ++i) xin[i] = p.x[i][(p.period[i] == total) ? e : (e % p.period[i])];
    out[e] = f(xin);
  }
}

// ---- SSA bytecode VM: value slots in LDS, slot s of thread t at lds[s*256 + t] ----------
template <class S>
struct VmIO {
  const S* x[8];
  long period[8];
};

template <class S>
__global__ __launch_bounds__(256) void ew_vm_kernel(VmIO<S> io, int n_in, const int32_t* __restrict__ code,
                                                    const S* __restrict__ consts, int n_instr,
                                                    int result_slot, S* __restrict__ out, long total) {
  extern __shared__ __attribute__((aligned(16))) char slots_raw[];
  S* slots = reinterpret_cast<S*>(slots_raw);
  const int t = threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long e = (long)blockIdx.x * blockDim.x + t; e < total; e += stride) {
    for (int i = 0; i < n_in; ++i) {
      const long pi = io.period[i];
      slots[i * 256 + t] = io.x[i][pi == total ? e : e % pi];
    }
    for (int i = 0; i < n_instr; ++i) {
      const int op = code[4 * i], d = code[4 * i + 1], ia = code[4 * i + 2], ib = code[4 * i + 3];
      S r;
      if (op == TO_X_CONST) {
        r = consts[ia];
      } else {
        const S a = slots[ia * 256 + t];
        const S b = slots[ib * 256 + t];
        switch (op) {
          case TO_X_ADD: r = a + b; break;
          case TO_X_SUB: r = a - b; break;
          case TO_X_MUL: r = a * b; break;
          case TO_X_DIV: r = a / b; break;
          case TO_X_NEG: r = -a; break;
          case TO_X_RECIP: r = S(1) / a; break;
          case TO_X_EXP: r = dexp(a); break;
          case TO_X_LOG: r = dlog(a); break;
          case TO_X_SQRT: r = dsqrt(a); break;
          case TO_X_ABS: r = a < S(0) ? -a : a; break;
          case TO_X_SIGNUM: r = (a > S(0)) ? S(1) : ((a < S(0)) ? S(-1) : a); break;
          case TO_X_SIN: r = sin(a); break;
          case TO_X_COS: r = cos(a); break;
          case TO_X_TANH: r = dtanh(a); break;
          case TO_X_POW: r = pow(a, b); break;
          case TO_X_MAX: r = fmax(a, b); break;
          case TO_X_MIN: r = fmin(a, b); break;
          default: r = S(__builtin_nanf("")); break;
        }
      }
      slots[d * 256 + t] = r;
    }
    out[e] = slots[result_slot * 256 + t];
  }
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <class S, int N, class F>
static void run(const EwArgs& a, F f, hipStream_t s) {
  if (a.total == 0) return;
  constexpr int V = 16 / sizeof(S);
  EwPtrs<S> p{};
  bool vec = (a.total % V == 0) && al16(a.out);
  for (int i = 0; i < N; ++i) {
    p.x[i] = static_cast<const S*>(a.x[i]);
    p.period[i] = a.period[i];
    vec = vec && al16(a.x[i]) && (a.period[i] % V == 0);
  }
  S* out = static_cast<S*>(a.out);
  // TOPS_EW_MODE (0 plain, 2 streaming) / TOPS_EW_BLOCKS override the policy (tuning knobs)
  static const int ew_mode_env = [] { const char* e = ab_getenv("TOPS_EW_MODE"); return e ? atoi(e) : -1; }();
  static const int ew_blocks_env = [] { const char* e = ab_getenv("TOPS_EW_BLOCKS"); return e ? atoi(e) : 0; }();
  if (vec) {
    const long totalv = a.total / V;
    const bool streaming = a.total * (long)sizeof(S) >= (64L << 20);
    const int mode = ew_mode_env >= 0 ? ew_mode_env : (streaming ? 2 : 0);
    // (512^3 map: 16384 -> 5.88, 24576..49152 -> 5.95-6.05 TB/s.  Round 6, tools/ew_time.py: beyond 2^25 quads a thread should make ONE
    //  trip of two pieces -- 5 x 10^8 floats 5.33 -> 6.30 TB/s, 10^9 5.56 -> 6.35 -- and with two inputs one piece: a + b over 512^3 5.31 ->
    //  6.47, over 10^9 5.07 -> 6.06)
    const long stream_cap = N >= 2 ? std::min<long>((totalv + 255) / 256, 1L << 20) : (totalv <= (1L << 25) ? 32768 : std::min<long>((totalv + 511) / 512, 1L << 20));
    const long cap = ew_blocks_env > 0 ? ew_blocks_env : (streaming ? stream_cap : 2048);
    long blocks = (totalv + 255) / 256;
    if (blocks > cap) blocks = cap;
    if (mode == 2 && totalv >= (1 << 20))
      launch_k((ew_stream_kernel<S, N, F>), dim3((unsigned)blocks), dim3(256), 0, s, p, out, totalv,
                         (long)a.total, f);
    else
      launch_k((ew_vec_kernel<S, N, F>), dim3((unsigned)blocks), dim3(256), 0, s, p, out, totalv,
                         (long)a.total, f);
  } else {
    long blocks = (a.total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    launch_k((ew_scalar_kernel<S, N, F>), dim3((unsigned)blocks), dim3(256), 0, s, p, out,
                       (long)a.total, f);
  }
  TO_HIP(hipGetLastError());
  count_launch();
}

template <class S>
static void launch_ewise_t(const EwArgs& a, hipStream_t s) {
  switch (a.kind) {
    case EW_AFFINE: {
      FAffine<S> f{};
      for (int i = 0; i < 4; ++i) f.a[i] = (S)a.coef[i];
      f.c = (S)a.c0;
      f.n = a.n;
      switch (a.n) {
        case 1: run<S, 1>(a, f, s); break;
        case 2: run<S, 2>(a, f, s); break;
        case 3: run<S, 3>(a, f, s); break;
        case 4: run<S, 4>(a, f, s); break;
        default: fail(TO_ERR_ARG, "affine arity");
      }
      return;
    }
    case EW_CONST: run<S, 0>(a, FConst<S>{(S)a.c0}, s); return;
    case EW_MUL: run<S, 2>(a, FMul<S>{}, s); return;
    case EW_DIV: run<S, 2>(a, FDiv<S>{}, s); return;
    case EW_EXP: run<S, 1>(a, FExp<S>{}, s); return;
    case EW_LOG: run<S, 1>(a, FLog<S>{}, s); return;
    case EW_RECIP: run<S, 1>(a, FRecip<S>{}, s); return;
    case EW_TANH: run<S, 1>(a, FTanh<S>{}, s); return;
    case EW_SQRT: run<S, 1>(a, FSqrt<S>{}, s); return;
    case EW_LOGISTIC: run<S, 1>(a, FLogistic<S>{}, s); return;
    case EW_MUL_DLOGISTIC: run<S, 2>(a, FMulDLogistic<S>{}, s); return;
    case EW_MUL_H1MH: run<S, 2>(a, FMulHOneMinusH<S>{}, s); return;
    case EW_MUL_DTANH: run<S, 2>(a, FMulDTanh<S>{}, s); return;
    case EW_MUL_1MH2: run<S, 2>(a, FMulOneMinusH2<S>{}, s); return;
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
  TO_CHECK(a.n_slots * sizeof(S) <= 96 * 4, TO_ERR_UNSUPPORTED, "expression needs too many live values for the VM");
  VmIO<S> io{};
  for (int i = 0; i < a.n; ++i) {
    io.x[i] = static_cast<const S*>(a.x[i]);
    io.period[i] = a.period[i];
  }
  long blocks = (a.total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const size_t lds = (size_t)a.n_slots * 256 * sizeof(S);
  launch_k(ew_vm_kernel<S>, dim3((unsigned)blocks), dim3(256), lds, s, io, a.n, a.d_code,
                     static_cast<const S*>(a.d_consts), a.n_instr, a.result_slot, static_cast<S*>(a.out),
                     (long)a.total);
  TO_HIP(hipGetLastError());
  count_launch();
}

void launch_ewise(const EwArgs& a, hipStream_t s) {
  if (a.dtype == TO_F64) launch_ewise_t<double>(a, s);
  else launch_ewise_t<float>(a, s);
}

template <class S>
__global__ void sgd_kernel(S* __restrict__ p, const S* __restrict__ g, S r, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = p[i] - r * g[i];
}

void launch_sgd(int dtype, void* p, const void* g, double r, int64_t n, hipStream_t s) {
  if (n == 0) return;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (dtype == TO_F64)
    launch_k(sgd_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, s, (double*)p, (const double*)g, r,
                       (long)n);
  else
    launch_k(sgd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (float*)p, (const float*)g,
                       (float)r, (long)n);
  TO_HIP(hipGetLastError());
  count_launch();
}

}  // namespace to
