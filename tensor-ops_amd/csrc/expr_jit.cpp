// Run-time specialisation of the elementwise kernel skeleton for closures that match none of
// the pre-fused functors: the reified SSA program (to_expr_compile) is printed as the body of
// a device function and compiled once with hiprtc for gfx950, giving arbitrary
// `forall a. RealFloat a =>` closures (src/TensorOps/Types.hs:114-117) the same float4 /
// two-quads-in-flight / nontemporal kernels the named functors get.  The bytecode VM
// (ewise.hip) stays as the fallback when compilation is disabled or fails.
#include <hip/hiprtc.h>

#include <cmath>
#include <cstdio>
#include <sstream>

#include "common.hpp"

namespace to {

struct JitKernels {
  hipModule_t mod = nullptr;
  hipFunction_t vec = nullptr, vecnt = nullptr, scalar = nullptr;
};

static std::string lit(double c) {
  const float f = (float)c;
  if (std::isnan(f)) return "__builtin_nanf(\"\")";
  if (std::isinf(f)) return f > 0 ? "__builtin_inff()" : "(-__builtin_inff())";
  char buf[64];
  std::snprintf(buf, sizeof buf, "%.9gf", (double)f);
  std::string s(buf);
  if (s.find('.') == std::string::npos && s.find('e') == std::string::npos && s.find("inf") == std::string::npos)
    s.insert(s.size() - 1, ".0");
  return s;
}

static std::string body(const to_expr_s& e) {
  std::ostringstream o;
  const int n = (int)(e.code.size() / 3);
  for (int i = 0; i < e.arity; ++i) o << "  const float v" << i << " = x[" << i << "];\n";
  for (int i = 0; i < n; ++i) {
    const int op = e.code[3 * i], a = e.code[3 * i + 1], b = e.code[3 * i + 2];
    const std::string A = "v" + std::to_string(a), B = "v" + std::to_string(b);
    o << "  const float v" << (e.arity + i) << " = ";
    switch (op) {
      case TO_X_CONST: o << lit(e.consts[a]); break;
      case TO_X_ADD: o << A << " + " << B; break;
      case TO_X_SUB: o << A << " - " << B; break;
      case TO_X_MUL: o << A << " * " << B; break;
      case TO_X_DIV: o << A << " / " << B; break;
      case TO_X_NEG: o << "-" << A; break;
      case TO_X_RECIP: o << "1.0f / " << A; break;
      case TO_X_EXP: o << "expf(" << A << ")"; break;
      case TO_X_LOG: o << "logf(" << A << ")"; break;
      case TO_X_SQRT: o << "sqrtf(" << A << ")"; break;
      case TO_X_ABS: o << "fabsf(" << A << ")"; break;
      case TO_X_SIGNUM: o << "(" << A << " > 0.f) ? 1.f : ((" << A << " < 0.f) ? -1.f : " << A << ")"; break;
      case TO_X_SIN: o << "sinf(" << A << ")"; break;
      case TO_X_COS: o << "cosf(" << A << ")"; break;
      case TO_X_TANH: o << "tanhf(" << A << ")"; break;
      case TO_X_POW: o << "powf(" << A << ", " << B << ")"; break;
      case TO_X_MAX: o << "fmaxf(" << A << ", " << B << ")"; break;
      case TO_X_MIN: o << "fminf(" << A << ", " << B << ")"; break;
      default: o << "__builtin_nanf(\"\")"; break;
    }
    o << ";\n";
  }
  o << "  return v" << (e.arity + n - 1) << ";\n";
  return o.str();
}

static std::string source(const to_expr_s& e) {
  std::ostringstream o;
  const int N = e.arity;
  o << "typedef float f32x4 __attribute__((ext_vector_type(4)));\n"
       "struct P { const float* x[8]; long period[8]; };\n"
       "__device__ __forceinline__ float F(const float* x) {\n"
    << body(e)
    << "}\n"
       "#define N " << N << "\n"
       "#define NN " << (N > 0 ? N : 1) << "\n"
    << R"SRC(
extern "C" __global__ __launch_bounds__(256) void ew_vec(P p, float* __restrict__ out, long total4, long total) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += stride) {
    const long e = q * 4;
    f32x4 v[NN];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const long ei = (p.period[i] == total) ? e : (e % p.period[i]);
      v[i] = *reinterpret_cast<const f32x4*>(p.x[i] + ei);
    }
    float xin[4][NN];
#pragma unroll
    for (int i = 0; i < N; ++i) { xin[0][i] = v[i].x; xin[1][i] = v[i].y; xin[2][i] = v[i].z; xin[3][i] = v[i].w; }
    f32x4 r;
    r.x = F(xin[0]); r.y = F(xin[1]); r.z = F(xin[2]); r.w = F(xin[3]);
    *reinterpret_cast<f32x4*>(out + e) = r;
  }
}
extern "C" __global__ __launch_bounds__(256) void ew_vecnt(P p, float* __restrict__ out, long total4, long total) {
  const long stride = (long)gridDim.x * blockDim.x;
  const long start = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long q = start; q < total4; q += 2 * stride) {
    const long q2 = q + stride;
    const bool two = q2 < total4;
    f32x4 v[2][NN];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long e = (u == 0 ? q : (two ? q2 : q)) * 4;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const long ei = (p.period[i] == total) ? e : (e % p.period[i]);
        v[u][i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.x[i] + ei));
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
      float xin[4][NN];
#pragma unroll
      for (int i = 0; i < N; ++i) { xin[0][i] = v[u][i].x; xin[1][i] = v[u][i].y; xin[2][i] = v[u][i].z; xin[3][i] = v[u][i].w; }
      f32x4 r;
      r.x = F(xin[0]); r.y = F(xin[1]); r.z = F(xin[2]); r.w = F(xin[3]);
      __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(out + (u == 0 ? q : q2) * 4));
    }
  }
}
extern "C" __global__ __launch_bounds__(256) void ew_scalar(P p, float* __restrict__ out, long total4, long total) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    float xin[NN];
#pragma unroll
    for (int i = 0; i < N; ++i) xin[i] = p.x[i][(p.period[i] == total) ? e : (e % p.period[i])];
    out[e] = F(xin);
  }
}
)SRC";
  return o.str();
}

void* jit_build(const to_expr_s& e, std::string* err) {
  const std::string src = source(e);
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, src.c_str(), "tensorops_expr.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
    if (err) *err = "hiprtcCreateProgram failed";
    return nullptr;
  }
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off"};
  const hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
  if (r != HIPRTC_SUCCESS) {
    size_t n = 0;
    hiprtcGetProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) hiprtcGetProgramLog(prog, &log[0]);
    if (err) *err = "hiprtc: " + log;
    hiprtcDestroyProgram(&prog);
    return nullptr;
  }
  size_t sz = 0;
  hiprtcGetCodeSize(prog, &sz);
  std::vector<char> code(sz);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  auto* k = new JitKernels();
  if (hipModuleLoadData(&k->mod, code.data()) != hipSuccess ||
      hipModuleGetFunction(&k->vec, k->mod, "ew_vec") != hipSuccess ||
      hipModuleGetFunction(&k->vecnt, k->mod, "ew_vecnt") != hipSuccess ||
      hipModuleGetFunction(&k->scalar, k->mod, "ew_scalar") != hipSuccess) {
    if (err) *err = "hipModuleLoadData/GetFunction failed";
    if (k->mod) (void)hipModuleUnload(k->mod);
    delete k;
    return nullptr;
  }
  return k;
}

void jit_release(void* h) {
  auto* k = static_cast<JitKernels*>(h);
  if (!k) return;
  if (k->mod) (void)hipModuleUnload(k->mod);
  delete k;
}

std::string jit_source_for_tests(const to_expr_s& e) { return source(e); }

void jit_launch(void* h, const EwArgs& a, hipStream_t s) {
  auto* k = static_cast<JitKernels*>(h);
  struct { const float* x[8]; long period[8]; } p{};
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
  bool vec = (a.total % 4 == 0) && al16(a.out);
  for (int i = 0; i < a.n; ++i) {
    p.x[i] = a.x[i];
    p.period[i] = a.period[i];
    vec = vec && al16(a.x[i]) && (a.period[i] % 4 == 0);
  }
  float* out = a.out;
  long total = a.total, total4 = a.total / 4;
  void* args[] = {&p, &out, &total4, &total};
  hipFunction_t f;
  long blocks;
  if (vec) {
    const bool streaming = a.total >= (16L << 20);
    f = streaming ? k->vecnt : k->vec;
    blocks = (total4 + 255) / 256;
    const long cap = streaming ? 16384 : 2048;
    if (blocks > cap) blocks = cap;
  } else {
    f = k->scalar;
    blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
  }
  TO_HIP(hipModuleLaunchKernel(f, (unsigned)blocks, 1, 1, 256, 1, 1, 0, s, args, nullptr));
  count_launch();
}

}  // namespace to
