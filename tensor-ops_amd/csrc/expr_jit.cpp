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

static std::string lit(double c, bool f64) {
  const double v = f64 ? c : (double)(float)c;
  if (std::isnan(v)) return "S(__builtin_nanf(\"\"))";
  if (std::isinf(v)) return v > 0 ? "S(__builtin_inff())" : "S(-__builtin_inff())";
  char buf[64];
  std::snprintf(buf, sizeof buf, f64 ? "%.17g" : "%.9g", v);
  std::string s(buf);
  if (s.find('.') == std::string::npos && s.find('e') == std::string::npos) s += ".0";
  return f64 ? s : s + "f";
}

static std::string body(const to_expr_s& e, bool f64) {
  std::ostringstream o;
  const int n = (int)(e.code.size() / 3);
  const char* sfx = f64 ? "" : "f";  // expf vs exp ...
  for (int i = 0; i < e.arity; ++i) o << "  const S v" << i << " = x[" << i << "];\n";
  for (int i = 0; i < n; ++i) {
    const int op = e.code[3 * i], a = e.code[3 * i + 1], b = e.code[3 * i + 2];
    const std::string A = "v" + std::to_string(a), B = "v" + std::to_string(b);
    o << "  const S v" << (e.arity + i) << " = ";
    switch (op) {
      case TO_X_CONST: o << lit(e.consts[a], f64); break;
      case TO_X_ADD: o << A << " + " << B; break;
      case TO_X_SUB: o << A << " - " << B; break;
      case TO_X_MUL: o << A << " * " << B; break;
      case TO_X_DIV: o << A << " / " << B; break;
      case TO_X_NEG: o << "-" << A; break;
      case TO_X_RECIP: o << "S(1) / " << A; break;
      case TO_X_EXP: o << "exp" << sfx << "(" << A << ")"; break;
      case TO_X_LOG: o << "log" << sfx << "(" << A << ")"; break;
      case TO_X_SQRT: o << "sqrt" << sfx << "(" << A << ")"; break;
      case TO_X_ABS: o << "fabs" << sfx << "(" << A << ")"; break;
      case TO_X_SIGNUM: o << "(" << A << " > S(0)) ? S(1) : ((" << A << " < S(0)) ? S(-1) : " << A << ")"; break;
      case TO_X_SIN: o << "sin" << sfx << "(" << A << ")"; break;
      case TO_X_COS: o << "cos" << sfx << "(" << A << ")"; break;
      case TO_X_TANH: o << "tanh" << sfx << "(" << A << ")"; break;
      case TO_X_POW: o << "pow" << sfx << "(" << A << ", " << B << ")"; break;
      case TO_X_MAX: o << "fmax" << sfx << "(" << A << ", " << B << ")"; break;
      case TO_X_MIN: o << "fmin" << sfx << "(" << A << ", " << B << ")"; break;
      default: o << "S(__builtin_nanf(\"\"))"; break;
    }
    o << ";\n";
  }
  o << "  return v" << (e.arity + n - 1) << ";\n";
  return o.str();
}

static std::string source(const to_expr_s& e, bool f64) {
  std::ostringstream o;
  const int N = e.arity;
  o << "typedef " << (f64 ? "double" : "float") << " S;\n"
       "#define V " << (f64 ? 2 : 4) << "\n"
       "typedef S vec __attribute__((ext_vector_type(V)));\n"
       "struct P { const S* x[8]; long period[8]; };\n"
       "__device__ __forceinline__ S F(const S* x) {\n"
    << body(e, f64)
    << "}\n"
       "#define N " << N << "\n"
       "#define NN " << (N > 0 ? N : 1) << "\n"
    << R"SRC(
extern "C" __global__ __launch_bounds__(256) void ew_vec(P p, S* __restrict__ out, long totalv, long total) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < totalv; q += stride) {
    const long e = q * V;
    vec v[NN];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const long ei = (p.period[i] == total) ? e : (e % p.period[i]);
      v[i] = *reinterpret_cast<const vec*>(p.x[i] + ei);
    }
    vec r;
#pragma unroll
    for (int c = 0; c < V; ++c) {
      S xin[NN];
#pragma unroll
      for (int i = 0; i < N; ++i) xin[i] = v[i][c];
      r[c] = F(xin);
    }
    *reinterpret_cast<vec*>(out + e) = r;
  }
}
extern "C" __global__ __launch_bounds__(256) void ew_vecnt(P p, S* __restrict__ out, long totalv, long total) {
  const long stride = (long)gridDim.x * blockDim.x;
  const long start = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long q = start; q < totalv; q += 2 * stride) {
    const long q2 = q + stride;
    const bool two = q2 < totalv;
    vec v[2][NN];
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
        S xin[NN];
#pragma unroll
        for (int i = 0; i < N; ++i) xin[i] = v[u][i][c];
        r[c] = F(xin);
      }
      __builtin_nontemporal_store(r, reinterpret_cast<vec*>(out + (u == 0 ? q : q2) * V));
    }
  }
}
extern "C" __global__ __launch_bounds__(256) void ew_scalar(P p, S* __restrict__ out, long totalv, long total) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    S xin[NN];
#pragma unroll
    for (int i = 0; i < N; ++i) xin[i] = p.x[i][(p.period[i] == total) ? e : (e % p.period[i])];
    out[e] = F(xin);
  }
}
)SRC";
  return o.str();
}

void* jit_build(const to_expr_s& e, int dtype, std::string* err) {
  const std::string src = source(e, dtype == TO_F64);
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

std::string jit_source_for_tests(const to_expr_s& e, int dtype) { return source(e, dtype == TO_F64); }
std::string jit_expr_body(const to_expr_s& e, bool f64) { return body(e, f64); }
std::string jit_literal(double c, bool f64) { return lit(c, f64); }

void jit_launch(void* h, const EwArgs& a, hipStream_t s) {
  auto* k = static_cast<JitKernels*>(h);
  struct { const void* x[8]; long period[8]; } p{};
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
  const long V = a.dtype == TO_F64 ? 2 : 4, esz = a.dtype == TO_F64 ? 8 : 4;
  bool vec = (a.total % V == 0) && al16(a.out);
  for (int i = 0; i < a.n; ++i) {
    p.x[i] = a.x[i];
    p.period[i] = a.period[i];
    vec = vec && al16(a.x[i]) && (a.period[i] % V == 0);
  }
  void* out = a.out;
  long total = a.total, total4 = a.total / V;
  void* args[] = {&p, &out, &total4, &total};
  hipFunction_t f;
  long blocks;
  if (vec) {
    const bool streaming = a.total * esz >= (64L << 20);
    f = (streaming && total4 >= (1 << 20)) ? k->vecnt : k->vec;
    blocks = (total4 + 255) / 256;
    const long cap = streaming ? 32768 : 2048;
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
