// Row programs: a row-local subgraph of the recorded class-method stream compiled at run time into ONE kernel.
//
// A loss head is row-local: everything between z = W a + b and its cotangent dz -- softmax (`map exp`, `sumRows`,
// `map recip`, an outer product with a scalar), a scale, squaredError or crossEntropy (a dot product, a negate) and the
// cotangents of all of these in whatever order the host's AD produced them -- touches one sample's row at a time
// (src/TensorOps/Learn/NeuralNet.hs:52-77, src/TensorOps/TOp.hs:151-159).  The two heads the small-GEMM kernel carries in
// its epilogue (rows of at most 16 outputs, csrc/gemm_small.hip) are recognised by probabilistic identity testing; every
// OTHER row-local subgraph -- a head of 20 or 784 outputs (an auto-encoder's squaredError over the whole input width,
// AutoEncoder.hs:87-142), softmax followed by a scale, a loss the library has no closed form for -- is printed as the
// body of a kernel, one wave per row, vectors in registers (ceil(N/64) elements per lane), row sums and dot products as
// wave reductions, closures inlined from their SSA programs (expr_jit.cpp), and compiled once with hiprtc.  What the
// kernel computes is exactly the recorded ops in recorded order: no identity is assumed.
#include <hip/hiprtc.h>

#include <map>
#include <sstream>

#include "ops.hpp"

namespace to {

std::string jit_expr_body(const to_expr_s& e, bool f64);
std::string jit_literal(double c, bool f64);

namespace {

struct RowModule {
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
};

std::map<std::string, RowModule*>& modules() {  // by source text: one compilation per distinct program
  static std::map<std::string, RowModule*> m;
  return m;
}

std::string source_of(const RowProg& rp) {
  const bool f64 = rp.dtype == TO_F64;
  const int64_t N = rp.N, EPL = (N + 63) / 64;
  std::ostringstream o;
  o << "typedef " << (f64 ? "double" : "float") << " S;\n#define NC " << N << "\n#define EPL " << EPL << "\n"
    << "struct P { const S* root; const S* ext[4]; S* out[4]; long rows; };\n"
       "__device__ __forceinline__ S wsum(S x) {\n#pragma unroll\n  for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);\n  return x;\n}\n";
  // the closures, one device function per distinct expression
  std::map<uint64_t, int> fid;
  for (const RowNode& n : rp.nodes)
    if (n.op == R_LIFT && !fid.count(n.f->uid)) {
      const int k = (int)fid.size();
      fid[n.f->uid] = k;
      o << "__device__ __forceinline__ S F" << k << "(const S* x) {\n" << jit_expr_body(*n.f, f64) << "}\n";
    }
  o << "extern \"C\" __global__ __launch_bounds__(256) void rowprog(P p) {\n"
       "  const int lane = threadIdx.x & 63;\n"
       "  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);\n"
       "  if (row >= p.rows) return;\n"
       "  S v0[EPL];\n#pragma unroll\n"
       "  for (int e = 0; e < EPL; ++e) v0[e] = lane + 64 * e < NC ? p.root[row * NC + lane + 64 * e] : S(0);\n";
  const int n_ext = (int)rp.ext_vec.size();
  std::vector<char> is_vec{1};
  for (int x = 0; x < n_ext; ++x) {
    const int id = 1 + x;
    is_vec.push_back(rp.ext_vec[x]);
    const char* base = rp.ext_rowwise[x] ? "row * " : "0 * ";
    if (rp.ext_vec[x])
      o << "  S v" << id << "[EPL];\n#pragma unroll\n  for (int e = 0; e < EPL; ++e) v" << id
        << "[e] = lane + 64 * e < NC ? p.ext[" << x << "][" << base << "NC + lane + 64 * e] : S(0);\n";
    else
      o << "  const S s" << id << " = p.ext[" << x << "][" << base << "1];\n";
  }
  auto val = [&](int id, const char* e) { return is_vec[id] ? "v" + std::to_string(id) + "[" + e + "]" : "s" + std::to_string(id); };
  for (size_t k = 0; k < rp.nodes.size(); ++k) {
    const RowNode& n = rp.nodes[k];
    const int id = 1 + n_ext + (int)k;
    is_vec.push_back(n.vec);
    const std::string me = (n.vec ? "v" : "s") + std::to_string(id);
    auto elementwise = [&](const std::string& rhs_e, const std::string& rhs_s) {
      if (n.vec) o << "  S " << me << "[EPL];\n#pragma unroll\n  for (int e = 0; e < EPL; ++e) { " << me << "[e] = " << rhs_e << "; }\n";
      else o << "  const S " << me << " = " << rhs_s << ";\n";
    };
    switch (n.op) {
      case R_CONST: elementwise(jit_literal(n.alpha, f64), jit_literal(n.alpha, f64)); break;
      case R_LIFT: {
        auto args = [&](const char* e) {
          std::string a = "{";
          for (size_t i = 0; i < n.in.size(); ++i) a += (i ? ", " : "") + val(n.in[i], e);
          return a + "}";
        };
        const std::string f = "F" + std::to_string(fid[n.f->uid]);
        if (n.vec)
          o << "  S " << me << "[EPL];\n#pragma unroll\n  for (int e = 0; e < EPL; ++e) { const S x[] = " << args("e") << "; " << me
            << "[e] = " << f << "(x); }\n";
        else
          o << "  S " << me << ";\n  { const S x[] = " << args("0") << "; " << me << " = " << f << "(x); }\n";
        break;
      }
      case R_DACT:
        if (n.alpha != 0.0)  // d * (1 - h^2)
          elementwise(val(n.in[0], "e") + " * (S(1) - " + val(n.in[1], "e") + " * " + val(n.in[1], "e") + ")",
                      val(n.in[0], "0") + " * (S(1) - " + val(n.in[1], "0") + " * " + val(n.in[1], "0") + ")");
        else
          elementwise(val(n.in[0], "e") + " * " + val(n.in[1], "e") + " * (S(1) - " + val(n.in[1], "e") + ")",
                      val(n.in[0], "0") + " * " + val(n.in[1], "0") + " * (S(1) - " + val(n.in[1], "0") + ")");
        break;
      case R_SUM: {
        auto sum = [&](const char* e) {
          std::string a = val(n.in[0], e);   // left fold, like sum' (Data/List/Util.hs:7-10)
          for (size_t i = 1; i < n.in.size(); ++i) a = "(" + a + " + " + val(n.in[i], e) + ")";
          return a;
        };
        elementwise(sum("e"), sum("0"));
        break;
      }
      case R_SCALE: elementwise(jit_literal(n.alpha, f64) + " * " + val(n.in[0], "e"), jit_literal(n.alpha, f64) + " * " + val(n.in[0], "0")); break;
      case R_MUL: elementwise(val(n.in[0], "e") + " * " + val(n.in[1], "e"), val(n.in[0], "0") + " * " + val(n.in[1], "0")); break;
      case R_SUM_ROWS:
        o << "  S " << me << " = S(0);\n#pragma unroll\n  for (int e = 0; e < EPL; ++e) if (lane + 64 * e < NC) " << me << " += "
          << val(n.in[0], "e") << ";\n  " << me << " = wsum(" << me << ");\n";
        break;
      case R_DOT:
        o << "  S " << me << " = S(0);\n#pragma unroll\n  for (int e = 0; e < EPL; ++e) if (lane + 64 * e < NC) " << me << " += "
          << val(n.in[0], "e") << " * " << val(n.in[1], "e") << ";\n  " << me << " = wsum(" << me << ");\n";
        break;
      case R_MAP_ROWS: elementwise(val(n.in[0], "e"), val(n.in[0], "0")); break;
      default: o << "#error unknown row op\n"; break;
    }
  }
  for (size_t j = 0; j < rp.outs.size(); ++j) {
    const int id = rp.outs[j];
    if (is_vec[id])
      o << "#pragma unroll\n  for (int e = 0; e < EPL; ++e) if (lane + 64 * e < NC) p.out[" << j << "][row * NC + lane + 64 * e] = v" << id
        << "[e];\n";
    else
      o << "  if (lane == 0) p.out[" << j << "][row] = s" << id << ";\n";
  }
  o << "}\n";
  return o.str();
}

}  // namespace

RowProg::~RowProg() {
  for (RowNode& n : nodes)
    if (n.f) expr_release(n.f);
}

bool rowprog_build(RowProg& rp) {
  if (rp.module) return true;
  if (rp.tried) return false;
  rp.tried = true;
  static const int enable = [] { const char* e = getenv("TOPS_ROWPROG"); return e ? atoi(e) : 1; }();
  if (!enable) {
    rp.err = "TOPS_ROWPROG=0";
    return false;
  }
  const std::string src = source_of(rp);
  auto it = modules().find(src);
  if (it != modules().end()) {
    rp.module = it->second;
    return rp.module != nullptr;
  }
  modules()[src] = nullptr;  // (a failed build is not retried)
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, src.c_str(), "tensorops_rowprog.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
    rp.err = "hiprtcCreateProgram failed";
    return false;
  }
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off"};
  if (hiprtcCompileProgram(prog, 3, opts) != HIPRTC_SUCCESS) {
    size_t n = 0;
    hiprtcGetProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) hiprtcGetProgramLog(prog, &log[0]);
    rp.err = "hiprtc: " + log;
    hiprtcDestroyProgram(&prog);
    if (getenv("TOPS_LAZY_DEBUG")) std::fprintf(stderr, "[rowprog] build failed: %s\n%s\n", rp.err.c_str(), src.c_str());
    return false;
  }
  size_t sz = 0;
  hiprtcGetCodeSize(prog, &sz);
  std::vector<char> code(sz);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  auto* m = new RowModule();
  if (hipModuleLoadData(&m->mod, code.data()) != hipSuccess || hipModuleGetFunction(&m->fn, m->mod, "rowprog") != hipSuccess) {
    rp.err = "hipModuleLoadData/GetFunction failed";
    if (m->mod) (void)hipModuleUnload(m->mod);
    delete m;
    return false;
  }
  modules()[src] = m;
  rp.module = m;
  return true;
}

void rowprog_launch(const RowProg& rp, const void* root, const void* const* ext, void* const* outs, int64_t rows, hipStream_t s) {
  auto* m = static_cast<RowModule*>(rp.module);
  TO_CHECK(m != nullptr, TO_ERR_STATE, "internal: row program was not built");
  if (rows <= 0) return;
  struct {
    const void* root;
    const void* ext[4];
    void* out[4];
    long rows;
  } p{};
  p.root = root;
  for (size_t i = 0; i < rp.ext_vec.size(); ++i) p.ext[i] = ext[i];
  for (size_t i = 0; i < rp.outs.size(); ++i) p.out[i] = outs[i];
  p.rows = rows;
  void* args[] = {&p};
  TO_HIP(hipModuleLaunchKernel(m->fn, (unsigned)((rows + 3) / 4), 1, 1, 256, 1, 1, 0, s, args, nullptr));
  count_launch();
}

}  // namespace to
