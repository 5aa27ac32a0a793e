// Elementwise expressions: the reified form of the reference's opaque closures
// (`liftT :: (Vec n e -> e) -> ...`, src/TensorOps/Types.hs:56-59; `VFunc`,
// :114-117).  A host shim instantiates the `RealFloat`-polymorphic closure at a
// symbolic element type and ships the resulting SSA program once.
//
// compile = validate + classify + slot-allocate.
// classify: probabilistic identity testing -- evaluate the program in double
// on fixed pseudo-random points and compare with the closed form of each
// pre-fused kernel (two real-analytic functions that agree on random points
// are identical with probability 1).  This is insensitive to the operation
// order the host's AD happened to produce, unlike structural matching.
#include <cmath>
#include <cstring>

#include <map>
#include <cstring>
#include "ops.hpp"

namespace to {

double expr_eval(const to_expr_s& e, const double* x) {
  const int n = (int)(e.code.size() / 3);
  std::vector<double> v(e.arity + n);
  for (int i = 0; i < e.arity; ++i) v[i] = x[i];
  for (int i = 0; i < n; ++i) {
    const int op = e.code[3 * i], ia = e.code[3 * i + 1], ib = e.code[3 * i + 2];
    double r;
    if (op == TO_X_CONST) {
      r = e.consts[ia];
    } else {
      const double a = v[ia], b = v[ib];
      switch (op) {
        case TO_X_ADD: r = a + b; break;
        case TO_X_SUB: r = a - b; break;
        case TO_X_MUL: r = a * b; break;
        case TO_X_DIV: r = a / b; break;
        case TO_X_NEG: r = -a; break;
        case TO_X_RECIP: r = 1.0 / a; break;
        case TO_X_EXP: r = std::exp(a); break;
        case TO_X_LOG: r = std::log(a); break;
        case TO_X_SQRT: r = std::sqrt(a); break;
        case TO_X_ABS: r = std::fabs(a); break;
        case TO_X_SIGNUM: r = (a > 0) ? 1.0 : ((a < 0) ? -1.0 : a); break;
        case TO_X_SIN: r = std::sin(a); break;
        case TO_X_COS: r = std::cos(a); break;
        case TO_X_TANH: r = std::tanh(a); break;
        case TO_X_POW: r = std::pow(a, b); break;
        case TO_X_MAX: r = std::fmax(a, b); break;
        case TO_X_MIN: r = std::fmin(a, b); break;
        default: r = NAN; break;
      }
    }
    v[e.arity + i] = r;
  }
  return v.empty() ? 0.0 : v.back();
}

static bool close(double a, double b) {
  if (!std::isfinite(a) || !std::isfinite(b)) return false;
  return std::fabs(a - b) <= 1e-10 * (1.0 + std::fabs(a) + std::fabs(b));
}

struct Lcg {
  uint64_t s = 0x7e500001ull;
  double next() {  // (0,1)
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return ((s >> 11) + 0.5) * (1.0 / 9007199254740992.0);
  }
};

template <class F>
static bool matches(const to_expr_s& e, F ref, bool positive_only) {
  Lcg g;
  for (int t = 0; t < 12; ++t) {
    double x[8];
    for (int i = 0; i < e.arity; ++i) {
      const double u = g.next();
      x[i] = positive_only ? 0.25 + 2.0 * u : -2.0 + 4.0 * u;
    }
    if (!close(expr_eval(e, x), ref(x))) return false;
  }
  return true;
}

static double sigm(double z) { return 1.0 / (1.0 + std::exp(-z)); }

// ABS, SIGNUM, MAX, MIN (and POW, whose domain depends on its operands) make a program piecewise: agreement with a
// closed form on the sample interval then proves nothing about the rest of the line -- `log (max x 1e-7)` equals
// `log x` on [0.25, 2.25], `max x (-3)` is the identity on [-2, 2].  Such programs are never classified; the
// run-time specialised kernel / the VM evaluates them exactly as written.
bool expr_is_smooth(const to_expr_s& e) {
  const int n = (int)(e.code.size() / 3);
  for (int i = 0; i < n; ++i) {
    const int op = e.code[3 * i];
    if (op == TO_X_ABS || op == TO_X_SIGNUM || op == TO_X_MAX || op == TO_X_MIN || op == TO_X_POW) return false;
  }
  return true;
}

static void classify(to_expr_s& e) {
  e.kind = EW_VM;
  const int n = e.arity;
  if (n > 0 && !expr_is_smooth(e)) return;
  if (n == 0) {
    e.kind = EW_CONST;
    e.c0_d = expr_eval(e, nullptr);
    return;
  }
  if (n <= 4) {  // affine: c + sum a_i x_i
    double zero[8] = {0};
    const double c = expr_eval(e, zero);
    double a[4] = {0, 0, 0, 0};
    bool ok = std::isfinite(c);
    for (int i = 0; i < n && ok; ++i) {
      double x[8] = {0};
      x[i] = 1.0;
      a[i] = expr_eval(e, x) - c;
      ok = std::isfinite(a[i]);
    }
    if (ok && matches(e, [&](const double* x) {
          double r = c;
          for (int i = 0; i < n; ++i) r += a[i] * x[i];
          return r;
        }, false)) {
      e.kind = EW_AFFINE;
      e.c0_d = c;
      for (int i = 0; i < 4; ++i) e.coef_d[i] = a[i];
      return;
    }
  }
  if (n == 1) {
    if (matches(e, [](const double* x) { return sigm(x[0]); }, false)) { e.kind = EW_LOGISTIC; return; }
    if (matches(e, [](const double* x) { return std::exp(x[0]); }, false)) { e.kind = EW_EXP; return; }
    if (matches(e, [](const double* x) { return std::tanh(x[0]); }, false)) { e.kind = EW_TANH; return; }
    if (matches(e, [](const double* x) { return 1.0 / x[0]; }, false)) { e.kind = EW_RECIP; return; }
    if (matches(e, [](const double* x) { return std::log(x[0]); }, true)) { e.kind = EW_LOG; return; }
    if (matches(e, [](const double* x) { return std::sqrt(x[0]); }, true)) { e.kind = EW_SQRT; return; }
  }
  if (n == 2) {
    if (matches(e, [](const double* x) { return x[0] * x[1]; }, false)) { e.kind = EW_MUL; return; }
    if (matches(e, [](const double* x) { return x[0] / x[1]; }, false)) { e.kind = EW_DIV; return; }
    if (matches(e, [](const double* x) {
          const double s = sigm(x[1]);
          return x[0] * (s * (1.0 - s));
        }, false)) { e.kind = EW_MUL_DLOGISTIC; return; }
    if (matches(e, [](const double* x) {
          const double t = std::tanh(x[1]);
          return x[0] * (1.0 - t * t);
        }, false)) { e.kind = EW_MUL_DTANH; return; }
  }
}

static void allocate_slots(to_expr_s& e) {
  const int n = (int)(e.code.size() / 3);
  const int nv = e.arity + n;
  std::vector<int> last_use(nv, -1);
  for (int i = 0; i < n; ++i) {
    const int op = e.code[3 * i];
    if (op == TO_X_CONST) continue;
    last_use[e.code[3 * i + 1]] = i;
    last_use[e.code[3 * i + 2]] = i;
  }
  if (nv > 0) last_use[nv - 1] = n;  // the result stays live
  std::vector<int> slot(nv, -1);
  std::vector<int> free_slots;
  int next = e.arity;
  for (int i = 0; i < e.arity; ++i) slot[i] = i;  // inputs keep slots 0..arity-1
  e.vm_code.assign(4 * (size_t)n, 0);
  for (int i = 0; i < n; ++i) {
    const int op = e.code[3 * i], ia = e.code[3 * i + 1], ib = e.code[3 * i + 2];
    int sa = 0, sb = 0;
    if (op != TO_X_CONST) { sa = slot[ia]; sb = slot[ib]; }
    // operands whose last use is this instruction free their slot (not inputs)
    if (op != TO_X_CONST) {
      if (ia >= e.arity && last_use[ia] == i) free_slots.push_back(slot[ia]);
      if (ib >= e.arity && ib != ia && last_use[ib] == i) free_slots.push_back(slot[ib]);
    }
    int d;
    if (!free_slots.empty()) { d = free_slots.back(); free_slots.pop_back(); }
    else d = next++;
    slot[e.arity + i] = d;
    e.vm_code[4 * i] = op;
    e.vm_code[4 * i + 1] = d;
    e.vm_code[4 * i + 2] = (op == TO_X_CONST) ? ia : sa;
    e.vm_code[4 * i + 3] = sb;
  }
  e.n_slots = next > 0 ? next : 1;
  e.result_slot = nv > 0 ? slot[nv - 1] : 0;
}


// A program's bytecode and constants (a few hundred bytes, once per expression and dtype) go to the device from a pinned
// bounce buffer, not from the heap vector that holds them: the runtime's pageable transfers are the one thing that has
// been seen to lose pieces on a shared device (DESIGN_HISTORY.md 11.1).  Synchronous on the null stream, as before -- legal
// inside a relaxed capture, and ordered before whatever launch uses the program.
// ONE bounce buffer for the life of the library (hipHostMalloc + hipHostFree per upload is a pinned allocation and a
// device-wide synchronisation per fresh closure -- a host that compiles a closure every step paid it on the step path).
// Grown on demand, freed by expr_shutdown; callers hold the library's lock, and the copy is synchronous, so the buffer is
// free again when upload_small returns.
static void* g_small_pin = nullptr;
static size_t g_small_pin_bytes = 0;
static void upload_small(void* dst, const void* src, size_t nbytes) {
  if (nbytes > g_small_pin_bytes) {
    if (g_small_pin) (void)hipHostFree(g_small_pin);
    g_small_pin = nullptr;
    g_small_pin_bytes = 0;
    size_t want = 4096;
    while (want < nbytes) want *= 2;
    TO_HIP(hipHostMalloc(&g_small_pin, want, hipHostMallocDefault));
    g_small_pin_bytes = want;
  }
  std::memcpy(g_small_pin, src, nbytes);
  TO_HIP(hipMemcpy(dst, g_small_pin, nbytes, hipMemcpyHostToDevice));
}
void expr_shutdown() {
  if (g_small_pin) (void)hipHostFree(g_small_pin);
  g_small_pin = nullptr;
  g_small_pin_bytes = 0;
}

// per-dtype resources of a VM-kind program, created on first use with that dtype
void expr_prepare(to_expr e, int dtype) {
  if (e->kind != EW_VM) return;
  const int di = dtype == TO_F64 ? 1 : 0;
  static const int jit_enabled = [] { const char* v = getenv("TOPS_EXPR_JIT"); return v ? atoi(v) : 1; }();
  if (jit_enabled && !e->jit_tried[di]) {
    e->jit_tried[di] = true;
    e->jit[di] = jit_build(*e, dtype, &e->jit_error);
  }
  if (e->jit[di]) return;
  if (!e->d_code && !e->vm_code.empty()) {
    const size_t cb = e->vm_code.size() * sizeof(int32_t);
    TO_HIP(hipMalloc(&e->d_code, cb));
    upload_small(e->d_code, e->vm_code.data(), cb);
  }
  if (!e->consts.empty()) {
    if (di == 0 && !e->d_consts_f32) {
      std::vector<float> c(e->consts.begin(), e->consts.end());
      TO_HIP(hipMalloc(&e->d_consts_f32, c.size() * sizeof(float)));
      upload_small(e->d_consts_f32, c.data(), c.size() * sizeof(float));
    }
    if (di == 1 && !e->d_consts_f64) {
      TO_HIP(hipMalloc(&e->d_consts_f64, e->consts.size() * sizeof(double)));
      upload_small(e->d_consts_f64, e->consts.data(), e->consts.size() * sizeof(double));
    }
  }
}

to_expr expr_compile(int arity, int n_instr, const int32_t* code, int n_consts,
                     const double* consts) {
  TO_CHECK(arity >= 0 && arity <= 8, TO_ERR_ARG, "expression arity must be 0..8");
  TO_CHECK(n_instr >= 0 && (n_instr == 0 || code), TO_ERR_ARG, "null code");
  TO_CHECK(arity + n_instr >= 1, TO_ERR_ARG, "empty expression");
  auto* e = new to_expr_s();
  static std::atomic<uint64_t> next_uid{1};
  e->uid = next_uid++;
  e->arity = arity;
  e->code.assign(code, code + 3 * (size_t)n_instr);
  e->consts.assign(consts, consts + (n_consts > 0 ? n_consts : 0));
  for (int i = 0; i < n_instr; ++i) {
    const int op = code[3 * i], a = code[3 * i + 1], b = code[3 * i + 2];
    bool ok = op >= 0 && op < TO_X_NOPS;
    if (ok && op == TO_X_CONST) ok = a >= 0 && a < n_consts;
    else if (ok) ok = a >= 0 && a < arity + i && b >= 0 && b < arity + i;
    if (!ok) {
      delete e;
      fail(TO_ERR_ARG, "malformed expression at instruction " + std::to_string(i));
    }
  }
  {
    // structure id: the same closure reified again (a new instance, a new uid) is the same program
    // Bounded: a host that generates fresh constants every step (a decaying rate inside a closure) would otherwise grow
    // this map for ever.  Past 65,536 structures the table starts over; ids keep counting, so "same id => same program"
    // still holds and a program met again merely looks new (one plan-cache miss).
    static std::map<std::vector<uint64_t>, uint64_t> interned;
    static uint64_t next_sid = 1;
    if (interned.size() >= 65536) interned.clear();
    std::vector<uint64_t> key;
    key.reserve(2 + e->code.size() + e->consts.size());
    key.push_back((uint64_t)arity);
    key.push_back((uint64_t)n_instr);
    for (int32_t c : e->code) key.push_back((uint64_t)(uint32_t)c);
    for (double d : e->consts) {
      uint64_t u;
      std::memcpy(&u, &d, 8);
      key.push_back(u);
    }
    auto it = interned.find(key);
    if (it == interned.end()) it = interned.emplace(std::move(key), next_sid++).first;
    e->sid = it->second;
  }
  classify(*e);
  allocate_slots(*e);
  return e;
}

void expr_retain(to_expr e) { e->refs.fetch_add(1); }

void expr_release(to_expr e) {
  if (!e) return;
  if (e->refs.fetch_sub(1) != 1) return;
  if (e->d_code) (void)hipFree(e->d_code);
  if (e->d_consts_f32) (void)hipFree(e->d_consts_f32);
  if (e->d_consts_f64) (void)hipFree(e->d_consts_f64);
  for (void* j : e->jit)
    if (j) jit_release(j);
  delete e;
}

}  // namespace to
