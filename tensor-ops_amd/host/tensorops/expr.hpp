// Symbolic element type + forward-mode AD: how `RealFloat`-polymorphic closures
// (`forall a. RealFloat a => Vec n a -> a`, src/TensorOps/Types.hs:114-117) are
// reified for a device backend.  `Expr` plays the role of `ElemT HipT` while a
// closure runs; arithmetic on it records an SSA program (the format of
// `to_expr_compile`, include/tensorops_hip.h).  `Dual<A>` is forward-mode AD
// over any scalar `A` -- what `Numeric.AD.diff`/`grad` provide to `TO.map` /
// `TO.zipN` (src/TensorOps/TOp.hs:209-213,241-247).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <tuple>
#include <vector>

#include "../../../include/tensorops_hip.h"

namespace tensorops {

struct Tape {
  int arity = 0;
  std::vector<int32_t> code;  // 3 per instruction
  std::vector<double> consts;
  std::map<std::tuple<int, int, int>, int> cse;
  std::map<uint64_t, int> const_ids;

  int emit(int op, int a, int b) {
    auto key = std::make_tuple(op, a, b);
    auto it = cse.find(key);
    if (it != cse.end()) return it->second;
    code.push_back(op);
    code.push_back(a);
    code.push_back(b);
    const int v = arity + (int)(code.size() / 3) - 1;
    cse[key] = v;
    return v;
  }
  int constant(double c) {
    uint64_t bits;
    std::memcpy(&bits, &c, 8);
    auto it = const_ids.find(bits);
    if (it != const_ids.end()) return it->second;
    consts.push_back(c);
    code.push_back(TO_X_CONST);
    code.push_back((int)consts.size() - 1);
    code.push_back(0);
    const int v = arity + (int)(code.size() / 3) - 1;
    const_ids[bits] = v;
    return v;
  }
};

class Expr {
 public:
  std::shared_ptr<Tape> tape;  // null: a literal constant
  int v = -1;
  double c = 0.0;

  Expr() = default;
  Expr(double x) : c(x) {}  // NOLINT: literals convert implicitly, like `fromRational`
  Expr(std::shared_ptr<Tape> t, int id) : tape(std::move(t)), v(id) {}
  bool is_const() const { return !tape; }

  int on(const std::shared_ptr<Tape>& t) const { return tape ? v : t->constant(c); }
};

inline Expr expr_bin(int op, const Expr& a, const Expr& b, double (*fold)(double, double)) {
  if (a.is_const() && b.is_const()) return Expr(fold(a.c, b.c));
  const auto& t = a.tape ? a.tape : b.tape;
  if (a.tape && b.tape && a.tape != b.tape) throw std::logic_error("Expr values from different closures");
  return Expr(t, t->emit(op, a.on(t), b.on(t)));
}
inline Expr expr_un(int op, const Expr& a, double (*fold)(double)) {
  if (a.is_const()) return Expr(fold(a.c));
  return Expr(a.tape, a.tape->emit(op, a.v, a.v));
}

inline Expr operator+(const Expr& a, const Expr& b) { return expr_bin(TO_X_ADD, a, b, [](double x, double y) { return x + y; }); }
inline Expr operator-(const Expr& a, const Expr& b) { return expr_bin(TO_X_SUB, a, b, [](double x, double y) { return x - y; }); }
inline Expr operator*(const Expr& a, const Expr& b) { return expr_bin(TO_X_MUL, a, b, [](double x, double y) { return x * y; }); }
inline Expr operator/(const Expr& a, const Expr& b) { return expr_bin(TO_X_DIV, a, b, [](double x, double y) { return x / y; }); }
inline Expr operator-(const Expr& a) { return expr_un(TO_X_NEG, a, [](double x) { return -x; }); }
inline Expr exp(const Expr& a) { return expr_un(TO_X_EXP, a, [](double x) { return std::exp(x); }); }
inline Expr log(const Expr& a) { return expr_un(TO_X_LOG, a, [](double x) { return std::log(x); }); }
inline Expr sqrt(const Expr& a) { return expr_un(TO_X_SQRT, a, [](double x) { return std::sqrt(x); }); }
inline Expr abs(const Expr& a) { return expr_un(TO_X_ABS, a, [](double x) { return std::fabs(x); }); }
inline Expr signum(const Expr& a) { return expr_un(TO_X_SIGNUM, a, [](double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : x); }); }
inline Expr sin(const Expr& a) { return expr_un(TO_X_SIN, a, [](double x) { return std::sin(x); }); }
inline Expr cos(const Expr& a) { return expr_un(TO_X_COS, a, [](double x) { return std::cos(x); }); }
inline Expr tanh(const Expr& a) { return expr_un(TO_X_TANH, a, [](double x) { return std::tanh(x); }); }
inline Expr recip(const Expr& a) { return expr_un(TO_X_RECIP, a, [](double x) { return 1.0 / x; }); }
inline Expr pow(const Expr& a, const Expr& b) { return expr_bin(TO_X_POW, a, b, [](double x, double y) { return std::pow(x, y); }); }
inline Expr max(const Expr& a, const Expr& b) { return expr_bin(TO_X_MAX, a, b, [](double x, double y) { return std::fmax(x, y); }); }
inline Expr min(const Expr& a, const Expr& b) { return expr_bin(TO_X_MIN, a, b, [](double x, double y) { return std::fmin(x, y); }); }

// the same vocabulary at `double`, so one generic closure serves host evaluation too
inline double exp(double x) { return std::exp(x); }
inline double log(double x) { return std::log(x); }
inline double sqrt(double x) { return std::sqrt(x); }
inline double abs(double x) { return std::fabs(x); }
inline double signum(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : x); }
inline double sin(double x) { return std::sin(x); }
inline double cos(double x) { return std::cos(x); }
inline double tanh(double x) { return std::tanh(x); }
inline double recip(double x) { return 1.0 / x; }
inline double pow(double x, double y) { return std::pow(x, y); }
inline double max(double x, double y) { return std::fmax(x, y); }
inline double min(double x, double y) { return std::fmin(x, y); }

// ---- forward-mode dual numbers ----------------------------------------------------------
template <class A>
struct Dual {
  A p;
  A t;
  bool has_t = false;  // false: a lifted constant (structurally zero tangent)
  Dual() = default;
  Dual(double x) : p(x), t(0.0) {}  // NOLINT
  Dual(A primal) : p(std::move(primal)), t(0.0) {}  // NOLINT
  Dual(A primal, A tangent) : p(std::move(primal)), t(std::move(tangent)), has_t(true) {}
};

template <class A> Dual<A> operator-(const Dual<A>& a) { return a.has_t ? Dual<A>(-a.p, -a.t) : Dual<A>(-a.p); }
template <class A> Dual<A> operator+(const Dual<A>& a, const Dual<A>& b) {
  if (!a.has_t && !b.has_t) return Dual<A>(a.p + b.p);
  if (!a.has_t) return Dual<A>(a.p + b.p, b.t);
  if (!b.has_t) return Dual<A>(a.p + b.p, a.t);
  return Dual<A>(a.p + b.p, a.t + b.t);
}
template <class A> Dual<A> operator-(const Dual<A>& a, const Dual<A>& b) { return a + (-b); }
template <class A> Dual<A> operator*(const Dual<A>& a, const Dual<A>& b) {
  if (!a.has_t && !b.has_t) return Dual<A>(a.p * b.p);
  if (!a.has_t) return Dual<A>(a.p * b.p, a.p * b.t);
  if (!b.has_t) return Dual<A>(a.p * b.p, a.t * b.p);
  return Dual<A>(a.p * b.p, a.t * b.p + a.p * b.t);
}
template <class A> Dual<A> operator/(const Dual<A>& a, const Dual<A>& b) {
  A q = a.p / b.p;
  if (!a.has_t && !b.has_t) return Dual<A>(q);
  if (!b.has_t) return Dual<A>(q, a.t / b.p);
  if (!a.has_t) return Dual<A>(q, -(q / b.p) * b.t);
  return Dual<A>(q, a.t / b.p - (q / b.p) * b.t);
}
#define TENSOROPS_DUAL_MIXED(op)                                                                \
  template <class A> Dual<A> operator op(const Dual<A>& a, double b) { return a op Dual<A>(b); } \
  template <class A> Dual<A> operator op(double a, const Dual<A>& b) { return Dual<A>(a) op b; }
TENSOROPS_DUAL_MIXED(+)
TENSOROPS_DUAL_MIXED(-)
TENSOROPS_DUAL_MIXED(*)
TENSOROPS_DUAL_MIXED(/)
#undef TENSOROPS_DUAL_MIXED

#define TENSOROPS_DUAL_UNARY(name, dexpr)                     \
  template <class A> Dual<A> name(const Dual<A>& a) {         \
    const A& x = a.p;                                         \
    A y = name(x);                                            \
    if (!a.has_t) return Dual<A>(y);                          \
    (void)x;                                                  \
    return Dual<A>(y, (dexpr) * a.t);                         \
  }
TENSOROPS_DUAL_UNARY(exp, y)
TENSOROPS_DUAL_UNARY(log, A(1.0) / x)
TENSOROPS_DUAL_UNARY(sqrt, A(0.5) / y)
TENSOROPS_DUAL_UNARY(sin, cos(x))
TENSOROPS_DUAL_UNARY(cos, -sin(x))
TENSOROPS_DUAL_UNARY(tanh, A(1.0) - y * y)
TENSOROPS_DUAL_UNARY(abs, signum(x))
TENSOROPS_DUAL_UNARY(recip, -(y * y))
#undef TENSOROPS_DUAL_UNARY
template <class A> Dual<A> pow(const Dual<A>& a, const Dual<A>& b) { return exp(b * log(a)); }

// `diff f x` (Numeric.AD) for a generic unary closure
template <class F>
Expr diff_at(const F& f, const Expr& x) {
  Dual<Expr> r = f(Dual<Expr>(x, Expr(1.0)));
  return r.has_t ? r.t : Expr(0.0);
}
// `grad f xs` for a generic n-ary closure over std::vector
template <class F>
std::vector<Expr> grad_at(const F& f, const std::vector<Expr>& xs) {
  std::vector<Expr> out;
  for (size_t i = 0; i < xs.size(); ++i) {
    std::vector<Dual<Expr>> args;
    for (size_t j = 0; j < xs.size(); ++j)
      args.push_back(j == i ? Dual<Expr>(xs[j], Expr(1.0)) : Dual<Expr>(xs[j]));
    Dual<Expr> r = f(args);
    out.push_back(r.has_t ? r.t : Expr(0.0));
  }
  return out;
}

// A closure given as data (SSA program, same format as to_expr_compile): how the Python
// harness hands arbitrary `forall a. RealFloat a =>` functions to this host layer.
struct SsaFn {
  int arity = 0;
  std::vector<int32_t> code;
  std::vector<double> consts;

  template <class A>
  A operator()(const std::vector<A>& xs) const {
    std::vector<A> v(xs.begin(), xs.end());
    const int n = (int)(code.size() / 3);
    for (int i = 0; i < n; ++i) {
      const int op = code[3 * i], ia = code[3 * i + 1], ib = code[3 * i + 2];
      if (op == TO_X_CONST) {
        v.push_back(A(consts[ia]));
        continue;
      }
      const A a = v[ia], b = v[ib];
      switch (op) {
        case TO_X_ADD: v.push_back(a + b); break;
        case TO_X_SUB: v.push_back(a - b); break;
        case TO_X_MUL: v.push_back(a * b); break;
        case TO_X_DIV: v.push_back(a / b); break;
        case TO_X_NEG: v.push_back(-a); break;
        case TO_X_RECIP: v.push_back(recip(a)); break;
        case TO_X_EXP: v.push_back(exp(a)); break;
        case TO_X_LOG: v.push_back(log(a)); break;
        case TO_X_SQRT: v.push_back(sqrt(a)); break;
        case TO_X_ABS: v.push_back(abs(a)); break;
        case TO_X_SIN: v.push_back(sin(a)); break;
        case TO_X_COS: v.push_back(cos(a)); break;
        case TO_X_TANH: v.push_back(tanh(a)); break;
        case TO_X_POW: v.push_back(pow(a, b)); break;
        default: throw std::invalid_argument("SsaFn: opcode not differentiable on the host");
      }
    }
    if (v.empty()) throw std::invalid_argument("SsaFn: empty program");
    return v.back();
  }
};

}  // namespace tensorops
