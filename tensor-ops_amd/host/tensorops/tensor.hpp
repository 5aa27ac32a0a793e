// `instance Tensor HipT`: the 13 methods of `class Tensor`
// (src/TensorOps/Types.hs:52-109) over device handles, each one or two calls
// into the C ABI (include/tensorops_hip.h).  This is the C++ rendering of what
// a Haskell shim would contain; it exists because no Haskell
// toolchain is available here.
#pragma once
#include <string>
#include <utility>

#include "expr.hpp"
#include "trace.hpp"

namespace tensorops {

struct TensorOpsError : std::runtime_error {
  int code;
  TensorOpsError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

inline void check(to_status s) {
  if (s != TO_OK) throw TensorOpsError(s, to_last_error());
}

using Dims = std::vector<int64_t>;

// `ElemT HipT`: the instance's element type is the runtime's default dtype
// (to_set_default_dtype); fp32 unless the caller selected the fp64 instance
inline int elem_dtype() {
  int dt = TO_F32;
  check(to_default_dtype(&dt));
  return dt;
}

// `t ns`: an immutable device tensor (ref-counted handle; copying shares it)
class T {
 public:
  T() = default;
  explicit T(to_tensor owned) : h_(owned) {}
  T(const T& o) : h_(o.h_) {
    if (h_) to_retain(h_);
  }
  T(T&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  T& operator=(T o) {
    std::swap(h_, o.h_);
    return *this;
  }
  ~T() {
    if (h_) to_release(h_);
  }
  to_tensor h() const { return h_; }
  explicit operator bool() const { return h_ != nullptr; }
  to_tensor release_handle() {
    to_tensor r = h_;
    h_ = nullptr;
    return r;
  }
  Dims dims() const {
    int rank = 0;
    int64_t d[TO_MAX_RANK];
    check(to_shape(h_, &rank, d, nullptr));
    return Dims(d, d + rank);
  }
  int64_t batch() const {
    int64_t b = 0;
    check(to_shape(h_, nullptr, nullptr, &b));
    return b;
  }
  bool batched() const { return batch() > 0; }

 private:
  to_tensor h_ = nullptr;
};

// Haskell values are thunks: a class-method argument that the callee ignores is never evaluated
class LT {  // a thunk of T
  struct Node {
    std::function<T()> f;
    T v;
    bool done = false;
    bool kept = false;  // part of a kept graph (Graph below): the closure stays, `reset` makes it a thunk again
  };

 public:
  // A thunk graph that is built once and evaluated many times.  GHC allocates the thunks of `gradTOp o xs` afresh on
  // every call and that costs it next to nothing; here each one is a std::function plus a shared node on the C++ heap,
  // and building the ~120 of a three-layer step was 7 us of a 40 us step.  A caller that evaluates the SAME expression
  // over the SAME leaves again (a trainer whose parameters are updated in place) builds it inside a `Recording`, forces
  // what it needs, and calls `reset()`: every thunk built under the recording forgets its value (the handles are
  // released, exactly where the thunks of a fresh graph would have died) and can be forced again -- the same closures,
  // hence the same class-method calls in the same order.  Thunks made while FORCING (rows of mapRows, recomputed
  // forward products) are ordinary ones and die with the value that holds them.
  class Graph {
   public:
    void reset() {
      for (auto& n : nodes_) {
        n->v = T();
        n->done = false;
      }
      for (auto& r : resets_) r();
    }
    void clear() {
      reset();
      for (auto& n : nodes_) n->kept = false;
      nodes_.clear();
      resets_.clear();
    }
    bool empty() const { return nodes_.empty(); }
    size_t size() const { return nodes_.size(); }
    // memo cells other than LT nodes (lazy_prod's) register how to forget their value
    void on_reset(std::function<void()> r) { resets_.push_back(std::move(r)); }

   private:
    friend class LT;
    std::vector<std::shared_ptr<Node>> nodes_;
    std::vector<std::function<void()>> resets_;
  };
  static Graph*& recording() {
    static thread_local Graph* g = nullptr;
    return g;
  }
  struct Recording {  // RAII: thunks constructed in this extent belong to `g`
    Graph* prev;
    explicit Recording(Graph* g) : prev(recording()) { recording() = g; }
    ~Recording() { recording() = prev; }
    Recording(const Recording&) = delete;
    Recording& operator=(const Recording&) = delete;
  };

  LT() = default;
  LT(T v) : n_(std::make_shared<Node>()) {  // NOLINT
    n_->v = std::move(v);
    n_->done = true;
  }
  explicit LT(std::function<T()> f) : n_(std::make_shared<Node>()) {
    n_->f = std::move(f);
    if (Graph* g = recording()) {
      n_->kept = true;
      g->nodes_.push_back(n_);
    }
  }
  const T& get() const {
    if (!n_->done) {
      // (forcing is not building: a thunk made by a closure that runs now is an ordinary one)
      Recording off(nullptr);
      n_->v = n_->f();
      n_->done = true;
      if (!n_->kept) n_->f = nullptr;
    }
    return n_->v;
  }

 private:
  std::shared_ptr<Node> n_;
};

// ---- closures -> compiled expressions, de-duplicated by program text ------------------------
using Closure = std::function<Expr(const std::vector<Expr>&)>;

class CompiledExpr {
 public:
  static to_expr get(int n, const Closure& f) {
    auto tape = std::make_shared<Tape>();
    tape->arity = n;
    std::vector<Expr> vars;
    for (int i = 0; i < n; ++i) vars.emplace_back(tape, i);
    Expr r = f(vars);
    int rv = r.on(tape);
    const int last = n + (int)(tape->code.size() / 3) - 1;
    if (rv != last) {  // make the result the last value (x + 0 keeps the value)
      const int z = tape->constant(0.0);
      tape->code.push_back(TO_X_ADD);
      tape->code.push_back(rv);
      tape->code.push_back(z);
    }
    Key key{n, tape->code, tape->consts};
    auto& c = cache();
    auto it = c.find(key);
    if (it != c.end()) return it->second;
    to_expr e = nullptr;
    check(to_expr_compile(n, (int)(tape->code.size() / 3), tape->code.data(), (int)tape->consts.size(),
                          tape->consts.data(), &e));
    c[key] = e;
    return e;
  }

 private:
  using Key = std::tuple<int, std::vector<int32_t>, std::vector<double>>;
  static std::map<Key, to_expr>& cache() {
    static std::map<Key, to_expr> m;
    return m;
  }
};

inline std::vector<std::string> idx_strings(const std::vector<int64_t>& idx) {
  std::vector<std::string> s;
  for (int64_t v : idx) s.push_back(std::to_string(v));
  return s;
}

// ---- the class methods ------------------------------------------------------------------------
struct HipT {
  // liftT (Types.hs:56-59)
  // `slot`: where a caller that applies the SAME closure again and again (a TOp instance, a trainer's update rule)
  // keeps the compiled expression -- reifying a closure (running it on symbolic scalars, de-duplicating the program
  // text) costs more than the call it leads to
  static T liftT(const Closure& f, const std::vector<T>& xs, to_expr* slot = nullptr) {
    to_expr e = slot && *slot ? *slot : CompiledExpr::get((int)xs.size(), f);
    if (slot) *slot = e;
    std::vector<to_tensor> hs;
    for (const T& x : xs) hs.push_back(x.h());
    to_tensor out = nullptr;
    check(to_lift(e, (int)hs.size(), hs.data(), &out));
    if (trace::on()) {
      // the closure's fingerprint: its value at two fixed points (literal Exprs fold to numbers)
      std::vector<std::string> fp{std::to_string(hs.size())};
      for (int k = 0; k < 2; ++k) {
        std::vector<Expr> pt;
        for (size_t i = 0; i < hs.size(); ++i) pt.emplace_back(k == 0 ? 0.3 + 0.17 * (double)i : 0.7 + 0.29 * (double)i);
        fp.push_back(trace::num(f(pt).c));
      }
      trace::call("liftT", fp, hs, out);
    }
    return T(out);
  }
  // gmul (Types.hs:60-66)
  static T gmul(int lm, int lo, int ln, const T& a, const T& b) {
    to_tensor out = nullptr;
    check(to_gmul(lm, lo, ln, a.h(), b.h(), &out));
    if (trace::on()) trace::call("gmul", {std::to_string(lm), std::to_string(lo), std::to_string(ln)}, {a.h(), b.h()}, out);
    return T(out);
  }
  // TT.inner / outer / outerV / dot / matVec / vecMat / matMat (Tensor.hs:132-185)
  static T inner(int lm, int ln, const T& a, const T& b) { return gmul(lm, 1, ln, a, b); }
  static T outer(int lm, int ln, const T& a, const T& b) { return gmul(lm, 0, ln, a, b); }
  static T outerV(const T& a, const T& b) { return gmul(1, 0, 1, a, b); }
  static T dot(const T& a, const T& b) { return gmul(0, 1, 0, a, b); }
  static T matVec(const T& a, const T& x) { return gmul(1, 1, 0, a, x); }
  static T vecMat(const T& x, const T& a) { return gmul(0, 1, 1, x, a); }
  static T matMat(const T& a, const T& b) { return gmul(1, 1, 1, a, b); }
  // gmul followed by the sum over the hidden batch (the cotangent of an unbatched operand)
  static T gmul_batch_sum(int lm, int lo, int ln, const T& a, const T& b) {
    to_tensor out = nullptr;
    check(to_gmul_batch_sum(lm, lo, ln, a.h(), b.h(), &out));
    if (trace::on()) trace::call("gmul_batch_sum", {std::to_string(lm), std::to_string(lo), std::to_string(ln)}, {a.h(), b.h()}, out);
    return T(out);
  }
  // sumT (Types.hs:69); dims = the `SingI o` evidence
  static T sumT(const std::vector<T>& xs, const Dims& dims) {
    std::vector<to_tensor> hs;
    for (const T& x : xs) hs.push_back(x.h());
    to_tensor out = nullptr;
    check(to_sum((int)hs.size(), hs.data(), (int)dims.size(), dims.data(), &out));
    if (trace::on()) trace::call("sumT", {std::to_string(hs.size())}, hs, out);
    return T(out);
  }
  static T scaleT(double alpha, const T& x) {  // Types.hs:70
    to_tensor out = nullptr;
    check(to_scale(alpha, x.h(), &out));
    if (trace::on()) trace::call("scaleT", {trace::num(alpha)}, {x.h()}, out);
    return T(out);
  }
  static T transp(const T& x) {  // Types.hs:71-73
    to_tensor out = nullptr;
    check(to_transp(x.h(), &out));
    if (trace::on()) trace::call("transp", {}, {x.h()}, out);
    return T(out);
  }
  static T sumRows(const T& x) {  // Types.hs:82-84
    to_tensor out = nullptr;
    check(to_sum_rows(x.h(), &out));
    if (trace::on()) trace::call("sumRows", {}, {x.h()}, out);
    return T(out);
  }
  // mapRows (Types.hs:77-81): a host traversal over zero-copy row views.  The view handed to `f` is a thunk, as in
  // Haskell: a closure that ignores its argument (`\_ -> dtdz`, TOp.hs:158) never makes the to_slice call.
  static T mapRows(int len_n, const std::function<T(const LT&)>& f, const T& x) {
    Dims d = x.dims();
    Dims lead(d.begin(), d.begin() + len_n);
    int64_t n = 1;
    for (int64_t v : lead) n *= v;
    std::vector<T> rows;
    std::vector<int64_t> idx(len_n, 0);
    for (int64_t r = 0; r < n; ++r) {
      rows.push_back(f(LT(std::function<T()>([x, idx, len_n]() {
        to_tensor v = nullptr;
        check(to_slice(x.h(), len_n, idx.data(), &v));
        if (trace::on()) trace::call("row", idx_strings(idx), {x.h()}, v);
        return T(v);
      }))));
      for (int k = len_n - 1; k >= 0; --k) {
        if (++idx[k] < lead[k]) break;
        idx[k] = 0;
      }
    }
    std::vector<to_tensor> hs;
    for (const T& t : rows) hs.push_back(t.h());
    to_tensor out = nullptr;
    check(to_stack(len_n, lead.data(), hs.data(), &out));
    if (trace::on()) {
      std::vector<to_tensor> ins{x.h()};
      ins.insert(ins.end(), hs.begin(), hs.end());
      trace::call("mapRows", {std::to_string(len_n)}, ins, out);
    }
    return T(out);
  }
  // ixRows (Types.hs:100-106) at Identity: like mapRows, the closure also gets the row's index and may change the row shape
  static T ixRows(int len_m, const std::function<T(const Dims&, const LT&)>& f, const T& x) {
    Dims d = x.dims();
    Dims lead(d.begin(), d.begin() + len_m);
    int64_t n = 1;
    for (int64_t v : lead) n *= v;
    std::vector<T> rows;
    std::vector<int64_t> idx(len_m, 0);
    for (int64_t r = 0; r < n; ++r) {
      rows.push_back(f(idx, LT(std::function<T()>([x, idx, len_m]() {
        to_tensor v = nullptr;
        check(to_slice(x.h(), len_m, idx.data(), &v));
        if (trace::on()) trace::call("row", idx_strings(idx), {x.h()}, v);
        return T(v);
      }))));
      for (int k = len_m - 1; k >= 0; --k) {
        if (++idx[k] < lead[k]) break;
        idx[k] = 0;
      }
    }
    std::vector<to_tensor> hs;
    for (const T& t : rows) hs.push_back(t.h());
    to_tensor out = nullptr;
    check(to_stack(len_m, lead.data(), hs.data(), &out));
    if (trace::on()) {
      std::vector<to_tensor> ins{x.h()};
      ins.insert(ins.end(), hs.begin(), hs.end());
      trace::call("ixRows", {std::to_string(len_m)}, ins, out);
    }
    return T(out);
  }
  static T diag(int rank, const T& x) {  // Types.hs:85-88
    to_tensor out = nullptr;
    check(to_diag(rank, x.h(), &out));
    if (trace::on()) trace::call("diag", {std::to_string(rank)}, {x.h()}, out);
    return T(out);
  }
  static T getDiag(const T& x) {  // Types.hs:89-92
    to_tensor out = nullptr;
    check(to_get_diag(x.h(), &out));
    if (trace::on()) trace::call("getDiag", {}, {x.h()}, out);
    return T(out);
  }
  // genRand (Types.hs:93-96)
  static T genRand(const Dims& dims, int dist, double a, double b, uint64_t seed, int64_t batch = 0) {
    to_tensor out = nullptr;
    check(to_rand(elem_dtype(), (int)dims.size(), dims.data(), batch, dist, a, b, seed, &out));
    return T(out);
  }
  // generateA (Types.hs:97-99) at Identity: build on the host, upload once
  static T generate(const Dims& dims, const std::function<double(const Dims&)>& f) {
    int64_t n = 1;
    for (int64_t v : dims) n *= v;
    const int dt = elem_dtype();
    std::vector<float> host(dt == TO_F32 ? (size_t)n : 0);
    std::vector<double> host64(dt == TO_F64 ? (size_t)n : 0);
    Dims idx(dims.size(), 0);
    for (int64_t e = 0; e < n; ++e) {
      if (dt == TO_F64) host64[(size_t)e] = f(idx);
      else host[(size_t)e] = (float)f(idx);
      for (int k = (int)dims.size() - 1; k >= 0; --k) {
        if (++idx[k] < dims[k]) break;
        idx[k] = 0;
      }
    }
    to_tensor out = nullptr;
    check(to_from_host(dt, (int)dims.size(), dims.data(), 0,
                       dt == TO_F64 ? (const void*)host64.data() : (const void*)host.data(), &out));
    if (trace::on()) {
      std::vector<std::string> vals;
      for (int64_t e = 0; e < n; ++e) vals.push_back(trace::num(dt == TO_F64 ? host64[(size_t)e] : (double)host[(size_t)e]));
      trace::call("generateA", vals, {}, out);
    }
    return T(out);
  }
  static T konst(const Dims& dims, double x) {  // TT.konst, Tensor.hs:49-54
    to_tensor out = nullptr;
    check(to_fill(elem_dtype(), (int)dims.size(), dims.data(), 0, x, &out));
    if (trace::on()) trace::call("konst", {trace::num(x)}, {}, out);
    return T(out);
  }
  static double index(const T& x, const Dims& i, int64_t sample = 0) {  // (!) Types.hs:107-109
    double v = 0;
    check(to_index(x.h(), i.data(), sample, &v));
    return v;
  }
  // TT.argMax (Tensor.hs:291-305): one index per sample (a single download)
  static std::vector<int64_t> argMax(const T& x) {
    const int64_t b = x.batch();
    std::vector<int64_t> out((size_t)(b > 0 ? b : 1));
    check(to_arg_max(x.h(), out.data()));
    return out;
  }
  static std::vector<int64_t> argMin(const T& x) {  // TT.argMin (Tensor.hs:307-321)
    const int64_t b = x.batch();
    std::vector<int64_t> out((size_t)(b > 0 ? b : 1));
    check(to_arg_min(x.h(), out.data()));
    return out;
  }
  // TT.oneHot (Tensor.hs:275-289) for a batch of class indices (empty batch argument = one vector)
  static T oneHot(int64_t n, double hot, double cold, const std::vector<int64_t>& idx, bool batched) {
    to_tensor out = nullptr;
    check(to_one_hot(elem_dtype(), n, hot, cold, batched ? (int64_t)idx.size() : 0, idx.data(), &out));
    return T(out);
  }
  static T batch_sum(const T& x) {
    to_tensor out = nullptr;
    check(to_batch_sum(x.h(), &out));
    if (trace::on()) trace::call("batch_sum", {}, {x.h()}, out);
    return T(out);
  }
};

// `rnf` of a product of tensors in one call (to_force_many): the values are planned together
inline void forceAll(const std::vector<T>& ts) {
  std::vector<to_tensor> hs;
  for (const T& t : ts) hs.push_back(t.h());
  check(to_force_many((int)hs.size(), hs.data()));
}

}  // namespace tensorops
