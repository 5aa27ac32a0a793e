// The batched training step over the C ABI: batched `gradTOp` of `net *>> loss` restricted to the
// parameters (what `netGrad` keeps, FeedForward.hs:187-199) landing in ONE flat gradient buffer,
// plus the SGD update `p - r*g` (FeedForward.hs:141-147) on the flat parameter buffer.
//
//  * scope  = to_memo_begin .. to_memo_end around one gradTOp: CSE of the forward passes the composition
//             recomputes (Types.hs:155), and -- TRAINER_FUSED -- the class-method stream is recorded and fused
//             by the library (csrc/lazy.cpp).  Nothing here knows what the network is made of: `Network` has
//             no activation tags, exactly like the reference's (FeedForward.hs:57-61).
//  * graph  = the step captured once as a HIP graph and replayed
//
// `trainAll` is `foldl' trainNetwork` (app/MNIST.hs:390-393, app/Dots.hs:74-80): per-sample online
// SGD over rows of a resident data set.
#pragma once
#include <cstdlib>
#include <memory>

#include "learn.hpp"

namespace tensorops {

// TRAINER_FUSED: let the library defer and fuse the recorded method stream (off: one launch per class-method call)
// TRAINER_FRESH_THUNKS: build gradTOp's thunk graph afresh on every directly-issued step (default: built once, kept)
enum { TRAINER_MEMO = 1, TRAINER_GRAPH = 2, TRAINER_FUSED = 4, TRAINER_FRESH_THUNKS = 8 };

inline int dtype_of(const T& t) {
  int dt = TO_F32;
  check(to_dtype(t.h(), &dt));
  return dt;
}

// the library's deferral switch for the duration of one call
struct LazyMode {
  int prev = 1;
  explicit LazyMode(bool on) { check(to_set_lazy(on ? 1 : 0, &prev)); }
  ~LazyMode() { to_set_lazy(prev, nullptr); }
};

class Trainer {
 public:
  Network net;  // params are views into `flat_p`
  TOp loss;
  double rate = 0;
  T x, y;
  T flat_p, flat_g;
  std::vector<int64_t> offs, sizes;
  std::vector<T> gviews;
  int64_t n_flat = 0;
  to_graph graph = nullptr;       // grad(): gradients into flat_g
  to_graph step_graph = nullptr;  // step(): the whole trainNetwork step, parameters updated in place
  bool use_memo = true;
  bool use_graph = false;
  bool fused = false;
  bool keep_thunks = true;
  int loss_id = 0;
  int dtype = TO_F32;
  int64_t launches = 0;       // kernel launches of one grad()
  int64_t step_launches = 0;  // ... of one step()
  to_expr update_expr = nullptr;  // `\p g -> p - r*g` compiled once (the rate is fixed for the trainer's lifetime)

  // A new batch: the kept thunk graph's closures hold copies of the old x / y leaves (and so their device buffers) -- they
  // are let go HERE, not at the next step's rebuild
  void set_data(const T& nx, const T& ny) {
    if (nx.h() != x.h() || ny.h() != y.h()) drop_kept();
    x = nx;
    y = ny;
  }

  Trainer() = default;
  Trainer(const Trainer&) = delete;
  Trainer& operator=(const Trainer&) = delete;
  ~Trainer() {
    if (graph) to_graph_release(graph);
    if (step_graph) to_graph_release(step_graph);
  }

  // elements of the flat buffers: every tensor starts on a 16-byte boundary (4 elements)
  static int64_t flat_size(const Network& n) {
    int64_t total = 0;
    for (const T& p : n.params) {
      int64_t sz = 1;
      for (int64_t d : p.dims()) sz *= d;
      total += (sz + 3) / 4 * 4;
    }
    return total;
  }

  // ext_params / ext_grads: caller-owned flat device buffers (e.g. torch tensors handed to
  // torch.distributed) or both null
  static std::unique_ptr<Trainer> create(const Network& n, int loss, double rate, const T& x, const T& y, int flags,
                                         void* ext_params = nullptr, void* ext_grads = nullptr) {
    auto t = std::make_unique<Trainer>();
    t->loss = loss_of(loss);
    t->loss_id = loss;
    t->rate = rate;
    t->x = x;
    t->y = y;
    t->use_memo = (flags & TRAINER_MEMO) != 0;
    t->use_graph = (flags & TRAINER_GRAPH) != 0;
    t->keep_thunks = (flags & TRAINER_FRESH_THUNKS) == 0;
    if (n.params.empty()) throw TensorOpsError(TO_ERR_ARG, "trainer: the network has no parameters");
    const int dt = t->dtype = dtype_of(n.params[0]);
    for (const T& p : n.params)
      if (dtype_of(p) != dt) throw TensorOpsError(TO_ERR_ARG, "trainer: parameters of different dtypes");
    const size_t es = dt == TO_F64 ? 8 : 4;
    t->fused = (flags & TRAINER_FUSED) && t->use_memo;  // deferral lives in the scope
    int64_t total = 0;
    for (const T& p : n.params) {
      int64_t sz = 1;
      for (int64_t d : p.dims()) sz *= d;
      t->offs.push_back(total);
      t->sizes.push_back(sz);
      total += (sz + 3) / 4 * 4;
    }
    t->n_flat = total;
    Dims fd{total};
    to_tensor fp = nullptr, fg = nullptr;
    if ((ext_params == nullptr) != (ext_grads == nullptr))
      throw TensorOpsError(TO_ERR_ARG, "give both external flat buffers or neither");
    if (ext_params) {
      check(to_wrap(ext_params, dt, 1, fd.data(), 0, &fp));
      t->flat_p = T(fp);
      check(to_wrap(ext_grads, dt, 1, fd.data(), 0, &fg));
      t->flat_g = T(fg);
    } else {
      check(to_fill(dt, 1, fd.data(), 0, 0.0, &fp));
      t->flat_p = T(fp);
      check(to_fill(dt, 1, fd.data(), 0, 0.0, &fg));
      t->flat_g = T(fg);
    }
    void *pp = nullptr, *gp = nullptr;
    check(to_data_ptr(fp, &pp));
    check(to_data_ptr(fg, &gp));
    t->net.op = n.op;
    for (size_t i = 0; i < n.params.size(); ++i) {
      const T& p = n.params[i];
      Dims d = p.dims();
      to_tensor pv = nullptr, gv = nullptr;
      check(to_wrap((char*)pp + t->offs[i] * es, dt, (int)d.size(), d.data(), 0, &pv));
      t->net.params.emplace_back(pv);
      check(to_wrap((char*)gp + t->offs[i] * es, dt, (int)d.size(), d.data(), 0, &gv));
      t->gviews.emplace_back(gv);
      check(to_copy_into(pv, p.h()));
    }
    check(to_sync());
    // warm-up run: compiles expressions, fills the pool, counts launches
    int64_t l0 = 0, l1 = 0;
    check(to_stats(nullptr, nullptr, &l0));
    t->body_in_scope(false);
    check(to_stats(nullptr, nullptr, &l1));
    t->launches = l1 - l0;
    check(to_sync());
    if (t->use_graph) t->graph = t->capture(false);
    return t;
  }

  // G <- summed parameter gradients
  void grad() {
    if (graph) check(to_graph_launch(graph));
    else body_in_scope(false);
  }
  // P <- P - rate * G (in place on the flat buffer)
  void apply() { check(to_sgd_step_inplace(flat_p.h(), flat_g.h(), rate)); }

  // One `trainNetwork` step (FeedForward.hs:131-148) on the batch, parameters updated in place.  With the
  // library's fusion on, the update is written exactly as the reference writes it -- `TT.zip (\p g -> p - r*g)`
  // on the gradient -- inside the same scope, and lands in the parameter buffer through to_copy_into: the
  // planner folds it into the weight-gradient launches, so `flat_g` is NOT written.  Otherwise grad(); apply().
  void step() {
    if (!fused) {
      grad();
      apply();
      return;
    }
    if (use_graph) {
      if (!step_graph) {
        count_step();                 // the first step runs directly (and counts its launches) ...
        step_graph = capture(true);   // ... later ones replay this capture (recording does not execute)
        return;
      }
      check(to_graph_launch(step_graph));
      return;
    }
    if (!step_launches) count_step();
    else body_in_scope(true);
  }

  // one graph = the gradient, or (with_update) the whole step
  to_graph capture(bool with_update) {
    check(to_graph_begin());
    try {
      body_in_scope(with_update);
    } catch (...) {
      to_graph g = nullptr;
      to_graph_end(&g);
      if (g) to_graph_release(g);
      throw;
    }
    to_graph g = nullptr;
    check(to_graph_end(&g));
    return g;
  }

 private:
  void count_step() {  // the first step() runs directly and counts its launches
    int64_t l0 = 0, l1 = 0;
    check(to_stats(nullptr, nullptr, &l0));
    body_in_scope(true);
    check(to_stats(nullptr, nullptr, &l1));
    step_launches = l1 - l0;
  }
  void body_in_scope(bool with_update) {
    LazyMode lm(fused);
    if (use_memo) check(to_memo_begin());
    try {
      body(with_update);
    } catch (...) {
      if (use_memo) to_memo_end();
      throw;
    }
    if (use_memo) check(to_memo_end());
  }
  // The thunk graph of `sumOverBatch (netGrad loss x y net)`, built once and evaluated by every step for as long as
  // x, y and the parameters are the handles it was built over (LT::Graph: a step's host cost was 40 us, 7 of them this
  // construction).  TRAINER_FRESH_THUNKS builds it afresh every step, as the reference's evaluator would.
  struct KeptGrad {
    LT::Graph graph;
    Prod g;
    std::vector<to_tensor> leaves;
    int rebuilds = 0;  // a caller that changes x / y every step (trainAll over row views) gains nothing: stop keeping
  } kept;
  std::vector<to_tensor> leaf_handles() const {
    std::vector<to_tensor> l{x.h(), y.h()};
    for (const T& p : net.params) l.push_back(p.h());
    return l;
  }
  void drop_kept() {
    kept.g.clear();
    kept.graph.clear();
    kept.leaves.clear();
  }
  void body(bool with_update) {
    // G_i = sum_b (gradTOp (net *>> loss) (x_b, p, y_b))_i -- the params are unbatched, so their cotangents are
    // summed over the samples where gradTOp returns (sumOverBatch; the library folds the sum into the GEMM)
    std::vector<T> outs;
    const bool keep = keep_thunks && kept.rebuilds < 4;
    if (!keep) drop_kept();   // (keeping abandoned: nothing of the last graph stays pinned)
    try {
      Prod fresh;
      if (keep) {
        std::vector<to_tensor> leaves = leaf_handles();
        if (kept.g.empty() || leaves != kept.leaves) {
          drop_kept();
          ++kept.rebuilds;
          LT::Recording rec(&kept.graph);
          kept.g = netGradBatch(loss, x, y, net);
          kept.leaves = std::move(leaves);
        }
      } else {
        fresh = netGradBatch(loss, x, y, net);
      }
      const Prod& g = keep ? kept.g : fresh;
      const double r = rate;
      for (size_t i = 0; i < net.params.size(); ++i) {
        T gi = g[i + 1].get();
        if (with_update)  // stepFunc (FeedForward.hs:145-147)
          outs.push_back(HipT::liftT([r](const std::vector<Expr>& v) { return v[0] - Expr(r) * v[1]; },
                                     {net.params[i], gi}, &update_expr));
        else
          outs.push_back(gi);
      }
    } catch (...) {
      drop_kept();
      throw;
    }
    // the thunks of the backward pass die here (a kept graph forgets their values): what they held is no longer
    // visible to the host
    if (keep) kept.graph.reset();
    std::vector<to_tensor> dst, src;
    for (size_t i = 0; i < net.params.size(); ++i) {
      dst.push_back(with_update ? net.params[i].h() : gviews[i].h());
      src.push_back(outs[i].h());
    }
    // land them in the flat buffer (deferred results are produced there directly)
    check(to_copy_into_many((int)dst.size(), dst.data(), src.data()));
  }
};

// `foldl' (\nt (i,o) -> trainNetwork loss rate i o nt)` over rows idx[0..n_idx) of the resident,
// contiguous, batched X / Y (idx null = rows 0..n_idx-1)
inline Network trainAll(const Network& n, int loss, double rate, const T& X, const T& Y, int64_t n_idx,
                        const int64_t* idx, int flags) {
  const Dims xd = X.dims(), yd = Y.dims();
  const int64_t xb = X.batch(), yb = Y.batch();
  const int xdt = dtype_of(X), ydt = dtype_of(Y);
  int xc = 0, yc = 0;
  check(to_is_contiguous(X.h(), &xc));
  check(to_is_contiguous(Y.h(), &yc));
  if (xb < 1 || xb != yb || !xc || !yc || xdt != ydt)
    throw TensorOpsError(TO_ERR_ARG,
                         "trainAll: x and y must be contiguous batched tensors of one batch size and dtype");
  if (n_idx < 0) throw TensorOpsError(TO_ERR_ARG, "trainAll: negative sample count");
  for (int64_t k = 0; k < n_idx; ++k) {
    const int64_t i = idx ? idx[k] : k;
    if (i < 0 || i >= xb) throw TensorOpsError(TO_ERR_SHAPE, "trainAll: sample index out of range");
  }
  int64_t xn = 1, yn = 1;
  for (int64_t d : xd) xn *= d;
  for (int64_t d : yd) yn *= d;
  const size_t es = xdt == TO_F64 ? 8 : 4;
  const bool fused = (flags & TRAINER_FUSED) && (flags & TRAINER_MEMO);
  // one-sample staging buffers: a hidden batch of 1 when the library fuses (rows of samples are what its
  // GEMM epilogues work on), plain unbatched tensors (exactly `trainNetwork`'s arguments) otherwise
  const int64_t sb = fused ? 1 : 0;
  to_tensor hx = nullptr, hy = nullptr;
  check(to_alloc(xdt, (int)xd.size(), xd.data(), sb, &hx));
  T xbuf(hx);
  check(to_alloc(ydt, (int)yd.size(), yd.data(), sb, &hy));
  T ybuf(hy);
  void *xp = nullptr, *yp = nullptr;
  check(to_data_ptr(X.h(), &xp));
  check(to_data_ptr(Y.h(), &yp));
  auto stage = [&](int64_t i) {
    to_tensor vx = nullptr, vy = nullptr;
    check(to_wrap((char*)xp + (size_t)i * xn * es, xdt, (int)xd.size(), xd.data(), sb, &vx));
    T tx(vx);
    check(to_wrap((char*)yp + (size_t)i * yn * es, ydt, (int)yd.size(), yd.data(), sb, &vy));
    T ty(vy);
    const to_tensor dst[2] = {xbuf.h(), ybuf.h()}, src[2] = {tx.h(), ty.h()};
    check(to_copy_into_many(2, dst, src));  // one launch for both
  };
  stage(n_idx > 0 ? (idx ? idx[0] : 0) : 0);
  auto tr = Trainer::create(n, loss, rate, xbuf, ybuf, flags & ~TRAINER_GRAPH);
  // one sample = one replay of a captured graph (gradTOp + update); TOPS_ONLINE_GRAPH=0 issues the launches
  // directly and reads row i of the resident data set through a view instead of staging it
  static const int online_graph = [] { const char* e = getenv("TOPS_ONLINE_GRAPH"); return e ? atoi(e) : -1; }();
  const bool use_graph = online_graph >= 0 ? online_graph != 0 : true;
  to_graph graph = nullptr;
  if (use_graph) {
    if (tr->fused) {
      graph = tr->capture(true);  // (capture records, it does not run: no sample is trained twice)
    } else {
      check(to_graph_begin());
      try {
        tr->grad();
        tr->apply();
      } catch (...) {
        to_graph g = nullptr;
        to_graph_end(&g);
        if (g) to_graph_release(g);
        throw;
      }
      check(to_graph_end(&graph));
    }
  }
  const bool views = !graph;
  // The library looks at what it made of the captured step: when that is the trainNetwork step of an ffLayer stack it
  // runs the whole stream of samples as one persistent launch (csrc/online_sgd.hip) -- nothing here says what `n` is.
  int handled = 0;
  if (graph && tr->fused) {
    to_status st = to_graph_online_sgd(graph, xbuf.h(), ybuf.h(), X.h(), Y.h(), n_idx, idx, &handled);
    if (st != TO_OK) {
      to_graph_release(graph);
      check(st);
    }
  }
  try {
    for (int64_t k = 0; k < n_idx && !handled; ++k) {
      const int64_t i = idx ? idx[k] : k;
      if (views) {
        to_tensor vx = nullptr, vy = nullptr;
        check(to_wrap((char*)xp + (size_t)i * xn * es, xdt, (int)xd.size(), xd.data(), sb, &vx));
        check(to_wrap((char*)yp + (size_t)i * yn * es, ydt, (int)yd.size(), yd.data(), sb, &vy));
        tr->set_data(T(vx), T(vy));
      } else {
        stage(i);
      }
      if (graph) {
        check(to_graph_launch(graph));
      } else {
        tr->step();
      }
    }
  } catch (...) {
    if (graph) to_graph_release(graph);
    throw;
  }
  if (graph) to_graph_release(graph);
  // the result owns fresh parameter tensors (the flat buffer dies with the trainer)
  Network res{tr->net.op, {}};
  for (const T& p : tr->net.params) {
    Dims d = p.dims();
    to_tensor c = nullptr;
    check(to_alloc(xdt, (int)d.size(), d.data(), 0, &c));
    T ct(c);
    check(to_copy_into(ct.h(), p.h()));
    res.params.push_back(ct);
  }
  check(to_sync());
  return res;
}

// `induceNum n t r iters x0` (app/MNIST.hs:399-411): `iters` steps of `induceNetwork loss r t n`
// (FeedForward.hs:150-164) -- gradient descent on the INPUT.  One step (gradTOp with the input's
// cotangent forced, the update, the copy back into the staging buffer) is captured as a HIP
// graph and replayed.
inline T induceNum(const Network& n, const TOp& loss, const T& target, double r, int iters, const T& x0) {
  Dims d = x0.dims();
  to_tensor hb = nullptr;
  check(to_alloc(dtype_of(x0), (int)d.size(), d.data(), x0.batch(), &hb));
  T xbuf(hb);
  check(to_copy_into(xbuf.h(), x0.h()));
  if (iters <= 0) return xbuf;
  auto step = [&]() {
    check(to_memo_begin());
    try {
      T x1 = induceNetwork(loss, r, target, n, xbuf);
      check(to_copy_into(xbuf.h(), x1.h()));
    } catch (...) {
      to_memo_end();
      throw;
    }
    check(to_memo_end());
  };
  step();  // warm-up: compiles the closures, fills the pool
  if (iters == 1) return xbuf;
  check(to_sync());
  check(to_graph_begin());
  try {
    step();
  } catch (...) {
    to_graph g = nullptr;
    to_graph_end(&g);
    if (g) to_graph_release(g);
    throw;
  }
  to_graph g = nullptr;
  check(to_graph_end(&g));
  // capture recorded the step without executing it: iterations 2..iters are replays
  for (int i = 1; i < iters; ++i) {
    to_status st = to_graph_launch(g);
    if (st != TO_OK) {
      to_graph_release(g);
      check(st);
    }
  }
  to_graph_release(g);
  check(to_sync());
  return xbuf;
}

}  // namespace tensorops
