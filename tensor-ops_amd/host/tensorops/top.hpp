// The `TOp` DSL (src/TensorOps/Types.hs:122-264, src/TensorOps/TOp.hs) over HipT.
//
// `TOp ns ms` is a record of two closures (Types.hs:122-125).  Products are
// vectors of LAZY tensors: Haskell's call-by-need matters here -- a backward
// closure that ignores its recomputed forward input (`add`, TOp.hs:218) never
// forces it, and cotangents nobody demands (the input's, which
// `trainNetwork` drops with `tail'`, FeedForward.hs:142) are never computed.
// Shapes are run-time values validated by the C ABI (type-level in Haskell).
//
// Rule for whoever writes a gradient builder here: NEVER evaluate a tensor while building the closures -- every value is
// taken inside a thunk, at force time.  The trainer keeps gradTOp's thunk graph across steps (trainer.hpp, LT::Graph) and
// validates it by the identity of its leaf handles only: a builder that captured a VALUE eagerly would hand later steps a
// stale one.  `tests/test_gpu_host_mirror.py::test_a_kept_thunk_graph_issues_the_same_calls_as_fresh_thunks` holds every
// activation / loss combination to it (kept graph vs fresh thunks over five steps whose parameters change in place: equal
// call counts, equal launches, bit-identical parameters -- a value captured at build time would show as a stale step).
//
// Batches (the one extension): nothing in this file knows about them, exactly like the reference's
// DSL.  With batched data the cotangent of an unbatched input simply comes back batched; the host sums it where
// `gradTOp` returns (`sumOverBatch` below = the shim's `batchSum`), and the library folds that sum into the
// recorded producer (to_batch_sum of a recorded gmul IS to_gmul_batch_sum, csrc/api.cpp).
#pragma once
#include "tensor.hpp"

namespace tensorops {

using Prod = std::vector<LT>;

// a lazily computed Prod of known length (the recomputed `f1 xs`, Types.hs:155)
inline Prod lazy_prod(size_t n, std::function<Prod()> f) {
  auto cell = std::make_shared<std::pair<bool, Prod>>(false, Prod());
  auto thunk = std::make_shared<std::function<Prod()>>(std::move(f));
  if (LT::Graph* g = LT::recording())  // part of a kept graph: the recomputed product is forgotten with the thunks
    g->on_reset([cell]() {
      cell->first = false;
      cell->second.clear();
    });
  Prod out;
  for (size_t i = 0; i < n; ++i)
    out.emplace_back(std::function<T()>([cell, thunk, i]() {
      if (!cell->first) {
        cell->second = (*thunk)();
        cell->first = true;
      }
      return cell->second[i].get();
    }));
  return out;
}

inline Prod slice(const Prod& p, size_t a, size_t b) { return Prod(p.begin() + a, p.begin() + b); }
inline Prod concat(Prod a, const Prod& b) {
  a.insert(a.end(), b.begin(), b.end());
  return a;
}

struct TOp {
  using RunFn = std::function<Prod(const Prod&)>;
  using GradFn = std::function<Prod(const Prod&, const Prod&)>;
  int n_in = 0, n_out = 0;  // `Known Length ns`, `Known Length ms`
  RunFn run;                // runTOp
  GradFn grad;              // gradTOp'
  TOp() = default;
  // The closures are held behind shared pointers: a combinator captures its operands BY VALUE,
  // so without sharing the in-memory size of a composition (and the cost of copying it, which
  // every backward pass does for its recompute thunks) would double per nesting level.
  TOp(int i, int o, RunFn r, GradFn g) : n_in(i), n_out(o) {
    auto pr = std::make_shared<const RunFn>(std::move(r));
    auto pg = std::make_shared<const GradFn>(std::move(g));
    run = [pr](const Prod& xs) { return (*pr)(xs); };
    grad = [pg](const Prod& xs, const Prod& ds) { return (*pg)(xs, ds); };
  }
};

inline void arity_check(bool ok, const char* what) {
  if (!ok) throw TensorOpsError(TO_ERR_SHAPE, std::string("TOp arity mismatch in ") + what);
}

inline Prod runTOp(const TOp& o, const Prod& xs) {
  arity_check((int)xs.size() == o.n_in, "runTOp");
  return o.run(xs);
}
// gradTOp (Types.hs:127-132): seed the scalar output with 1
inline Prod gradTOp(const TOp& o, const Prod& xs) {
  arity_check((int)xs.size() == o.n_in && o.n_out == 1, "gradTOp");
  // `only (getI $ generateA (\_ -> I 1))` (Types.hs:132): the seed goes through generateA like any other built value
  // (in a kept graph it is a thunk like the rest, generated again by every evaluation)
  auto seed = []() { return HipT::generate({}, [](const Dims&) { return 1.0; }); };
  return o.grad(xs, Prod{LT::recording() ? LT(std::function<T()>(seed)) : LT(seed())});
}

// The batching rule, applied where a gradient is handed back to the host (never inside the DSL): the cotangent of an
// unbatched input is the sum of its per-sample cotangents.  In Haskell this is `batchSum <$> gradTOp o xs` in the
// shim's `trainBatch` (hs/TensorOps/Backend/HipTensor.hs); a no-op when nothing is batched.
inline Prod sumOverBatch(const Prod& g, const Prod& xs) {
  arity_check(g.size() <= xs.size(), "sumOverBatch");
  Prod out;
  for (size_t i = 0; i < g.size(); ++i) {
    LT gi = g[i], xi = xs[i];
    out.emplace_back(std::function<T()>([gi, xi]() {
      const T& c = gi.get();
      return !xi.get().batched() && c.batched() ? HipT::batch_sum(c) : c;
    }));
  }
  return out;
}

// ---- Category and products (Types.hs:135-264) --------------------------------------------------
inline TOp idOp(int n) {
  return TOp{n, n, [](const Prod& xs) { return xs; }, [](const Prod&, const Prod& ds) { return ds; }};
}
// f1 >>> f2 ;  g3 xs ds = g1 xs (g2 (f1 xs) ds)   (Types.hs:139-156)
inline TOp compose(const TOp& f1, const TOp& f2) {
  arity_check(f1.n_out == f2.n_in, ">>>");
  return TOp{f1.n_in, f2.n_out,
             [f1, f2](const Prod& xs) { return f2.run(f1.run(xs)); },
             [f1, f2](const Prod& xs, const Prod& ds) {
               Prod mid = lazy_prod((size_t)f1.n_out, [f1, xs]() { return f1.run(xs); });
               return f1.grad(xs, f2.grad(mid, ds));
             }};
}
inline TOp operator>>(const TOp& a, const TOp& b) { return compose(a, b); }
// firstOp (Types.hs:165-182) with the pass-through arity `os` explicit
inline TOp firstOp(const TOp& o, int n_pass) {
  return TOp{o.n_in + n_pass, o.n_out + n_pass,
             [o](const Prod& xs) { return concat(o.run(slice(xs, 0, o.n_in)), slice(xs, o.n_in, xs.size())); },
             [o](const Prod& xs, const Prod& ds) {
               return concat(o.grad(slice(xs, 0, o.n_in), slice(ds, 0, o.n_out)), slice(ds, o.n_out, ds.size()));
             }};
}
// secondOp (Types.hs:184-201)
inline TOp secondOp(int n_skip, const TOp& o) {
  return TOp{n_skip + o.n_in, n_skip + o.n_out,
             [o, n_skip](const Prod& xs) { return concat(slice(xs, 0, n_skip), o.run(slice(xs, n_skip, xs.size()))); },
             [o, n_skip](const Prod& xs, const Prod& ds) {
               return concat(slice(ds, 0, n_skip), o.grad(slice(xs, n_skip, xs.size()), slice(ds, n_skip, ds.size())));
             }};
}
// t1 *>> t2 = firstOp t1 >>> t2   (Types.hs:204-209)
inline TOp then_first(const TOp& t1, const TOp& t2) {
  arity_check(t2.n_in >= t1.n_out, "*>>");
  return compose(firstOp(t1, t2.n_in - t1.n_out), t2);
}
// *** (Types.hs:221-241)
inline TOp par(const TOp& a, const TOp& b) {
  return TOp{a.n_in + b.n_in, a.n_out + b.n_out,
             [a, b](const Prod& xs) { return concat(a.run(slice(xs, 0, a.n_in)), b.run(slice(xs, a.n_in, xs.size()))); },
             [a, b](const Prod& xs, const Prod& ds) {
               return concat(a.grad(slice(xs, 0, a.n_in), slice(ds, 0, a.n_out)),
                             b.grad(slice(xs, a.n_in, xs.size()), slice(ds, a.n_out, ds.size())));
             }};
}
// &&& (Types.hs:243-264): backward sums the two cotangents with sumT [gxy, gxz]
inline TOp fanout(const TOp& a, const TOp& b) {
  arity_check(a.n_in == b.n_in, "&&&");
  return TOp{a.n_in, a.n_out + b.n_out,
             [a, b](const Prod& xs) { return concat(a.run(xs), b.run(xs)); },
             [a, b](const Prod& xs, const Prod& ds) {
               Prod g1 = a.grad(xs, slice(ds, 0, a.n_out)), g2 = b.grad(xs, slice(ds, a.n_out, ds.size()));
               Prod out;
               for (size_t i = 0; i < g1.size(); ++i) {
                 LT x = xs[i], p = g1[i], q = g2[i];
                 out.emplace_back(std::function<T()>([x, p, q]() {
                   return HipT::sumT({p.get(), q.get()}, x.get().dims());  // `SingI as` evidence = x's dims
                 }));
               }
               return out;
             }};
}

// ---- op vocabulary (src/TensorOps/TOp.hs) ---------------------------------------------------------
// VFunc (Types.hs:114-117) at the symbolic element type
struct VFunc {
  int n = 0;
  std::function<Expr(const std::vector<Expr>&)> f;
  std::function<std::vector<Expr>(const std::vector<Expr>&)> g;
};

// liftOp (TOp.hs:42-54) + TT.gradLift (Tensor.hs:119-129):
// dx_i = liftT (\(d :* x) -> d * (vfGrad f x)_i) (dtdy :* xs) -- one (n+1)-ary pass per input
inline TOp liftOp(const VFunc& vf) {
  const int n = vf.n;
  arity_check(n >= 1, "liftOp (use konst for n = 0)");
  // the compiled forms of this op's closures (forward, one backward per input), filled on first use
  auto compiled = std::make_shared<std::vector<to_expr>>((size_t)n + 1, nullptr);
  return TOp{n, 1,
             [vf, compiled](const Prod& xs) {
               return Prod{LT(std::function<T()>([vf, xs, compiled]() {
                 std::vector<T> v;
                 for (const LT& x : xs) v.push_back(x.get());
                 return HipT::liftT(vf.f, v, &(*compiled)[0]);
               }))};
             },
             [vf, n, compiled](const Prod& xs, const Prod& ds) {
               Prod out;
               for (int i = 0; i < n; ++i) {
                 LT d = ds[0];
                 out.emplace_back(std::function<T()>([vf, xs, d, i, compiled]() {
                   std::vector<T> v{d.get()};
                   for (const LT& x : xs) v.push_back(x.get());
                   return HipT::liftT(
                       [vf, i](const std::vector<Expr>& dx) {
                         std::vector<Expr> x(dx.begin() + 1, dx.end());
                         return dx[0] * vf.g(x)[i];
                       },
                       v, &(*compiled)[(size_t)i + 1]);
                 }));
               }
               return out;
             }};
}

// gmul (TOp.hs:56-94)
inline TOp gmul(int lm, int lo, int ln) {
  return TOp{2, 1,
             [=](const Prod& xs) {
               LT x = xs[0], y = xs[1];
               return Prod{LT(std::function<T()>([=]() { return HipT::gmul(lm, lo, ln, x.get(), y.get()); }))};
             },
             [=](const Prod& xs, const Prod& ds) {
               LT x = xs[0], y = xs[1], d = ds[0];
               // dx = gmul lM lN lO dtdz (transp y)                       (TOp.hs:81)
               LT dx(std::function<T()>([=]() { return HipT::gmul(lm, ln, lo, d.get(), HipT::transp(y.get())); }));
               // dy = gmul (rev lO) (rev lM) lN (transp x) dtdz           (TOp.hs:86-88)
               LT dy(std::function<T()>([=]() { return HipT::gmul(lo, lm, ln, HipT::transp(x.get()), d.get()); }));
               return Prod{dx, dy};
             }};
}

// transpOp (TOp.hs:97-104)
inline TOp transpOp() {
  auto t = [](const LT& x) { return LT(std::function<T()>([x]() { return HipT::transp(x.get()); })); };
  return TOp{1, 1, [t](const Prod& xs) { return Prod{t(xs[0])}; },
             [t](const Prod&, const Prod& ds) { return Prod{t(ds[0])}; }};
}

inline LT lazy_sum(std::vector<LT> parts, LT like) {
  return LT(std::function<T()>([parts, like]() {
    std::vector<T> v;
    for (const LT& p : parts) v.push_back(p.get());
    return HipT::sumT(v, like.get().dims());
  }));
}

// shuffle (TOp.hs:106-131)
inline TOp shuffle(const std::vector<int>& idx, int n_in) {
  return TOp{n_in, (int)idx.size(),
             [idx](const Prod& xs) {
               Prod out;
               for (int i : idx) out.push_back(xs[i]);
               return out;
             },
             [idx, n_in](const Prod& xs, const Prod& ds) {
               Prod out;
               for (int i = 0; i < n_in; ++i) {
                 std::vector<LT> parts;
                 for (size_t k = 0; k < idx.size(); ++k)
                   if (idx[k] == i) parts.push_back(ds[k]);
                 out.push_back(lazy_sum(parts, xs[i]));
               }
               return out;
             }};
}

// sumRows (TOp.hs:151-159): backward = mapRows (LS LZ) (\_ -> dtdz) x
inline TOp sumRows() {
  return TOp{1, 1,
             [](const Prod& xs) {
               LT x = xs[0];
               return Prod{LT(std::function<T()>([x]() { return HipT::sumRows(x.get()); }))};
             },
             [](const Prod& xs, const Prod& ds) {
               LT x = xs[0], d = ds[0];
               // the general class method with a closure that ignores its row (TOp.hs:158): the row views are never
               // forced, and every element of the stacked result is the same handle
               return Prod{LT(std::function<T()>([x, d]() {
                 return HipT::mapRows(1, [d](const LT&) { return d.get(); }, x.get());
               }))};
             }};
}

// sumOp (TOp.hs:161-169)
inline TOp sumOp(int n, const Dims& dims) {
  return TOp{n, 1,
             [dims](const Prod& xs) {
               return Prod{LT(std::function<T()>([xs, dims]() {
                 std::vector<T> v;
                 for (const LT& x : xs) v.push_back(x.get());
                 return HipT::sumT(v, dims);
               }))};
             },
             [n](const Prod& xs, const Prod& ds) {
               Prod out;
               for (int i = 0; i < n; ++i) {
                 out.push_back(ds[0]);
               }
               return out;
             }};
}

// scale (TOp.hs:171-177), negate (:194-196)
inline TOp scale(double alpha) {
  auto s = [alpha](const LT& x) {
    return LT(std::function<T()>([alpha, x]() { return HipT::scaleT(alpha, x.get()); }));
  };
  return TOp{1, 1, [s](const Prod& xs) { return Prod{s(xs[0])}; },
             [s](const Prod&, const Prod& ds) { return Prod{s(ds[0])}; }};
}
inline TOp negate() { return scale(-1.0); }

// konst (TOp.hs:185-192)
inline TOp konst(int n, const Dims& dims, double x) {
  return TOp{0, n,
             [=](const Prod&) {
               Prod out;
               for (int i = 0; i < n; ++i) out.emplace_back(HipT::konst(dims, x));
               return out;
             },
             [](const Prod&, const Prod&) { return Prod{}; }};
}

// map' (TOp.hs:198-206), map = map' f (diff f) (:209-213)
template <class F, class DF>
TOp map_with(F f, DF fprime) {
  VFunc vf;
  vf.n = 1;
  vf.f = [f](const std::vector<Expr>& x) { return f(x[0]); };
  vf.g = [fprime](const std::vector<Expr>& x) { return std::vector<Expr>{fprime(x[0])}; };
  return liftOp(vf);
}
template <class F>
TOp map(F f) {
  return map_with(f, [f](const Expr& x) { return diff_at(f, x); });
}

// add (TOp.hs:215-221), add3 (:223-229)
inline TOp addN(int n) {
  return TOp{n, 1,
             [](const Prod& xs) {
               return Prod{LT(std::function<T()>([xs]() {
                 std::vector<T> v;
                 for (const LT& x : xs) v.push_back(x.get());
                 return HipT::sumT(v, v[0].dims());
               }))};
             },
             [n](const Prod& xs, const Prod& ds) {
               Prod out;
               for (int i = 0; i < n; ++i) {
                 out.push_back(ds[0]);
               }
               return out;
             }};
}
inline TOp add() { return addN(2); }
inline TOp add3() { return addN(3); }

// zipN' (TOp.hs:232-239), zipN = zipN' u f (grad f) (:241-247); f takes std::vector<A>
template <class F>
TOp zipN(int n, F f) {
  VFunc vf;
  vf.n = n;
  vf.f = [f](const std::vector<Expr>& x) { return f(x); };
  vf.g = [f](const std::vector<Expr>& x) { return grad_at(f, x); };
  return liftOp(vf);
}

// zipN' with the gradient given explicitly (TOp.hs:232-239); g returns the n partial derivatives
template <class F, class G>
TOp zipN_with(int n, F f, G g) {
  VFunc vf;
  vf.n = n;
  vf.f = [f](const std::vector<Expr>& x) { return f(x); };
  vf.g = [g](const std::vector<Expr>& x) { return g(x); };
  return liftOp(vf);
}
// zip / zip' (TOp.hs:249-266), zip3 / zip3' (:268-285): the 2- and 3-ary special cases
template <class F>
TOp zip(F f) {
  return zipN(2, [f](const auto& v) { return f(v[0], v[1]); });
}
template <class F, class G>
TOp zip_with(F f, G g) {  // g x y = (df/dx, df/dy)
  return zipN_with(2, [f](const std::vector<Expr>& v) { return f(v[0], v[1]); },
                   [g](const std::vector<Expr>& v) {
                     auto d = g(v[0], v[1]);
                     return std::vector<Expr>{d.first, d.second};
                   });
}
template <class F>
TOp zip3(F f) {
  return zipN(3, [f](const auto& v) { return f(v[0], v[1], v[2]); });
}

// replicate (TOp.hs:287-293), duplicate (:295-302)
inline TOp replicate(int n) {
  return TOp{1, n,
             [n](const Prod& xs) { return Prod((size_t)n, xs[0]); },
             [](const Prod& xs, const Prod& ds) {
               return Prod{lazy_sum(std::vector<LT>(ds.begin(), ds.end()), xs[0])};
             }};
}
inline TOp duplicate() { return replicate(2); }

// inner/outer/dot/matVec/vecMat/matMat (TOp.hs:304-343)
inline TOp inner(int lm, int ln) { return gmul(lm, 1, ln); }
inline TOp outer(int lm, int ln) { return gmul(lm, 0, ln); }
inline TOp dot() { return inner(0, 0); }
inline TOp matVec() { return inner(1, 0); }
inline TOp vecMat() { return inner(0, 1); }
inline TOp matMat() { return inner(1, 1); }

// swap (TOp.hs:346-351)
inline TOp swap() {
  return TOp{2, 2, [](const Prod& xs) { return Prod{xs[1], xs[0]}; },
             [](const Prod&, const Prod& ds) { return Prod{ds[1], ds[0]}; }};
}

// swap' (TOp.hs:353-360) = shuffleF swapProd swapProd: ns ++ ms -> ms ++ ns, a re-ordering both ways
inline TOp swap_n(int n_front, int n_back) {
  return TOp{n_front + n_back, n_front + n_back,
             [n_front](const Prod& xs) { return concat(slice(xs, n_front, xs.size()), slice(xs, 0, n_front)); },
             [n_back](const Prod&, const Prod& ds) {
               return concat(slice(ds, n_back, ds.size()), slice(ds, 0, n_back));
             }};
}

// drop / take (TOp.hs:362-381): the discarded inputs get sumT [] = zeros
inline TOp drop(int n_drop, int n) {
  return TOp{n, n - n_drop,
             [n_drop](const Prod& xs) { return slice(xs, n_drop, xs.size()); },
             [n_drop](const Prod& xs, const Prod& ds) {
               Prod out;
               for (int i = 0; i < n_drop; ++i) out.push_back(lazy_sum({}, xs[i]));
               return concat(out, ds);
             }};
}
inline TOp take(int n_take, int n) {
  return TOp{n, n_take,
             [n_take](const Prod& xs) { return slice(xs, 0, n_take); },
             [n_take, n](const Prod& xs, const Prod& ds) {
               Prod out = ds;
               for (int i = n_take; i < n; ++i) out.push_back(lazy_sum({}, xs[i]));
               return out;
             }};
}

}  // namespace tensorops
