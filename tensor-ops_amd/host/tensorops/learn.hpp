// TensorOps.Learn.NeuralNet (src/TensorOps/Learn/NeuralNet.hs:15-77) and
// TensorOps.Learn.NeuralNet.FeedForward (…/FeedForward.hs:57-235) over HipT.
#pragma once
#include "top.hpp"

namespace tensorops {

// ---- NeuralNet.hs -------------------------------------------------------------------------------
struct Logistic {  // logistic x = 1 / (1 + exp (-x))   (NeuralNet.hs:42-44)
  template <class A>
  A operator()(const A& x) const { return A(1.0) / (A(1.0) + exp(-x)); }
};
struct LogisticPrime {  // logistic' x = logix * (1 - logix)   (NeuralNet.hs:46-50)
  template <class A>
  A operator()(const A& x) const {
    A l = Logistic()(x);
    return l * (A(1.0) - l);
  }
};
struct ExpF { template <class A> A operator()(const A& x) const { return exp(x); } };
struct LogF { template <class A> A operator()(const A& x) const { return log(x); } };
struct RecipF { template <class A> A operator()(const A& x) const { return A(1.0) / x; } };
struct TanhF { template <class A> A operator()(const A& x) const { return tanh(x); } };

using Activation = std::function<TOp()>;  // `Activation k` (NeuralNet.hs:15-19)

template <class F> Activation actMap(F f) { return [f]() { return map(f); }; }                          // :21-25
template <class F, class DF> Activation actMapWith(F f, DF df) { return [f, df]() { return map_with(f, df); }; }  // :27-32
inline Activation actLogistic() { return actMapWith(Logistic(), LogisticPrime()); }                      // :38-40

// softmax = map exp >>> duplicate >>> firstOp (sumRows >>> map recip) >>> outer LZ (LS LZ)   (:52-59)
inline TOp softmax() {
  // `>>>` is infixr 1: a >>> (b >>> (c >>> d))
  return map(ExpF()) >> (duplicate() >> (firstOp(sumRows() >> map(RecipF()), 1) >> outer(0, 1)));
}
inline Activation actSoftmax() { return []() { return softmax(); }; }                                    // :34-36

// squaredError = negate *>> add >>> duplicate >>> dot   (:61-68)
// (`*>>` infixr 0 binds looser than `>>>` infixr 1)
inline TOp squaredError() { return then_first(negate(), add() >> (duplicate() >> dot())); }
// crossEntropy = map log *>> dot >>> negate   (:71-77); second input is the target
inline TOp crossEntropy() { return then_first(map(LogF()), dot() >> negate()); }

// ids of the activations / losses the C entry points and `genNet` bookkeeping use (same values
// as TOH_ACT_* / TOH_LOSS_* in tensorops_host.h)
enum { ACT_LOGISTIC = 0, ACT_MAP_LOGISTIC = 1, ACT_SOFTMAX = 2, ACT_MAP_TANH = 3 };
enum { LOSS_SQUARED_ERROR = 0, LOSS_CROSS_ENTROPY = 1 };
inline Activation act_of(int id) {
  switch (id) {
    case ACT_LOGISTIC: return actLogistic();
    case ACT_MAP_LOGISTIC: return actMap(Logistic());
    case ACT_SOFTMAX: return actSoftmax();
    case ACT_MAP_TANH: return actMap(TanhF());
    default: throw TensorOpsError(TO_ERR_ARG, "unknown activation id");
  }
}
inline TOp loss_of(int id) {
  switch (id) {
    case LOSS_SQUARED_ERROR: return squaredError();
    case LOSS_CROSS_ENTROPY: return crossEntropy();
    default: throw TensorOpsError(TO_ERR_ARG, "unknown loss id");
  }
}

// ---- FeedForward.hs -------------------------------------------------------------------------------
// `Network t i o` (FeedForward.hs:57-61): an op and its parameters, nothing else.  The reference carries no
// record of which activations built `op`, so neither does this: what the backend can fuse it has to find in
// the class-method stream itself (csrc/lazy.cpp).
struct Network {
  TOp op;                 // ('[i] ': ps) -> '[ '[o] ]
  std::vector<T> params;  // Prod t ps
};

inline Network seq(const Network& a, const Network& b) {  // ~*~ (:82-90)
  Network n{then_first(a.op, b.op), a.params};
  n.params.insert(n.params.end(), b.params.begin(), b.params.end());
  return n;
}
inline Network then(const Network& n, const TOp& f) { return Network{n.op >> f, n.params}; }  // *~ (:103-108)

// f ~* n = N (f *>> o) p (:96-101);  liftNet o = buildNet o Ø (:110-113);  nmap f n = n *~ TO.map f (:115-121)
inline Network after(const TOp& f, const Network& n) { return Network{then_first(f, n.op), n.params}; }
inline Network buildNet(const TOp& o, const std::vector<T>& params) { return Network{o, params}; }
inline Network liftNet(const TOp& o) { return Network{o, {}}; }
template <class F>
Network nmap(F f, const Network& n) { return then(n, map(f)); }

// ffLayer' = firstOp (swap >>> matVec) >>> add   on [x, W, b]   (:209-213)
inline TOp ffLayerOp() { return firstOp(swap() >> matVec(), 1) >> add(); }
inline Network ffLayer(const T& w, const T& b) { return Network{ffLayerOp(), {w, b}}; }  // weights are inputs
// ffLayer with the reference's initial distribution: W, b ~ normalDistr 0 0.5 (:205-207)
inline Network ffLayerRand(int64_t i, int64_t o, uint64_t seed) {
  return ffLayer(HipT::genRand({o, i}, 1, 0.0, 0.5, seed), HipT::genRand({o}, 1, 0.0, 0.5, seed + 1));
}

// genNet (:216-235): go [] = ffLayer *~ f ; go ((x,f'):xs) = (ffLayer *~ f') ~*~ go xs
inline Network genNet(const std::vector<std::pair<T, T>>& weights, const Activation& hidden,
                      const Activation& out, size_t from = 0) {
  const auto& wb = weights[from];
  if (from + 1 == weights.size()) return then(ffLayer(wb.first, wb.second), out());
  return seq(then(ffLayer(wb.first, wb.second), hidden()), genNet(weights, hidden, out, from + 1));
}

inline T runNetwork(const Network& n, const T& x) {  // (:123-129)
  Prod in{LT(x)};
  for (const T& p : n.params) in.emplace_back(p);
  return runTOp(n.op, in)[0].get();
}

// netGrad (:178-199): gradTOp (o *>> loss) (x :< p >: y), keep x's and the params' cotangents
inline Prod netGrad(const TOp& loss, const T& x, const T& y, const Network& n) {
  TOp o = then_first(n.op, loss);
  Prod in{LT(x)};
  for (const T& p : n.params) in.emplace_back(p);
  in.emplace_back(y);
  Prod g = gradTOp(o, in);
  return slice(g, 0, 1 + n.params.size());
}

// networkGradient (:166-176): the parameters' cotangents only (`tail'`)
inline Prod networkGradient(const TOp& loss, const T& x, const T& y, const Network& n) {
  Prod g = netGrad(loss, x, y, n);
  return slice(g, 1, g.size());
}

// trainNetwork (:131-148): p' = zip (\o g -> o - r*g) p (tail' grads); x's cotangent is never forced
inline Network trainNetwork(const TOp& loss, double r, const T& x, const T& y, const Network& n) {
  Prod g = netGrad(loss, x, y, n);
  Network out{n.op, {}};
  for (size_t i = 0; i < n.params.size(); ++i)
    out.params.push_back(HipT::liftT(
        [r](const std::vector<Expr>& v) { return v[0] - Expr(r) * v[1]; }, {n.params[i], g[i + 1].get()}));
  return out;
}

// ---- batched data (the shim's `trainBatch`, hs/TensorOps/Backend/HipTensor.hs) ------------------------------------
// The reference's functions above are per-sample.  On a batch they are called UNCHANGED; the host then sums the
// cotangents of the (unbatched) parameters over the samples -- the one place where the batching extension shows.
inline Prod netGradBatch(const TOp& loss, const T& x, const T& y, const Network& n) {
  Prod in{LT(x)};
  for (const T& p : n.params) in.emplace_back(p);
  return sumOverBatch(netGrad(loss, x, y, n), in);
}
// trainNetwork's body (:131-148) with `batchSum` on the gradient; the new parameters are forced together inside the
// caller's scope (`forceAll ps'`), so that the step is one plan
inline Network trainBatch(const TOp& loss, double r, const T& x, const T& y, const Network& n) {
  Prod g = netGradBatch(loss, x, y, n);
  Network out{n.op, {}};
  for (size_t i = 0; i < n.params.size(); ++i)
    out.params.push_back(HipT::liftT(
        [r](const std::vector<Expr>& v) { return v[0] - Expr(r) * v[1]; }, {n.params[i], g[i + 1].get()}));
  return out;
}

// induceNetwork (:150-164): gradient step on the INPUT
inline T induceNetwork(const TOp& loss, double r, const T& y, const Network& n, const T& x) {
  Prod g = netGrad(loss, x, y, n);
  return HipT::liftT([r](const std::vector<Expr>& v) { return v[0] - Expr(r) * v[1]; }, {x, g[0].get()});
}

}  // namespace tensorops
