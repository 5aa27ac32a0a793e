// src/TensorOps/Learn/NeuralNet/Recurrent.hs over HipT: stateful networks, `~*~`, the BPTT
// unrolling (`unroll` / `rollup`, :392-463) and `trainNetwork'` (:326-356).
//
// The unrolled graph uses every parameter n times; the accumulation of their cotangents is
// `&&&`'s `sumT [gxy, gxz]` (Types.hs:259) at every time step -- on this backend an n-ary fused
// add on the device instead of the reference's GEMM-with-`eye` matrix addition
// (BTensor.hs:110-113).  Order conventions are the reference's: inputs are fed REVERSED in
// time and outputs come back reversed (:289-293).
//
// Batch rule: the x_t / y_t may carry a hidden batch of B independent sequences; the initial
// state and the parameters are unbatched, so their cotangents are the batch sums.
#pragma once
#include "learn.hpp"

namespace tensorops {
namespace recurrent {

struct Network {  // `Network t i o` (Recurrent.hs:66-72): op : ('[i] ': ss ++ ps) -> ('[o] ': ss)
  TOp op;
  std::vector<T> state;   // Prod t ss
  std::vector<T> params;  // Prod t ps
  int n_s() const { return (int)state.size(); }
  int n_p() const { return (int)params.size(); }
};

// `fc` (:108-118) on [x, s, W', W, b] -> [z, act z],  z = W x + W' s + b
// (the output is the pre-activation sum, the new state its activation)
inline TOp fullyConnectedOp(const Activation& act) {
  TOp inner = firstOp(swap() >> matVec(), 2) >> firstOp(swap(), 1);
  return secondOp(1, inner) >> firstOp(swap() >> matVec(), 2) >> add3() >> duplicate() >> secondOp(1, act());
}
// fullyConnected (:91-119) with given values; params = (W' :< W :< b), state = (s)
inline Network fullyConnected(const Activation& act, const T& s, const T& w_state, const T& w, const T& b) {
  return Network{fullyConnectedOp(act), {s}, {w_state, w, b}};
}
// ... drawing everything from normalDistr 0 0.5 on the device, like the reference
inline Network fullyConnectedRand(const Activation& act, int64_t i, int64_t o, uint64_t seed) {
  return fullyConnected(act, HipT::genRand({o}, 1, 0.0, 0.5, seed), HipT::genRand({o, o}, 1, 0.0, 0.5, seed + 1),
                        HipT::genRand({o, i}, 1, 0.0, 0.5, seed + 2), HipT::genRand({o}, 1, 0.0, 0.5, seed + 3));
}
// stateless (:126-131)
inline Network stateless(const tensorops::Network& ff) { return Network{ff.op, {}, ff.params}; }
inline Network ffLayer(const T& w, const T& b) { return stateless(tensorops::ffLayer(w, b)); }  // :133-138

// ~*~ (:170-222): states ss2 ++ ss1, params ps1 ++ ps2
inline Network seq(const Network& n1, const Network& n2) {
  const int s1 = n1.n_s(), s2 = n2.n_s(), p1 = n1.n_p(), p2 = n2.n_p();
  TOp o = secondOp(1, firstOp(swap_n(s2, s1 + p1), p2)) >> firstOp(n1.op, s2 + p2) >>
          secondOp(1, swap_n(s1, s2 + p2)) >> firstOp(n2.op, s1);
  Network n{o, n2.state, n1.params};
  n.state.insert(n.state.end(), n1.state.begin(), n1.state.end());
  n.params.insert(n.params.end(), n2.params.begin(), n2.params.end());
  return n;
}
// *~ (:247-252) and ~* (:240-245)
inline Network then(const Network& n, const TOp& f) { return Network{n.op >> firstOp(f, n.n_s()), n.state, n.params}; }
inline Network after(const TOp& f, const Network& n) { return Network{then_first(f, n.op), n.state, n.params}; }

// runNetwork (:224-232): (y, network carrying the new state)
inline std::pair<T, Network> runNetwork(const Network& n, const T& x) {
  Prod in{LT(x)};
  for (const T& s : n.state) in.emplace_back(s);
  for (const T& p : n.params) in.emplace_back(p);
  Prod out = runTOp(n.op, in);
  Network next{n.op, {}, n.params};
  for (size_t i = 1; i < out.size(); ++i) next.state.push_back(out[i].get());
  return {out[0].get(), next};
}

// unroll (:392-431): Replicate n '[i] ++ ss ++ ps -> ss ++ Replicate n '[o]; the LAST input is
// consumed first and its output lands last
inline TOp unroll(const TOp& o, int ls, int lp, int n) {
  if (n == 0) return take(ls, ls + lp);
  const int m = n - 1;
  TOp step = fanout(o, drop(1 + ls, 1 + ls + lp)) >> swap_n(1, ls + lp);
  return secondOp(m, step) >> firstOp(unroll(o, ls, lp, m), 1);
}
// rollup (:434-463): Replicate n '[o] ++ Replicate n '[o] -> '[ '[] ]
inline TOp rollup(const TOp& loss, int n) {
  if (n == 0) return konst(1, {}, 0.0);
  if (n == 1) return loss;
  const int m = n - 1;
  return secondOp(m, firstOp(loss, m) >> swap_n(1, m)) >> firstOp(rollup(loss, m), 1) >> add();
}

struct Grads {
  Prod inputs;  // in the order of the REVERSED inputs, like the reference (:283)
  Prod state, params;
};
// netGrad (:265-324)
inline Grads netGrad(const TOp& loss, const std::vector<T>& xs, const std::vector<T>& ys, const Network& net) {
  const int n = (int)xs.size(), ls = net.n_s(), lp = net.n_p();
  if (ys.size() != xs.size()) throw TensorOpsError(TO_ERR_ARG, "netGrad: inputs and targets differ in length");
  TOp unrolled = unroll(net.op, ls, lp, n) >> drop(ls, ls + n);
  TOp o = firstOp(unrolled, n) >> rollup(loss, n);
  Prod in;
  for (int t = n - 1; t >= 0; --t) in.emplace_back(xs[(size_t)t]);
  for (const T& s : net.state) in.emplace_back(s);
  for (const T& p : net.params) in.emplace_back(p);
  for (const T& y : ys) in.emplace_back(y);
  Prod g = gradTOp(o, in);
  return Grads{slice(g, 0, n), slice(g, n, n + ls), slice(g, n + ls, n + ls + lp)};
}
// netGrad on B independent sequences: the cotangents of the (unbatched) initial state and parameters summed over them
inline Grads netGradBatch(const TOp& loss, const std::vector<T>& xs, const std::vector<T>& ys, const Network& net) {
  Grads g = netGrad(loss, xs, ys, net);
  Prod s, p;
  for (const T& v : net.state) s.emplace_back(v);
  for (const T& v : net.params) p.emplace_back(v);
  return Grads{g.inputs, sumOverBatch(g.state, s), sumOverBatch(g.params, p)};
}
// trainNetwork' (:326-356): separate rates for the initial state and the parameters
inline Network trainNetwork(const TOp& loss, double r_s, double r_p, const std::vector<T>& xs,
                            const std::vector<T>& ys, const Network& net) {
  Grads g = netGradBatch(loss, xs, ys, net);  // (= netGrad when nothing is batched)
  auto step = [](double r, const T& p, const T& gr) {
    return HipT::liftT([r](const std::vector<Expr>& v) { return v[0] - Expr(r) * v[1]; }, {p, gr});
  };
  Network out{net.op, {}, {}};
  for (size_t i = 0; i < net.state.size(); ++i) out.state.push_back(step(r_s, net.state[i], g.state[i].get()));
  for (size_t i = 0; i < net.params.size(); ++i) out.params.push_back(step(r_p, net.params[i], g.params[i].get()));
  return out;
}

}  // namespace recurrent

// ---- src/TensorOps/Learn/NeuralNet/AutoEncoder.hs ----------------------------------------------------
namespace autoencoder {

struct Encoder {  // `Encoder t i o` (:37-40)
  tensorops::Network enc, dec;
};
inline T encode(const Encoder& e, const T& x) { return runNetwork(e.enc, x); }                 // :42-48
inline T decode(const Encoder& e, const T& y) { return runNetwork(e.dec, y); }                 // :50-56
inline tensorops::Network encoderNet(const Encoder& e) { return seq(e.enc, e.dec); }          // :83-87
inline T encodeDecode(const Encoder& e, const T& x) { return runNetwork(encoderNet(e), x); }  // :58-63

// the objective of :130-137 / :73-79: duplicate x, encode+decode one copy, swap, loss
inline TOp objective(const TOp& loss, const Encoder& e) {
  const int pe = (int)e.enc.params.size(), pd = (int)e.dec.params.size();
  return firstOp(duplicate(), pe + pd) >> secondOp(1, firstOp(e.enc.op, pd) >> e.dec.op) >> swap() >> loss;
}
inline Prod inputs(const Encoder& e, const T& x) {
  Prod in{LT(x)};
  for (const T& p : e.enc.params) in.emplace_back(p);
  for (const T& p : e.dec.params) in.emplace_back(p);
  return in;
}
inline T testEncoder(const TOp& loss, const Encoder& e, const T& x) {  // :65-81
  return runTOp(objective(loss, e), inputs(e, x))[0].get();
}
// encGrad (:112-142): tail' of gradTOp -- x's cotangent is never forced
inline Prod encGrad(const TOp& loss, const T& x, const Encoder& e) {
  Prod g = gradTOp(objective(loss, e), inputs(e, x));
  return slice(g, 1, g.size());
}
inline Prod encGradBatch(const TOp& loss, const T& x, const Encoder& e) {  // parameters' cotangents summed over the batch
  Prod in = inputs(e, x);
  return sumOverBatch(encGrad(loss, x, e), slice(in, 1, in.size()));
}
inline Encoder trainEncoder(const TOp& loss, double r, const T& x, const Encoder& e) {  // :89-110
  Prod g = encGradBatch(loss, x, e);
  auto step = [r](const T& p, const T& gr) {
    return HipT::liftT([r](const std::vector<Expr>& v) { return v[0] - Expr(r) * v[1]; }, {p, gr});
  };
  Encoder out{tensorops::Network{e.enc.op, {}},
              tensorops::Network{e.dec.op, {}}};
  size_t k = 0;
  for (const T& p : e.enc.params) out.enc.params.push_back(step(p, g[k++].get()));
  for (const T& p : e.dec.params) out.dec.params.push_back(step(p, g[k++].get()));
  return out;
}

}  // namespace autoencoder
}  // namespace tensorops
