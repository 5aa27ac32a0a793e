// C entry points of the C++ host mirror (tensorops_host.h).
#include "tensorops_host.h"

#include <cstring>
#include <memory>

#include "tensorops/recurrent.hpp"
#include "tensorops/trainer.hpp"

using namespace tensorops;

struct toh_op_s {
  TOp op;
};
struct toh_net_s {
  Network net;
};
struct toh_rnn_s {
  recurrent::Network net;
};
struct toh_trainer_s {
  std::unique_ptr<Trainer> t;
};
static_assert((int)TOH_ACT_LOGISTIC == (int)ACT_LOGISTIC && (int)TOH_ACT_MAP_LOGISTIC == (int)ACT_MAP_LOGISTIC &&
                  (int)TOH_ACT_SOFTMAX == (int)ACT_SOFTMAX && (int)TOH_ACT_MAP_TANH == (int)ACT_MAP_TANH &&
                  (int)TOH_LOSS_SQUARED_ERROR == (int)LOSS_SQUARED_ERROR &&
                  (int)TOH_LOSS_CROSS_ENTROPY == (int)LOSS_CROSS_ENTROPY && (int)TOH_TRAINER_MEMO == (int)TRAINER_MEMO &&
                  (int)TOH_TRAINER_GRAPH == (int)TRAINER_GRAPH && (int)TOH_TRAINER_FUSED == (int)TRAINER_FUSED &&
                  (int)TOH_TRAINER_FRESH_THUNKS == (int)TRAINER_FRESH_THUNKS,
              "C ids mirror the C++ ones");

static thread_local std::string g_herr;

#define H_BEGIN try {
#define H_END                               \
  return TO_OK;                             \
  }                                         \
  catch (const TensorOpsError& e) {         \
    g_herr = e.what();                      \
    return e.code;                          \
  }                                         \
  catch (const std::exception& e) {         \
    g_herr = e.what();                      \
    return TO_ERR_ARG;                      \
  }
#define H_NONNULL(p) \
  if (!(p)) throw TensorOpsError(TO_ERR_ARG, "null argument: " #p)

static SsaFn make_ssa(int arity, int n_instr, const int32_t* code, int n_consts, const double* consts) {
  SsaFn f;
  f.arity = arity;
  f.code.assign(code, code + 3 * (size_t)n_instr);
  f.consts.assign(consts, consts + (n_consts > 0 ? n_consts : 0));
  for (int i = 0; i < n_instr; ++i) {
    const int op = code[3 * i], a = code[3 * i + 1], b = code[3 * i + 2];
    bool ok = op >= 0 && op < TO_X_NOPS;
    if (ok && op == TO_X_CONST) ok = a >= 0 && a < n_consts;
    else if (ok) ok = a >= 0 && a < arity + i && b >= 0 && b < arity + i;
    if (!ok) throw TensorOpsError(TO_ERR_ARG, "malformed SSA program");
  }
  if (arity + n_instr < 1) throw TensorOpsError(TO_ERR_ARG, "empty SSA program");
  return f;
}

static Prod to_prod(int n, const to_tensor* hs) {
  Prod p;
  for (int i = 0; i < n; ++i) {
    H_NONNULL(hs[i]);
    check(to_retain(hs[i]));
    p.emplace_back(T(hs[i]));
  }
  return p;
}

static T borrow(to_tensor h) {
  check(to_retain(h));
  return T(h);
}

extern "C" {

const char* toh_last_error(void) { return g_herr.c_str(); }

to_status toh_op_named(const char* name, int iarg, double darg, toh_op* out) {
  H_BEGIN
  H_NONNULL(name);
  H_NONNULL(out);
  const std::string s(name);
  TOp o;
  if (s == "idOp") o = idOp(iarg);
  else if (s == "add") o = add();
  else if (s == "add3") o = add3();
  else if (s == "addN") o = addN(iarg);
  else if (s == "duplicate") o = duplicate();
  else if (s == "replicate") o = replicate(iarg);
  else if (s == "swap") o = swap();
  else if (s == "negate") o = negate();
  else if (s == "scale") o = scale(darg);
  else if (s == "sumRows") o = sumRows();
  else if (s == "transpOp") o = transpOp();
  else if (s == "dot") o = dot();
  else if (s == "matVec") o = matVec();
  else if (s == "vecMat") o = vecMat();
  else if (s == "matMat") o = matMat();
  else if (s == "softmax") o = softmax();
  else if (s == "squaredError") o = squaredError();
  else if (s == "crossEntropy") o = crossEntropy();
  else if (s == "actLogistic") o = actLogistic()();
  else if (s == "mapLogistic") o = map(Logistic());
  else if (s == "mapExp") o = map(ExpF());
  else if (s == "mapLog") o = map(LogF());
  else if (s == "mapRecip") o = map(RecipF());
  else if (s == "mapTanh") o = map(TanhF());
  else if (s == "ffLayer") o = ffLayerOp();
  else throw TensorOpsError(TO_ERR_ARG, "unknown op name: " + s);
  *out = new toh_op_s{o};
  H_END
}

to_status toh_op_gmul(int lm, int lo, int ln, toh_op* out) {
  H_BEGIN
  H_NONNULL(out);
  *out = new toh_op_s{gmul(lm, lo, ln)};
  H_END
}

to_status toh_op_map(int n_instr, const int32_t* code, int n_consts, const double* consts, toh_op* out) {
  H_BEGIN
  H_NONNULL(out);
  SsaFn f = make_ssa(1, n_instr, code, n_consts, consts);
  auto unary = [f](auto x) { return f(std::vector<decltype(x)>{x}); };
  *out = new toh_op_s{map(unary)};
  H_END
}

to_status toh_op_map_with(int n_f, const int32_t* f, int nc_f, const double* c_f, int n_df,
                          const int32_t* df, int nc_df, const double* c_df, toh_op* out) {
  H_BEGIN
  H_NONNULL(out);
  SsaFn ff = make_ssa(1, n_f, f, nc_f, c_f), dd = make_ssa(1, n_df, df, nc_df, c_df);
  *out = new toh_op_s{map_with([ff](const Expr& x) { return ff(std::vector<Expr>{x}); },
                               [dd](const Expr& x) { return dd(std::vector<Expr>{x}); })};
  H_END
}

to_status toh_op_zipN(int n, int n_instr, const int32_t* code, int n_consts, const double* consts,
                      toh_op* out) {
  H_BEGIN
  H_NONNULL(out);
  SsaFn f = make_ssa(n, n_instr, code, n_consts, consts);
  *out = new toh_op_s{zipN(n, f)};
  H_END
}

to_status toh_op_zipN_with(int n, int n_f, const int32_t* f, int nc_f, const double* c_f, const int32_t* n_g,
                           const int32_t* const* g, const int32_t* nc_g, const double* const* c_g, toh_op* out) {
  H_BEGIN
  H_NONNULL(out); H_NONNULL(n_g); H_NONNULL(g); H_NONNULL(nc_g); H_NONNULL(c_g);
  SsaFn ff = make_ssa(n, n_f, f, nc_f, c_f);
  std::vector<SsaFn> gs;
  for (int i = 0; i < n; ++i) gs.push_back(make_ssa(n, n_g[i], g[i], nc_g[i], c_g[i]));
  *out = new toh_op_s{zipN_with(n, ff, [gs](const std::vector<Expr>& x) {
    std::vector<Expr> d;
    for (const SsaFn& gi : gs) d.push_back(gi(x));
    return d;
  })};
  H_END
}

to_status toh_op_sumOp(int n, int rank, const int64_t* dims, toh_op* out) {
  H_BEGIN
  H_NONNULL(out);
  *out = new toh_op_s{sumOp(n, Dims(dims, dims + rank))};
  H_END
}

to_status toh_op_konst(int n, int rank, const int64_t* dims, double x, toh_op* out) {
  H_BEGIN
  H_NONNULL(out);
  *out = new toh_op_s{konst(n, Dims(dims, dims + rank), x)};
  H_END
}

to_status toh_op_shuffle(int n_in, int n_idx, const int32_t* idx, toh_op* out) {
  H_BEGIN
  H_NONNULL(out);
  std::vector<int> v(idx, idx + n_idx);
  for (int i : v)
    if (i < 0 || i >= n_in) throw TensorOpsError(TO_ERR_ARG, "shuffle: index out of range");
  *out = new toh_op_s{shuffle(v, n_in)};
  H_END
}

to_status toh_op_drop(int n_drop, int n, toh_op* out) {
  H_BEGIN
  H_NONNULL(out);
  *out = new toh_op_s{drop(n_drop, n)};
  H_END
}

to_status toh_op_take(int n_take, int n, toh_op* out) {
  H_BEGIN
  H_NONNULL(out);
  *out = new toh_op_s{take(n_take, n)};
  H_END
}

to_status toh_op_compose(toh_op a, toh_op b, toh_op* out) {
  H_BEGIN
  H_NONNULL(a); H_NONNULL(b); H_NONNULL(out);
  *out = new toh_op_s{compose(a->op, b->op)};
  H_END
}

to_status toh_op_first(toh_op o, int n_pass, toh_op* out) {
  H_BEGIN
  H_NONNULL(o); H_NONNULL(out);
  *out = new toh_op_s{firstOp(o->op, n_pass)};
  H_END
}

to_status toh_op_second(int n_skip, toh_op o, toh_op* out) {
  H_BEGIN
  H_NONNULL(o); H_NONNULL(out);
  *out = new toh_op_s{secondOp(n_skip, o->op)};
  H_END
}

to_status toh_op_then_first(toh_op a, toh_op b, toh_op* out) {
  H_BEGIN
  H_NONNULL(a); H_NONNULL(b); H_NONNULL(out);
  *out = new toh_op_s{then_first(a->op, b->op)};
  H_END
}

to_status toh_op_par(toh_op a, toh_op b, toh_op* out) {
  H_BEGIN
  H_NONNULL(a); H_NONNULL(b); H_NONNULL(out);
  *out = new toh_op_s{par(a->op, b->op)};
  H_END
}

to_status toh_op_fanout(toh_op a, toh_op b, toh_op* out) {
  H_BEGIN
  H_NONNULL(a); H_NONNULL(b); H_NONNULL(out);
  *out = new toh_op_s{fanout(a->op, b->op)};
  H_END
}

to_status toh_op_arity(toh_op o, int* n_in, int* n_out) {
  H_BEGIN
  H_NONNULL(o);
  if (n_in) *n_in = o->op.n_in;
  if (n_out) *n_out = o->op.n_out;
  H_END
}

to_status toh_op_release(toh_op o) {
  delete o;
  return TO_OK;
}

to_status toh_run(toh_op o, int n_in, const to_tensor* xs, to_tensor* ys) {
  H_BEGIN
  H_NONNULL(o); H_NONNULL(ys);
  Prod r = runTOp(o->op, to_prod(n_in, xs));
  std::vector<T> forced;
  for (const LT& y : r) forced.push_back(y.get());
  for (size_t i = 0; i < forced.size(); ++i) ys[i] = forced[i].release_handle();
  H_END
}

static void force_into(const Prod& g, const int32_t* want, to_tensor* dxs) {
  std::vector<T> forced(g.size());
  for (size_t i = 0; i < g.size(); ++i)
    if (!want || want[i]) forced[i] = g[i].get();
  for (size_t i = 0; i < g.size(); ++i) dxs[i] = forced[i] ? forced[i].release_handle() : nullptr;
}

to_status toh_grad(toh_op o, int n_in, const to_tensor* xs, const to_tensor* ds, const int32_t* want,
                   to_tensor* dxs) {
  H_BEGIN
  H_NONNULL(o); H_NONNULL(dxs);
  arity_check(n_in == o->op.n_in, "toh_grad");
  Prod in = to_prod(n_in, xs);
  Prod g = sumOverBatch(o->op.grad(in, to_prod(o->op.n_out, ds)), in);
  force_into(g, want, dxs);
  H_END
}

to_status toh_gradTOp(toh_op o, int n_in, const to_tensor* xs, const int32_t* want, to_tensor* dxs) {
  H_BEGIN
  H_NONNULL(o); H_NONNULL(dxs);
  Prod in = to_prod(n_in, xs);
  Prod g = sumOverBatch(gradTOp(o->op, in), in);
  force_into(g, want, dxs);
  H_END
}

// ---- Learn ---------------------------------------------------------------------------------------
to_status toh_genNet(int n_layers, const to_tensor* ws, const to_tensor* bs, int hidden_act,
                     int out_act, toh_net* out) {
  H_BEGIN
  H_NONNULL(ws); H_NONNULL(bs); H_NONNULL(out);
  if (n_layers < 1) throw TensorOpsError(TO_ERR_ARG, "genNet needs at least one layer");
  std::vector<std::pair<T, T>> w;
  for (int i = 0; i < n_layers; ++i) w.emplace_back(borrow(ws[i]), borrow(bs[i]));
  Network net = genNet(w, act_of(hidden_act), act_of(out_act));
  *out = new toh_net_s{net};
  H_END
}

to_status toh_genNet_rand(int n_sizes, const int64_t* sizes, int hidden_act, int out_act,
                          uint64_t seed, toh_net* out) {
  H_BEGIN
  H_NONNULL(sizes); H_NONNULL(out);
  if (n_sizes < 2) throw TensorOpsError(TO_ERR_ARG, "genNet needs input and output sizes");
  std::vector<std::pair<T, T>> w;
  for (int i = 0; i + 1 < n_sizes; ++i) {
    Network l = ffLayerRand(sizes[i], sizes[i + 1], seed + 2 * (uint64_t)i);
    w.emplace_back(l.params[0], l.params[1]);
  }
  Network net = genNet(w, act_of(hidden_act), act_of(out_act));
  *out = new toh_net_s{net};
  H_END
}

to_status toh_buildNet(toh_op o, int n_params, const to_tensor* params, toh_net* out) {
  H_BEGIN
  H_NONNULL(o); H_NONNULL(out);
  arity_check(o->op.n_in == 1 + n_params && o->op.n_out == 1, "buildNet");
  std::vector<T> ps;
  for (int i = 0; i < n_params; ++i) {
    H_NONNULL(params[i]);
    ps.push_back(borrow(params[i]));
  }
  *out = new toh_net_s{buildNet(o->op, ps)};
  H_END
}

to_status toh_net_seq(toh_net a, toh_net b, toh_net* out) {
  H_BEGIN
  H_NONNULL(a); H_NONNULL(b); H_NONNULL(out);
  *out = new toh_net_s{seq(a->net, b->net)};
  H_END
}

to_status toh_net_after_op(toh_op f, toh_net n, toh_net* out) {
  H_BEGIN
  H_NONNULL(f); H_NONNULL(n); H_NONNULL(out);
  arity_check(f->op.n_in == 1 && f->op.n_out == 1, "~*");
  *out = new toh_net_s{after(f->op, n->net)};
  H_END
}

to_status toh_net_then_op(toh_net n, toh_op f, toh_net* out) {
  H_BEGIN
  H_NONNULL(f); H_NONNULL(n); H_NONNULL(out);
  arity_check(f->op.n_in == 1 && f->op.n_out == 1, "*~");
  *out = new toh_net_s{then(n->net, f->op)};
  H_END
}

to_status toh_net_release(toh_net n) {
  delete n;
  return TO_OK;
}

to_status toh_net_n_params(toh_net n, int* out) {
  H_BEGIN
  H_NONNULL(n); H_NONNULL(out);
  *out = (int)n->net.params.size();
  H_END
}

to_status toh_net_params(toh_net n, to_tensor* out) {
  H_BEGIN
  H_NONNULL(n); H_NONNULL(out);
  for (size_t i = 0; i < n->net.params.size(); ++i) {
    check(to_retain(n->net.params[i].h()));
    out[i] = n->net.params[i].h();
  }
  H_END
}

to_status toh_runNetwork(toh_net n, to_tensor x, to_tensor* out) {
  H_BEGIN
  H_NONNULL(n); H_NONNULL(x); H_NONNULL(out);
  *out = runNetwork(n->net, borrow(x)).release_handle();
  H_END
}

to_status toh_netGrad(toh_net n, int loss, to_tensor x, to_tensor y, int want_x, to_tensor* grads) {
  H_BEGIN
  H_NONNULL(n); H_NONNULL(x); H_NONNULL(y); H_NONNULL(grads);
  Prod g = netGradBatch(loss_of(loss), borrow(x), borrow(y), n->net);
  std::vector<int32_t> want(g.size(), 1);
  want[0] = want_x ? 1 : 0;
  force_into(g, want.data(), grads);
  H_END
}

to_status toh_trainNetwork(toh_net n, int loss, double rate, to_tensor x, to_tensor y, toh_net* out) {
  H_BEGIN
  H_NONNULL(n); H_NONNULL(x); H_NONNULL(y); H_NONNULL(out);
  // per sample this IS trainNetwork (FeedForward.hs:131-148); on a batch the gradient is summed over the samples
  *out = new toh_net_s{trainBatch(loss_of(loss), rate, borrow(x), borrow(y), n->net)};
  H_END
}

// ---- batched, replayed step (tensorops/trainer.hpp) -------------------------------------------------------
to_status toh_trainer_flat_size(toh_net n, int64_t* n_floats) {
  H_BEGIN
  H_NONNULL(n); H_NONNULL(n_floats);
  *n_floats = Trainer::flat_size(n->net);
  H_END
}

to_status toh_trainer_create(toh_net n, int loss, double rate, to_tensor x_batched,
                             to_tensor y_batched, int use_memo, int use_graph, toh_trainer* out) {
  return toh_trainer_create_ext(n, loss, rate, x_batched, y_batched, use_memo, use_graph, nullptr,
                                nullptr, out);
}

to_status toh_trainer_create_ext(toh_net n, int loss, double rate, to_tensor x_batched,
                                 to_tensor y_batched, int use_memo, int use_graph, void* ext_params,
                                 void* ext_grads, toh_trainer* out) {
  return toh_trainer_create_opts(n, loss, rate, x_batched, y_batched,
                                 (use_memo ? TOH_TRAINER_MEMO : 0) | (use_graph ? TOH_TRAINER_GRAPH : 0) |
                                     TOH_TRAINER_FUSED,
                                 ext_params, ext_grads, out);
}

to_status toh_trainer_create_opts(toh_net n, int loss, double rate, to_tensor x_batched,
                                  to_tensor y_batched, int flags, void* ext_params, void* ext_grads,
                                  toh_trainer* out) {
  H_BEGIN
  H_NONNULL(n); H_NONNULL(x_batched); H_NONNULL(y_batched); H_NONNULL(out);
  *out = new toh_trainer_s{Trainer::create(n->net, loss, rate, borrow(x_batched), borrow(y_batched), flags,
                                           ext_params, ext_grads)};
  H_END
}

to_status toh_trainer_is_fused(toh_trainer t, int* out) {
  H_BEGIN
  H_NONNULL(t); H_NONNULL(out);
  *out = t->t->fused ? 1 : 0;
  H_END
}

to_status toh_trainer_is_graph(toh_trainer t, int* out) {
  H_BEGIN
  H_NONNULL(t); H_NONNULL(out);
  *out = t->t->graph ? 1 : 0;
  H_END
}

to_status toh_trainer_release(toh_trainer t) {
  delete t;
  return TO_OK;
}

to_status toh_trainer_grad(toh_trainer t) {
  H_BEGIN
  H_NONNULL(t);
  t->t->grad();
  H_END
}

to_status toh_trainer_apply(toh_trainer t) {
  H_BEGIN
  H_NONNULL(t);
  t->t->apply();
  H_END
}

to_status toh_trainer_step(toh_trainer t) {
  H_BEGIN
  H_NONNULL(t);
  t->t->step();
  H_END
}

to_status toh_trainer_flat(toh_trainer t, void** params, void** grads, int64_t* n_floats) {
  H_BEGIN
  H_NONNULL(t);
  if (params) check(to_data_ptr(t->t->flat_p.h(), params));
  if (grads) check(to_data_ptr(t->t->flat_g.h(), grads));
  if (n_floats) *n_floats = t->t->n_flat;
  H_END
}

to_status toh_trainer_net(toh_trainer t, toh_net* out) {
  H_BEGIN
  H_NONNULL(t); H_NONNULL(out);
  *out = new toh_net_s{t->t->net};
  H_END
}

to_status toh_trainer_launches_per_step(toh_trainer t, int64_t* out) {
  H_BEGIN
  H_NONNULL(t); H_NONNULL(out);
  *out = t->t->launches;
  H_END
}

to_status toh_trainer_step_launches(toh_trainer t, int64_t* out) {
  H_BEGIN
  H_NONNULL(t); H_NONNULL(out);
  *out = t->t->step_launches;
  H_END
}

// ---- online SGD over a resident data set (app/MNIST.hs:390-393) ------------------------------------------------
to_status toh_trainAll(toh_net n, int loss, double rate, to_tensor x_batched, to_tensor y_batched,
                       int64_t n_idx, const int64_t* idx, int flags, toh_net* out) {
  H_BEGIN
  H_NONNULL(n); H_NONNULL(x_batched); H_NONNULL(y_batched); H_NONNULL(out);
  *out = new toh_net_s{trainAll(n->net, loss, rate, borrow(x_batched), borrow(y_batched), n_idx, idx, flags)};
  H_END
}

// ---- Recurrent.hs -----------------------------------------------------------------------------------------
to_status toh_rnn_fullyConnected(int state_act, to_tensor s, to_tensor w_state, to_tensor w, to_tensor b,
                                 toh_rnn* out) {
  H_BEGIN
  H_NONNULL(s); H_NONNULL(w_state); H_NONNULL(w); H_NONNULL(b); H_NONNULL(out);
  *out = new toh_rnn_s{recurrent::fullyConnected(act_of(state_act), borrow(s), borrow(w_state), borrow(w), borrow(b))};
  H_END
}

to_status toh_rnn_fullyConnected_rand(int state_act, int64_t i, int64_t o, uint64_t seed, toh_rnn* out) {
  H_BEGIN
  H_NONNULL(out);
  *out = new toh_rnn_s{recurrent::fullyConnectedRand(act_of(state_act), i, o, seed)};
  H_END
}

to_status toh_rnn_ffLayer(to_tensor w, to_tensor b, toh_rnn* out) {
  H_BEGIN
  H_NONNULL(w); H_NONNULL(b); H_NONNULL(out);
  *out = new toh_rnn_s{recurrent::ffLayer(borrow(w), borrow(b))};
  H_END
}

to_status toh_rnn_stateless(toh_net ff, toh_rnn* out) {
  H_BEGIN
  H_NONNULL(ff); H_NONNULL(out);
  *out = new toh_rnn_s{recurrent::stateless(ff->net)};
  H_END
}

to_status toh_rnn_seq(toh_rnn a, toh_rnn b, toh_rnn* out) {
  H_BEGIN
  H_NONNULL(a); H_NONNULL(b); H_NONNULL(out);
  *out = new toh_rnn_s{recurrent::seq(a->net, b->net)};
  H_END
}

to_status toh_rnn_then_act(toh_rnn n, int act, toh_rnn* out) {
  H_BEGIN
  H_NONNULL(n); H_NONNULL(out);
  *out = new toh_rnn_s{recurrent::then(n->net, act_of(act)())};
  H_END
}

to_status toh_rnn_then_op(toh_rnn n, toh_op f, toh_rnn* out) {
  H_BEGIN
  H_NONNULL(n); H_NONNULL(f); H_NONNULL(out);
  arity_check(f->op.n_in == 1 && f->op.n_out == 1, "*~");
  *out = new toh_rnn_s{recurrent::then(n->net, f->op)};
  H_END
}

to_status toh_rnn_after_op(toh_op f, toh_rnn n, toh_rnn* out) {
  H_BEGIN
  H_NONNULL(n); H_NONNULL(f); H_NONNULL(out);
  arity_check(f->op.n_in == 1 && f->op.n_out == 1, "~*");
  *out = new toh_rnn_s{recurrent::after(f->op, n->net)};
  H_END
}

to_status toh_rnn_release(toh_rnn n) {
  delete n;
  return TO_OK;
}

to_status toh_rnn_counts(toh_rnn n, int* n_state, int* n_params) {
  H_BEGIN
  H_NONNULL(n);
  if (n_state) *n_state = n->net.n_s();
  if (n_params) *n_params = n->net.n_p();
  H_END
}

static void retained(const std::vector<T>& ts, to_tensor* out) {
  for (size_t i = 0; i < ts.size(); ++i) {
    check(to_retain(ts[i].h()));
    out[i] = ts[i].h();
  }
}

to_status toh_rnn_state(toh_rnn n, to_tensor* out) {
  H_BEGIN
  H_NONNULL(n); H_NONNULL(out);
  retained(n->net.state, out);
  H_END
}

to_status toh_rnn_params(toh_rnn n, to_tensor* out) {
  H_BEGIN
  H_NONNULL(n); H_NONNULL(out);
  retained(n->net.params, out);
  H_END
}

to_status toh_rnn_run(toh_rnn n, to_tensor x, to_tensor* y, toh_rnn* next) {
  H_BEGIN
  H_NONNULL(n); H_NONNULL(x); H_NONNULL(y); H_NONNULL(next);
  auto r = recurrent::runNetwork(n->net, borrow(x));
  *next = new toh_rnn_s{r.second};
  *y = r.first.release_handle();
  H_END
}

static std::vector<T> borrow_all(int n, const to_tensor* hs) {
  std::vector<T> v;
  for (int i = 0; i < n; ++i) {
    H_NONNULL(hs[i]);
    v.push_back(borrow(hs[i]));
  }
  return v;
}

to_status toh_rnn_netGrad(toh_rnn n, int loss, int n_steps, const to_tensor* xs, const to_tensor* ys,
                          to_tensor* g_inputs, to_tensor* g_state, to_tensor* g_params) {
  H_BEGIN
  H_NONNULL(n); H_NONNULL(g_state); H_NONNULL(g_params);
  if (n_steps < 0 || (n_steps > 0 && (!xs || !ys))) throw TensorOpsError(TO_ERR_ARG, "netGrad: bad step list");
  recurrent::Grads g = recurrent::netGradBatch(loss_of(loss), borrow_all(n_steps, xs), borrow_all(n_steps, ys), n->net);
  force_into(g.state, nullptr, g_state);
  force_into(g.params, nullptr, g_params);
  if (g_inputs) force_into(g.inputs, nullptr, g_inputs);
  H_END
}

to_status toh_rnn_trainNetwork(toh_rnn n, int loss, double rate_state, double rate_params, int n_steps,
                               const to_tensor* xs, const to_tensor* ys, toh_rnn* out) {
  H_BEGIN
  H_NONNULL(n); H_NONNULL(out);
  if (n_steps < 0 || (n_steps > 0 && (!xs || !ys))) throw TensorOpsError(TO_ERR_ARG, "trainNetwork: bad step list");
  *out = new toh_rnn_s{recurrent::trainNetwork(loss_of(loss), rate_state, rate_params, borrow_all(n_steps, xs),
                                               borrow_all(n_steps, ys), n->net)};
  H_END
}

// ---- AutoEncoder.hs ---------------------------------------------------------------------------------------
to_status toh_ae_encode(toh_net enc, toh_net dec, to_tensor x, to_tensor* out) {
  H_BEGIN
  H_NONNULL(enc); H_NONNULL(dec); H_NONNULL(x); H_NONNULL(out);
  *out = autoencoder::encode({enc->net, dec->net}, borrow(x)).release_handle();
  H_END
}

to_status toh_ae_decode(toh_net enc, toh_net dec, to_tensor y, to_tensor* out) {
  H_BEGIN
  H_NONNULL(enc); H_NONNULL(dec); H_NONNULL(y); H_NONNULL(out);
  *out = autoencoder::decode({enc->net, dec->net}, borrow(y)).release_handle();
  H_END
}

to_status toh_ae_encodeDecode(toh_net enc, toh_net dec, to_tensor x, to_tensor* out) {
  H_BEGIN
  H_NONNULL(enc); H_NONNULL(dec); H_NONNULL(x); H_NONNULL(out);
  *out = autoencoder::encodeDecode({enc->net, dec->net}, borrow(x)).release_handle();
  H_END
}

to_status toh_ae_testEncoder(toh_net enc, toh_net dec, int loss, to_tensor x, to_tensor* out) {
  H_BEGIN
  H_NONNULL(enc); H_NONNULL(dec); H_NONNULL(x); H_NONNULL(out);
  *out = autoencoder::testEncoder(loss_of(loss), {enc->net, dec->net}, borrow(x)).release_handle();
  H_END
}

to_status toh_ae_encGrad(toh_net enc, toh_net dec, int loss, to_tensor x, to_tensor* grads) {
  H_BEGIN
  H_NONNULL(enc); H_NONNULL(dec); H_NONNULL(x); H_NONNULL(grads);
  force_into(autoencoder::encGradBatch(loss_of(loss), borrow(x), {enc->net, dec->net}), nullptr, grads);
  H_END
}

to_status toh_ae_trainEncoder(toh_net enc, toh_net dec, int loss, double rate, to_tensor x, toh_net* enc_out,
                              toh_net* dec_out) {
  H_BEGIN
  H_NONNULL(enc); H_NONNULL(dec); H_NONNULL(x); H_NONNULL(enc_out); H_NONNULL(dec_out);
  autoencoder::Encoder e = autoencoder::trainEncoder(loss_of(loss), rate, borrow(x), {enc->net, dec->net});
  *enc_out = new toh_net_s{e.enc};
  *dec_out = new toh_net_s{e.dec};
  H_END
}

// ---- call trace -------------------------------------------------------------------------------------------
to_status toh_trace_begin(int n_leaves, const to_tensor* leaves) {
  H_BEGIN
  if (n_leaves > 0) H_NONNULL(leaves);
  trace::begin(n_leaves, leaves);
  H_END
}

to_status toh_trace_end(char* buf, int64_t cap, int64_t* length) {
  H_BEGIN
  H_NONNULL(length);
  static thread_local std::string pending;  // a too-small buffer does not lose the log
  if (trace::on() || pending.empty()) pending = trace::end();
  *length = (int64_t)pending.size();
  if (buf && cap > (int64_t)pending.size()) {
    std::memcpy(buf, pending.c_str(), pending.size() + 1);
    pending.clear();
  }
  H_END
}

}  // extern "C"
