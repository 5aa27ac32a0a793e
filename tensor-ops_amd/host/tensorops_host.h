/* tensorops_host.h -- C entry points of the C++ host mirror (libtensorops_host.so).
 *
 * The reference's host side (TOp DSL, Learn layer) is Haskell and stays Haskell in
 * a real integration; this library is its C++ rendering so the path above the
 * drop-in boundary (include/tensorops_hip.h) can be driven and tested here, where
 * no Haskell toolchain exists.  Names follow src/TensorOps/TOp.hs,
 * src/TensorOps/Types.hs and src/TensorOps/Learn/NeuralNet{,/FeedForward,/Recurrent,/AutoEncoder}.hs.
 * Closures (`forall a. RealFloat a => ...`) cross as SSA programs in the
 * `to_expr_compile` format; derivatives are taken on this side by forward-mode AD.
 */
#ifndef TENSOROPS_HOST_H
#define TENSOROPS_HOST_H
#include "../../include/tensorops_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

typedef struct toh_op_s* toh_op;           /* a TOp ns ms */
typedef struct toh_net_s* toh_net;         /* a Network t i o */
typedef struct toh_trainer_s* toh_trainer; /* replayed batched gradTOp step */

const char* toh_last_error(void);

/* ---- op vocabulary (src/TensorOps/TOp.hs) ---- */
/* name in: idOp(i) add add3 addN(i) duplicate replicate(i) swap negate scale(d) sumRows transpOp
 *          dot matVec vecMat matMat softmax squaredError crossEntropy
 *          actLogistic mapLogistic mapExp mapLog mapRecip mapTanh */
to_status toh_op_named(const char* name, int iarg, double darg, toh_op* out);
to_status toh_op_gmul(int len_m, int len_o, int len_n, toh_op* out);
/* map f = map' f (diff f) ; zipN n f = zipN' f (grad f): f as an SSA program */
to_status toh_op_map(int n_instr, const int32_t* code, int n_consts, const double* consts, toh_op* out);
to_status toh_op_map_with(int n_f, const int32_t* f, int nc_f, const double* c_f, int n_df,
                          const int32_t* df, int nc_df, const double* c_df, toh_op* out);
to_status toh_op_zipN(int n, int n_instr, const int32_t* code, int n_consts, const double* consts,
                      toh_op* out);
/* zipN' (TOp.hs:232-239): f and its n partial derivatives given explicitly, n + 1 SSA programs of arity n */
to_status toh_op_zipN_with(int n, int n_f, const int32_t* f, int nc_f, const double* c_f, const int32_t* n_g,
                           const int32_t* const* g, const int32_t* nc_g, const double* const* c_g, toh_op* out);
to_status toh_op_sumOp(int n, int rank, const int64_t* dims, toh_op* out);
to_status toh_op_konst(int n, int rank, const int64_t* dims, double x, toh_op* out);
to_status toh_op_shuffle(int n_in, int n_idx, const int32_t* idx, toh_op* out);
to_status toh_op_drop(int n_drop, int n, toh_op* out);
to_status toh_op_take(int n_take, int n, toh_op* out);
/* ---- combinators (src/TensorOps/Types.hs:135-264) ---- */
to_status toh_op_compose(toh_op first, toh_op then, toh_op* out); /* first >>> then */
to_status toh_op_first(toh_op o, int n_pass, toh_op* out);        /* firstOp */
to_status toh_op_second(int n_skip, toh_op o, toh_op* out);       /* secondOp */
to_status toh_op_then_first(toh_op a, toh_op b, toh_op* out);     /* a *>> b */
to_status toh_op_par(toh_op a, toh_op b, toh_op* out);            /* a *** b */
to_status toh_op_fanout(toh_op a, toh_op b, toh_op* out);         /* a &&& b */
to_status toh_op_arity(toh_op o, int* n_in, int* n_out);
to_status toh_op_release(toh_op o);
/* runTOp / gradTOp' / gradTOp.  `want` (nullable) = which cotangents to force; the
 * others stay unevaluated thunks (NULL handles), as under Haskell's laziness. */
to_status toh_run(toh_op o, int n_in, const to_tensor* xs, to_tensor* ys);
to_status toh_grad(toh_op o, int n_in, const to_tensor* xs, const to_tensor* ds, const int32_t* want,
                   to_tensor* dxs);
to_status toh_gradTOp(toh_op o, int n_in, const to_tensor* xs, const int32_t* want, to_tensor* dxs);

/* ---- Learn layer ---- */
enum { TOH_ACT_LOGISTIC = 0, TOH_ACT_MAP_LOGISTIC = 1, TOH_ACT_SOFTMAX = 2, TOH_ACT_MAP_TANH = 3 };
enum { TOH_LOSS_SQUARED_ERROR = 0, TOH_LOSS_CROSS_ENTROPY = 1 };
/* genNet with the weights given (FeedForward.hs:216-235) */
to_status toh_genNet(int n_layers, const to_tensor* ws, const to_tensor* bs, int hidden_act,
                     int out_act, toh_net* out);
/* genNet drawing W,b ~ normalDistr 0 0.5 on the device (FeedForward.hs:205-207) */
to_status toh_genNet_rand(int n_sizes, const int64_t* sizes, int hidden_act, int out_act,
                          uint64_t seed, toh_net* out);
/* buildNet / liftNet (params = 0) (:68-73, :110-113); n1 ~*~ n2 (:82-90); f ~* n (:96-101); n *~ f (:103-108) */
to_status toh_buildNet(toh_op o, int n_params, const to_tensor* params, toh_net* out);
to_status toh_net_seq(toh_net a, toh_net b, toh_net* out);
to_status toh_net_after_op(toh_op f, toh_net n, toh_net* out);
to_status toh_net_then_op(toh_net n, toh_op f, toh_net* out);
to_status toh_net_release(toh_net n);
to_status toh_net_n_params(toh_net n, int* out);
to_status toh_net_params(toh_net n, to_tensor* out /* retained handles */);
to_status toh_runNetwork(toh_net n, to_tensor x, to_tensor* out);
to_status toh_netGrad(toh_net n, int loss, to_tensor x, to_tensor y, int want_x,
                      to_tensor* grads /* [1 + n_params]; grads[0] NULL unless want_x */);
to_status toh_trainNetwork(toh_net n, int loss, double rate, to_tensor x, to_tensor y, toh_net* out);

/* ---- batched gradTOp step: G = sum_b gradTOp(x_b, p, y_b) over a fixed batch buffer ---- */
/* Parameters are moved into ONE flat buffer (what the data-parallel all-reduce
 * exchanges); the step is run once inside a memo scope (CSE of the recomputed
 * forward passes, Types.hs:155), captured as a HIP graph, and replayed. */
to_status toh_trainer_create(toh_net n, int loss, double rate, to_tensor x_batched,
                             to_tensor y_batched, int use_memo, int use_graph, toh_trainer* out);
/* same, with the flat parameter/gradient buffers owned by the caller (e.g. torch tensors
 * handed to torch.distributed); toh_trainer_flat_size gives the float count they need */
to_status toh_trainer_flat_size(toh_net n, int64_t* n_floats);
to_status toh_trainer_create_ext(toh_net n, int loss, double rate, to_tensor x_batched,
                                 to_tensor y_batched, int use_memo, int use_graph, void* ext_params,
                                 void* ext_grads, toh_trainer* out);
/* flags: 1 = memo scope (CSE), 2 = HIP-graph replay, 4 = let the library defer and fuse the class-method
 * stream inside the scope (csrc/lazy.cpp); without it every class-method call is one launch; 8 = build gradTOp's
 * thunks afresh on every directly-issued step, as the reference's evaluator does (default: the thunk graph is built
 * once and re-evaluated, the same class-method calls for 7 us less host time a step) */
enum { TOH_TRAINER_MEMO = 1, TOH_TRAINER_GRAPH = 2, TOH_TRAINER_FUSED = 4, TOH_TRAINER_FRESH_THUNKS = 8 };
to_status toh_trainer_create_opts(toh_net n, int loss, double rate, to_tensor x_batched,
                                  to_tensor y_batched, int flags, void* ext_params, void* ext_grads,
                                  toh_trainer* out);
to_status toh_trainer_is_fused(toh_trainer t, int* out);
to_status toh_trainer_is_graph(toh_trainer t, int* out); /* 1 when grad() replays a captured HIP graph */
to_status toh_trainer_release(toh_trainer t);
to_status toh_trainer_grad(toh_trainer t);  /* G <- summed parameter gradients */
to_status toh_trainer_apply(toh_trainer t); /* P <- P - rate * G (in place)     */
/* one trainNetwork step, parameters updated in place; with fusion on, the update `p - rate*g` is part of the
 * recorded stream and ends up inside the weight-gradient launches (the flat gradient buffer is not written) */
to_status toh_trainer_step(toh_trainer t);
to_status toh_trainer_flat(toh_trainer t, void** params, void** grads, int64_t* n_floats);
to_status toh_trainer_net(toh_trainer t, toh_net* out); /* network over the flat parameters */
to_status toh_trainer_launches_per_step(toh_trainer t, int64_t* out); /* kernel launches of one grad() */
to_status toh_trainer_step_launches(toh_trainer t, int64_t* out);     /* ... of one step() (0 before the first) */

/* `trainAll = foldl' (\nt (i,o) -> trainNetwork crossEntropy rate i o nt)` (app/MNIST.hs:390-393):
 * per-sample ONLINE SGD over the listed rows of a resident data set, in the given order
 * (idx NULL = rows 0..n_idx-1).  One captured HIP graph (gradTOp + update) is replayed per
 * sample on a one-sample staging buffer; flags as for toh_trainer_create_opts (GRAPH ignored). */
to_status toh_trainAll(toh_net n, int loss, double rate, to_tensor x_batched, to_tensor y_batched,
                       int64_t n_idx, const int64_t* idx, int flags, toh_net* out);

/* ---- recurrent networks (src/TensorOps/Learn/NeuralNet/Recurrent.hs) ---- */
typedef struct toh_rnn_s* toh_rnn; /* a recurrent Network t i o: op, initial state, params */
/* fullyConnected (:91-119): z = W x + W' s + b, output z, new state act(z); values given */
to_status toh_rnn_fullyConnected(int state_act, to_tensor s, to_tensor w_state, to_tensor w,
                                 to_tensor b, toh_rnn* out);
/* ... drawn from normalDistr 0 0.5 on the device */
to_status toh_rnn_fullyConnected_rand(int state_act, int64_t i, int64_t o, uint64_t seed, toh_rnn* out);
to_status toh_rnn_ffLayer(to_tensor w, to_tensor b, toh_rnn* out);   /* stateless ffLayer (:133-138) */
to_status toh_rnn_stateless(toh_net ff, toh_rnn* out);               /* stateless (:126-131)          */
to_status toh_rnn_seq(toh_rnn a, toh_rnn b, toh_rnn* out);           /* a ~*~ b (:170-222)            */
to_status toh_rnn_then_act(toh_rnn n, int act, toh_rnn* out);        /* n *~ getAct act (:247-252)    */
to_status toh_rnn_then_op(toh_rnn n, toh_op f, toh_rnn* out);        /* n *~ f                        */
to_status toh_rnn_after_op(toh_op f, toh_rnn n, toh_rnn* out);       /* f ~* n (:240-245)             */
to_status toh_rnn_release(toh_rnn n);
to_status toh_rnn_counts(toh_rnn n, int* n_state, int* n_params);
to_status toh_rnn_state(toh_rnn n, to_tensor* out /* retained */);
to_status toh_rnn_params(toh_rnn n, to_tensor* out /* retained */);
/* runNetwork (:224-232): y and the network carrying the new state */
to_status toh_rnn_run(toh_rnn n, to_tensor x, to_tensor* y, toh_rnn* next);
/* netGrad (:265-324) over n_steps (x_t, y_t) pairs given in TIME order; g_inputs comes back in the
 * reference's order (reversed time) and may be NULL = never forced */
to_status toh_rnn_netGrad(toh_rnn n, int loss, int n_steps, const to_tensor* xs, const to_tensor* ys,
                          to_tensor* g_inputs, to_tensor* g_state, to_tensor* g_params);
/* trainNetwork' (:326-356) */
to_status toh_rnn_trainNetwork(toh_rnn n, int loss, double rate_state, double rate_params, int n_steps,
                               const to_tensor* xs, const to_tensor* ys, toh_rnn* out);

/* ---- auto-encoder (src/TensorOps/Learn/NeuralNet/AutoEncoder.hs) ---- */
to_status toh_ae_encode(toh_net enc, toh_net dec, to_tensor x, to_tensor* out);        /* :42-48 */
to_status toh_ae_decode(toh_net enc, toh_net dec, to_tensor y, to_tensor* out);        /* :50-56 */
to_status toh_ae_encodeDecode(toh_net enc, toh_net dec, to_tensor x, to_tensor* out);  /* :58-63 */
to_status toh_ae_testEncoder(toh_net enc, toh_net dec, int loss, to_tensor x, to_tensor* out); /* :65-81 */
/* encGrad (:112-142): gradients of the encoder's then the decoder's parameters */
to_status toh_ae_encGrad(toh_net enc, toh_net dec, int loss, to_tensor x, to_tensor* grads);
to_status toh_ae_trainEncoder(toh_net enc, toh_net dec, int loss, double rate, to_tensor x,
                              toh_net* enc_out, toh_net* dec_out);                    /* :89-110 */

/* ---- call trace (tensorops/trace.hpp): the class-method stream this mirror issues, for tests ----
 * Between begin and end every `class Tensor` method the mirror calls on this thread is logged with the
 * identities of its operands; `leaves` name the program's inputs (ids 0..n-1).  end copies the log (text, one call
 * per line) into buf when it fits and always reports its length (without the terminating NUL). */
to_status toh_trace_begin(int n_leaves, const to_tensor* leaves);
to_status toh_trace_end(char* buf, int64_t cap, int64_t* length);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
