/* tensorops_hip.h -- C ABI of the MI355X (gfx950) backend for mstksg/tensor-ops.
 *
 * This is the drop-in boundary: exactly the entry points a Haskell
 * `instance Tensor HipT` (class at src/TensorOps/Types.hs:52-109) or
 * `instance BLAS HipB` (class at src/TensorOps/BLAS.hs:90-173) would bind with
 * `foreign import ccall` (stubs in INTEGRATION.md).  Plain pointers and sizes
 * only; no C++ or torch types.  All citations are relative to the reference
 * repository root.
 *
 * Conventions
 *  - Every function returns `to_status`: 0 = ok, nonzero = error; the message is
 *    in `to_last_error()` (thread-local).  The reference cannot have shape
 *    errors (dims are type-level); here every dim is re-validated and a
 *    mismatch is TO_ERR_SHAPE.
 *  - Tensors are immutable values behind opaque ref-counted handles, matching
 *    the pure class methods: every op returns a NEW handle in `*out`
 *    (refcount 1); inputs are never written.  A Haskell shim wraps handles in
 *    `ForeignPtr` with `to_release` as finaliser.
 *  - Layout: logical row-major, first dim slowest (`genBTensorA`,
 *    src/TensorOps/Backend/BTensor.hs:503-511).  Handles carry dims+strides so
 *    `transp` is a zero-copy view.
 *  - Hidden batch dimension (new capability): a handle may
 *    carry `batch` = B > 0 independent samples of the same logical shape,
 *    stored sample-major.  `batch` = 0 means "one value shared by all samples"
 *    (parameters).  Ops broadcast unbatched operands over the batch.
 *  - Scalars (`ElemT t` / `ElemB b`) cross as `double` and are rounded to the
 *    tensor dtype inside.
 *  - All work is enqueued on ONE HIP stream (`to_set_stream`); functions that
 *    return host data synchronise that stream.  The library is thread-safe
 *    (one global lock), so it can be called from any Haskell capability.
 */
#ifndef TENSOROPS_HIP_H
#define TENSOROPS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden */

typedef int32_t to_status;
typedef struct to_tensor_s* to_tensor; /* opaque, ref-counted */
typedef struct to_expr_s* to_expr;     /* compiled elementwise expression */
typedef struct to_graph_s* to_graph;   /* captured HIP graph */

enum {
  TO_OK = 0,
  TO_ERR_ARG = 1,    /* null pointer, bad enum, bad rank ...            */
  TO_ERR_SHAPE = 2,  /* dims do not satisfy the op's type-level contract */
  TO_ERR_HIP = 3,    /* a HIP runtime call failed                        */
  TO_ERR_STATE = 4,  /* not initialised, capture misuse ...              */
  TO_ERR_UNSUPPORTED = 5
};

/* dtype = `ElemT t` / `ElemB b`.  TO_F32 is the primary type (north_star parity 1e-5, MFMA and
 * fused paths); TO_F64 is what the reference's apps instantiate (`HMat Double`,
 * src/TensorOps/BLAS/HMat.hs:35): every op, fp64-MFMA GEMM, parity 1e-12.  Operands of one op
 * must share a dtype. */
enum { TO_F32 = 0, TO_F64 = 1 };

#define TO_MAX_RANK 8

/* ---- runtime ------------------------------------------------------------------ */
to_status to_init(int device);        /* idempotent; selects the device          */
to_status to_shutdown(void);          /* frees the pool; handles become invalid  */
const char* to_last_error(void);
to_status to_device_count(int* out);
to_status to_set_stream(void* hip_stream); /* NULL = library-owned stream       */
to_status to_get_stream(void** out);
/* `rnf` of NFData (app/Dots.hs:50-53, app/MNIST.hs:233-235) = wait for the stream */
to_status to_sync(void);
/* live handles / bytes held by the pool (leak checks in tests) */
to_status to_stats(int64_t* live_handles, int64_t* pool_bytes, int64_t* kernel_launches);
/* *ab_knobs = 1 in a development build (the A/B knobs of DESIGN.md are compiled in and read from the environment),
 * 0 in a product build, which reads the documented product switches only */
to_status to_build_info(int* ab_knobs);
/* The element type of the instance (`ElemT t`, src/TensorOps/Types.hs:54): values that
 * have no operand to take a dtype from (`sumT []`) and the host shims' constructors use it.
 * TO_F32 by default; TO_F64 selects the fp64 instance (HMat's element type). */
to_status to_set_default_dtype(int dtype);
to_status to_default_dtype(int* dtype);

/* ---- handles -------------------------------------------------------------------- */
to_status to_alloc(int dtype, int rank, const int64_t* dims, int64_t batch, to_tensor* out);
/* non-owning view of caller-owned device memory (e.g. a torch tensor), contiguous */
to_status to_wrap(void* device_ptr, int dtype, int rank, const int64_t* dims, int64_t batch,
                  to_tensor* out);
to_status to_retain(to_tensor t);
to_status to_release(to_tensor t);
to_status to_shape(to_tensor t, int* rank, int64_t* dims /*[TO_MAX_RANK]*/, int64_t* batch);
to_status to_dtype(to_tensor t, int* dtype);
to_status to_is_contiguous(to_tensor t, int* out);
to_status to_data_ptr(to_tensor t, void** out); /* base pointer of the view */
/* host <-> device, logical row-major order (sample-major when batched).
 * `generateA` / `fromList` (src/TensorOps/Tensor.hs:187-197) build on the host
 * and upload once; `toList`/`ixRows` traversals download once.
 * All three are synchronous (on return the bytes are where they go and `host` may be reused or freed).  `host` may be
 * any memory of the process -- a Haskell `Storable` vector, malloc, the stack: the copy engine never touches it.  The
 * bytes pass through the library's own page-locked staging buffers (two 4 MiB chunks, the next chunk's DMA overlapping
 * the CPU copy); memory the caller has page-locked itself (hipHostMalloc / hipHostRegister, a torch pinned tensor) is
 * transferred in place.  TOPS_PINNED_STAGING=0 hands `host` to the runtime's hipMemcpyAsync as rounds 1-4 did -- under
 * several processes sharing one GPU that path returned downloads with pieces of the destination unwritten (DESIGN_HISTORY.md 11.1). */
to_status to_upload(to_tensor t, const void* host, int64_t nbytes);
to_status to_download(to_tensor t, void* host, int64_t nbytes);
to_status to_from_host(int dtype, int rank, const int64_t* dims, int64_t batch,
                       const void* host, to_tensor* out);
/* transfers of caller memory since start: through the staging buffers / in place (caller-pinned or TOPS_PINNED_STAGING=0) */
to_status to_transfer_stats(int64_t* staged_calls, int64_t* staged_bytes, int64_t* direct_calls, int64_t* direct_bytes);
to_status to_fill(int dtype, int rank, const int64_t* dims, int64_t batch, double value,
                  to_tensor* out); /* `TT.konst`, src/TensorOps/Tensor.hs:49-54 */
/* `genRand` (Types.hs:93-96): counter-based generator, seed+element index -> value.
 * dist 0 = uniform[a,b) (`uniformDistr`), 1 = normal(mean a, std-dev b) (`normalDistr`, FeedForward.hs:206),
 * 2 = exponential(rate a), 3 = cauchy(location a, scale b), 4 = laplace(location a, scale b) -- the `ContGen`
 * instances of `statistics` with a closed-form inverse CDF.  Any other `ContGen d` goes the way the reference's
 * own backends go (BTensor.hs:841: `generateA (\_ -> genContVar d g)`): draw on the host, one to_from_host. */
to_status to_rand(int dtype, int rank, const int64_t* dims, int64_t batch, int dist, double a,
                  double b, uint64_t seed, to_tensor* out);

/* ---- class Tensor (src/TensorOps/Types.hs:52-109) --------------------------------- */
/* gmul (Types.hs:60-66): a : ms++os, b : Reverse os ++ ns -> ms++ns,
 * C[m,n] = sum_o A[m,o1..oq] * B[oq..o1,n] (src/Data/Nested.hs:465-472).
 * The three Length witnesses cross as small ints.
 * One product is DEFERRED in every mode, also outside a scope and with TOPS_LAZY=0: `len_o == 0` on two BATCHED operands
 * -- the per-sample outer products `gradTOp` hands back for a weight matrix (src/TensorOps/TOp.hs:86-88), B*o*i numbers
 * whose only use in a gradient is their sum over the batch.  The returned handle has a shape and no storage; to_sum and
 * to_scale of it record behind it; to_batch_sum of it IS to_gmul_batch_sum (one GEMM with K = B); anything else that
 * asks for its elements produces it then (TO_ERR_UNSUPPORTED above TOPS_OUTER_MAX_BYTES).  The operands are read when
 * that happens: the library's own in-place entry points order themselves after such readers, and when either operand
 * is caller-owned memory (to_wrap) outside a scope the product is computed at once instead, so memory the library
 * cannot watch is never read late.  (to_batch_sum, to_gmul, to_sum, to_scale may therefore plan and launch: `safe` imports.) */
to_status to_gmul(int len_m, int len_o, int len_n, to_tensor a, to_tensor b, to_tensor* out);
/* liftT (Types.hs:56-59): n-ary elementwise map of a compiled expression */
to_status to_lift(to_expr f, int n, const to_tensor* xs, to_tensor* out);
/* sumT (Types.hs:69): left fold of n same-shaped tensors; n == 0 -> zeros of
 * (rank, dims), the `SingI o` evidence (src/Data/List/Util.hs:7-10) */
to_status to_sum(int n, const to_tensor* xs, int rank, const int64_t* dims, to_tensor* out);
to_status to_scale(double alpha, to_tensor x, to_tensor* out); /* scaleT (Types.hs:70) */
to_status to_transp(to_tensor x, to_tensor* out);               /* transp (Types.hs:71-73), zero-copy */
to_status to_sum_rows(to_tensor x, to_tensor* out);             /* sumRows (Types.hs:82-84) */
/* mapRows (Types.hs:77-81) with a constant function -- the only use on the hot
 * path (`TO.sumRows` gradient, src/TensorOps/TOp.hs:155-158): every ms-slice
 * under `like`'s leading `len_n` dims := row */
to_status to_map_rows_const(int len_n, to_tensor row, to_tensor like, to_tensor* out);
/* general mapRows / ixRows (Types.hs:77-81,100-106) are host traversals over
 * zero-copy row views: slice -> user function (device ops) -> stack */
to_status to_slice(to_tensor x, int len_m, const int64_t* index, to_tensor* out);
to_status to_stack(int rank_m, const int64_t* dims_m, const to_tensor* rows, to_tensor* out);
to_status to_diag(int rank, to_tensor x, to_tensor* out);     /* diag (Types.hs:85-88) */
to_status to_get_diag(to_tensor x, to_tensor* out);           /* getDiag (Types.hs:89-92) */
to_status to_index(to_tensor x, const int64_t* index, int64_t sample, double* out); /* (!) (Types.hs:107-109) */
/* `TT.argMax` (src/TensorOps/Tensor.hs:291-305) of a vector, per sample when batched: the
 * index of the maximum, EARLIEST index on ties (`Max (Arg x j)` keeps its left argument).
 * Writes max(batch,1) indices to host memory (one download instead of the reference's
 * per-element `ixRows` traversal, Tensor.hs:220-230). */
to_status to_arg_max(to_tensor x, int64_t* host_out);
/* `TT.argMin` (Tensor.hs:307-321): same, minimum; ties -> the earliest index (Min over Arg keeps the left) */
to_status to_arg_min(to_tensor x, int64_t* host_out);
/* `TT.oneHot` (Tensor.hs:275-289) for a batch of indices: out[b][j] = (j == idx[b]) ? hot : cold */
to_status to_one_hot(int dtype, int64_t n, double hot, double cold, int64_t batch,
                     const int64_t* host_idx, to_tensor* out);

/* ---- class BLAS (src/TensorOps/BLAS.hs:90-173); rank-1/2 handles only ------------- */
to_status to_blas_axpy(double alpha, to_tensor x, to_tensor y_or_null, to_tensor* out); /* :97-101 */
to_status to_blas_dot(to_tensor x, to_tensor y, double* out);                           /* :102-104 */
to_status to_blas_ger(to_tensor x, to_tensor y, to_tensor* out);                        /* :108-110 */
to_status to_blas_gemv(double alpha, to_tensor a, to_tensor x, double beta,
                       to_tensor y_or_null, to_tensor* out);                           /* :111-116 */
to_status to_blas_gemm(double alpha, to_tensor a, to_tensor b, double beta,
                       to_tensor c_or_null, to_tensor* out);                           /* :118-123 */
to_status to_blas_scale(double alpha, to_tensor x, to_tensor* out);                     /* scaleB :124-127 */
to_status to_blas_add(to_tensor x, to_tensor y, to_tensor* out);                        /* addB :128 */
to_status to_blas_index_row(int64_t i, to_tensor a, to_tensor* out);                    /* indexRowB :133-136 */
to_status to_blas_transp(to_tensor a, to_tensor* out);                                  /* transpB :137-139 */
to_status to_blas_eye(int dtype, int64_t n, to_tensor* out);                            /* eye :160-161 */
to_status to_blas_trace(to_tensor a, double* out);                                      /* traceB :162-164 */
to_status to_blas_diag(to_tensor x, to_tensor* out);                                    /* diagB :165-167 */
to_status to_blas_get_diag(to_tensor a, to_tensor* out);                                /* getDiagB :168-170 */
to_status to_blas_sum(to_tensor x, double* out);                                        /* sumB :171-173 */
/* liftB (:92-96) = to_lift; indexB (:129-132) = to_index; iRowsB/iElemsB/bgenA/
 * bgenRowsA (:140-159) are host traversals built from to_download/to_from_host. */

/* ---- elementwise expressions ("reified closures") ---------------- */
/* SSA program: value v < arity is input v; value arity+i is the result of
 * instruction i = {op, a, b}; TO_X_CONST takes consts[a].  The last value is
 * the result.  `RealFloat`-polymorphic closures (Types.hs:114-117) are reified
 * by instantiating them at a symbolic element type that emits this code. */
enum {
  TO_X_CONST = 0, TO_X_ADD, TO_X_SUB, TO_X_MUL, TO_X_DIV, TO_X_NEG, TO_X_RECIP, TO_X_EXP,
  TO_X_LOG, TO_X_SQRT, TO_X_ABS, TO_X_SIGNUM, TO_X_SIN, TO_X_COS, TO_X_TANH, TO_X_POW,
  TO_X_MAX, TO_X_MIN, TO_X_NOPS
};
to_status to_expr_compile(int arity, int n_instr, const int32_t* code /*[3*n_instr]*/,
                          int n_consts, const double* consts, to_expr* out);
to_status to_expr_release(to_expr e);
/* which kernel runs it: 0 = bytecode VM, 1..99 = a pre-fused functor, 100 = a kernel specialised
 * for this program at run time (hiprtc; TOPS_EXPR_JIT=0 disables) */
to_status to_expr_kind(to_expr e, int* kind);

/* ---- batching extension (G = sum_b gradTOp(x_b, p, y_b)) ---------- */
to_status to_batch_sum(to_tensor x, to_tensor* out);     /* [B; ns] -> ns, sum over samples */
to_status to_batch_bcast(to_tensor x, int64_t batch, to_tensor* out); /* ns -> [B; ns] */
to_status to_batch_select(to_tensor x, int64_t sample, to_tensor* out); /* view of one sample */
/* zero-copy view of samples [start, start+count) (a `V.splitAt` chunk, app/MNIST.hs:319) */
to_status to_batch_slice(to_tensor x, int64_t start, int64_t count, to_tensor* out);
/* out[k] = x[idx[k]]: a re-ordered / sub-sampled data set in one pass (the `uniformShuffle`d
 * queue, app/MNIST.hs:308); idx is host memory */
to_status to_batch_gather(to_tensor x, int64_t n_idx, const int64_t* host_idx, to_tensor* out);
/* gmul followed by the sum over samples, fused (the cotangent of an unbatched
 * operand): out = sum_b gmul(a_b, b_b); e.g. dW = sum_b dz_b (x) x_b = dZ^T X */
to_status to_gmul_batch_sum(int len_m, int len_o, int len_n, to_tensor a, to_tensor b,
                            to_tensor* out);

/* ---- purity-based CSE and graph replay ---------------------------------------------- */
/* Inside a memo scope a pure op called again with the same input handles
 * returns the same result handle (values are immutable, so this is exact); it
 * removes the reference's forward recomputation (Types.hs:155). */
to_status to_memo_begin(void);
to_status to_memo_end(void);
/* The scope is also a FUSION scope, and it belongs to the calling thread.  Inside it the pure class methods
 * (to_gmul, to_gmul_batch_sum, to_lift, to_sum, to_scale, to_sum_rows, to_map_rows_const, to_batch_sum,
 * to_fill; to_transp and the slicing views of their results) validate their shapes and return at once with a
 * DEFERRED handle: the op is recorded, nothing is launched.  The recorded graph runs -- with bias, activation,
 * loss head, row sums and the `p - r*g` update folded into the GEMM launches where the kernels allow -- when a
 * value is needed: to_download / to_index / to_data_ptr / any eager entry point taking it, to_force,
 * to_force_many, to_copy_into (which lets the source be produced straight into the destination).  Closing a scope,
 * ending a capture and to_sync demand NOTHING: a handle that is still deferred then stays deferred and is produced when it is asked for (under
 * a garbage collector every intermediate of a step is still "held" until its finaliser has run; launching those
 * would re-run the step unfused).  So a host forces what a step produces -- all of it in ONE to_force_many call,
 * which plans the results together -- before it closes the scope.  Results nobody asks for are never computed
 * (call-by-need, like the reference).  What a value IS never changes; only when it is computed does.  Caller-owned memory (to_wrap) read by recorded ops must not be changed behind the
 * library's back while deferred results derived from it are alive; the library's own in-place entry points
 * (to_upload, to_copy_into, to_sgd_step_inplace, to_comm_allreduce_sum ...) order themselves after such
 * readers.  TOPS_LAZY=0 makes every call eager again. */
/* Numerical contract of the fused loss heads.  A recognised head (found by evaluating the recorded row-local subgraph
 * on three random rows against the closed forms the kernel carries, z in [-2, 2], targets in [0.05, 1.05], agreement to
 * 1e-9; only programs without abs/signum/max/min/pow are considered, so agreement on random points means identity) runs
 * as DIFFERENT CODE from the ops it replaces: softmax >>> crossEntropy's backward as softmax(z) * sum(y) - y with the
 * row maximum subtracted before exp, logistic >>> squaredError's as -2 (y - s) s (1 - s).  Wherever the recorded ops are
 * finite the two agree within rounding (tests hold the fused step to 1e-5 / 1e-11 of the fp64 oracle); beyond the range
 * of a literal evaluation (a logit above ln(max float): 88.7 in fp32, 709.8 in fp64) the fused head returns the limit
 * value of the same formula where the recorded ops -- and the reference -- return inf/NaN
 * (tests/test_gpu_lazy.py::test_extreme_logits_state_the_loss_heads_contract). */
/* switch for the deferral on the CALLING THREAD (default: TOPS_LAZY, on); returns the previous setting */
to_status to_set_lazy(int on, int* previous_or_null);
/* The loss-head recognition above is an identity test, not a proof; a host that prefers the recorded ops as they are (a row
 * program, or one launch per op) turns it off -- for everything planned from now on, process-wide (default: TOPS_LOSS_HEAD_MATCH,
 * on); returns the previous setting.  Plans cached under the other setting are not reused. */
to_status to_set_loss_head_match(int on, int* previous_or_null);
/* `rnf` of ONE value for a lazy host (`instance NFData (HipT ns)`): make t's storage exist (enqueue, not wait) */
to_status to_force(to_tensor t);
/* `rnf` of a product of values (the new parameters of a training step): one plan, so that launches shared between
 * them -- a weight gradient and its bias gradient, the pair of weight-gradient GEMMs -- are shared */
to_status to_force_many(int n, const to_tensor* ts);
/* counters since start: ops recorded, fused GEMM launches, recorded ops that never got storage of their own, plans */
to_status to_lazy_stats(int64_t* recorded, int64_t* fused_launches, int64_t* elided, int64_t* flushes);
/* host time spent planning / in plans + their launches, nanoseconds since start */
to_status to_lazy_time(int64_t* plan_ns, int64_t* flush_ns);
/* The plan cache: a scope that records the same graph as an earlier one (same ops, wiring, operand layouts, aliasing,
 * demands -- a training loop) reuses that plan instead of planning again; nothing the host has to do.  TOPS_PLAN_CACHE=0
 * turns it off.  Counters since start / drop every cached plan. */
to_status to_plan_cache_stats(int64_t* hits, int64_t* misses, int64_t* entries);
to_status to_plan_cache_clear(void);
/* host time spent inside this library's entry points and how many were called, since start (the rest of a step's
 * host time is the caller's own) */
to_status to_api_time(int64_t* ns, int64_t* calls);
/* stream capture of everything enqueued between begin/end into a HIP graph */
to_status to_graph_begin(void);
to_status to_graph_end(to_graph* out);
to_status to_graph_launch(to_graph g);
/* nodes of the captured step, and whether a replay issues them as plain kernel launches (a short step of
 * kernels only: cheaper than hipGraphLaunch on this stack; TOPS_REPLAY_LIST_MAX=0 forces the HIP graph) */
to_status to_graph_info(to_graph g, int* n_launches, int* replays_as_launch_list);
to_status to_graph_release(to_graph g);

/* ---- in-place parameter update (program-level, NOT a class method) ------------------ */
/* p <- p - r*g on caller-owned buffers: the only mutation in the library; used
 * by the replayed training step where parameters live at fixed addresses.
 * Same arithmetic as `stepFunc` (FeedForward.hs:145-147). */
to_status to_sgd_step_inplace(to_tensor p, to_tensor g, double rate);
/* dst <- src on caller-owned storage (lands a freshly computed gradient in the
 * flat buffer the data-parallel all-reduce works on) */
to_status to_copy_into(to_tensor dst, to_tensor src);
/* the same for n (dst, src) pairs in ONE launch (landing every parameter gradient of a step) */
to_status to_copy_into_many(int n, const to_tensor* dsts, const to_tensor* srcs);

/* ---- data-parallel exchange ---------------------------------------- */
/* One process per GPU.  Rank 0 calls to_comm_unique_id and hands the 128 bytes to every rank over any
 * transport the host has; every rank then calls to_comm_init(rank, world, id).  to_comm_allreduce_sum
 * sums a contiguous tensor (the flat gradient buffer) over all ranks in place, enqueued on the library
 * stream (RCCL over xGMI; librccl.so is loaded on first use, TOPS_RCCL_LIB overrides the path). */
to_status to_comm_unique_id(void* out_128_bytes);
to_status to_comm_init(int rank, int world, const void* id_128_bytes);
to_status to_comm_allreduce_sum(to_tensor t);
to_status to_comm_world(int* world); /* the ranks RCCL reports for the communicator (ncclCommCount); 0 when none exists */
to_status to_comm_shutdown(void);

/* The same exchange without RCCL: a one-shot two-phase all-reduce over hipIpc-mapped peer buffers (every rank
 * pushes 1/world of its vector to each peer over all xGMI links at once, reduces its own slice in rank order,
 * pushes the sum to everybody).  to_p2p_create allocates this rank's exchange buffer for vectors of up to
 * max_elems elements and returns its 64-byte IPC handle; the ranks all-gather the handles over any transport and
 * call to_p2p_connect(rank, handles[world][64]).  to_p2p_allreduce_sum(g): g <- sum over ranks, in place, one
 * launch on the library stream.  to_p2p_allreduce_sgd: p <- p - rate * sum (the update of FeedForward.hs:141-147
 * in the same launch; g also receives the sum when also_write_g).  Every rank must pass vectors of the same
 * length.  Alignment: a vector whose length is a multiple of 16 bytes is exchanged 16 bytes per lane -- the slice
 * layout of the inbox / outbox has to be the same on every rank, so it is decided by the LENGTH alone -- and its
 * local address (g, and p) must then be 16-byte aligned too: an offset view of a flat buffer that breaks this is
 * TO_ERR_ARG, not a silent slow path (pad the view, or exchange the whole buffer).  A peer that does not show up within TOPS_P2P_TIMEOUT_S (5 s) makes the launch give up:
 * to_p2p_status then reports a nonzero code and further exchanges are refused. */
to_status to_p2p_create(int64_t max_elems, int dtype, int world, void* out_ipc_handle_64_bytes);
to_status to_p2p_connect(int rank, const void* handles_world_x_64_bytes);
to_status to_p2p_allreduce_sum(to_tensor g);
to_status to_p2p_allreduce_sgd(to_tensor p, to_tensor g, double rate, int also_write_g);
to_status to_p2p_status(int* world, int* timed_out_code);
to_status to_p2p_shutdown(void);

/* ---- pre-fused ffLayer stack (program-level, like the two calls above) --------------- */
/* Batched parameter gradients of `genNet` stacks (FeedForward.hs:216-235),
 *   a_l = act_l (W_l a_{l-1} + b_l),  l = 1..n_layers,  loss(a_L, y),
 * summed over the hidden batch of x/y, written into the caller's gW[l] / gb[l]
 * (e.g. views of the flat all-reduce buffer).  Mathematically the same as gradTOp on
 * the generic path; the per-op launches are collapsed into GEMMs with fused epilogues.
 * hidden_act: TO_ACT_LOGISTIC; (out_act, loss): (TO_ACT_SOFTMAX, TO_LOSS_CROSS_ENTROPY)
 * or (TO_ACT_LOGISTIC, TO_LOSS_SQUARED_ERROR).  losses_or_null: per-sample loss [B]. */
enum { TO_ACT_LOGISTIC = 0, TO_ACT_SOFTMAX = 2 };
enum { TO_LOSS_SQUARED_ERROR = 0, TO_LOSS_CROSS_ENTROPY = 1 };
to_status to_fflayer_stack_grad(int n_layers, const to_tensor* w, const to_tensor* b, int hidden_act,
                                int out_act, int loss, to_tensor x, to_tensor y, const to_tensor* gw,
                                const to_tensor* gb, to_tensor losses_or_null);
/* The whole `trainNetwork` step (FeedForward.hs:131-148: p <- p - rate * gradTOp ...) of the same stack on
 * one batch, parameters updated in place: the weight-gradient launches subtract rate * gradient in their
 * epilogue, so the step has no separate update launch.  For a single process (data-parallel ranks need the
 * gradient itself for the all-reduce: to_fflayer_stack_grad + to_sgd_step_inplace).  TO_ERR_UNSUPPORTED,
 * with the parameters untouched, when a weight gradient is not in the range of the fused-epilogue kernel. */
to_status to_fflayer_stack_sgd(int n_layers, const to_tensor* w, const to_tensor* b, int hidden_act, int out_act,
                               int loss, to_tensor x, to_tensor y, double rate, to_tensor losses_or_null);

/* Per-sample online SGD -- `foldl' (\nt (i,o) -> trainNetwork loss rate i o nt)` (app/MNIST.hs:390-396,
 * app/Dots.hs:74-80) -- of the same stacks over rows idx[0..n_idx) (null: rows 0..n_idx-1) of the resident batched X / Y,
 * parameters updated in place, as ONE persistent launch: the workgroups keep the parameters in LDS between samples,
 * layer 1 split by rows and layer 2 by columns over up to 32 workgroups of one XCD, one exchange per sample
 * (csrc/online_sgd.hip).  fp32 or fp64, 2..6 layers, input <= 2048, head <= 64 outputs, everything a workgroup holds within
 * 160 KiB of LDS: TO_ERR_UNSUPPORTED otherwise, with the parameters untouched (the generic path -- one recorded and
 * fused step per sample -- always works).  Blocks until the stream of samples is done. */
to_status to_fflayer_stack_online_sgd(int n_layers, const to_tensor* w, const to_tensor* b, int hidden_act, int out_act,
                                      int loss, to_tensor X, to_tensor Y, int64_t n_idx, const int64_t* idx_or_null,
                                      double rate);

/* The same, found by the library itself.  `g` is a captured ONE-SAMPLE training step (whatever the host's DSL recorded
 * in a scope: gradTOp of its network, the update, to_copy_into of the new parameters), x_buf / y_buf the buffers that
 * step reads its sample from.  If the launches the planner made of that step are exactly the trainNetwork step of an
 * ffLayer stack -- GEMVs with bias + logistic, a recognised loss head, the cotangents back through the layers, every
 * layer's outer-product update in place -- and the stack fits the persistent kernel, the samples idx[0..n_idx) of X / Y
 * are trained in one launch and *handled = 1; otherwise *handled = 0 and nothing has been done (replay `g` per sample).
 * The host says nothing about what its network is made of.  TOPS_ONLINE_KERNEL=0 disables it. */
to_status to_graph_online_sgd(to_graph g, to_tensor x_buf, to_tensor y_buf, to_tensor X, to_tensor Y, int64_t n_idx,
                              const int64_t* idx_or_null, int* handled);

/* how often to_graph_online_sgd recognised a captured step and ran the persistent kernel, and over how many samples */
to_status to_online_sgd_stats(int64_t* runs, int64_t* samples);

/* ---- measurement ---------------------------------------------------------------------- */
/* Average duration (ms) of kernels enqueued between the two calls, measured with
 * HIP events on the library's stream. */
to_status to_timer_start(void);
to_status to_timer_stop(float* ms);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* TENSOROPS_HIP_H */
