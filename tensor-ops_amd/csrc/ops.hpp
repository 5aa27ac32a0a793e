// Host-side op implementations shared by the eager entry points (api.cpp) and the deferred
// executor (lazy.cpp).  Everything here takes MATERIALISED handles (ptr != nullptr).
#pragma once
#include <map>
#include <unordered_map>
#include <memory>
#include <string>
#include <vector>

#include "common.hpp"

namespace to {

to_expr expr_compile(int arity, int n_instr, const int32_t* code, int n_consts, const double* consts);
void expr_release(to_expr e);
void expr_retain(to_expr e);
void expr_prepare(to_expr e, int dtype);
void expr_shutdown();   // (the pinned bounce buffer of upload_small)
double expr_eval(const to_expr_s& e, const double* x);  // the program in double on the host
bool expr_is_smooth(const to_expr_s& e);                // no ABS / SIGNUM / MAX / MIN / POW anywhere
uint64_t fresh_id();

hipStream_t S();
void require_init();
void no_capture(const char* what);

// ---- GEMM routing ----------------------------------------------------------------------------------
void run_gemm(const GemmProblem& p);
// would run_gemm honour alpha/beta/Cin/bias/act/dact of this problem (every kernel it may pick carries them)?
bool gemm_epilogue_ok(const GemmProblem& p);
// is the problem in the range of the small-GEMM kernel, the only one with rowsum / loss head / tail epilogues?
bool gemm_small_route(const GemmProblem& p);

// ---- gmul planning -----------------------------------------------------------------------------------
// `gmul lM lO lN a b` as ONE GemmProblem.  dry = true: shapes only -- validates, derives
// the output shape and the problem's extents/strides from the handles' dims and strides without touching
// memory (operands may be deferred); `exact` is false when the real plan would first pack or pre-sum an
// operand (then extents are right but strides are not final).
struct GmulPlan {
  GemmProblem p{};
  int out_rank = 0;
  int64_t odims[TO_MAX_RANK] = {0};
  int64_t out_batch = 0;
  int dtype = TO_F32;
  bool zero = false;   // empty contraction: the result is all zeros
  bool exact = true;
  bool rows_are_samples = false;  // C is [B rows of one sample each] x N (batch folded into M; one GEMM)
  Holder ha, hb;       // packed / pre-summed operands (real plans only)
};
void gmul_plan(GmulPlan& gp, int lm, int lo, int ln, to_tensor a, to_tensor b, bool reduce, bool dry);

// ---- eager implementations (fresh contiguous results, refcount 1) ------------------------------------
to_tensor gmul_impl(int lm, int lo, int ln, to_tensor a, to_tensor b, bool reduce);
void lift_check(to_expr f, int n, const to_tensor* xs, int64_t* batch, int* dtype);
to_tensor lift_impl(to_expr f, int n, const to_tensor* xs, int rank_hint, const int64_t* dims_hint,
                    int dtype_hint = TO_F32);
void lift_launch_raw(to_expr f, int n, const void* const* xs, void* out, int64_t total, int dtype);
to_tensor affine_impl(int n, const to_tensor* xs, const double* coef, double c);
to_tensor kind_impl(int kind, int n, const to_tensor* xs);  // a pre-fused functor by EwKind
to_tensor sum_impl(int n, const to_tensor* xs, int rank, const int64_t* dims, int dtype0);
to_tensor transp_impl(to_tensor x);
to_tensor sum_rows_impl(to_tensor x);
to_tensor batch_sum_impl(to_tensor x);
void map_rows_const_check(int len_n, to_tensor row, to_tensor like);
to_tensor map_rows_const_impl(int len_n, to_tensor row, to_tensor like);
// the result of a `mapRows` / `ixRows` traversal: rows laid out under the leading dims dims_m (validates; returns the
// result's dims and batch)
void stack_check(int rank_m, const int64_t* dims_m, const to_tensor* rows, int64_t* nrows, int64_t* odims, int64_t* batch);
to_tensor stack_impl(int rank_m, const int64_t* dims_m, const to_tensor* rows);

// ---- deferred execution (lazy.cpp) ---------------------------------------------------------------------
enum NodeOp {
  N_GMUL = 1,      // lm, lo, ln, reduce ; in = {a, b}
  N_LIFT,          // f ; in = xs
  N_SUM,           // in = xs (n >= 2)
  N_SCALE,         // alpha ; in = {x}
  N_SUM_ROWS,      // in = {x}
  N_MAP_ROWS,      // len_n ; in = {row, like}
  N_BATCH_SUM,     // in = {x}
  N_FILL,          // alpha = value ; no inputs
  N_DACT,          // in = {d, h}: d * h (1 - h)  (planner rewrite of `d * logistic'(z)`); lm = 1: d * (1 - h^2) (of `d * tanh'(z)`)
  N_STACK,         // len_n = how many leading dims the rows are laid out under ; in = the rows, row-major
};
struct NodeDesc {
  int op = 0;
  int lm = 0, lo = 0, ln = 0;
  bool reduce = false;
  to_expr f = nullptr;
  double alpha = 0.0;
  int len_n = 0;
};
// is the calling thread inside a fusion scope (to_memo_begin .. to_memo_end) with deferral enabled?
bool lazy_active();
int lazy_set(int on);  // returns the previous setting
int lazy_set_loss_head_match(int on);  // the planner's loss-head recognition (process-wide); returns the previous setting
// record an op: returns a deferred handle of the given shape (refcount 1)
to_tensor lazy_record(const NodeDesc& d, int n_in, const to_tensor* in, int rank, const int64_t* dims,
                      int64_t batch, int dtype);
// the recorded op a deferred handle stands for (false: t has storage, is a view, or was not recorded)
bool lazy_node_of(to_tensor t, NodeDesc* d, std::vector<to_tensor>* in);
// make t's storage exist (runs the recorded graph it depends on, fused where the kernels allow)
void ensure(to_tensor t);
void ensure_all(int n, const to_tensor* ts);
// dst[i] <- src[i]; deferred sources are produced straight into the destination when nothing else needs them
void lazy_copy_into(int n, const to_tensor* dsts, const to_tensor* srcs);
// an in-place write to t's memory is about to happen: first run every recorded op that still reads it
void before_write(to_tensor t);
// launch every deferred result of this thread that the host still holds and no recorded op consumes
void lazy_flush_sinks();
void lazy_flush_all();  // every thread's (graph replay, shutdown)

// ---- fusion scope + CSE memo (thread-local) -----------------------------------------------------------------
struct MemoKey {
  std::vector<uint64_t> k;
  bool operator==(const MemoKey& o) const { return k == o.k; }
};
struct MemoKeyHash {
  size_t operator()(const MemoKey& m) const {
    uint64_t h = 1469598103934665603ull;
    for (uint64_t v : m.k) {
      h ^= v;
      h *= 1099511628211ull;
      h ^= h >> 31;
    }
    return (size_t)h;
  }
};
to_tensor memo_find(const MemoKey& key);
void memo_put(const MemoKey& key, to_tensor t);
void scope_begin();
void scope_end();
void scope_reset_all();  // to_shutdown
// ---- row programs (rowprog.cpp): a row-local subgraph of the recorded stream as ONE run-time compiled kernel ---------
// Values are per row (sample): vectors of N elements or scalars.  Value ids: 0 = the root vector, 1 .. n_ext = existing
// tensors read by the program (vectors [N] or scalars, per row or shared by all rows), then one per node, in order.
enum RowOp { R_CONST = 1, R_LIFT, R_DACT, R_SUM, R_SCALE, R_SUM_ROWS, R_MAP_ROWS, R_DOT, R_MUL };
struct RowNode {
  int op = 0;
  bool vec = false;          // the result is a vector (else a scalar)
  std::vector<int> in;       // value ids
  to_expr f = nullptr;       // R_LIFT (retained by the program)
  double alpha = 0.0;        // R_SCALE factor / R_CONST value
};
struct RowProg {
  int dtype = TO_F32;
  int64_t N = 0;
  std::vector<char> ext_vec, ext_rowwise;  // per external: vector?  one per row (else shared by all rows)?
  std::vector<RowNode> nodes;
  std::vector<int> outs;     // value ids written back (each a buffer of its own)
  void* module = nullptr;    // compiled form (null: not built / failed, see err)
  bool tried = false;
  std::string err;
  ~RowProg();
};
bool rowprog_build(RowProg& rp);  // false: no run-time compiler or the build failed (rp.err)
void rowprog_launch(const RowProg& rp, const void* root, const void* const* ext, void* const* outs, int64_t rows,
                    hipStream_t s);

// What a captured step consisted of, launch by launch (filled while a capture is recording): lets the library see that a
// captured one-sample step IS an ffLayer stack's trainNetwork step -- from the plan it made of the class-method stream,
// not from anything the host says about its network (to_graph_online_sgd, api.cpp).
struct StepDesc {
  int kind = 0;       // 0: one GEMM launch with its epilogue (p), 1: the outer-product updates of all layers, 2: anything else
  GemmProblem p{};
  int n = 0;          // kind 1
  const void* dz[RANK1_MAX_LAYERS] = {nullptr};
  const void* a[RANK1_MAX_LAYERS] = {nullptr};
  void* w[RANK1_MAX_LAYERS] = {nullptr};
  void* b[RANK1_MAX_LAYERS] = {nullptr};
  const void* w_in[RANK1_MAX_LAYERS] = {nullptr};
  const void* b_in[RANK1_MAX_LAYERS] = {nullptr};
  double alpha[RANK1_MAX_LAYERS] = {0};
  int64_t rows[RANK1_MAX_LAYERS] = {0}, cols[RANK1_MAX_LAYERS] = {0};
};
void lazy_describe_into(std::vector<StepDesc>* v);  // null: stop describing
void lazy_cache_stats(int64_t* hits, int64_t* misses, int64_t* entries);  // the plan cache (lazy.cpp)
void lazy_cache_clear();
int64_t lazy_stat(int which);  // 0 recorded nodes, 1 fused groups launched, 2 nodes elided, 3 flushes

}  // namespace to
