// Shared internals of libtensorops_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/tensorops_hip.h"

struct to_tensor_s;

// Switches.  The PRODUCT switches -- what a user may set -- are read with getenv and listed in DESIGN.md section 3:
// TOPS_LAZY, TOPS_LAZY_FUSE, TOPS_LAZY_DEBUG, TOPS_EXPR_JIT, TOPS_ROWPROG, TOPS_PLAN_CACHE, TOPS_STEP_SEAM,
// TOPS_ONLINE_KERNEL, TOPS_ONLINE_GRAPH, TOPS_REPLAY_LIST_MAX, TOPS_OUTER_MAX_BYTES, TOPS_RCCL_LIB, TOPS_P2P_TIMEOUT_S,
// TOPS_ONLINE_TIMEOUT_S, TOPS_PINNED_STAGING, TOPS_GEMM_KW_KSPLIT, TOPS_LOSS_HEAD_MATCH (0: the planner's loss-head recognition off),
// TOPS_SIBLING_BATCH (0: sibling products / lifts of a plan one launch each), TOPS_GEMV (0: matVec / vecMat / outer / tall column sums
// on the routes they had before csrc/gemv.hip); tests/test_gpu_switches.py walks every one of them.  Everything else -- the A/B knobs the
// measurements in DESIGN.md and profiles/README.md were made with, per-kernel debug stamps -- exists in a development
// build only (TOPS_BUILD_AB=1 python tensor-ops_amd/build.py: -DTOPS_AB_KNOBS): a product build does not read them, so they
// are not routes the product can be steered onto.
// (internal linkage: tools/build_ab_lib.py links objects compiled with and without TOPS_AB_KNOBS into one library, and two
//  different definitions of one inline function would be an ODR violation)
static inline const char* ab_getenv(const char* name) {
#ifdef TOPS_AB_KNOBS
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

namespace to {
struct Node;  // lazy.cpp: the recorded op a deferred handle stands for

struct Error : std::runtime_error {
  to_status code;
  Error(to_status c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void fail(to_status c, const std::string& m) { throw Error(c, m); }

#define TO_HIP(expr)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess)                                                              \
      ::to::fail(TO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
  } while (0)

#define TO_CHECK(cond, code, msg)                  \
  do {                                             \
    if (!(cond)) ::to::fail((code), (msg));        \
  } while (0)

// ---- device memory ---------------------------------------------------------------
struct Buffer {  // one pool allocation, shared by views
  void* ptr = nullptr;
  size_t bytes = 0;   // size class
  bool owned = true;  // false: wrapped caller memory
  std::atomic<int> refs{1};
};

struct Runtime {
  bool inited = false;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool capturing = false;
  std::vector<std::vector<void*>> free_lists;  // per size class (log2)
  int64_t pool_bytes = 0;
  int64_t live_handles = 0;
  int64_t launches = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // second stream + fork/join events for independent kernels of one step (to_fflayer_stack_grad)
  hipStream_t side = nullptr;
  hipEvent_t fork_ev[8] = {nullptr}, join_ev = nullptr;
  int default_dtype = TO_F32;  // dtype of values that have no operand to take it from (`sumT []`)
  // every tensor created while capturing stays reserved for the graph's lifetime: a replay
  // rewrites those buffers, so they must never be handed to another live value
  std::vector<to_tensor_s*> capture_kept;
};
Runtime& rt();
std::recursive_mutex& lock();

Buffer* pool_alloc(size_t bytes);
void buffer_release(Buffer* b);

// Transfers between the CALLER's memory and the device (runtime.cpp).  Both are synchronous: on return the destination holds
// the bytes and the source may be reused.  The device never touches pageable caller memory: the bytes pass through the
// library's own pinned staging buffers (TOPS_PINNED_STAGING=0: the runtime's hipMemcpyAsync on the caller's pointer, as in
// rounds 1-4).  Memory the caller has pinned itself (hipHostMalloc / hipHostRegister) is transferred in place.
void host_to_device(void* dst, const void* host, size_t nbytes, hipStream_t s);
void device_to_host(void* host, const void* src, size_t nbytes, hipStream_t s);
void staging_shutdown();
// a small host array (a pointer table of a batched launch) -> device memory, stream-ordered, through pinned memory; the
// returned device address is valid for launches enqueued on `s` before the ring comes round again (4 uploads)
const void* table_upload(const void* host, size_t bytes, hipStream_t s);
// consecutive slices of ONE pool allocation for n handles without storage (each a multiple of 16 bytes): results of
// sibling ops that the next batched launch can read as one range
void alloc_storage_shared(int n, const to_tensor* ts);
struct TransferStats { long long staged_calls, staged_bytes, direct_calls, direct_bytes; };
TransferStats transfer_stats();

// ---- tensor handle -------------------------------------------------------------------
}  // namespace to

struct to_tensor_s {
  std::atomic<int> refs{1};
  to::Buffer* buf = nullptr;
  void* ptr = nullptr;  // base of the view
  int dtype = TO_F32;
  size_t esize() const { return dtype == TO_F64 ? 8 : 4; }
  void* at(int64_t elems) const { return static_cast<char*>(ptr) + elems * (int64_t)esize(); }
  float* f32() const { return static_cast<float*>(ptr); }
  int rank = 0;
  int64_t dims[TO_MAX_RANK] = {0};
  int64_t strides[TO_MAX_RANK] = {0};  // elements
  int64_t batch = 0;                   // 0 = unbatched (shared by all samples)
  int64_t bstride = 0;                 // elements between samples
  uint64_t id = 0;                     // identity for the memo table
  // Deferred values (lazy.cpp).  Inside a fusion scope the pure class methods return handles whose
  // shape is known but whose storage does not exist yet: `ptr == nullptr` and either `node` (the op
  // that will produce it; owned) or `view_base` (a view of such a handle; retained).  Every entry
  // point that touches memory calls to::ensure() first.  A fresh value is always contiguous, so
  // dims/strides are final from the start.
  to::Node* node = nullptr;
  to_tensor_s* view_base = nullptr;
  int64_t view_off = 0;                // elements into view_base
  int int_refs = 0;                    // how many of `refs` are held by the library (nodes, views, memo)
  std::vector<to_tensor_s*> dviews;    // the deferred views of this handle (each holds one of int_refs)
  bool pending() const { return ptr == nullptr; }

  int64_t numel() const {
    int64_t n = 1;
    for (int i = 0; i < rank; ++i) n *= dims[i];
    return n;
  }
  int64_t total() const { return numel() * (batch > 0 ? batch : 1); }
  bool inner_contiguous() const {
    int64_t s = 1;
    for (int i = rank - 1; i >= 0; --i) {
      if (dims[i] != 1 && strides[i] != s) return false;
      s *= dims[i];
    }
    return true;
  }
  bool contiguous() const {
    return inner_contiguous() && (batch <= 1 || bstride == numel());
  }
};

namespace to {

to_tensor new_tensor(int rank, const int64_t* dims, int64_t batch, int dtype = TO_F32);  // fresh contiguous
to_tensor new_deferred(int rank, const int64_t* dims, int64_t batch, int dtype);         // shape only, no storage
void alloc_storage(to_tensor t);                                                         // gives a deferred handle its buffer
void adopt_storage(to_tensor t, to_tensor from);  // t takes (shares) from's buffer; from must be contiguous, same shape
to_tensor new_view(to_tensor base, int rank, const int64_t* dims, const int64_t* strides,
                   int64_t batch, int64_t bstride, int64_t offset);
to_tensor contiguous(to_tensor x);  // retained x if already contiguous, else a packed copy
void retain(to_tensor t);
void release(to_tensor t);
bool same_shape(to_tensor a, to_tensor b);
std::string shape_str(to_tensor t);

struct Holder {  // RAII for temporaries
  to_tensor t;
  explicit Holder(to_tensor x = nullptr) : t(x) {}
  ~Holder() {
    if (t) release(t);
  }
  Holder(const Holder&) = delete;
  Holder& operator=(const Holder&) = delete;
  to_tensor take() {
    to_tensor r = t;
    t = nullptr;
    return r;
  }
};

inline void count_launch() { rt().launches++; }

// ---- launch records ------------------------------------------------------------------------------------------
// Every kernel goes out through launch_k.  While a step is being captured (to_graph_begin .. to_graph_end) each
// launch is also remembered with its arguments: a captured step that consists of a handful of kernels is
// replayed by issuing those launches again, which on this stack is cheaper than hipGraphLaunch (three launches
// of the config-3 step: 26.5 us issued directly, 31.5 us as a HIP graph).
struct LaunchRec {
  virtual void replay(hipStream_t s) const = 0;
  virtual ~LaunchRec() {}
};
template <class... KA>
struct LaunchRecT final : LaunchRec {
  void (*kern)(KA...);
  dim3 grid, block;
  size_t lds;
  std::tuple<std::decay_t<KA>...> args;
  LaunchRecT(void (*k)(KA...), dim3 g, dim3 b, size_t l, std::tuple<std::decay_t<KA>...> a)
      : kern(k), grid(g), block(b), lds(l), args(std::move(a)) {}
  void replay(hipStream_t s) const override {
    std::apply([&](const auto&... x) { hipLaunchKernelGGL(kern, grid, block, lds, s, x...); }, args);
  }
};
std::vector<std::unique_ptr<LaunchRec>>* launch_recorder();  // non-null while a capture is recording
void set_launch_recorder(std::vector<std::unique_ptr<LaunchRec>>* r);

template <class... KA, class... A>
inline void launch_k(void (*kern)(KA...), dim3 grid, dim3 block, size_t lds, hipStream_t s, A&&... a) {
  if (auto* rec = launch_recorder())
    rec->emplace_back(new LaunchRecT<KA...>(kern, grid, block, lds, std::tuple<std::decay_t<KA>...>(a...)));
  hipLaunchKernelGGL(kern, grid, block, lds, s, a...);
}

// comm.cpp: the data-parallel exchange (RCCL, loaded on first use)
void comm_unique_id(void* out128);
void comm_init(int rank, int world, const void* id128);
void comm_allreduce_sum(to_tensor t, hipStream_t s);
void comm_shutdown();
int comm_world();
// p2p.hip: the one-shot all-reduce over hipIpc-mapped peer buffers
void p2p_create(int64_t max_elems, int dtype, int world, void* out_handle64);
void p2p_connect(int rank, const void* handles);
void p2p_allreduce(to_tensor g, to_tensor p, double rate, bool write_g, hipStream_t s);
int p2p_status();
int p2p_world();
void p2p_shutdown();

// ---- kernels (each .hip file) ---------------------------------------------------------
struct GemmProblem {
  int dtype = TO_F32;
  const void* A;
  const void* B;
  void* C;
  int64_t M, N, K;
  int64_t a_sm, a_sk;  // element strides
  int64_t b_sk, b_sn;
  int64_t c_sm;        // C row stride (c_sn == 1)
  int64_t batch;       // >= 1
  int64_t a_sb, b_sb, c_sb;
  int reduce_batch;    // 1: C = sum_b A_b B_b (c_sb ignored)
  double alpha, beta;  // C = alpha*A*B + beta*Cin
  const void* Cin;     // same layout as C; may be null when beta == 0
  // fused epilogue (the pre-fused ffLayer path): v = alpha*acc + beta*Cin ; v += bias[n] ;
  // act 1: v = logistic(v) ; dact: v *= h*(1-h) with h = dact[m*c_sm + n] (same layout as C)
  // (element type = dtype; the tiled fp64 kernel has no fused epilogue, the small-GEMM kernel has it for both)
  const void* bias = nullptr;
  int act = 0;               // 1: logistic, 2: tanh
  const void* dact = nullptr;
  int dact_kind = 0;         // what `dact` holds: 0: h = logistic(z), v *= h (1 - h);  1: h = tanh(z), v *= 1 - h^2
  void* rowsum = nullptr;  // optional [M]: sum_k A[m,k], produced by the small-GEMM kernel only
  // rowsum_acc: rowsum[m] += rowsum_alpha * sum_k A[m,k] instead (the bias update of the fused SGD step)
  bool rowsum_acc = false;
  double rowsum_alpha = 1.0;
  const void* rowsum_in = nullptr;  // rowsum_acc: rowsum[m] = rowsum_in[m] + rowsum_alpha * sum (null: rowsum itself, in place)
  // loss head fused into the last layer's GEMM (small-GEMM kernel, N <= 16, see gemm_small_fuses_loss):
  // 1: C = softmax(v) * sum(target row) - target (softmax >>> crossEntropy backward)
  // 2: C = -2 (t - s) s (1 - s), s = logistic(v)   (logistic >>> squaredError backward)
  int loss_rows = 0;
  const void* target = nullptr;  // [M][N], same layout as C
  void* loss_out = nullptr;      // optional [M]: the per-row loss value
  // ... and, behind the loss head, the cotangent of the previous layer for the same rows:
  // tail_out[M][tail_n] = (dz[M][N] . tail_w[N][tail_n]) * h (1 - h),  h = tail_h[M][tail_n]
  // (`dZ_{L-1} = dZ_L . W_L (.) logistic'`), which removes one launch from the step
  const void* tail_w = nullptr;
  const void* tail_h = nullptr;
  void* tail_out = nullptr;
  int tail_n = 0;
  // Sibling products in one launch (lazy.cpp, round 6): `M` rows are `M / a_table_rows` matrices of a_table_rows rows each
  // that live in SEPARATE allocations -- matrix i's rows start at a_table[i] (a device array of pointers; same strides for
  // all) -- and share B; C is one tall matrix.  Understood by the short-K streaming kernel only (gemm_skinnyk.hip).
  const void* a_table = nullptr;
  int64_t a_table_rows = 0;
};
bool gemm_kw_long_k(const GemmProblem& p);   // gemm_kwave.hip: fewer than 100 tiles under K >= 1,536 that it splits over workgroups / streams
int gemv_form(const GemmProblem& p, bool standalone = false);                      // gemv.hip: 0 not there, 1 matVec / vecMat (M or N = 1), 2 outer product (K = 1); standalone: asked by run_gemm
void launch_gemv(const GemmProblem& p, hipStream_t s);
bool launch_column_sum(int dtype, const void* x, void* out, int64_t R, int64_t J, int64_t si, hipStream_t s);   // out[j] = sum_i x[i si + j]; false: not here
bool gemm_small_fuses_loss(const GemmProblem& p);
bool gemm_small_fuses_tail(const GemmProblem& p, int64_t tail_n);
void launch_gemm_f64(const GemmProblem& p, hipStream_t s);
bool gemm_t32_applicable(const GemmProblem& p);  // gemm_t32.hip: ~one round of 32x32 tiles, four waves each, DMA-fed (the training step's two big contractions)
void launch_gemm_t32(const GemmProblem& p, hipStream_t s);
bool launch_gemm_t32_pair(const GemmProblem& p1, const GemmProblem& p2, hipStream_t s);   // two weight-gradient contractions, one launch
bool launch_gemm_t32_head(const GemmProblem& fwd, const GemmProblem& head, hipStream_t s); // forward layer + the loss-head launch reading its output, one launch
void gemm_t32_init();   // its row-block counters (to_init)
int gemm_t32_take_failure();   // nonzero once after a joined launch gave up waiting (its outputs are invalid; the form is off from then on)
bool gemm_kw_applicable(const GemmProblem& p);  // gemm_kwave.hip: 64x64 tiles, K split over the waves of a workgroup
void launch_gemm_kw(const GemmProblem& p, hipStream_t s);
bool gemm_kw16_applicable(const GemmProblem& p);  // gemm_kw16.hip: the same design on 48x48 / 48x64 / 64x48 / 80x80 tiles of 16x16 MFMA blocks
void launch_gemm_kw16(const GemmProblem& p, hipStream_t s);
bool gemm_kw64_applicable(const GemmProblem& p);  // gemm_kwave_f64.hip: the same design on v_mfma_f64_16x16x4_f64
void launch_gemm_kw64(const GemmProblem& p, hipStream_t s);
bool gemm_skinnyk64_applicable(const GemmProblem& p);  // gemm_skinnyk_f64.hip: config 5's shape class in Double
void launch_gemm_skinnyk64(const GemmProblem& p, hipStream_t s);
struct GemmEpilogue {
  const float* bias;
  const float* dact;
  int act;
};
void launch_gemm_mfma(const GemmProblem& p, hipStream_t s);
void launch_gemm_small(const GemmProblem& p, hipStream_t s);
bool gemm_small_applicable(const GemmProblem& p);
bool gemm_small_can(const GemmProblem& p);
bool launch_gemm_small_pair(const GemmProblem& p1, const GemmProblem& p2, hipStream_t s);
// forward layer + the loss-head launch reading its output, joined by an intra-XCD seam (gemm_small.hip)
bool launch_gemm_small_seam(const GemmProblem& fwd, const GemmProblem& head, hipStream_t s);
void gemm_small_seam_init();
void gemm_kw_pair_init();   // gemm_kwave.hip: workspace + counters of the two-workgroups-per-tile form
// forward / output layer + loss head / the two weight gradients of a batched step as ONE launch with grid barriers
bool launch_gemm_small_chain(const GemmProblem& pa, const GemmProblem& pb, const GemmProblem& pc1, const GemmProblem& pc2,
                             hipStream_t s);
int gemm_small_chain_status();  // nonzero: a grid barrier of a chained launch timed out (its results are invalid)
// nonzero ONCE after a joined forward + loss-head launch (TOPS_STEP_SEAM) gave up waiting for a row block: its outputs are
// invalid; the seam is off for the rest of the process (checked by to_sync and by every flush of recorded ops)
int gemm_small_seam_take_failure();
// "workgroup b of a grid runs on XCD b % 8": probed once with the grid shape the kernels use (gemm_kwave.hip).  Observed
// behaviour, not a contract: the seam launch and the online-SGD kernel, whose hand-overs meet in ONE XCD's L2, refuse to
// run without it; the wave-split GEMM's several-workgroups-per-tile form asks for it for speed only.
bool xcd_placement_probe();
void launch_gemm_naive(const GemmProblem& p, hipStream_t s);
// short-K streaming GEMM (gemm_skinnyk.hip): B resident in LDS, barrier-free wave streams; alpha, bias, act
bool gemm_skinnyk_applicable(const GemmProblem& p);
void launch_gemm_skinnyk(const GemmProblem& p, hipStream_t s);
bool gemm_mfma_worthwhile(const GemmProblem& p);
bool gemm_w4_full_rounds(const GemmProblem& p);
bool gemm_w4_edge_whole(const GemmProblem& p);  // ragged (multiples of 4) but worth running whole on the pinned kernel
bool gemm_f64_w4_full_rounds(const GemmProblem& p);

// elementwise
enum EwKind {
  EW_VM = 0,
  EW_AFFINE = 1,       // c + sum_i a_i x_i   (n <= 4)
  EW_MUL = 2,          // x0 * x1
  EW_EXP = 3,
  EW_LOG = 4,
  EW_RECIP = 5,
  EW_LOGISTIC = 6,     // 1/(1+exp(-x0))
  EW_MUL_DLOGISTIC = 7,  // x0 * s(x1)(1-s(x1))
  EW_TANH = 8,
  EW_SQRT = 9,
  EW_DIV = 10,         // x0 / x1
  EW_CONST = 11,       // arity 0 or constant function
  EW_MUL_H1MH = 12,    // x0 * x1 (1 - x1): d * logistic'(z) written on h = logistic(z) (planner rewrite, lazy.cpp)
  EW_MUL_DTANH = 13,   // x0 * (1 - tanh(x1)^2)
  EW_MUL_1MH2 = 14,    // x0 * (1 - x1^2): d * tanh'(z) written on h = tanh(z) (planner rewrite)
};
struct EwArgs {
  int dtype;
  int kind;
  int n;                    // inputs (<= 8)
  const void* x[8];
  int64_t period[8];        // element count of input i (index = e % period); == total if full
  void* out;
  int64_t total;
  double coef[4];           // EW_AFFINE
  double c0;                // EW_AFFINE constant / EW_CONST value
  // VM
  const int32_t* d_code;    // device copy [3*n_instr] (slot-allocated: dst,a,b packed, see expr.cpp)
  const void* d_consts;     // float or double copy, matching dtype
  int n_instr;
  int n_slots;
  int result_slot;
  void* jit;                // hiprtc-built kernels for this program (null: run the VM)
};
void launch_ewise(const EwArgs& a, hipStream_t s);
struct to_expr_s_fwd;
void jit_launch(void* h, const EwArgs& a, hipStream_t s);
void jit_release(void* h);

// reductions / layout
// out[o, j] = sum_i x[o*so + i*si + j*sj], o<O, i<R, j<J ; out contiguous [O,J]
// (every launcher takes the element type as `dtype` and dispatches to a float / double instantiation)
void launch_sum_axis(int dtype, const void* x, void* out, int64_t O, int64_t R, int64_t J, int64_t so,
                     int64_t si, int64_t sj, hipStream_t s);
// C[m,n] (row stride c_sm) = sum_split work[split][m][n]  (fp32, N % 4 == 0)
void launch_sum_splits_strided(const void* work, void* C, int splits, int64_t M, int64_t N, int64_t c_sm, hipStream_t s);
// out[o, i, j] = d[o*dso + j], contiguous out [O,R,J]
void launch_bcast_axis(int dtype, const void* d, void* out, int64_t O, int64_t R, int64_t J, int64_t dso,
                       hipStream_t s);
// packed row-major copy of a strided view (batch folded in as leading dim by caller)
void launch_copy_strided(int dtype, const void* src, void* dst, int rank, const int64_t* dims,
                         const int64_t* strides, hipStream_t s);
void launch_fill(int dtype, void* dst, int64_t n, double v, hipStream_t s);
void launch_rand(int dtype, void* dst, int64_t n, int dist, double a, double b, uint64_t seed, hipStream_t s);
void launch_diag(int dtype, const void* x, void* out, int64_t n, int rank, hipStream_t s);   // out pre-zeroed
void launch_get_diag(int dtype, const void* x, void* out, int64_t n, int64_t step, hipStream_t s);
void launch_sgd(int dtype, void* p, const void* g, double r, int64_t n, hipStream_t s);
void launch_arg_max_rows(int dtype, const void* x, long long* out, int64_t B, int64_t n, int64_t bstride,
                         int64_t stride, hipStream_t s, bool minimum = false);
void launch_multi_copy(int n, const void* const* srcs, void* const* dsts, const int64_t* dwords, hipStream_t s);
// out[b][k][:] = rows[k][b][:], k < n <= 16 (one launch); out points at row r0 of the stacked result
void launch_stack_rows(int n, const void* const* rows, const int64_t* row_sb_dwords, void* out, int64_t B,
                       int64_t row_dwords, int64_t out_sb_dwords, hipStream_t s);
void launch_gather_rows(const void* x, void* out, const long long* idx, int64_t n_rows, int64_t row_bytes,
                        hipStream_t s);
void launch_one_hot(int dtype, void* out, const long long* idx, int64_t B, int64_t n, double hot, double cold,
                    hipStream_t s);
constexpr int RANK1_MAX_LAYERS = 8;
// gW_l = dz_l (x) a_l, gb_l = dz_l for n layers in one launch (acc: P += alpha * gradient in place)
void launch_rank1_many(int dtype, int n, const void* const* dz, const void* const* a, void* const* w, void* const* b,
                       const int64_t* rows, const int64_t* cols, double alpha, bool acc, hipStream_t s);
// per layer: W = (w_in ? w_in : 0) + alpha * dz (x) a ; b likewise (b may be null)
void launch_rank1_general(int dtype, int n, const void* const* dz, const void* const* a, void* const* w, void* const* b,
                          const void* const* w_in, const void* const* b_in, const double* alpha, const int64_t* rows,
                          const int64_t* cols, hipStream_t s);
// online_sgd.hip: per-sample SGD over a stream of samples as one persistent launch (fp32 ffLayer stacks)
bool online_sgd_plan(int dtype, int L, const int64_t* dims, int* G_out, int* rpw_out, size_t* lds_out);
void launch_online_sgd(int dtype, int L, const int64_t* dims, void* const* W, void* const* b, const void* X, const void* Y,
                       const long long* idx_dev, int64_t n, double rate, int head, hipStream_t s);
bool online_sgd_placement_ok(hipStream_t s);   // probed once: do workgroups b, b + 8, ... of a grid share an XCD?
int online_sgd_status();
void online_sgd_reset_status();
void launch_loss_grad_rows(int dtype, const void* z, const void* y, void* dz, void* loss, int64_t B, int64_t n,
                           int kind, hipStream_t s);

}  // namespace to

// ---- expression handle ------------------------------------------------------------------
struct to_expr_s {
  std::atomic<int> refs{1};     // the host's reference + one per recorded node
  uint64_t uid = 0;             // never reused (memo keys; an address can be)
  uint64_t sid = 0;             // structure id: equal for two instances with the same arity, code and constants (interned
                                // exactly, no hashing: plan-cache signatures are keyed on it)
  int arity = 0;
  std::vector<int32_t> code;    // 3 per instr: op, a, b  (SSA value ids)
  std::vector<double> consts;
  int kind = 0;                 // EwKind
  double coef_d[4] = {0, 0, 0, 0};  // EW_AFFINE coefficients / constant (rounded to the dtype at launch)
  double c0_d = 0;
  // VM form: slot-allocated
  std::vector<int32_t> vm_code;  // 4 per instr: op, dst_slot, a_slot, b_slot (CONST: a = const idx)
  int n_slots = 0, result_slot = 0;
  int32_t* d_code = nullptr;
  float* d_consts_f32 = nullptr;
  double* d_consts_f64 = nullptr;
  void* jit[2] = {nullptr, nullptr};  // JitKernels per dtype (expr_jit.cpp), built on first use
  bool jit_tried[2] = {false, false};
  std::string jit_error;              // why the JIT was not used (empty if it was, or never tried)
};
namespace to {
void* jit_build(const to_expr_s& e, int dtype, std::string* err);
}
