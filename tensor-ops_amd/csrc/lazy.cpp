// Deferred execution of the `class Tensor` method stream, and its fusion into the GEMM kernels' epilogues.
//
// Why this exists.  The reference's `Network` is `{Sing ps, TOp, Prod t ps}` -- no activation tags
// (src/TensorOps/Learn/NeuralNet/FeedForward.hs:57-61) -- and `trainNetwork` only ever calls `gradTOp` on
// opaque closures (FeedForward.hs:131-148, src/TensorOps/Types.hs:127-132).  All a backend sees is a stream
// of class-method calls (src/TensorOps/Types.hs:52-109): gmul, liftT, sumT, scaleT, transp, sumRows, mapRows.
// Run one kernel per call and the MNIST step is 29 launches.  The methods are pure and their results are
// immutable values, so inside a fusion scope (to_memo_begin .. to_memo_end) they are RECORDED instead: each
// call returns a handle with a shape and no storage, and when a result is demanded (a download, `to_force`,
// `to_copy_into`, the end of the scope for results the host still holds) the recorded graph is planned:
//
//   gmul -> (+ unbatched vector) -> logistic                  one GEMM launch, bias and activation in its epilogue
//   gmul -> scale / p - r*g with p of the result's shape     alpha, beta*Cin in the epilogue (the SGD update)
//   d * logistic'(z) where h = logistic(z) is in the graph    rewritten to d * h(1-h): z need not be stored
//   the row-local subgraph between z = gmul+bias [B x N<=16] and dz (softmax, log, dot, ... and their
//   cotangents, whatever order the host's AD produced)        evaluated on the host on random rows and compared
//                                                             with the closed forms the small-GEMM kernel carries
//                                                             (softmax>>>crossEntropy, logistic>>>squaredError):
//                                                             the loss head of that launch
//   gmul(W^T, dz) -> * h(1-h)                                 the tail of the loss-head launch
//   gmul_batch_sum(dz, a) next to batch_sum(dz)               the weight-gradient GEMM with its row sums
//   two independent weight-gradient GEMMs                     one launch (gemm_small_pair_kernel)
//   to_copy_into(dst, result)                                 the result is produced in dst
//
// Anything that matches no rule runs through the same eager implementation as outside a scope, so the recorded
// graph always has a correct execution; rules only remove launches and intermediate stores.  Results nobody
// demands are never computed (the reference gets this from Haskell's laziness: the input's cotangent that
// `trainNetwork` drops with `tail'`, FeedForward.hs:142).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <unordered_map>

#include "ops.hpp"

namespace to {

// ---- recorded ops ---------------------------------------------------------------------------------------------
struct Node {
  NodeDesc d;
  uint64_t seq = 0;    // recording order: inputs always have smaller numbers
  uint64_t owner = 0;  // recording thread
  std::vector<to_tensor> in;  // retained (counted in int_refs)
  to_tensor_s* out = nullptr; // the handle that owns this node
  Node *prev = nullptr, *next = nullptr;  // the global list of pending nodes
  uint64_t plan_epoch = 0;  // position in the plan being built (valid while plan_epoch == the plan's epoch)
  uint64_t write_mark = 0;  // stale_after_write's own mark (its own field: a plan may be live when a write hazard is checked)
  int plan_idx = -1;
};

static Node* g_head = nullptr;
static uint64_t g_seq = 0;
static int64_t g_stats[6] = {0, 0, 0, 0, 0, 0};  // ..., [4] ns spent planning, [5] ns spent in flushes in all

static uint64_t this_thread() {
  static std::atomic<uint64_t> next{1};
  static thread_local uint64_t id = next++;
  return id;
}

int64_t lazy_stat(int which) { return which >= 0 && which < 6 ? g_stats[which] : 0; }

static void retain_int(to_tensor t) {
  t->refs.fetch_add(1);
  t->int_refs++;
}
static void release_int(to_tensor t) {
  t->int_refs--;
  release(t);
}

static void unlink(Node* n) {
  if (n->prev) n->prev->next = n->next;
  else if (g_head == n) g_head = n->next;
  if (n->next) n->next->prev = n->prev;
  n->prev = n->next = nullptr;
}

// frees t's node (the value exists now, or the handle died): releases the inputs it kept alive
void lazy_drop_node(to_tensor t) {
  Node* n = t->node;
  if (!n) return;
  t->node = nullptr;
  unlink(n);
  std::vector<to_tensor> in;
  in.swap(n->in);
  if (n->d.f) expr_release(n->d.f);
  delete n;
  for (to_tensor x : in) release_int(x);
}

// ---- fusion scope: thread-local depth + CSE memo -----------------------------------------------------------------
struct Scope {
  int depth = 0;
  std::unordered_map<MemoKey, to_tensor, MemoKeyHash> memo;
};
// (heap-allocated and never freed: to_shutdown may walk the list after the thread is gone; a thread that exits
//  drops its memo table)
struct ScopeOwner {
  Scope* s = new Scope();
  ~ScopeOwner();
};
static Scope& scope() {
  static thread_local ScopeOwner o;
  return *o.s;
}
// the scopes of all threads that ever opened one, so that to_shutdown can drop their memo tables
static std::vector<Scope*>& all_scopes() {
  static std::vector<Scope*> v;
  return v;
}

// The deferral switch: TOPS_LAZY gives the default, to_set_lazy overrides it FOR THE CALLING THREAD (scopes and memo
// tables are per thread too: one thread turning fusion off for a call must not change what another thread's open scope
// records).
static int& lazy_override() {
  static thread_local int v = -1;
  return v;
}
static bool lazy_enabled() {
  static const int dflt = [] { const char* e = getenv("TOPS_LAZY"); return e ? atoi(e) : 1; }();
  const int o = lazy_override();
  return (o >= 0 ? o : dflt) != 0;
}
int lazy_set(int on) {
  const int prev = lazy_enabled() ? 1 : 0;
  lazy_override() = on ? 1 : 0;
  return prev;
}

bool lazy_active() { return lazy_enabled() && scope().depth > 0; }

to_tensor memo_find(const MemoKey& key) {
  Scope& s = scope();
  if (s.depth == 0) return nullptr;
  auto it = s.memo.find(key);
  if (it == s.memo.end()) return nullptr;
  retain(it->second);
  return it->second;
}

void memo_put(const MemoKey& key, to_tensor t) {
  Scope& s = scope();
  if (s.depth == 0) return;
  auto it = s.memo.find(key);
  if (it != s.memo.end()) release_int(it->second);
  retain_int(t);
  s.memo[key] = t;
}

static void memo_clear(Scope& s) {
  std::unordered_map<MemoKey, to_tensor, MemoKeyHash> m;
  m.swap(s.memo);
  for (auto& kv : m) release_int(kv.second);
}

ScopeOwner::~ScopeOwner() {
  std::lock_guard<std::recursive_mutex> guard(lock());
  memo_clear(*s);
  s->depth = 0;
}

void scope_begin() {
  Scope& s = scope();
  if (s.depth == 0) {
    auto& all = all_scopes();
    if (std::find(all.begin(), all.end(), &s) == all.end()) all.push_back(&s);
  }
  ++s.depth;
}

void scope_end() {
  Scope& s = scope();
  TO_CHECK(s.depth > 0, TO_ERR_STATE, "to_memo_end without to_memo_begin");
  // Closing a scope demands nothing.  A handle the host still holds stays deferred and is produced if and when
  // it is asked for: under a garbage collector "still held" says nothing about "still wanted" -- every intermediate
  // of a step is reachable from the Haskell heap until the finaliser of its ForeignPtr has run, and launching those
  // here would re-run the whole step unfused.
  if (--s.depth == 0) memo_clear(s);
}

void scope_reset_all() {
  for (Scope* s : all_scopes()) {
    memo_clear(*s);
    s->depth = 0;
  }
  // handles that are still deferred keep their nodes; they die with their handles
}

to_tensor lazy_record(const NodeDesc& d, int n_in, const to_tensor* in, int rank, const int64_t* dims,
                      int64_t batch, int dtype) {
  to_tensor t = new_deferred(rank, dims, batch, dtype);
  auto* n = new Node();
  n->d = d;
  if (n->d.f) expr_retain(n->d.f);
  n->seq = ++g_seq;
  n->owner = this_thread();
  n->in.assign(in, in + n_in);
  for (to_tensor x : n->in) retain_int(x);
  n->out = t;
  n->next = g_head;
  if (g_head) g_head->prev = n;
  g_head = n;
  t->node = n;
  g_stats[0]++;
  return t;
}

bool lazy_node_of(to_tensor t, NodeDesc* d, std::vector<to_tensor>* in) {
  if (t->ptr || t->view_base || !t->node) return false;
  *d = t->node->d;
  *in = t->node->in;
  return true;
}

// ---- handles and memory ---------------------------------------------------------------------------------------------
// the pending node a handle's value comes from (nullptr: the value exists; a pending view gets resolved here)
static void resolve_view(to_tensor t) {
  to_tensor_s* b = t->view_base;
  if (!b || !b->ptr) return;
  t->buf = b->buf;
  if (t->buf) t->buf->refs.fetch_add(1);
  t->ptr = b->at(t->view_off);
  t->view_base = nullptr;
  for (size_t i = 0; i < b->dviews.size(); ++i)
    if (b->dviews[i] == t) {
      b->dviews[i] = b->dviews.back();
      b->dviews.pop_back();
      break;
    }
  release_int(b);
}

static Node* producer(to_tensor t) {
  if (t->ptr) return nullptr;
  if (t->view_base) {
    if (t->view_base->ptr) {
      resolve_view(t);
      return nullptr;
    }
    TO_CHECK(t->view_base->node != nullptr, TO_ERR_STATE, "deferred view of a handle without a recorded op");
    return t->view_base->node;
  }
  TO_CHECK(t->node != nullptr, TO_ERR_STATE, "handle has neither storage nor a recorded op");
  return t->node;
}

static bool is_live(to_tensor t) { return t->refs.load() > t->int_refs; }
// can the host still ask for this value -- through the handle, or through a view of it that it holds?
static bool host_reachable(to_tensor t) {
  if (is_live(t)) return true;
  for (to_tensor_s* v : t->dviews)
    if (is_live(v)) return true;
  return false;
}

// does `in` denote exactly the value of handle h, element for element in the same order?
static bool same_value_layout(to_tensor in, to_tensor h) {
  if (in == h) return true;
  if (in->view_base != h || in->view_off != 0 || in->rank != h->rank || in->batch != h->batch) return false;
  for (int i = 0; i < h->rank; ++i)
    if (in->dims[i] != h->dims[i] || (h->dims[i] != 1 && in->strides[i] != h->strides[i])) return false;
  return h->batch <= 1 || in->bstride == h->bstride;
}

// byte range a materialised handle can touch
static void mem_range(to_tensor t, const char** lo, const char** hi) {
  int64_t last = 0;
  for (int i = 0; i < t->rank; ++i)
    if (t->dims[i] > 0) last += (t->dims[i] - 1) * t->strides[i];
  if (t->batch > 1) last += (t->batch - 1) * t->bstride;
  *lo = static_cast<const char*>(t->ptr);
  *hi = *lo + (last + 1) * (int64_t)t->esize();
  if (t->total() == 0) *hi = *lo;
}
static bool overlaps(to_tensor a, to_tensor b) {
  if (!a->ptr || !b->ptr) return false;
  const char *al, *ah, *bl, *bh;
  mem_range(a, &al, &ah);
  mem_range(b, &bl, &bh);
  return al < bh && bl < ah;
}

// ---- tiny host tensors: the planner evaluates candidate loss heads on them -------------------------------------------
struct HT {
  int rank = 0;
  int64_t dims[TO_MAX_RANK] = {0};
  int64_t batch = 0;
  std::vector<double> v;
  int64_t numel() const {
    int64_t n = 1;
    for (int i = 0; i < rank; ++i) n *= dims[i];
    return n;
  }
  double at(int64_t b, int64_t e) const { return v[(size_t)((batch > 0 ? b : 0) * numel() + e)]; }
};
static HT ht_like(to_tensor t, int64_t B) {
  HT h;
  h.rank = t->rank;
  for (int i = 0; i < t->rank; ++i) h.dims[i] = t->dims[i];
  h.batch = t->batch > 0 ? B : 0;
  h.v.assign((size_t)(h.numel() * (h.batch > 0 ? h.batch : 1)), 0.0);
  return h;
}

// ---- the plan of one flush ---------------------------------------------------------------------------------------------
struct PN {
  Node* n = nullptr;
  to_tensor h = nullptr;
  std::vector<int> prod;  // per input: producing PN, -1 = the value exists
  std::vector<int> cons;  // distinct consuming PNs
  bool demanded = false;  // must exist in storage of its own after the flush
  to_tensor copy_dst = nullptr;  // to_copy_into destination (a root that need not have storage of its own)
  int group = -1;
  bool is_const = false;  // every element equals cval (FILL and what is computed from FILLs alone)
  double cval = 0.0;
  bool stored = false;    // storage was produced by this flush
  bool fwd = false;       // planned: produce it straight into copy_dst
  bool copied = false;    // ... and that happened
};

// "input `in` of plan node `node`": how a group names an operand so that a cached plan can be re-bound to new handles
struct Ref {
  int node = -1, in = -1;
};

struct Gr {
  bool gemm = false;
  std::vector<int> mem;  // members in recording order; those that are no output are never stored
  int anchor = -1, out = -1;
  double alpha = 1.0, beta = 0.0;
  to_tensor cin = nullptr, bias = nullptr, dact = nullptr;
  Ref r_cin, r_bias, r_dact, r_rs_in, r_target, r_tail_w, r_tail_h;  // where the operand pointers of this group came from
  int act = 0;       // 1: logistic, 2: tanh
  int dact_kind = 0; // 0: * h (1 - h), 1: * (1 - h^2)
  int rs = -1;  // PN receiving the row sums of the A operand (batch_sum(dz), or p_b - r * that)
  to_tensor rs_in = nullptr;
  double rs_alpha = 1.0;
  int loss_kind = 0, loss_node = -1;
  to_tensor target = nullptr;
  int tail = -1;  // PN receiving (dz . W) * h(1-h)
  to_tensor tail_w = nullptr, tail_h = nullptr;
  bool wgrad_like = false;
  int pair = -1;          // the other weight-gradient group launched together with this one
  int r1 = -1;            // leader of the rank-1 unit this group belongs to (K = 1 weight gradients of a one-sample
                          // step: all layers' outer-product updates in ONE launch)
  std::vector<int> r1_members;  // on the leader
  std::vector<int> deps;  // groups whose outputs (or whose reads of a forwarding destination) come first
  bool done = false;
  // a row program (rowprog.cpp): the members are a row-local subgraph hanging off node rp_root, run as ONE compiled kernel
  std::shared_ptr<RowProg> rowprog;
  int rp_root = -1;
  std::vector<int> rp_outs;       // plan nodes whose values the program writes back (parallel to rowprog->outs)
  std::vector<Ref> rp_ext_ref;    // where the existing tensors it reads came from (a cached plan re-binds them) ...
  std::vector<to_tensor> rp_ext;  // ... and the tensors themselves
};

struct Plan {
  std::vector<PN> ns;
  uint64_t epoch = 0;
  std::vector<std::vector<uint64_t>> anc;  // ancestor bitsets (over ns)
  std::vector<Gr> gs;
  int words = 0;
  bool is_anc(int a, int of) const { return (anc[of][a >> 6] >> (a & 63)) & 1; }
};

static int pn_of(Plan& pl, to_tensor t) {
  Node* p = producer(t);
  return p && p->plan_epoch == pl.epoch ? p->plan_idx : -1;
}

// would adding a node with these inputs to a group with these members close a cycle through other groups?
// (an outside input that descends from a member would have to run both after and before the group)
static bool inputs_clear_of(const Plan& pl, const std::vector<int>& members, int cand) {
  for (int q : pl.ns[cand].prod) {
    if (q < 0) continue;
    if (std::find(members.begin(), members.end(), q) != members.end()) continue;
    for (int m : members)
      if (m == q || pl.is_anc(m, q)) return false;
  }
  return true;
}

static bool sole_consumer(const Plan& pl, int i) {
  return pl.ns[i].cons.size() == 1 && !pl.ns[i].demanded && !pl.ns[i].copy_dst;
}

static bool full_like(to_tensor x, to_tensor like) {  // same per-sample shape AND same batch, contiguous
  return same_shape(x, like) && x->batch == like->batch && x->dtype == like->dtype && x->contiguous();
}

// ---- loss-head recognition ---------------------------------------------------------------------------------------------
static bool ht_eval_node(const Plan& pl, int i, const std::unordered_map<int, HT>& env, to_tensor target,
                         const HT& target_val, int64_t B, HT* out) {
  const PN& pn = pl.ns[i];
  const Node* n = pn.n;
  std::vector<HT> tmp;
  tmp.reserve(n->in.size());
  std::vector<const HT*> xs;
  for (size_t k = 0; k < n->in.size(); ++k) {
    const int q = pn.prod[k];
    if (q >= 0) {
      if (pl.ns[q].is_const) {
        HT c = ht_like(n->in[k], B);
        std::fill(c.v.begin(), c.v.end(), pl.ns[q].cval);
        tmp.push_back(std::move(c));
        xs.push_back(nullptr);  // fixed up below (tmp may reallocate)
        continue;
      }
      auto it = env.find(q);
      if (it == env.end()) return false;
      xs.push_back(&it->second);
    } else {
      if (!target || n->in[k]->ptr != target->ptr) return false;
      xs.push_back(&target_val);
    }
  }
  {
    size_t t = 0;
    for (size_t k = 0; k < xs.size(); ++k)
      if (!xs[k]) xs[k] = &tmp[t++];
  }
  HT r = ht_like(pn.h, B);
  const int64_t ne = r.numel(), nb = r.batch > 0 ? r.batch : 1;
  switch (n->d.op) {
    case N_LIFT: {
      double x[8];
      for (int64_t b = 0; b < nb; ++b)
        for (int64_t e = 0; e < ne; ++e) {
          for (size_t k = 0; k < xs.size(); ++k) x[k] = xs[k]->at(b, e);
          r.v[(size_t)(b * ne + e)] = expr_eval(*n->d.f, x);
        }
      break;
    }
    case N_DACT:
      for (int64_t b = 0; b < nb; ++b)
        for (int64_t e = 0; e < ne; ++e) {
          const double d = xs[0]->at(b, e), h = xs[1]->at(b, e);
          r.v[(size_t)(b * ne + e)] = n->d.lm ? d * (1.0 - h * h) : d * h * (1.0 - h);
        }
      break;
    case N_SUM:
      for (int64_t b = 0; b < nb; ++b)
        for (int64_t e = 0; e < ne; ++e) {
          double a = 0.0;
          for (const HT* x : xs) a += x->at(b, e);
          r.v[(size_t)(b * ne + e)] = a;
        }
      break;
    case N_SCALE:
      for (int64_t b = 0; b < nb; ++b)
        for (int64_t e = 0; e < ne; ++e) r.v[(size_t)(b * ne + e)] = n->d.alpha * xs[0]->at(b, e);
      break;
    case N_SUM_ROWS: {
      const int64_t R = xs[0]->dims[0];
      for (int64_t b = 0; b < nb; ++b)
        for (int64_t e = 0; e < ne; ++e) {
          double a = 0.0;
          for (int64_t q = 0; q < R; ++q) a += xs[0]->at(b, q * ne + e);
          r.v[(size_t)(b * ne + e)] = a;
        }
      break;
    }
    case N_MAP_ROWS: {
      const int64_t J = xs[0]->numel();
      for (int64_t b = 0; b < nb; ++b)
        for (int64_t e = 0; e < ne; ++e) r.v[(size_t)(b * ne + e)] = xs[0]->at(b, J ? e % J : 0);
      break;
    }
    case N_GMUL: {
      if (n->d.reduce || n->d.lo > 1) return false;
      int64_t M = 1, K = 1, N = 1;
      for (int k = 0; k < n->d.lm; ++k) M *= xs[0]->dims[k];
      for (int k = 0; k < n->d.lo; ++k) K *= xs[0]->dims[n->d.lm + k];
      for (int k = 0; k < n->d.ln; ++k) N *= xs[1]->dims[n->d.lo + k];
      for (int64_t b = 0; b < nb; ++b)
        for (int64_t m = 0; m < M; ++m)
          for (int64_t c = 0; c < N; ++c) {
            double a = 0.0;
            for (int64_t k = 0; k < K; ++k) a += xs[0]->at(b, m * K + k) * xs[1]->at(b, k * N + c);
            r.v[(size_t)(b * ne + m * N + c)] = a;
          }
      break;
    }
    default: return false;
  }
  *out = std::move(r);
  return true;
}

static bool ht_close(double a, double b) {
  if (!std::isfinite(a) || !std::isfinite(b)) return false;
  return std::fabs(a - b) <= 1e-9 * (1.0 + std::fabs(a) + std::fabs(b));
}

// The row-local subgraph hanging off `root` ([B x N], the result of gmul + bias): if the only thing the rest of
// the graph needs from it is dz [B x N] (and possibly a per-row loss) and dz(z, t) is one of the two closed forms
// the small-GEMM kernel's loss head computes, fill in the group.  Probabilistic identity testing, as for
// closures (expr.cpp): only smooth programs are considered, so agreement on random rows means identity.
// The recognition is an identity TEST (three scales of logits, three rows each, 1e-9), not a proof: a false positive would be
// a silently wrong gradient.  A host that would rather pay the launches can turn it off: TOPS_LOSS_HEAD_MATCH=0 for the
// process, to_set_loss_head_match for what is planned from now on (the state is part of a plan's signature, so a cached plan
// made under the other setting is not reused).  Off, the same subgraph runs as a row program or op by op.
static int g_loss_head_match = -1;
static bool loss_head_match_on() {
  if (g_loss_head_match < 0) {
    const char* e = getenv("TOPS_LOSS_HEAD_MATCH");
    g_loss_head_match = !(e && e[0] == '0');
  }
  return g_loss_head_match != 0;
}
int lazy_set_loss_head_match(int on) {
  const int prev = loss_head_match_on() ? 1 : 0;
  g_loss_head_match = on ? 1 : 0;
  return prev;
}

static bool match_loss_head(Plan& pl, Gr& g, int root) {
  if (!loss_head_match_on()) return false;
  to_tensor rh = pl.ns[root].h;
  // (batched: one row per sample; unbatched: the single row of a per-sample step)
  if (rh->rank != 1 || rh->dims[0] < 1 || rh->dims[0] > 16) return false;
  const int64_t N = rh->dims[0], Bfull = rh->batch;
  std::vector<int> S{root}, K;  // members, constants they use
  std::vector<char> inS(pl.ns.size(), 0);
  inS[root] = 1;
  to_tensor target = nullptr;
  Ref target_ref;
  for (size_t i = (size_t)root + 1; i < pl.ns.size(); ++i) {
    PN& pn = pl.ns[i];
    if (pn.group >= 0 || pn.is_const) continue;
    const Node* n = pn.n;
    const int op = n->d.op;
    if (!(op == N_LIFT || op == N_DACT || op == N_SUM || op == N_SCALE || op == N_SUM_ROWS || op == N_MAP_ROWS ||
          (op == N_GMUL && !n->d.reduce && n->d.lo <= 1)))
      continue;
    if (op == N_LIFT && !expr_is_smooth(*n->d.f)) continue;
    // the result: one row (or one number) per sample
    if (pn.h->batch != Bfull || pn.h->rank > 1 || (pn.h->rank == 1 && pn.h->dims[0] != N)) continue;
    bool any_in = false, ok = true;  // any_in: reads a member or the target rows
    to_tensor tgt = target;
    Ref tgt_ref = target_ref;
    for (size_t k = 0; k < n->in.size() && ok; ++k) {
      to_tensor x = n->in[k];
      const int q = pn.prod[k];
      if (q >= 0) {
        if (inS[q]) {
          any_in = true;
          ok = same_value_layout(x, pl.ns[q].h);
        } else if (pl.ns[q].is_const) {
          ok = x->rank <= 1;
        } else {
          ok = false;
        }
      } else {
        // an existing value: the target rows (one operand only)
        ok = x->batch == Bfull && x->rank == 1 && x->dims[0] == N && x->contiguous() && x->dtype == rh->dtype &&
             (!tgt || tgt->ptr == x->ptr);
        if (ok) {
          if (!tgt) tgt_ref = Ref{(int)i, (int)k};
          tgt = x;
          any_in = true;
        }
      }
    }
    if (!ok || !any_in) continue;
    target = tgt;
    target_ref = tgt_ref;
    inS[i] = 1;
    S.push_back((int)i);
  }
  if (S.size() < 2 || !target) return false;
  // what the rest of the graph reads from S
  int dz = -1, loss = -1;
  for (int i : S) {
    const PN& pn = pl.ns[i];
    bool outside = pn.demanded || pn.copy_dst;
    for (int c : pn.cons)
      if (!inS[c]) outside = true;
    if (!outside) continue;
    if (pn.h->rank == 1 && dz < 0 && i != root) dz = i;
    else if (pn.h->rank == 0 && loss < 0) loss = i;
    else return false;
  }
  if (dz < 0) return false;
  // Evaluate on random rows at three scales -- logits in (-2, 2), (-6, 6) and (-0.1, 0.1), three rows each (one each
  // for the single row of an unbatched step).  Members are compositions of +, *, /, exp, log, tanh ... (expr_is_smooth:
  // nothing piecewise), i.e. real-analytic in (z, t) on the connected domain where they are defined, and so are the
  // closed forms: two analytic maps that agree on a set with an accumulation point are the same map, and a map that is
  // NOT the closed form differs from it everywhere except on a set of measure zero -- points in general position at
  // three scales do not lie on it.  What can slip through is a program within 1e-9 relative of the closed form at every
  // scale, whose gradient is then wrong by that much.
  const int64_t B = 3;
  struct Lcg {
    uint64_t s = 0x7e500002ull;
    double next() {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      return ((s >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    }
  } rng;
  int kind = 0;
  const double half_width[3] = {2.0, 6.0, 0.1};
  for (int trial = 0; trial < 3; ++trial) {
    HT z = ht_like(rh, B), t = ht_like(target, B);
    for (double& x : z.v) x = half_width[trial] * (2.0 * rng.next() - 1.0);
    for (double& x : t.v) x = 0.05 + rng.next();
    std::unordered_map<int, HT> env;
    env[root] = z;
    for (size_t k = 1; k < S.size(); ++k) {
      HT r;
      if (!ht_eval_node(pl, S[k], env, target, t, B, &r)) return false;
      env[S[k]] = std::move(r);
    }
    const HT& got = env[dz];
    int kind_here = 0;
    for (int cand = 1; cand <= 2 && !kind_here; ++cand) {
      bool ok = true;
      for (int64_t b = 0; b < B && ok; ++b) {
        double se = 0.0, sy = 0.0, mx = -1e300, lossv = 0.0;
        for (int64_t j = 0; j < N; ++j) mx = std::max(mx, z.at(b, j));
        for (int64_t j = 0; j < N; ++j) {
          se += std::exp(z.at(b, j) - mx);
          sy += t.at(b, j);
        }
        for (int64_t j = 0; j < N && ok; ++j) {
          double want;
          if (cand == 1) {
            const double pr = std::exp(z.at(b, j) - mx) / se;
            want = pr * sy - t.at(b, j);
            lossv += -t.at(b, j) * std::log(pr);
          } else {
            const double s = 1.0 / (1.0 + std::exp(-z.at(b, j))), e = t.at(b, j) - s;
            want = -2.0 * e * s * (1.0 - s);
            lossv += e * e;
          }
          ok = ht_close(got.at(b, j), want);
        }
        if (ok && loss >= 0) ok = ht_close(env[loss].at(b, 0), lossv);
      }
      if (ok) kind_here = cand;
    }
    if (!kind_here || (trial > 0 && kind_here != kind)) return false;
    kind = kind_here;
  }
  if (!kind) return false;
  // constants used only inside S ride along (never stored when the head is fused)
  for (int i : S)
    for (int q : pl.ns[i].prod)
      if (q >= 0 && pl.ns[q].is_const && pl.ns[q].group < 0 && !pl.ns[q].demanded && !pl.ns[q].copy_dst) {
        bool all_in = true;
        for (int c : pl.ns[q].cons)
          if (!inS[c]) all_in = false;
        if (all_in && std::find(K.begin(), K.end(), q) == K.end()) K.push_back(q);
      }
  // (constants of constants: one more level covers `negate` of the seed)
  for (size_t k = 0; k < K.size(); ++k)
    for (int q : pl.ns[K[k]].prod)
      if (q >= 0 && pl.ns[q].is_const && pl.ns[q].group < 0 && !pl.ns[q].demanded && !pl.ns[q].copy_dst) {
        bool all_in = true;
        for (int c : pl.ns[q].cons)
          if (!inS[c] && std::find(K.begin(), K.end(), c) == K.end()) all_in = false;
        if (all_in && std::find(K.begin(), K.end(), q) == K.end()) K.push_back(q);
      }
  for (size_t k = 1; k < S.size(); ++k) g.mem.push_back(S[k]);
  for (int q : K) g.mem.push_back(q);
  g.loss_kind = kind;
  g.target = target;
  g.r_target = target_ref;
  g.loss_node = loss;
  g.out = dz;
  return true;
}

// ---- grouping -----------------------------------------------------------------------------------------------------------
static void dry_plan(const Node* n, GmulPlan& gp) {
  gmul_plan(gp, n->d.lm, n->d.lo, n->d.ln, n->in[0], n->in[1], n->d.reduce, true);
}

// `d * logistic'(z)` (EW_MUL_DLOGISTIC on [d, z]) where h = logistic(z) is part of the graph: consume h instead.
// The forward value is always there (the next layer needed it), and z = gmul + bias then has one consumer less --
// which is what lets it stay inside the GEMM launch.
static void apply_dlogistic(Plan& pl, int i, int c);
static void rewrite_dlogistic(Plan& pl, std::vector<std::pair<int, int>>& done) {
  for (size_t i = 0; i < pl.ns.size(); ++i) {
    Node* n = pl.ns[i].n;
    if (n->d.op != N_LIFT || (n->d.f->kind != EW_MUL_DLOGISTIC && n->d.f->kind != EW_MUL_DTANH) || n->in.size() != 2) continue;
    const int fwd_kind = n->d.f->kind == EW_MUL_DTANH ? EW_TANH : EW_LOGISTIC;
    const int zq = pl.ns[i].prod[1];
    if (zq < 0) continue;
    to_tensor z = n->in[1];
    if (!same_value_layout(z, pl.ns[zq].h)) continue;
    for (int c : pl.ns[zq].cons) {
      Node* m = pl.ns[c].n;
      if ((size_t)c == i || m->d.op != N_LIFT || m->d.f->kind != fwd_kind || !same_value_layout(m->in[0], pl.ns[zq].h))
        continue;
      if (!full_like(pl.ns[c].h, pl.ns[i].h)) continue;
      apply_dlogistic(pl, (int)i, c);  // rewrite in place: the node now reads h
      done.emplace_back((int)i, c);
      break;
    }
  }
}

static void form_gemm_group(Plan& pl, int a) {
  PN& an = pl.ns[a];
  Node* n = an.n;
  GmulPlan gp;
  dry_plan(n, gp);
  Gr g;
  g.gemm = true;
  g.anchor = a;
  g.mem.push_back(a);
  int cur = a;
  const bool plain_layout = gp.exact && !gp.zero && gp.p.batch == 1 && !gp.p.reduce_batch;
  // C as the kernels see it: [p.M x p.N] row-major in the result's own storage
  const GemmProblem& p = gp.p;
  int stage = 0;  // 0 linear part, 1 bias added, 2 activation applied, 3 dact applied
  // the loss head (and its tail) behind a bare gmul + bias whose rows fit one 16-lane group
  bool head_done = false;
  auto try_loss_head = [&]() {
    if (head_done || !(plain_layout && gp.rows_are_samples && stage <= 1 && !g.cin && g.beta == 0.0 && p.N <= 16)) return false;
    GemmProblem q = p;
    q.alpha = g.alpha;
    if (!gemm_small_fuses_loss(q)) return false;
    Gr trial = g;
    if (!match_loss_head(pl, trial, cur)) return false;
    g = trial;
    head_done = true;
    // tail: T = (gmul W^T dz) * h (1 - h), one 16-row block of T per workgroup of the same launch
    const int dz = g.out;
    for (int c : pl.ns[dz].cons) {
      PN& cn = pl.ns[c];
      Node* m = cn.n;
      if (cn.group >= 0 || m->d.op != N_GMUL || m->d.reduce || !sole_consumer(pl, c)) continue;
      if (cn.prod[1] != dz || !same_value_layout(m->in[1], pl.ns[dz].h) || cn.prod[0] >= 0) continue;
      const int tq = cn.cons[0];
      PN& tn = pl.ns[tq];
      if (tn.group >= 0 || tn.n->d.op != N_DACT || tn.n->d.lm != 0 || tn.prod[0] != c || !same_value_layout(tn.n->in[0], cn.h)) continue;
      if (!full_like(tn.n->in[1], tn.h) || tn.h->rank != 1 || tn.h->batch != pl.ns[dz].h->batch) continue;
      GmulPlan tp;
      dry_plan(m, tp);
      const int64_t tail_n = tn.h->dims[0];
      if (!tp.exact || tp.zero || !tp.rows_are_samples || tp.p.batch != 1 || tp.p.K != p.N || tp.p.N != tail_n ||
          tp.p.b_sk != tail_n || tp.p.b_sn != 1 || tp.p.a_sk != 1 || tp.p.a_sm != p.N)
        continue;
      if (!gemm_small_fuses_tail(q, tail_n)) continue;
      std::vector<int> with = g.mem;
      with.push_back(c);
      if (!inputs_clear_of(pl, with, tq) || !inputs_clear_of(pl, g.mem, c)) continue;
      g.mem.push_back(c);
      g.mem.push_back(tq);
      g.tail = tq;
      g.tail_w = m->in[0];
      g.tail_h = tn.n->in[1];
      g.r_tail_w = Ref{c, 0};
      g.r_tail_h = Ref{tq, 1};
      break;
    }
    return true;
  };
  while (plain_layout && sole_consumer(pl, cur) && !head_done) {
    const int c = pl.ns[cur].cons[0];
    PN& cn = pl.ns[c];
    if (cn.group >= 0) break;
    Node* m = cn.n;
    // which input is the running value?
    int pos = -1;
    for (size_t k = 0; k < m->in.size(); ++k)
      if (cn.prod[k] == cur && same_value_layout(m->in[k], pl.ns[cur].h)) pos = (int)k;
    if (pos < 0) break;
    int uses = 0;
    for (int q : cn.prod) uses += q == cur;
    if (uses != 1) break;
    if (!full_like(cn.h, pl.ns[cur].h)) break;
    if (!inputs_clear_of(pl, g.mem, c)) break;
    bool took = false;
    const int op = m->d.op;
    if (op == N_SCALE && stage == 0) {
      g.alpha *= m->d.alpha;
      g.beta *= m->d.alpha;
      took = true;
    } else if ((op == N_SUM && m->in.size() == 2) ||
               (op == N_LIFT && m->d.f->kind == EW_AFFINE && m->in.size() == 2 && m->d.f->c0_d == 0.0)) {
      to_tensor other = m->in[1 - pos];
      double ca = 1.0, co = 1.0;  // coefficients of the running value / of the other operand
      if (op == N_LIFT) {
        ca = m->d.f->coef_d[pos];
        co = m->d.f->coef_d[1 - pos];
      }
      const bool bias_like = gp.rows_are_samples && other->batch == 0 && other->rank == 1 && cn.h->rank == 1 &&
                             other->dims[0] == p.N && other->contiguous() && co == 1.0;
      if (stage == 0 && bias_like && ca != 0.0) {
        g.alpha *= ca;
        g.beta *= ca;
        g.bias = other;
        g.r_bias = Ref{c, 1 - pos};
        stage = 1;
        took = true;
      } else if (stage == 0 && !g.cin && full_like(other, cn.h) && ca != 0.0) {
        g.alpha *= ca;
        g.cin = other;
        g.r_cin = Ref{c, 1 - pos};
        g.beta = co;
        took = true;
      }
    } else if (op == N_LIFT && m->d.f->kind == EW_AFFINE && m->in.size() == 1 && m->d.f->c0_d == 0.0 && stage == 0) {
      g.alpha *= m->d.f->coef_d[0];
      g.beta *= m->d.f->coef_d[0];
      took = true;
    } else if (op == N_LIFT && m->d.f->kind == EW_LOGISTIC && stage <= 1) {
      // (an output layer: logistic >>> squaredError's backward reads this value -- the loss head takes all of it)
      if (try_loss_head()) break;
      g.act = 1;
      stage = 2;
      took = true;
    } else if (op == N_LIFT && m->d.f->kind == EW_TANH && stage <= 1) {
      g.act = 2;
      stage = 2;
      took = true;
    } else if (op == N_DACT && pos == 0 && stage <= 1 && full_like(m->in[1], cn.h)) {
      g.dact = m->in[1];
      g.dact_kind = m->d.lm;
      g.r_dact = Ref{c, 1};
      stage = 3;
      took = true;
    }
    if (!took) break;
    g.mem.push_back(c);
    cur = c;
  }
  if (!head_done) {
    g.out = cur;
    try_loss_head();
  }
  // the weight-gradient form dW = sum_b dz_b (x) a_b = dZ^T A next to db = sum_b dz_b: the row sums of the
  // A operand come out of the same launch
  const bool outer1 = !n->d.reduce && n->d.lm == 1 && n->d.lo == 0 && n->d.ln == 1 && n->in[0]->batch == 0 &&
                      n->in[1]->batch == 0;  // dz (x) a of a per-sample step
  if (plain_layout && (n->d.reduce || outer1) && !g.loss_kind && g.act == 0 && !g.dact && !g.bias) {
    g.wgrad_like = true;
    to_tensor dzh = n->in[0];
    // The row sums ride along in the small-GEMM kernel only.  A weight gradient beyond its range (a 4096 -> 4096 layer: dW is
    // 4096 x B x 4096) keeps its own epilogue -- W - r dW as alpha A B + beta Cin, produced in place -- and leaves the bias
    // gradient to a launch of its own; with the sibling attached the whole group used to fall apart into GEMM, update, sum,
    // update and a copy of W (tools/step_scan.py: 4096-4096-10 at 32 rows 193 us a step, torch 125).
    const bool rs_rides = gemm_small_route(p);
    if (rs_rides && outer1 && an.prod[0] >= 0 && p.K == 1) {
      // the bias update of the same layer reads dz itself (no batch to sum over): b - r*dz rides along as the
      // "row sums" of the one-column A operand
      const int dq = an.prod[0];
      for (int c : pl.ns[dq].cons) {
        PN& cn = pl.ns[c];
        Node* m = cn.n;
        if (g.rs >= 0 || cn.group >= 0 || c == a || m->d.op != N_LIFT || m->d.f->kind != EW_AFFINE || m->in.size() != 2 ||
            m->d.f->c0_d != 0.0)
          continue;
        int pos = -1;
        for (int k = 0; k < 2; ++k)
          if (cn.prod[k] == dq && same_value_layout(m->in[k], pl.ns[dq].h) && same_value_layout(dzh, pl.ns[dq].h)) pos = k;
        if (pos < 0 || cn.prod[1 - pos] == dq || m->d.f->coef_d[1 - pos] != 1.0 || !full_like(m->in[1 - pos], cn.h) ||
            !full_like(cn.h, pl.ns[dq].h) || m->d.f->coef_d[pos] != g.alpha)
          continue;
        if (!inputs_clear_of(pl, g.mem, c)) continue;
        bool indep = true;
        for (int mm : g.mem)
          if (pl.is_anc(mm, c) || pl.is_anc(c, mm)) indep = false;
        if (!indep) continue;
        g.mem.push_back(c);
        g.rs = c;
        g.rs_in = m->in[1 - pos];
        g.r_rs_in = Ref{c, 1 - pos};
        g.rs_alpha = m->d.f->coef_d[pos];
      }
    }
    if (rs_rides && dzh->batch > 0 && dzh->rank == 1 && n->d.lm == 1 && n->d.lo == 0 && (p.a_sm == 1 || p.M == 1) &&
        (p.a_sk == p.M || p.K == 1) && p.K == dzh->batch) {
      const int dq = an.prod[0];
      // siblings: batch_sum of the same value
      auto try_sibling = [&](int r) {
        PN& rn = pl.ns[r];
        if (rn.group >= 0 || rn.n->d.op != N_BATCH_SUM || r == a) return false;
        to_tensor x = rn.n->in[0];
        if (!(x == dzh || (x->ptr && x->ptr == dzh->ptr && full_like(x, dzh)) ||
              (dq >= 0 && rn.prod[0] == dq && same_value_layout(x, pl.ns[dq].h) && same_value_layout(dzh, pl.ns[dq].h))))
          return false;
        std::vector<int> with = g.mem;
        if (!inputs_clear_of(pl, with, r)) return false;
        for (int mm : g.mem)
          if (pl.is_anc(mm, r) || pl.is_anc(r, mm)) return false;
        g.mem.push_back(r);
        g.rs = r;
        // p_b - rate * db: the bias update in the same epilogue
        if (sole_consumer(pl, r)) {
          const int c = rn.cons[0];
          PN& cn = pl.ns[c];
          Node* m = cn.n;
          if (cn.group < 0 && m->d.op == N_LIFT && m->d.f->kind == EW_AFFINE && m->in.size() == 2 && m->d.f->c0_d == 0.0) {
            int pos = -1;
            for (int k = 0; k < 2; ++k)
              if (cn.prod[k] == r && same_value_layout(m->in[k], rn.h)) pos = k;
            if (pos >= 0 && cn.prod[1 - pos] != r && m->d.f->coef_d[1 - pos] == 1.0 && full_like(m->in[1 - pos], rn.h) &&
                full_like(cn.h, rn.h) && inputs_clear_of(pl, g.mem, c)) {
              g.mem.push_back(c);
              g.rs = c;
              g.rs_in = m->in[1 - pos];
              g.r_rs_in = Ref{c, 1 - pos};
              g.rs_alpha = m->d.f->coef_d[pos];
            }
          }
        }
        return true;
      };
      bool found = false;
      if (dq >= 0) {
        for (int r : pl.ns[dq].cons)
          if (!found && try_sibling(r)) found = true;
      } else {
        for (size_t r = 0; r < pl.ns.size() && !found; ++r)
          if (try_sibling((int)r)) found = true;
      }
    }
  }
  std::sort(g.mem.begin(), g.mem.end());
  const int gi = (int)pl.gs.size();
  for (int m : g.mem) pl.ns[m].group = gi;
  pl.gs.push_back(std::move(g));
}

// ---- TOPS_LAZY_DEBUG=1: the plan of every flush on stderr -----------------------------------------------------------------
static bool debug_on() {
  static const int on = [] { const char* e = getenv("TOPS_LAZY_DEBUG"); return e ? atoi(e) : 0; }();
  return on != 0;
}
static const char* op_name(int op) {
  switch (op) {
    case N_GMUL: return "gmul";
    case N_LIFT: return "lift";
    case N_SUM: return "sum";
    case N_SCALE: return "scale";
    case N_SUM_ROWS: return "sumRows";
    case N_MAP_ROWS: return "mapRows";
    case N_BATCH_SUM: return "batchSum";
    case N_FILL: return "fill";
    case N_DACT: return "dact";
    case N_STACK: return "stack";
    default: return "?";
  }
}
static void dump_plan(const Plan& pl) {
  std::fprintf(stderr, "[lazy] flush: %zu nodes, %zu groups\n", pl.ns.size(), pl.gs.size());
  for (size_t i = 0; i < pl.ns.size(); ++i) {
    const PN& pn = pl.ns[i];
    std::fprintf(stderr, "  n%-3zu g%-3d %-8s%s %s <-", i, pn.group, op_name(pn.n->d.op),
                 pn.n->d.op == N_LIFT ? (" k" + std::to_string(pn.n->d.f->kind)).c_str() : "", shape_str(pn.h).c_str());
    for (size_t k = 0; k < pn.prod.size(); ++k) {
      if (pn.prod[k] >= 0) std::fprintf(stderr, " n%d", pn.prod[k]);
      else std::fprintf(stderr, " %s", shape_str(pn.n->in[k]).c_str());
    }
    std::fprintf(stderr, "  refs %d/%d v%zu%s%s%s%s\n", (int)pn.h->refs.load(), pn.h->int_refs, pn.h->dviews.size(), pn.demanded ? "  DEMANDED" : "", pn.copy_dst ? "  ->dst" : "", pn.fwd ? "(in place)" : "",
                 pn.is_const ? "  const" : "");
  }
  for (size_t gi = 0; gi < pl.gs.size(); ++gi) {
    const Gr& g = pl.gs[gi];
    if (g.rowprog) {
      std::fprintf(stderr, "  g%zu: row program off n%d, %zu ops, %zu existing tensors, outputs", gi, g.rp_root, g.rowprog->nodes.size(),
                   g.rp_ext.size());
      for (int o : g.rp_outs) std::fprintf(stderr, " n%d", o);
      std::fprintf(stderr, "\n");
      continue;
    }
    if (!g.gemm || g.mem.size() == 1) continue;
    std::fprintf(stderr, "  g%zu: gemm n%d out n%d alpha %g beta %g%s%s%s%s rs n%d loss %d tail n%d pair g%d deps", gi, g.anchor,
                 g.out, g.alpha, g.beta, g.cin ? " cin" : "", g.bias ? " bias" : "", g.act ? " act" : "", g.dact ? " dact" : "",
                 g.rs, g.loss_kind, g.tail, g.pair);
    for (int d : g.deps) std::fprintf(stderr, " g%d", d);
    std::fprintf(stderr, "\n");
  }
}

// ---- execution ---------------------------------------------------------------------------------------------------------
static std::vector<StepDesc>* g_describe = nullptr;
void lazy_describe_into(std::vector<StepDesc>* v) { g_describe = v; }
static void describe_gemm(const GemmProblem& p) {
  if (!g_describe) return;
  StepDesc d;
  d.kind = 0;
  d.p = p;
  g_describe->push_back(d);
}
static void describe_other() {
  if (!g_describe) return;
  StepDesc d;
  d.kind = 2;
  g_describe->push_back(d);
}

struct Exec {
  Plan& pl;
  std::vector<to_tensor> finish;  // handles whose value now exists: their nodes are dropped at the end
  const char* why = "";
  explicit Exec(Plan& p) : pl(p) {}

  void in_ready(const Node* n) {
    for (to_tensor x : n->in) {
      if (!x->ptr) resolve_view(x);
      TO_CHECK(x->ptr != nullptr, TO_ERR_STATE, "internal: input of a recorded op was not produced first");
    }
  }
  void stored(int i) {
    if (pl.ns[i].stored) return;
    pl.ns[i].stored = true;
    finish.push_back(pl.ns[i].h);
  }

  // one recorded op through the eager implementation
  void run_single(int i) {
    PN& pn = pl.ns[i];
    if (pn.stored || pn.h->ptr) return;
    Node* n = pn.n;
    in_ready(n);
    describe_other();
    Holder r;
    switch (n->d.op) {
      case N_GMUL: r.t = gmul_impl(n->d.lm, n->d.lo, n->d.ln, n->in[0], n->in[1], n->d.reduce); break;
      case N_LIFT: r.t = lift_impl(n->d.f, (int)n->in.size(), n->in.data(), 0, nullptr); break;
      case N_DACT: r.t = kind_impl(n->d.lm ? EW_MUL_1MH2 : EW_MUL_H1MH, 2, n->in.data()); break;
      case N_SUM: r.t = sum_impl((int)n->in.size(), n->in.data(), pn.h->rank, pn.h->dims, pn.h->dtype); break;
      case N_SCALE: r.t = affine_impl(1, n->in.data(), &n->d.alpha, 0.0); break;
      case N_SUM_ROWS: r.t = sum_rows_impl(n->in[0]); break;
      case N_MAP_ROWS: r.t = map_rows_const_impl(n->d.len_n, n->in[0], pn.h); break;
      case N_BATCH_SUM: r.t = batch_sum_impl(n->in[0]); break;
      case N_STACK: r.t = stack_impl(n->d.len_n, pn.h->dims, n->in.data()); break;
      case N_FILL:
        alloc_storage(pn.h);
        launch_fill(pn.h->dtype, pn.h->ptr, pn.h->total(), n->d.alpha, S());
        stored(i);
        return;
      default: fail(TO_ERR_STATE, "internal: unknown recorded op");
    }
    if (!r.t->contiguous() || r.t->batch != pn.h->batch) {
      // (sum of one operand / batch_sum of an unbatched value return their argument: give the handle the
      //  contiguous layout it promised)
      Holder c(contiguous(r.t));
      TO_CHECK(c.t->batch == pn.h->batch, TO_ERR_STATE, "internal: result batch differs from the recorded shape");
      adopt_storage(pn.h, c.t);
    } else {
      adopt_storage(pn.h, r.t);
    }
    stored(i);
  }

  void run_members(const Gr& g) {
    if (debug_on() && g.mem.size() > 1) std::fprintf(stderr, "[lazy] group of n%d: NOT fused (%s), running %zu ops one by one\n", g.anchor, why, g.mem.size());
    for (int m : g.mem) {
      pl.ns[m].fwd = false;
      run_single(m);
    }
  }

  struct Launch {
    GemmProblem p;
    GmulPlan gp;  // keeps packed operands alive
  };

  // build the fused problem of a GEMM group; false = the kernels cannot take it as planned
  bool build(const Gr& g, Launch& L) {
    const PN& an = pl.ns[g.anchor];
    Node* n = an.n;
    for (int m : g.mem)
      if (!pl.ns[m].is_const) {
        // inputs produced inside the group do not exist (that is the point); everything else must
        for (size_t k = 0; k < pl.ns[m].n->in.size(); ++k) {
          const int q = pl.ns[m].prod[k];
          if (q >= 0 && pl.ns[q].group == pl.ns[m].group && !pl.ns[q].stored) continue;
          to_tensor x = pl.ns[m].n->in[k];
          if (!x->ptr) resolve_view(x);
          TO_CHECK(x->ptr != nullptr, TO_ERR_STATE, "internal: input of a fused group was not produced first");
        }
      }
    // The real plan may LAUNCH (a pack of a non-collapsible operand, the pre-sum of a batch-reduced one).  An operand
    // produced by a launch that is still held back in `queue` has storage but no contents yet: issue the queue first.
    if (!queue.empty()) {
      GmulPlan dry;
      dry_plan(n, dry);
      bool from_queue = false;
      for (size_t k = 0; k < an.prod.size(); ++k) from_queue = from_queue || (an.prod[k] >= 0 && in_queue(an.prod[k]));
      if (!dry.exact && from_queue) drain();
    }
    gmul_plan(L.gp, n->d.lm, n->d.lo, n->d.ln, n->in[0], n->in[1], n->d.reduce, false);
    why = "empty contraction";
    if (L.gp.zero) return false;
    GemmProblem& p = L.p;
    p = L.gp.p;
    const bool epi = g.mem.size() > 1;
    why = "batched GEMM form";
    if (epi && (p.batch != 1 || p.reduce_batch)) return false;
    p.alpha = g.alpha;
    p.beta = g.cin ? g.beta : 0.0;
    p.Cin = g.cin ? g.cin->ptr : nullptr;
    p.bias = g.bias ? g.bias->ptr : nullptr;
    why = "bias does not run along the columns";
    if (g.bias && (p.N != g.bias->dims[0] || p.c_sm != p.N)) return false;
    p.act = g.act;
    p.dact = g.dact ? g.dact->ptr : nullptr;
    p.dact_kind = g.dact_kind;
    const bool needs_small = g.rs >= 0 || g.loss_kind != 0;
    why = "outside the small-GEMM range";
    if (needs_small && !gemm_small_route(p)) return false;
    why = "no kernel with a fused epilogue for this shape";
    if (epi && !needs_small && !gemm_epilogue_ok(p)) return false;
    if (g.loss_kind) {
      why = "loss head does not fit the kernel";
      if (!gemm_small_fuses_loss(p)) return false;
      p.loss_rows = g.loss_kind;
      p.target = g.target->ptr;
      if (g.tail >= 0) {
        const int64_t tn = pl.ns[g.tail].h->dims[0];
        if (!gemm_small_fuses_tail(p, tn)) return false;
        p.tail_w = g.tail_w->ptr;
        p.tail_h = g.tail_h->ptr;
        p.tail_n = (int)tn;
      }
    }
    return true;
  }

  void* out_ptr(int i) {
    PN& pn = pl.ns[i];
    if (pn.fwd) return pn.copy_dst->ptr;  // produced in place: the handle itself stays deferred
    if (!pn.h->ptr) alloc_storage(pn.h);
    return pn.h->ptr;
  }

  void bind_outputs(const Gr& g, Launch& L) {
    GemmProblem& p = L.p;
    p.C = out_ptr(g.out);
    if (g.rs >= 0) {
      p.rowsum = out_ptr(g.rs);
      if (g.rs_in) {
        p.rowsum_acc = true;
        p.rowsum_in = g.rs_in->ptr;
        p.rowsum_alpha = g.rs_alpha;
      }
    }
    if (g.loss_node >= 0) p.loss_out = out_ptr(g.loss_node);
    if (g.tail >= 0) p.tail_out = out_ptr(g.tail);
  }

  void mark_outputs(const Gr& g) {
    const int outs[4] = {g.out, g.rs, g.loss_node, g.tail};
    for (int o : outs)
      if (o >= 0) {
        if (pl.ns[o].fwd) pl.ns[o].copied = true;
        else stored(o);
      }
    if (g.mem.size() > 1) g_stats[1]++;
    for (int m : g.mem)
      if (m != g.out && m != g.rs && m != g.loss_node && m != g.tail) g_stats[2]++;
  }

  void launch_one(const Gr& g, Launch& L) {
    describe_gemm(L.p);
    if (g.mem.size() > 1 && gemm_small_route(L.p)) launch_gemm_small(L.p, S());
    else run_gemm(L.p);
  }

  // Small-GEMM launches are held back for a moment: three in a row -- a forward launch, the output layer with
  // its loss head, the pair of weight gradients -- are the batched training step, which goes out as ONE launch
  // with grid barriers (gemm_small_chain_kernel) when its shapes pick the configurations that kernel is built from.
  struct Queued {
    std::unique_ptr<Launch> a, b;  // b: the second problem of a pair
    const Gr *ga = nullptr, *gb = nullptr;
  };
  std::vector<Queued> queue;

  // is PN i an output of a launch that has been planned but not issued yet?
  bool in_queue(int i) const {
    for (const Queued& e : queue)
      for (const Gr* g : {e.ga, e.gb})
        if (g && (i == g->out || i == g->rs || i == g->loss_node || i == g->tail)) return true;
    return false;
  }

  void drain() {
    if (queue.empty()) return;
    std::vector<Queued> q;
    q.swap(queue);
    for (Queued& e : q) {
      describe_gemm(e.a->p);
      if (e.b) describe_gemm(e.b->p);
    }
    if (q.size() == 3 && !q[0].b && !q[1].b && q[2].b && q[1].a->p.loss_rows && !q[0].a->p.loss_rows &&
        (launch_gemm_small_chain(q[0].a->p, q[1].a->p, q[2].a->p, q[2].b->p, S()) ||
         launch_gemm_small_chain(q[0].a->p, q[1].a->p, q[2].b->p, q[2].a->p, S()))) {
      g_stats[1] -= 2;  // one launch, not three
      return;
    }
    for (size_t qi = 0; qi < q.size(); ++qi) {
      Queued& e = q[qi];
      // a forward layer directly followed by the loss-head launch that reads its output: one launch, joined inside
      // each XCD (gemm_small_seam_kernel); the pair kernel's refusal costs nothing
      if (!e.b && qi + 1 < q.size() && !q[qi + 1].b && q[qi + 1].a->p.loss_rows && !e.a->p.loss_rows &&
          launch_gemm_small_seam(e.a->p, q[qi + 1].a->p, S())) {
        g_stats[1]--;  // one launch, not two
        ++qi;
        continue;
      }
      if (e.b) {
        if (launch_gemm_small_pair(e.a->p, e.b->p, S()) || launch_gemm_small_pair(e.b->p, e.a->p, S())) continue;
        launch_gemm_small(e.a->p, S());
        launch_gemm_small(e.b->p, S());
        g_stats[1]++;
      } else {
        launch_gemm_small(e.a->p, S());
      }
    }
  }

  void run_gemm_group(Gr& g) {
    std::unique_ptr<Launch> L(new Launch());
    if (!build(g, *L)) {
      drain();
      run_members(g);
      return;
    }
    bind_outputs(g, *L);
    if (g.mem.size() > 1 && gemm_small_route(L->p)) {
      Queued e;
      e.a = std::move(L);
      e.ga = &g;
      queue.push_back(std::move(e));
    } else {
      drain();
      describe_gemm(L->p);
      run_gemm(L->p);
    }
    mark_outputs(g);
  }

  void run_pair(Gr& g1, Gr& g2) {
    std::unique_ptr<Launch> L1(new Launch()), L2(new Launch());
    const bool ok1 = build(g1, *L1), ok2 = build(g2, *L2);
    if (ok1) bind_outputs(g1, *L1);
    if (ok2) bind_outputs(g2, *L2);
    if (ok1 && ok2 && gemm_small_route(L1->p) && gemm_small_route(L2->p)) {
      Queued e;
      e.a = std::move(L1);
      e.b = std::move(L2);
      e.ga = &g1;
      e.gb = &g2;
      queue.push_back(std::move(e));
      mark_outputs(g1);
      mark_outputs(g2);
      g_stats[1]--;  // one launch, not two (drain() corrects this if the pair kernel refuses the shapes)
      return;
    }
    drain();
    if (ok1) { launch_one(g1, *L1); mark_outputs(g1); } else run_members(g1);
    if (ok2) { launch_one(g2, *L2); mark_outputs(g2); } else run_members(g2);
  }

  // all outer-product weight gradients of a one-sample step in one launch
  void run_rank1_unit(const std::vector<int>& members) {
    std::vector<std::unique_ptr<Launch>> L;
    bool ok = true;
    for (int gi : members) {
      L.emplace_back(new Launch());
      ok = ok && build(pl.gs[gi], *L.back());
      const GemmProblem& p = L.back()->p;
      // (a one-row / one-column operand has no stride to speak of: the output layer of tensor-ops-dots is 1 x 8)
      ok = ok && p.K == 1 && p.batch == 1 && (p.a_sm == 1 || p.M == 1) && (p.b_sn == 1 || p.N == 1) &&
           (p.beta == 0.0 || p.beta == 1.0) &&
           p.c_sm == p.N;
    }
    drain();
    if (!ok) {
      for (int gi : members) run_gemm_group(pl.gs[gi]);
      return;
    }
    const void *dz[RANK1_MAX_LAYERS], *a[RANK1_MAX_LAYERS], *w_in[RANK1_MAX_LAYERS], *b_in[RANK1_MAX_LAYERS];
    void *w[RANK1_MAX_LAYERS], *b[RANK1_MAX_LAYERS];
    double alpha[RANK1_MAX_LAYERS];
    int64_t rows[RANK1_MAX_LAYERS], cols[RANK1_MAX_LAYERS];
    for (size_t k = 0; k < members.size(); ++k) {
      Gr& g = pl.gs[members[k]];
      bind_outputs(g, *L[k]);
      const GemmProblem& p = L[k]->p;
      dz[k] = p.A; a[k] = p.B; w[k] = p.C;
      w_in[k] = p.beta == 1.0 ? p.Cin : nullptr;
      b[k] = p.rowsum;
      b_in[k] = p.rowsum_acc ? p.rowsum_in : nullptr;
      alpha[k] = p.alpha;
      // (the bias update carries its own factor; the kernel has one per layer: they are the same -rate in every
      //  network the DSL can build, and a mismatch falls back below)
      if (p.rowsum && (p.rowsum_acc ? p.rowsum_alpha : 1.0) != p.alpha) ok = false;
      rows[k] = p.M; cols[k] = p.N;
    }
    if (!ok) {
      for (size_t k = 0; k < members.size(); ++k) { launch_one(pl.gs[members[k]], *L[k]); mark_outputs(pl.gs[members[k]]); }
      return;
    }
    if (g_describe) {
      StepDesc d;
      d.kind = 1;
      d.n = (int)members.size();
      d.p.dtype = L[0]->p.dtype;
      for (int k = 0; k < d.n; ++k) {
        d.dz[k] = dz[k]; d.a[k] = a[k]; d.w[k] = w[k]; d.b[k] = b[k]; d.w_in[k] = w_in[k]; d.b_in[k] = b_in[k];
        d.alpha[k] = alpha[k]; d.rows[k] = rows[k]; d.cols[k] = cols[k];
      }
      g_describe->push_back(d);
    }
    launch_rank1_general(L[0]->p.dtype, (int)members.size(), dz, a, w, b, w_in, b_in, alpha, rows, cols, S());
    for (int gi : members) mark_outputs(pl.gs[gi]);
    g_stats[1] -= (int64_t)members.size() - 1;
  }

  // a row-local subgraph as one compiled kernel (or, without a run-time compiler, op by op)
  void run_rowprog(Gr& g) {
    RowProg& rp = *g.rowprog;
    to_tensor root = pl.ns[g.rp_root].h;
    if (!root->ptr) resolve_view(root);
    bool ok = root->ptr && root->contiguous() && rowprog_build(rp);
    for (to_tensor e : g.rp_ext) {
      if (e && !e->ptr) resolve_view(e);  // (a peer's output: produced earlier in this plan)
      ok = ok && e && e->ptr && e->contiguous();
    }
    if (!ok) {
      why = rp.err.empty() ? "row program: operands not ready" : rp.err.c_str();
      run_members(g);
      return;
    }
    const void* ext[4] = {nullptr, nullptr, nullptr, nullptr};
    void* outs[4] = {nullptr, nullptr, nullptr, nullptr};
    for (size_t i = 0; i < g.rp_ext.size(); ++i) ext[i] = g.rp_ext[i]->ptr;
    for (size_t i = 0; i < g.rp_outs.size(); ++i) {
      PN& on = pl.ns[g.rp_outs[i]];
      if (!on.h->ptr) alloc_storage(on.h);
      outs[i] = on.h->ptr;
    }
    describe_other();
    rowprog_launch(rp, root->ptr, ext, outs, root->batch > 0 ? root->batch : 1, S());
    for (int o : g.rp_outs) stored(o);
    g_stats[1]++;
    for (int m : g.mem)
      if (std::find(g.rp_outs.begin(), g.rp_outs.end(), m) == g.rp_outs.end()) g_stats[2]++;
  }

  // ---- sibling batches (round 6) ------------------------------------------------------------------------------------------
  // The reference's BTensor maps a GEMM over the trailing matrices of a rank > 2 operand (`mapBTM`, BTensor.hs:703-710) and a
  // `liftB` over every leaf (:345-369): through the inner boundary config 5 arrives as 512 `gemm` calls that share B and 512
  // `liftB` calls that share a closure (README.md:150-154 prescribes exactly this integration) -- 1,024 launches for what the
  // outer boundary does in two.  Deferral fuses a value with its consumers (vertically); siblings it has to find here:
  //  * plain products (no epilogue) of equal shape with the same right operand, in the range of the short-K streaming kernel
  //    once their rows are counted together: ONE launch, the A operands through a device table of pointers, the results in
  //    consecutive slices of one allocation;
  //  * lifts of the same closure whose operands lie one behind the other in memory (which is how the launch above leaves
  //    them): ONE launch over the whole range, results again in one allocation.
  // A batch runs at the place of its first member in the plan's order, so every member's inputs must have been produced by
  // then (its group's dependencies all lie earlier).  Whatever does not qualify runs as before.  Not while a step is being
  // captured or described (a table upload is no kernel launch; the step recognisers read single launches).
  static bool batching_on() {
    static const bool v = [] { const char* e = getenv("TOPS_SIBLING_BATCH"); return !(e && e[0] == '0'); }();
    return v;
  }
  std::vector<int> pos;   // group -> its place in the order (filled by run_all for plans worth looking at)

  bool deps_before(const Gr& g, int k) const {
    for (int d : g.deps)
      if (pos[(size_t)d] >= k) return false;
    return true;
  }
  bool plain_single(const Gr& g) const {
    return !g.done && g.mem.size() == 1 && g.pair < 0 && g.r1 < 0 && !g.rowprog && !pl.ns[g.mem[0]].fwd && !pl.ns[g.mem[0]].stored &&
           !pl.ns[g.mem[0]].h->ptr && !pl.ns[g.mem[0]].is_const;
  }
  // a product alone, or with nothing but an activation (and a scale) fused behind it: what the short-K kernel's epilogue carries
  bool product_family_member(const Gr& g) const {
    if (!g.gemm || g.done || g.pair >= 0 || g.r1 >= 0 || g.rowprog || g.mem.size() > 3) return false;
    if (g.rs >= 0 || g.loss_kind || g.tail >= 0 || g.bias || g.cin || g.dact || g.act > 1) return false;
    const PN& o = pl.ns[g.out];
    return !o.fwd && !o.stored && !o.h->ptr && !o.is_const && !pl.ns[g.anchor].n->d.reduce;
  }
  static bool same_epilogue(const Gr& a, const Gr& b) { return a.act == b.act && a.alpha == b.alpha && a.mem.size() == b.mem.size(); }

  bool try_gemm_batch(const std::vector<int>& order, int k) {
    Gr& g0 = pl.gs[order[(size_t)k]];
    if (!product_family_member(g0)) return false;
    std::unique_ptr<Launch> L0(new Launch());
    if (!build(g0, *L0)) return false;
    const GemmProblem& p0 = L0->p;
    if (p0.dtype != TO_F32 || p0.batch != 1 || p0.reduce_batch || p0.beta != 0.0 || p0.bias || p0.act > 1 || p0.dact ||
        p0.a_sk != 1 || p0.M % 32 != 0 || (p0.M * p0.N * 4) % 16 != 0 || (reinterpret_cast<uintptr_t>(p0.A) & 15u))
      return false;
    std::vector<int> mem{order[(size_t)k]};
    std::vector<const void*> atab{p0.A};
    std::vector<std::unique_ptr<Launch>> keep;   // (packed operands stay alive until the launch is enqueued)
    for (size_t j = (size_t)k + 1; j < order.size() && atab.size() < 4096; ++j) {
      Gr& g = pl.gs[order[j]];
      if (!product_family_member(g) || !same_epilogue(g, g0) || !deps_before(g, k)) continue;
      const Node* n = pl.ns[g.anchor].n;
      const Node* n0 = pl.ns[g0.anchor].n;
      if (n->in[1] != n0->in[1] || n->d.lm != n0->d.lm || n->d.lo != n0->d.lo || n->d.ln != n0->d.ln) continue;   // (the same B handle)
      // the common case without a plan of its own: a left operand with storage and exactly the first one's layout gives the
      // first one's problem with another A (512 full plans were a third of this flush's time on the host)
      {
        to_tensor a = n->in[0], a0 = n0->in[0];
        if (a->ptr && a0->ptr && a0->ptr == p0.A && a->dtype == a0->dtype && a->batch == a0->batch && a->bstride == a0->bstride &&
            same_shape(a, a0) && std::equal(a->strides, a->strides + a->rank, a0->strides) && !(reinterpret_cast<uintptr_t>(a->ptr) & 15u) &&
            same_shape(pl.ns[g.out].h, pl.ns[g0.out].h) && pl.ns[g.out].h->batch == pl.ns[g0.out].h->batch) {
          mem.push_back(order[j]);
          atab.push_back(a->ptr);
          continue;
        }
      }
      std::unique_ptr<Launch> L(new Launch());
      if (!build(g, *L)) continue;
      const GemmProblem& p = L->p;
      if (p.dtype != p0.dtype || p.B != p0.B || p.b_sk != p0.b_sk || p.b_sn != p0.b_sn || p.M != p0.M || p.N != p0.N || p.K != p0.K ||
          p.a_sm != p0.a_sm || p.a_sk != 1 || p.batch != 1 || p.reduce_batch || p.alpha != p0.alpha || p.beta != 0.0 || p.act != p0.act ||
          p.bias || p.dact || (reinterpret_cast<uintptr_t>(p.A) & 15u))
        continue;
      mem.push_back(order[j]);
      atab.push_back(p.A);
      keep.push_back(std::move(L));
    }
    if (mem.size() < 2) return false;
    GemmProblem all = p0;
    all.M = p0.M * (int64_t)mem.size();
    all.c_sm = p0.N;
    all.C = reinterpret_cast<void*>(16);   // (placeholder with the alignment the result will have: the applicability test reads it)
    if (!gemm_skinnyk_applicable(all)) return false;
    drain();
    std::vector<to_tensor> outs;
    for (int gi : mem) outs.push_back(pl.ns[pl.gs[gi].out].h);
    alloc_storage_shared((int)outs.size(), outs.data());
    all.C = outs[0]->ptr;
    all.a_table = table_upload(atab.data(), atab.size() * sizeof(void*), S());
    all.a_table_rows = p0.M;
    describe_gemm(all);
    launch_gemm_skinnyk(all, S());
    for (int gi : mem) {
      pl.gs[gi].done = true;
      mark_outputs(pl.gs[gi]);   // (the output exists; what was fused behind the product is counted as elided)
    }
    if (debug_on()) std::fprintf(stderr, "[lazy] sibling batch: %zu products %lld x %lld x %lld with one right operand -> one launch\n", mem.size(),
                                 (long long)p0.M, (long long)p0.K, (long long)p0.N);
    return true;
  }

  bool try_lift_batch(const std::vector<int>& order, int k) {
    Gr& g0 = pl.gs[order[(size_t)k]];
    if (g0.gemm || !plain_single(g0)) return false;
    const int i0 = g0.mem[0];
    const Node* n0 = pl.ns[i0].n;
    if (n0->d.op != N_LIFT || n0->in.empty() || n0->in.size() > 8) return false;
    to_tensor h0 = pl.ns[i0].h;
    const int64_t total = h0->total();
    const size_t bytes = (size_t)total * h0->esize();
    if (total == 0 || bytes % 16 != 0) return false;
    auto operands_ok = [&](const Node* n, to_tensor h) {
      if (n->d.op != N_LIFT || n->d.f != n0->d.f || n->in.size() != n0->in.size() || h->dtype != h0->dtype || h->total() != total ||
          h->batch != h0->batch || !same_shape(h, h0))
        return false;
      for (to_tensor x : n->in) {
        if (!x->ptr) resolve_view(x);
        if (!x->ptr || !x->contiguous() || x->total() != total || x->dtype != h0->dtype) return false;   // (no broadcast operand)
      }
      return true;
    };
    if (!operands_ok(n0, h0)) return false;
    std::vector<int> mem{order[(size_t)k]};
    for (size_t j = (size_t)k + 1; j < order.size(); ++j) {
      Gr& g = pl.gs[order[j]];
      if (g.gemm || !plain_single(g) || !deps_before(g, k)) continue;
      if (!operands_ok(pl.ns[g.mem[0]].n, pl.ns[g.mem[0]].h)) continue;
      mem.push_back(order[j]);
    }
    if (mem.size() < 2) return false;
    // in the order of their first operand's address; every operand then has to advance by one tensor per member
    std::sort(mem.begin(), mem.end(), [&](int a, int b) {
      return pl.ns[pl.gs[a].mem[0]].n->in[0]->ptr < pl.ns[pl.gs[b].mem[0]].n->in[0]->ptr;
    });
    const Node* nf = pl.ns[pl.gs[mem[0]].mem[0]].n;
    // the longest run from the front that is consecutive in every operand (what does not belong runs on its own later)
    size_t run = 1;
    for (; run < mem.size(); ++run) {
      const Node* n = pl.ns[pl.gs[mem[run]].mem[0]].n;
      bool ok = true;
      for (size_t q = 0; q < nf->in.size() && ok; ++q)
        ok = static_cast<const char*>(n->in[q]->ptr) == static_cast<const char*>(nf->in[q]->ptr) + run * bytes;
      if (!ok) break;
    }
    // (the run has to contain the group whose turn it is: it is the one that must be done when this returns)
    bool has_k = false;
    for (size_t r = 0; r < run; ++r) has_k = has_k || mem[r] == order[(size_t)k];
    if (run < 2 || !has_k) return false;
    mem.resize(run);
    drain();
    std::vector<to_tensor> outs;
    for (int gi : mem) outs.push_back(pl.ns[pl.gs[gi].mem[0]].h);
    alloc_storage_shared((int)outs.size(), outs.data());
    const void* xs[8];
    for (size_t q = 0; q < nf->in.size(); ++q) xs[q] = nf->in[q]->ptr;
    describe_other();
    lift_launch_raw(nf->d.f, (int)nf->in.size(), xs, outs[0]->ptr, total * (int64_t)run, h0->dtype);
    for (int gi : mem) {
      pl.gs[gi].done = true;
      stored(pl.gs[gi].mem[0]);
    }
    if (debug_on()) std::fprintf(stderr, "[lazy] sibling batch: %zu lifts of one closure over %lld elements each -> one launch\n", run, (long long)total);
    return true;
  }

  void run_all(const std::vector<int>& order) {
    const bool look = order.size() >= 8 && batching_on() && !g_describe && !launch_recorder() && !rt().capturing;
    if (look) {
      pos.assign(pl.gs.size(), -1);
      for (size_t k = 0; k < order.size(); ++k) pos[(size_t)order[k]] = (int)k;
    }
    for (size_t k = 0; k < order.size(); ++k) {
      if (look && !pl.gs[order[k]].done && (try_gemm_batch(order, (int)k) || try_lift_batch(order, (int)k))) continue;
      run_group(order[k]);
    }
  }

  void run_group(int gi) {
    Gr& g = pl.gs[gi];
    if (g.done) return;
    if (g.r1 >= 0) {
      const std::vector<int> members = pl.gs[g.r1].r1_members;
      for (int m : members) pl.gs[m].done = true;
      run_rank1_unit(members);
      return;
    }
    g.done = true;
    if (g.pair >= 0) pl.gs[g.pair].done = true;
    if (g.rowprog) {
      drain();
      run_rowprog(g);
    } else if (!g.gemm) {
      drain();
      run_members(g);
    }
    else if (g.pair >= 0) run_pair(g, pl.gs[g.pair]);
    else run_gemm_group(g);
  }
};

// ---- one flush -----------------------------------------------------------------------------------------------------------
static uint64_t g_plan_epoch = 0;

// the recorded ops the roots depend on, in recording order, with their producer / consumer links
static void collect(Plan& pl, const std::vector<to_tensor>& roots) {
  pl.epoch = ++g_plan_epoch;
  std::vector<Node*> stack, all;
  auto push = [&](to_tensor t) {
    Node* p = producer(t);
    if (p && p->plan_epoch != pl.epoch) {
      p->plan_epoch = pl.epoch;
      p->plan_idx = -1;
      stack.push_back(p);
    }
  };
  for (to_tensor t : roots) push(t);
  while (!stack.empty()) {
    Node* n = stack.back();
    stack.pop_back();
    all.push_back(n);
    for (to_tensor x : n->in) push(x);
  }
  std::sort(all.begin(), all.end(), [](const Node* a, const Node* b) { return a->seq < b->seq; });
  pl.ns.resize(all.size());
  for (size_t i = 0; i < all.size(); ++i) {
    pl.ns[i].n = all[i];
    pl.ns[i].h = all[i]->out;
    all[i]->plan_idx = (int)i;
  }
  for (size_t i = 0; i < all.size(); ++i) {
    PN& pn = pl.ns[i];
    Node* n = pn.n;
    pn.prod.resize(n->in.size());
    bool all_const = true;
    for (size_t k = 0; k < n->in.size(); ++k) {
      const int q = pn_of(pl, n->in[k]);
      pn.prod[k] = q;
      if (q >= 0) {
        if (std::find(pl.ns[q].cons.begin(), pl.ns[q].cons.end(), (int)i) == pl.ns[q].cons.end())
          pl.ns[q].cons.push_back((int)i);
        if (!pl.ns[q].is_const) all_const = false;
      } else {
        all_const = false;
      }
    }
    // constants: FILL, and scale / smooth closures / sums over constants (the seed of gradTOp and its negation)
    const int op = n->d.op;
    if (op == N_FILL) {
      pn.is_const = true;
      pn.cval = n->d.alpha;
    } else if (all_const && !n->in.empty() && n->in.size() <= 8 && (op == N_SCALE || op == N_SUM || op == N_LIFT)) {
      double x[8] = {0};
      for (size_t k = 0; k < n->in.size(); ++k) x[k] = pl.ns[pn.prod[k]].cval;
      bool same_shapes = true;
      for (to_tensor in : n->in) same_shapes = same_shapes && same_shape(in, pn.h);
      if (same_shapes) {
        pn.is_const = true;
        if (op == N_SCALE) pn.cval = n->d.alpha * x[0];
        else if (op == N_SUM) {
          pn.cval = 0.0;
          for (size_t k = 0; k < n->in.size(); ++k) pn.cval += pl.ns[pn.prod[k]].cval;
        } else {
          pn.cval = expr_eval(*n->d.f, x);
        }
      }
    }
  }
}

// ancestor bitsets (planning only: a plan that comes out of the cache does not need them)
static void compute_ancestors(Plan& pl) {
  const size_t N = pl.ns.size();
  pl.words = (int)((N + 63) / 64);
  pl.anc.assign(N, std::vector<uint64_t>((size_t)pl.words, 0));
  for (size_t i = 0; i < N; ++i)
    for (int q : pl.ns[i].prod)
      if (q >= 0) {
        for (int w = 0; w < pl.words; ++w) pl.anc[i][w] |= pl.anc[q][w];
        pl.anc[i][q >> 6] |= 1ull << (q & 63);
      }
}

// ---- plan cache --------------------------------------------------------------------------------------------------------
// A training loop records the same graph every step.  Everything the planner's decisions depend on goes into a signature:
// the recorded ops with their static arguments and expression ids, the graph's wiring, the layout of every operand that
// is not simply "the contiguous result of another node" (views, existing tensors), which existing tensors are the same
// memory or overlap, and what is demanded or copied where.  A flush whose signature has been seen takes its groups, their
// order, the forwarding decisions and the node rewrites from the cache and goes straight to execution; operands are
// named by position ("input k of node i"), so the plan binds to this step's handles.  Nothing that depends on addresses
// (alignment-driven kernel variants, packing) is cached: Exec::build works that out per launch as before.
struct CachedPlan {
  std::vector<uint64_t> sig;
  std::vector<Gr> gs;                // operand pointers cleared; the Refs name them
  std::vector<int> group, order;     // per node / execution order of the groups
  std::vector<char> fwd;             // per node: produced straight into its copy destination
  std::vector<std::pair<int, int>> dlogistic;  // rewrite_dlogistic: node i reads the value of node c instead of z
  mutable uint64_t used = 0;         // when it was last found (eviction is least-recently-used)
};
static uint64_t g_plan_clock = 0;
static std::unordered_map<uint64_t, std::vector<std::unique_ptr<CachedPlan>>>& plan_cache() {
  static std::unordered_map<uint64_t, std::vector<std::unique_ptr<CachedPlan>>> m;
  return m;
}
static size_t g_plan_cache_entries = 0;
static int64_t g_plan_cache_hits = 0, g_plan_cache_misses = 0;
void lazy_cache_stats(int64_t* hits, int64_t* misses, int64_t* entries) {
  if (hits) *hits = g_plan_cache_hits;
  if (misses) *misses = g_plan_cache_misses;
  if (entries) *entries = (int64_t)g_plan_cache_entries;
}
void lazy_cache_clear() {
  plan_cache().clear();
  g_plan_cache_entries = 0;
}
static bool plan_cache_on() {
  static const int on = [] { const char* e = getenv("TOPS_PLAN_CACHE"); return e ? atoi(e) : 1; }();
  return on != 0;
}

static uint64_t dbits(double d) {
  uint64_t u;
  std::memcpy(&u, &d, 8);
  return u;
}

static void sig_layout(std::vector<uint64_t>& s, to_tensor t) {
  s.push_back(((uint64_t)t->rank << 32) | ((uint64_t)t->dtype << 16) | (t->batch > 0 ? 1u : 0u));
  s.push_back((uint64_t)t->batch);
  s.push_back((uint64_t)t->bstride);
  for (int i = 0; i < t->rank; ++i) {
    s.push_back((uint64_t)t->dims[i]);
    s.push_back((uint64_t)t->strides[i]);
  }
}

static void plan_signature(Plan& pl, const std::vector<std::pair<to_tensor, to_tensor>>& copies, std::vector<uint64_t>& s) {
  s.clear();
  s.reserve(pl.ns.size() * 24);
  std::vector<to_tensor> ext;  // existing tensors read by the plan, and the copy destinations
  // (found by hashing beyond a handful: the 513 operands of 512 sibling products were 131k pointer compares here and as many
  //  range tests below -- 0.65 ms of a flush whose two launches take 0.33)
  std::unordered_map<to_tensor, size_t> ext_ix;
  auto ext_slot = [&](to_tensor x) {
    if (ext.size() < 16) {
      for (size_t i = 0; i < ext.size(); ++i)
        if (ext[i] == x) return i;
    } else {
      if (ext_ix.empty())
        for (size_t i = 0; i < ext.size(); ++i) ext_ix.emplace(ext[i], i);
      auto it = ext_ix.find(x);
      if (it != ext_ix.end()) return it->second;
      ext_ix.emplace(x, ext.size());
    }
    ext.push_back(x);
    return ext.size() - 1;
  };
  s.push_back(pl.ns.size() | (loss_head_match_on() ? 0ull : 1ull << 62));
  for (size_t i = 0; i < pl.ns.size(); ++i) {
    const PN& pn = pl.ns[i];
    const Node* n = pn.n;
    s.push_back(((uint64_t)n->d.op << 48) | ((uint64_t)n->d.lm << 40) | ((uint64_t)n->d.lo << 32) | ((uint64_t)n->d.ln << 24) |
                ((uint64_t)n->d.reduce << 16) | (uint64_t)(n->d.len_n & 0xffff));
    s.push_back(dbits(n->d.alpha));
    // (the STRUCTURE of the closure, not the instance: a host that reifies its closures anew every step -- the Haskell
    //  shim's liftH does unless it caches them -- still repeats itself as far as a plan is concerned)
    s.push_back(n->d.f ? n->d.f->sid : 0);
    s.push_back(((uint64_t)n->in.size() << 8) | (pn.demanded ? 1u : 0u) | (pn.copy_dst ? 2u : 0u));
    sig_layout(s, pn.h);  // (a fresh result is contiguous: dims, batch and dtype are what matters)
    for (size_t k = 0; k < n->in.size(); ++k) {
      to_tensor x = n->in[k];
      const int q = pn.prod[k];
      if (q >= 0) {
        s.push_back(0x1000000000000000ull | (uint64_t)q);
        if (x == pl.ns[q].h) continue;
        s.push_back((uint64_t)x->view_off);  // a view of that node's value
        sig_layout(s, x);
      } else {
        s.push_back(0x2000000000000000ull | (uint64_t)ext_slot(x));
        sig_layout(s, x);
      }
    }
    if (pn.copy_dst) {
      s.push_back(0x3000000000000000ull | (uint64_t)ext_slot(pn.copy_dst));
      sig_layout(s, pn.copy_dst);
    }
  }
  // which existing tensors are the same memory / overlap (the target rows found twice, Cin aliasing a copy destination,
  // readers of memory that a forwarded result overwrites)
  s.push_back(0x4000000000000000ull | (uint64_t)ext.size());
  if (ext.size() < 16) {
    for (size_t a = 0; a < ext.size(); ++a)
      for (size_t b = a + 1; b < ext.size(); ++b) {
        const uint64_t rel = (ext[a]->ptr == ext[b]->ptr ? 1u : 0u) | (overlaps(ext[a], ext[b]) ? 2u : 0u);
        if (rel) s.push_back((a << 40) | (b << 8) | rel);
      }
  } else {
    // the same relation words in the same (a, b) order, found by a sweep over the address ranges instead of every pair
    struct R { const char *lo, *hi; size_t i; };
    std::vector<R> rs;
    rs.reserve(ext.size());
    for (size_t i = 0; i < ext.size(); ++i) {
      R r{nullptr, nullptr, i};
      if (ext[i]->ptr) mem_range(ext[i], &r.lo, &r.hi);
      rs.push_back(r);
    }
    std::sort(rs.begin(), rs.end(), [](const R& x, const R& y) { return x.lo < y.lo || (x.lo == y.lo && x.i < y.i); });
    std::vector<uint64_t> rel;
    for (size_t x = 0; x < rs.size(); ++x)
      for (size_t y = x + 1; y < rs.size(); ++y) {
        const bool same_ptr = rs[y].lo == rs[x].lo;                          // (null == null included, as in the pairwise form)
        const bool over = rs[x].lo && rs[y].lo < rs[x].hi && rs[x].lo < rs[y].hi;
        if (!same_ptr && !(rs[x].lo && rs[y].lo < rs[x].hi)) break;           // (sorted by lo: nothing further can touch x)
        const uint64_t w = (same_ptr ? 1u : 0u) | (over ? 2u : 0u);
        if (!w) continue;
        const size_t a = std::min(rs[x].i, rs[y].i), b = std::max(rs[x].i, rs[y].i);
        rel.push_back(((uint64_t)a << 40) | ((uint64_t)b << 8) | w);
      }
    std::sort(rel.begin(), rel.end());
    s.insert(s.end(), rel.begin(), rel.end());
  }
  (void)copies;
}

static uint64_t sig_hash(const std::vector<uint64_t>& s) {
  uint64_t h = 1469598103934665603ull;
  for (uint64_t v : s) {
    h ^= v;
    h *= 1099511628211ull;
    h ^= h >> 29;
  }
  return h;
}

static to_tensor bind_ref(const Plan& pl, const Ref& r) { return r.node >= 0 ? pl.ns[r.node].n->in[(size_t)r.in] : nullptr; }

// node i of the plan reads h = logistic(z), the value of node c, instead of z: `d * logistic'(z)` becomes d * h (1 - h)
static void apply_dlogistic(Plan& pl, int i, int c) {
  Node* n = pl.ns[i].n;
  to_tensor z = n->in[1], h = pl.ns[c].h;
  const int zq = pl.ns[i].prod[1];
  retain_int(h);
  n->in[1] = h;
  const int tanh_form = n->d.f->kind == EW_MUL_DTANH ? 1 : 0;
  expr_release(n->d.f);
  n->d.f = nullptr;
  n->d.op = N_DACT;
  n->d.lm = tanh_form;
  pl.ns[i].prod[1] = c;
  if (zq >= 0 && pl.ns[i].prod[0] != zq) {  // (`d * logistic'(d)`: the node still reads z as its first input)
    auto& zc = pl.ns[zq].cons;
    zc.erase(std::remove(zc.begin(), zc.end(), i), zc.end());
  }
  if (std::find(pl.ns[c].cons.begin(), pl.ns[c].cons.end(), i) == pl.ns[c].cons.end()) pl.ns[c].cons.push_back(i);
  release_int(z);
}

static void plan_store(const Plan& pl, const std::vector<int>& order, std::vector<uint64_t>&& sig, uint64_t hash,
                       const std::vector<std::pair<int, int>>& dlog) {
  if (g_plan_cache_entries >= 512) {  // (a host that never repeats itself): the least recently used quarter goes
    std::vector<uint64_t> stamps;
    for (auto& kv : plan_cache())
      for (auto& c : kv.second) stamps.push_back(c->used);
    std::nth_element(stamps.begin(), stamps.begin() + stamps.size() / 4, stamps.end());
    const uint64_t cut = stamps[stamps.size() / 4];
    for (auto it = plan_cache().begin(); it != plan_cache().end();) {
      auto& v = it->second;
      const size_t before = v.size();
      v.erase(std::remove_if(v.begin(), v.end(), [&](const std::unique_ptr<CachedPlan>& c) { return c->used <= cut; }), v.end());
      g_plan_cache_entries -= before - v.size();
      it = v.empty() ? plan_cache().erase(it) : std::next(it);
    }
  }
  auto cp = std::make_unique<CachedPlan>();
  cp->used = ++g_plan_clock;
  cp->sig = std::move(sig);
  cp->gs = pl.gs;
  for (Gr& g : cp->gs) {
    g.cin = g.bias = g.dact = g.rs_in = g.target = g.tail_w = g.tail_h = nullptr;
    g.done = false;
    for (to_tensor& e : g.rp_ext) e = nullptr;
  }
  cp->order = order;
  cp->dlogistic = dlog;
  for (const PN& pn : pl.ns) {
    cp->group.push_back(pn.group);
    cp->fwd.push_back(pn.fwd ? 1 : 0);
  }
  plan_cache()[hash].push_back(std::move(cp));
  ++g_plan_cache_entries;
}

static const CachedPlan* plan_find(const std::vector<uint64_t>& sig, uint64_t hash) {
  auto it = plan_cache().find(hash);
  if (it == plan_cache().end()) return nullptr;
  for (const auto& cp : it->second)
    if (cp->sig == sig) {
      cp->used = ++g_plan_clock;
      return cp.get();
    }
  return nullptr;
}

static void plan_instantiate(const CachedPlan& cp, Plan& pl, std::vector<int>& order) {
  for (const auto& r : cp.dlogistic) apply_dlogistic(pl, r.first, r.second);
  pl.gs = cp.gs;
  for (Gr& g : pl.gs) {
    g.cin = bind_ref(pl, g.r_cin);
    g.bias = bind_ref(pl, g.r_bias);
    g.dact = bind_ref(pl, g.r_dact);
    g.rs_in = bind_ref(pl, g.r_rs_in);
    g.target = bind_ref(pl, g.r_target);
    g.tail_w = bind_ref(pl, g.r_tail_w);
    g.tail_h = bind_ref(pl, g.r_tail_h);
    for (size_t e = 0; e < g.rp_ext.size(); ++e) g.rp_ext[e] = bind_ref(pl, g.rp_ext_ref[e]);
  }
  for (size_t i = 0; i < pl.ns.size(); ++i) {
    pl.ns[i].group = cp.group[i];
    pl.ns[i].fwd = cp.fwd[i] != 0;
  }
  order = cp.order;
}

static bool path_between(const Plan& pl, const Gr& from, const Gr& to) {  // does `to` depend on `from`?
  for (int a : from.mem)
    for (int b : to.mem)
      if (a == b || pl.is_anc(a, b)) return true;
  return false;
}

// ---- row programs: what hangs off a GEMM group's output and only ever touches one row at a time -------------------------
// (loss heads wider than the 16 lanes of the small-GEMM epilogue, heads the library has no closed form for: softmax >>>
//  scale >>> squaredError, an auto-encoder's squaredError over the whole input width ...)
static bool form_row_program_impl(Plan& pl, int root, bool allow_peers);
static void form_row_program(Plan& pl, int root) {
  // operands produced by OTHER launches of the same plan (the second GEMM of `W x + W' s + b`, Recurrent.hs:108-118) may
  // be read like existing tensors -- unless that closes a cycle through the program's own outputs; then without them
  if (!form_row_program_impl(pl, root, true)) form_row_program_impl(pl, root, false);
}
// true: done (a group was formed, or there is nothing to form); false: try again without peers
static bool form_row_program_impl(Plan& pl, int root, bool allow_peers) {
  to_tensor rh = pl.ns[root].h;
  if (rh->rank != 1 || rh->dims[0] < 1 || rh->dims[0] > 1024) return true;
  const int64_t N = rh->dims[0], Bfull = rh->batch;
  std::vector<to_tensor> ext;
  std::vector<Ref> ext_ref;
  std::vector<int> ext_q;  // producing plan node of a peer operand, -1 for an existing tensor
  std::vector<char> inT(pl.ns.size(), 0), inS(pl.ns.size(), 0);
  const char* inT_ptr = inT.data();
  // a peer: the stored output of a GEMM group that has already been formed
  auto peer_ok = [&](int q, to_tensor x) {
    if (!allow_peers || pl.ns[q].group < 0) return false;
    const Gr& pg = pl.gs[pl.ns[q].group];
    if (!(pg.gemm && (pg.out == q || pg.tail == q) && same_value_layout(x, pl.ns[q].h) && (x->batch == Bfull || x->batch == 0)))
      return false;
    // a peer that itself needs something this program computes would have to run both before and after it
    for (size_t m = (size_t)root + 1; m < (size_t)q; ++m)
      if (inT_ptr[m] && pl.is_anc((int)m, q)) return false;
    return true;
  };
  auto row_shaped = [&](to_tensor t) { return t->rank == 0 || (t->rank == 1 && t->dims[0] == N); };
  auto same_ext = [](to_tensor e, to_tensor x) { return e == x || (e->ptr && e->ptr == x->ptr && e->batch == x->batch && e->rank == x->rank); };
  // pass 1: T = row-local ops whose operands are the root, other members of T, constants or existing row-shaped tensors
  inT[root] = 1;
  for (size_t i = (size_t)root + 1; i < pl.ns.size(); ++i) {
    PN& pn = pl.ns[i];
    if (pn.group >= 0 || pn.is_const || pn.copy_dst) continue;
    const Node* n = pn.n;
    const int op = n->d.op;
    if (!(op == N_LIFT || op == N_DACT || op == N_SUM || op == N_SCALE || op == N_SUM_ROWS || op == N_MAP_ROWS ||
          (op == N_GMUL && !n->d.reduce && n->d.lo <= 1)))
      continue;
    if (pn.h->batch != Bfull || !row_shaped(pn.h) || pn.h->dtype != rh->dtype || n->in.size() > 8) continue;
    if (op == N_MAP_ROWS && n->d.len_n != 1) continue;
    if (op == N_GMUL && !((n->d.lo == 1 && n->in[0]->rank == 1 && n->in[1]->rank == 1) ||
                          (n->d.lo == 0 && n->in[0]->rank + n->in[1]->rank <= 1)))
      continue;
    bool ok = true;
    for (size_t k = 0; k < n->in.size() && ok; ++k) {
      to_tensor x = n->in[k];
      const int q = pn.prod[k];
      if (!row_shaped(x) || x->dtype != rh->dtype) ok = false;
      else if (q >= 0) ok = inT[q] ? same_value_layout(x, pl.ns[q].h) : (pl.ns[q].is_const || peer_ok(q, x));
      else ok = x->ptr && x->contiguous() && (x->batch == Bfull || x->batch == 0);  // per row, or shared by all rows
    }
    if (ok) inT[i] = 1;
  }
  // pass 2: what the root reaches inside T; pass 3: plus what those need from T (the target's side of a loss:
  // `-y * seed` depends on no member, the cotangent that consumes it does)
  inS[root] = 1;
  for (size_t i = (size_t)root + 1; i < pl.ns.size(); ++i)
    if (inT[i])
      for (int q : pl.ns[i].prod)
        if (q >= 0 && inS[q]) inS[i] = 1;
  for (size_t i = pl.ns.size(); i-- > (size_t)root + 1;)
    if (inS[i])
      for (int q : pl.ns[i].prod)
        if (q >= 0 && inT[q]) inS[q] = 1;
  std::vector<int> S{root};
  for (size_t i = (size_t)root + 1; i < pl.ns.size(); ++i) {
    if (!inS[i]) continue;
    S.push_back((int)i);
    const Node* n = pl.ns[i].n;
    for (size_t k = 0; k < n->in.size(); ++k) {
      const int q = pl.ns[i].prod[k];
      if (q >= 0 && (inS[q] || pl.ns[q].is_const)) continue;
      bool known = false;
      for (size_t e = 0; e < ext.size(); ++e) known = known || (q >= 0 ? ext_q[e] == q : (ext_q[e] < 0 && same_ext(ext[e], n->in[k])));
      if (!known) {
        ext.push_back(n->in[k]);
        ext_ref.push_back(Ref{(int)i, (int)k});
        ext_q.push_back(q);
      }
    }
  }
  if (ext.size() > 4) return !allow_peers;
  if (S.size() < 3) return true;  // (the root and a single op: that op is one launch already)
  // a peer that itself needs something this program produces would have to run both before and after it
  for (int q : ext_q)
    if (q >= 0)
      for (size_t k = 1; k < S.size(); ++k)
        if (pl.is_anc(S[k], q)) return false;
  // what the rest of the graph needs from it
  std::vector<int> outs;
  for (size_t k = 1; k < S.size(); ++k) {
    const PN& pn = pl.ns[S[k]];
    bool outside = pn.demanded;
    for (int c : pn.cons)
      if (!inS[c]) outside = true;
    if (outside) outs.push_back(S[k]);
  }
  if (outs.empty() || outs.size() > 4) return true;
  // the program: value ids 0 = root, 1.. = existing tensors, then the nodes (constants are re-stated as literals)
  auto rp = std::make_shared<RowProg>();
  rp->dtype = rh->dtype;
  rp->N = N;
  for (to_tensor e : ext) {
    rp->ext_vec.push_back(e->rank == 1);
    rp->ext_rowwise.push_back(e->batch > 0 || Bfull == 0);
  }
  std::unordered_map<int, int> id_of;  // plan node -> value id
  id_of[root] = 0;
  std::unordered_map<int, int> const_id;
  auto ext_id = [&](to_tensor x, int q) {
    for (size_t e = 0; e < ext.size(); ++e)
      if (q >= 0 ? ext_q[e] == q : (ext_q[e] < 0 && same_ext(ext[e], x))) return 1 + (int)e;
    return -1;
  };
  const int base = 1 + (int)ext.size();
  for (size_t k = 1; k < S.size(); ++k) {
    const PN& pn = pl.ns[S[k]];
    const Node* n = pn.n;
    std::vector<int> in;
    for (size_t j = 0; j < n->in.size(); ++j) {
      const int q = pn.prod[j];
      if (q >= 0 && inS[q]) in.push_back(id_of[q]);
      else if (q >= 0 && !pl.ns[q].is_const) in.push_back(ext_id(n->in[j], q));  // a peer's output
      else if (q >= 0) {  // a constant
        auto it = const_id.find(q);
        if (it == const_id.end()) {
          RowNode c;
          c.op = R_CONST;
          c.vec = pl.ns[q].h->rank == 1;
          c.alpha = pl.ns[q].cval;
          rp->nodes.push_back(c);
          it = const_id.emplace(q, base + (int)rp->nodes.size() - 1).first;
        }
        in.push_back(it->second);
      } else {
        in.push_back(ext_id(n->in[j], -1));
      }
    }
    RowNode r;
    r.vec = pn.h->rank == 1;
    r.in = in;
    switch (n->d.op) {
      case N_LIFT: r.op = R_LIFT; r.f = n->d.f; expr_retain(r.f); break;
      case N_DACT: r.op = R_DACT; r.alpha = n->d.lm; break;
      case N_SUM: r.op = R_SUM; break;
      case N_SCALE: r.op = R_SCALE; r.alpha = n->d.alpha; break;
      case N_SUM_ROWS: r.op = R_SUM_ROWS; break;
      case N_MAP_ROWS: r.op = R_MAP_ROWS; break;
      default: r.op = n->d.lo == 1 ? R_DOT : R_MUL; break;  // gmul: a dot product, or a product with a scalar
    }
    rp->nodes.push_back(r);
    id_of[S[k]] = base + (int)rp->nodes.size() - 1;
  }
  for (int o : outs) rp->outs.push_back(id_of[o]);
  Gr g;
  g.rowprog = rp;
  g.rp_root = root;
  g.rp_outs = outs;
  g.rp_ext = ext;
  g.rp_ext_ref = ext_ref;
  for (size_t k = 1; k < S.size(); ++k) g.mem.push_back(S[k]);
  // constants used only in here never get storage
  for (auto& kv : const_id) {
    const PN& cn = pl.ns[kv.first];
    bool all_in = cn.group < 0 && !cn.demanded && !cn.copy_dst;
    for (int c : cn.cons)
      if (!inS[c]) all_in = false;
    if (all_in) g.mem.push_back(kv.first);
  }
  std::sort(g.mem.begin(), g.mem.end());
  g.out = outs[0];
  const int gi = (int)pl.gs.size();
  for (int m : g.mem) pl.ns[m].group = gi;
  pl.gs.push_back(std::move(g));
  return true;
}

static void plan_groups(Plan& pl, std::vector<std::pair<int, int>>& dlog) {
  static const int fuse = [] { const char* e = getenv("TOPS_LAZY_FUSE"); return e ? atoi(e) : 1; }();
  if (fuse && pl.ns.size() <= 8192) {
    rewrite_dlogistic(pl, dlog);
    // a contraction of two per-row vectors / scalars (a dot product, a product with a scalar: the small change of a loss
    // head) is no GEMM: it is left for the row programs below, and becomes a launch of its own only if none takes it
    auto row_local = [&](int i) {
      const Node* n = pl.ns[i].n;
      return !n->d.reduce && n->d.lo <= 1 && n->in[0]->rank <= 1 && n->in[1]->rank <= 1 && pl.ns[i].h->rank <= 1;
    };
    for (size_t i = 0; i < pl.ns.size(); ++i)
      if (pl.ns[i].group < 0 && pl.ns[i].n->d.op == N_GMUL && !row_local((int)i)) form_gemm_group(pl, (int)i);
    // what is left hanging off the output of a GEMM group, row by row
    const size_t n_gemm_groups = pl.gs.size();
    for (size_t gi = 0; gi < n_gemm_groups; ++gi)
      if (pl.gs[gi].gemm && !pl.gs[gi].loss_kind && pl.gs[gi].out >= 0) form_row_program(pl, pl.gs[gi].out);
    for (size_t i = 0; i < pl.ns.size(); ++i)
      if (pl.ns[i].group < 0 && pl.ns[i].n->d.op == N_GMUL) form_gemm_group(pl, (int)i);
  }
  for (size_t i = 0; i < pl.ns.size(); ++i)
    if (pl.ns[i].group < 0) {
      Gr g;
      g.mem.push_back((int)i);
      g.out = (int)i;
      pl.ns[i].group = (int)pl.gs.size();
      pl.gs.push_back(std::move(g));
    }
  // dependencies: outputs of other groups read by members
  for (size_t gi = 0; gi < pl.gs.size(); ++gi) {
    Gr& g = pl.gs[gi];
    for (int m : g.mem)
      for (int q : pl.ns[m].prod)
        if (q >= 0 && pl.ns[q].group != (int)gi &&
            std::find(g.deps.begin(), g.deps.end(), pl.ns[q].group) == g.deps.end())
          g.deps.push_back(pl.ns[q].group);
  }
  if (!fuse) return;
  // one-sample steps: every weight gradient is an outer product (K = 1).  All mutually independent ones go out
  // as ONE launch (rank1_many_kernel), with their `p - r*g` and bias updates
  {
    std::vector<int> r1;
    for (size_t gi = 0; gi < pl.gs.size(); ++gi) {
      Gr& g = pl.gs[gi];
      if (!g.gemm || !g.wgrad_like || g.act || g.dact || g.bias || g.loss_kind || g.tail >= 0) continue;
      if (g.cin && g.beta != 1.0) continue;
      GmulPlan gp;
      dry_plan(pl.ns[g.anchor].n, gp);
      if (!gp.exact || gp.zero || gp.p.K != 1 || gp.p.batch != 1 || (gp.p.a_sm != 1 && gp.p.M != 1) ||
          (gp.p.b_sn != 1 && gp.p.N != 1))
        continue;
      bool indep = true;
      for (int o : r1) indep = indep && !path_between(pl, pl.gs[o], g) && !path_between(pl, g, pl.gs[o]);
      if (indep && (int)r1.size() < RANK1_MAX_LAYERS) r1.push_back((int)gi);
    }
    if (r1.size() >= 2) {
      std::vector<int> deps;
      for (int gi : r1)
        for (int d : pl.gs[gi].deps)
          if (std::find(deps.begin(), deps.end(), d) == deps.end()) deps.push_back(d);
      for (int gi : r1) {
        pl.gs[gi].r1 = r1[0];
        pl.gs[gi].deps = deps;
        pl.gs[gi].wgrad_like = false;  // not a pair candidate any more
      }
      pl.gs[r1[0]].r1_members = r1;
    }
  }
  // two independent weight-gradient GEMMs go out as one launch when the pair kernel takes their shapes
  std::vector<int> wg;
  for (size_t gi = 0; gi < pl.gs.size(); ++gi)
    if (pl.gs[gi].gemm && pl.gs[gi].wgrad_like) wg.push_back((int)gi);
  for (size_t a = 0; a < wg.size(); ++a)
    for (size_t b = a + 1; b < wg.size(); ++b) {
      Gr &g1 = pl.gs[wg[a]], &g2 = pl.gs[wg[b]];
      if (g1.pair >= 0 || g2.pair >= 0) continue;
      if (path_between(pl, g1, g2) || path_between(pl, g2, g1)) continue;
      // a third group between them (g1 -> x -> g2) is impossible without a path g1 -> g2
      GmulPlan p1, p2;
      dry_plan(pl.ns[g1.anchor].n, p1);
      dry_plan(pl.ns[g2.anchor].n, p2);
      if (!p1.exact || !p2.exact || !gemm_small_route(p1.p) || !gemm_small_route(p2.p)) continue;
      g1.pair = wg[b];
      g2.pair = wg[a];
      for (int d : g2.deps)
        if (std::find(g1.deps.begin(), g1.deps.end(), d) == g1.deps.end()) g1.deps.push_back(d);
      g2.deps = g1.deps;
    }
}

// to_copy_into destinations: produce the source straight into the destination when the source is an output of a
// fused launch, nothing else needs it, and every other reader of the destination's memory in this flush can be
// ordered before the launch
static void plan_forwarding(Plan& pl) {
  for (size_t i = 0; i < pl.ns.size(); ++i) {
    PN& pn = pl.ns[i];
    if (!pn.copy_dst || pn.demanded || !pn.cons.empty()) continue;
    Gr& g = pl.gs[pn.group];
    if (!g.gemm || !((int)i == g.out || (int)i == g.rs || (int)i == g.tail || (int)i == g.loss_node)) continue;
    to_tensor d = pn.copy_dst;
    bool ok = true;
    std::vector<int> first;  // groups that must run before this one
    for (size_t k = 0; k < pl.ns.size() && ok; ++k) {
      const PN& o = pl.ns[k];
      for (size_t j = 0; j < o.n->in.size() && ok; ++j) {
        to_tensor x = o.n->in[j];
        if (o.prod[j] >= 0 || !x->ptr || !overlaps(x, d)) continue;
        if (o.group == pn.group) {
          // inside the launch only an element-for-element alias is safe: Cin (or the bias being updated)
          const bool alias = x->ptr == d->ptr && full_like(x, d) &&
                             (((int)i == g.out && g.cin == x) || ((int)i == g.rs && g.rs_in == x));
          if (!alias) ok = false;
        } else if (g.r1 >= 0 && pl.gs[o.group].r1 == g.r1) {
          ok = false;  // another layer of the same launch reads it (never the case for a network's own parameters)
        } else if (path_between(pl, g, pl.gs[o.group]) || (g.pair >= 0 && path_between(pl, pl.gs[g.pair], pl.gs[o.group]))) {
          ok = false;  // that reader needs this launch's result: it cannot come first
        } else {
          first.push_back(o.group);
        }
      }
    }
    if (!ok) continue;
    pn.fwd = true;
    for (int f : first) {
      if (g.r1 >= 0)
        for (int m : pl.gs[g.r1].r1_members)
          if (m != f && std::find(pl.gs[m].deps.begin(), pl.gs[m].deps.end(), f) == pl.gs[m].deps.end())
            pl.gs[m].deps.push_back(f);
      if (std::find(g.deps.begin(), g.deps.end(), f) == g.deps.end()) g.deps.push_back(f);
      if (g.pair >= 0 && f != g.pair) {
        Gr& h = pl.gs[g.pair];
        if (std::find(h.deps.begin(), h.deps.end(), f) == h.deps.end()) h.deps.push_back(f);
      }
    }
  }
}

static bool topo_order(Plan& pl, std::vector<int>& order) {
  const int G = (int)pl.gs.size();
  std::vector<int> state(G, 0);
  order.clear();
  // iterative DFS; a pair is one unit (deps were merged)
  for (int root = 0; root < G; ++root) {
    if (state[root]) continue;
    std::vector<std::pair<int, size_t>> st{{root, 0}};
    state[root] = 1;
    while (!st.empty()) {
      auto& [g, k] = st.back();
      if (k < pl.gs[g].deps.size()) {
        int d = pl.gs[g].deps[k++];
        if (pl.gs[g].pair == d) continue;
        if (state[d] == 1) return false;  // cycle
        if (state[d] == 0) {
          state[d] = 1;
          st.push_back({d, 0});
        }
      } else {
        state[g] = 2;
        order.push_back(g);
        st.pop_back();
      }
    }
  }
  return true;
}

static void flush(const std::vector<to_tensor>& demand, const std::vector<std::pair<to_tensor, to_tensor>>& copies) {
  const auto t_begin = std::chrono::steady_clock::now();
  struct Timer {
    std::chrono::steady_clock::time_point t0;
    ~Timer() { g_stats[5] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
  } timer{t_begin};
  std::vector<to_tensor> roots = demand;
  for (auto& c : copies) roots.push_back(c.second);
  TO_CHECK(gemm_small_chain_status() == 0, TO_ERR_HIP,
           "a grid barrier of a chained step launch timed out (code " + std::to_string(gemm_small_chain_status()) +
               "): the results of that step are invalid; set TOPS_STEP_CHAIN=0");
  TO_CHECK(gemm_t32_take_failure() == 0, TO_ERR_HIP,
           "the joined forward + loss-head launch gave up waiting for a row block's tiles (are CUs masked below one round of the "
           "grid?): the outputs of that launch are invalid; the joined form is off for the rest of this process (TOPS_STEP_SEAM=0 "
           "turns it off from the start)");
  TO_CHECK(gemm_small_seam_take_failure() == 0, TO_ERR_HIP,
           "the joined forward + loss-head launch (TOPS_STEP_SEAM) gave up waiting for a row block: the outputs of that launch "
           "are invalid; the seam is off for the rest of this process");
  Plan pl;
  collect(pl, roots);
  if (pl.ns.empty()) return;
  g_stats[3]++;
  for (to_tensor t : demand) {
    const int i = pn_of(pl, t);
    if (i >= 0) pl.ns[i].demanded = true;
  }
  for (auto& c : copies) {
    const int i = pn_of(pl, c.second);
    if (i < 0) continue;
    pl.ns[i].copy_dst = c.first;  // (lazy_copy_into passes each deferred result once, never a view)
  }
  // The plan holds a reference to every handle it touches until it is done.  A handle in it may otherwise die midway:
  // resolving a deferred view releases the view's reference to its base (an executed `gmul` whose only holder was its
  // `transp` view), and dropping one node releases the inputs it kept alive -- while `finish` and `pl.ns` still point there.
  // (Inside a scope the memo table used to hide this; values demanded after the scope closed do not have that cover.)
  // (library-side references: they do not make a handle look held by the host)
  struct Held {
    std::vector<to_tensor> v;
    ~Held() {
      for (to_tensor h : v) release_int(h);
    }
  } held;
  for (PN& pn : pl.ns) {
    retain_int(pn.h);
    held.v.push_back(pn.h);
  }
  std::vector<int> order;
  std::vector<uint64_t> sig;
  uint64_t hash = 0;
  const CachedPlan* hit = nullptr;
  if (plan_cache_on()) {
    plan_signature(pl, copies, sig);
    hash = sig_hash(sig);
    hit = plan_find(sig, hash);
  }
  if (hit) {
    ++g_plan_cache_hits;
    plan_instantiate(*hit, pl, order);
  } else {
    ++g_plan_cache_misses;
    std::vector<std::pair<int, int>> dlog;
    compute_ancestors(pl);
    plan_groups(pl, dlog);
    plan_forwarding(pl);
    if (!topo_order(pl, order)) {
      // ordering readers of a forwarding destination first closed a cycle: give the forwarding up
      for (PN& pn : pl.ns) pn.fwd = false;
      for (Gr& g : pl.gs) g.deps.clear();
      for (size_t gi = 0; gi < pl.gs.size(); ++gi) {
        Gr& g = pl.gs[gi];
        for (int m : g.mem)
          for (int q : pl.ns[m].prod)
            if (q >= 0 && pl.ns[q].group != (int)gi &&
                std::find(g.deps.begin(), g.deps.end(), pl.ns[q].group) == g.deps.end())
              g.deps.push_back(pl.ns[q].group);
        if (g.pair >= 0) pl.gs[g.pair].pair = -1, g.pair = -1;
        g.r1 = -1;
        g.r1_members.clear();
      }
      TO_CHECK(topo_order(pl, order), TO_ERR_STATE, "internal: recorded graph has a cycle");
    }
    if (plan_cache_on()) plan_store(pl, order, std::move(sig), hash, dlog);
  }
  if (debug_on()) dump_plan(pl);
  g_stats[4] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_begin).count();
  Exec ex(pl);
  std::exception_ptr err;
  try {
    ex.run_all(order);
    ex.drain();
    // sources that could not be produced in place: one copy launch for all of them
    std::vector<const void*> sp;
    std::vector<void*> dp;
    std::vector<int64_t> dw;
    for (PN& pn : pl.ns)
      if (pn.copy_dst && !pn.copied) {
        if (!pn.h->ptr) ex.run_single((int)(&pn - pl.ns.data()));
        if (pn.copy_dst->total() == 0) continue;
        sp.push_back(pn.h->ptr);
        dp.push_back(pn.copy_dst->ptr);
        dw.push_back(pn.copy_dst->total() * (int64_t)pn.copy_dst->esize() / 4);
      }
    for (size_t b = 0; b < sp.size(); b += 16) {
      const int m = (int)std::min<size_t>(16, sp.size() - b);
      describe_other();
      launch_multi_copy(m, sp.data() + b, dp.data() + b, dw.data() + b, S());
    }
    // A result produced straight into its destination still stands for a VALUE.  If its handle is asked for later (the
    // host holds it, or an op recorded afterwards reads it) the recorded op must not run again: one of its inputs may be
    // the very destination it has just overwritten (b' = b - r g produced into b would apply the update twice).  From
    // here on the handle means "the contents of the destination": a recorded `1 * dst`, which the write hazards
    // (before_write / stale_after_write) run before dst changes again; nothing is launched unless someone asks.
    for (PN& pn : pl.ns) {
      if (!pn.fwd || !pn.copied || pn.h->ptr || !pn.h->node) continue;
      Node* n = pn.n;
      to_tensor d = pn.copy_dst;
      if (full_like(d, pn.h)) {
        retain_int(d);
        for (to_tensor x : n->in) release_int(x);
        n->in.assign(1, d);
        if (n->d.f) expr_release(n->d.f);
        n->d = NodeDesc{};
        n->d.op = N_SCALE;
        n->d.alpha = 1.0;
      } else {
        // (a destination of another shape, e.g. a flat parameter view: the handle gets a copy of its own -- always:
        //  even when the host has let go of it, the scope's memo table may hand it out again, and re-running its
        //  recorded op would read the destination it has just overwritten)
        alloc_storage(pn.h);
        const void* sp1 = d->ptr;
        void* dp1 = pn.h->ptr;
        int64_t dw1 = d->total() * (int64_t)d->esize() / 4;
        describe_other();
        if (d->total() > 0) launch_multi_copy(1, &sp1, &dp1, &dw1, S());
        ex.finish.push_back(pn.h);
      }
    }
  } catch (...) {
    err = std::current_exception();
    try {
      ex.drain();  // what was already planned into held-back launches still has to produce its outputs
    } catch (...) {
    }
  }
  // values that exist now no longer need their recorded op (this releases the inputs the op kept alive)
  for (to_tensor h : ex.finish) lazy_drop_node(h);
  if (err) std::rethrow_exception(err);
}

void ensure(to_tensor t) {
  if (t->ptr) return;
  if (producer(t) == nullptr) return;  // a view that could be resolved
  to_tensor base = t->view_base ? t->view_base : t;
  flush({base}, {});
  if (!t->ptr) resolve_view(t);
  TO_CHECK(t->ptr != nullptr, TO_ERR_STATE, "internal: deferred value was not produced");
}

void ensure_all(int n, const to_tensor* ts) {
  std::vector<to_tensor> need;
  for (int i = 0; i < n; ++i)
    if (ts[i] && !ts[i]->ptr && producer(ts[i])) need.push_back(ts[i]->view_base ? ts[i]->view_base : ts[i]);
  if (!need.empty()) flush(need, {});
  for (int i = 0; i < n; ++i)
    if (ts[i] && !ts[i]->ptr) {
      resolve_view(ts[i]);
      TO_CHECK(ts[i]->ptr != nullptr, TO_ERR_STATE, "internal: deferred value was not produced");
    }
}

// live results of `owner` (0: of every thread) that no recorded op consumes
static std::vector<to_tensor> live_sinks(uint64_t owner) {
  std::unordered_map<Node*, char> consumed;
  for (Node* n = g_head; n; n = n->next)
    for (to_tensor x : n->in) {
      if (x->ptr) continue;
      to_tensor_s* b = x->view_base ? x->view_base : x;
      if (b->node) consumed[b->node] = 1;
    }
  std::vector<to_tensor> out;
  for (Node* n = g_head; n; n = n->next)
    if ((owner == 0 || n->owner == owner) && !consumed.count(n) && is_live(n->out)) out.push_back(n->out);
  return out;
}

void lazy_flush_sinks() {
  if (!g_head) return;
  std::vector<to_tensor> s = live_sinks(this_thread());
  if (!s.empty()) flush(s, {});
}

void lazy_flush_all() {
  if (!g_head) return;
  std::vector<to_tensor> s = live_sinks(0);
  if (!s.empty()) flush(s, {});
}

// recorded ops that read memory about to be overwritten, and everything reachable from them that the host can
// still ask for (a handle it holds, or a view of one): they must see the old contents
static std::vector<to_tensor> stale_after_write(int n, const to_tensor* dsts, const std::vector<to_tensor>& except) {
  if (!g_head) return {};
  // the destinations' byte ranges, once; most operands are rejected against their hull
  std::vector<std::pair<const char*, const char*>> dr;
  const char *hull_lo = nullptr, *hull_hi = nullptr;
  for (int i = 0; i < n; ++i) {
    if (!dsts[i]->ptr) continue;
    const char *lo, *hi;
    mem_range(dsts[i], &lo, &hi);
    if (lo == hi) continue;
    dr.emplace_back(lo, hi);
    if (!hull_lo || lo < hull_lo) hull_lo = lo;
    if (!hull_hi || hi > hull_hi) hull_hi = hi;
  }
  if (dr.empty()) return {};
  auto reads_dst = [&](to_tensor x) {
    const char* p = static_cast<const char*>(x->ptr);
    if (p >= hull_hi) return false;
    const char *lo, *hi;
    mem_range(x, &lo, &hi);
    if (hi <= hull_lo) return false;
    for (auto& r : dr)
      if (lo < r.second && r.first < hi) return true;
    return false;
  };
  // pending nodes in recording order (the list is newest first)
  std::vector<Node*> nodes;
  for (Node* q = g_head; q; q = q->next) nodes.push_back(q);
  bool sorted = true;
  for (size_t i = 1; i < nodes.size() && sorted; ++i) sorted = nodes[i - 1]->seq > nodes[i]->seq;
  if (sorted) std::reverse(nodes.begin(), nodes.end());
  else std::sort(nodes.begin(), nodes.end(), [](const Node* a, const Node* b) { return a->seq < b->seq; });
  static uint64_t g_write_epoch = 0;
  const uint64_t epoch = ++g_write_epoch;  // (marks "hit" nodes in Node::write_mark: pn_of's plan_epoch is not touched)
  bool any = false;
  for (Node* q : nodes) {
    bool h = false;
    for (to_tensor x : q->in) {
      if (x->ptr) {
        h = reads_dst(x);
      } else {
        to_tensor_s* b = x->view_base ? x->view_base : x;
        h = b->node && b->node->write_mark == epoch;
      }
      if (h) break;
    }
    if (h) {
      q->write_mark = epoch;
      any = true;
    }
  }
  std::vector<to_tensor> out;
  if (!any) return out;
  for (Node* q : nodes)
    if (q->write_mark == epoch && host_reachable(q->out) &&
        std::find(except.begin(), except.end(), q->out) == except.end())
      out.push_back(q->out);
  return out;
}

void before_write(to_tensor t) {
  if (!g_head || !t->ptr) return;
  std::vector<to_tensor> s = stale_after_write(1, &t, {});
  if (!s.empty()) flush(s, {});
}

void lazy_copy_into(int n, const to_tensor* dsts, const to_tensor* srcs) {
  std::vector<std::pair<to_tensor, to_tensor>> pending;
  std::vector<to_tensor> psrc;
  for (int i = 0; i < n; ++i) {
    ensure(dsts[i]);
    dsts[i]->id = fresh_id();  // new contents
    if (!srcs[i]->ptr && srcs[i]->node && std::find(psrc.begin(), psrc.end(), srcs[i]) == psrc.end()) {
      pending.push_back({dsts[i], srcs[i]});
      psrc.push_back(srcs[i]);
    }
  }
  if (g_head) {
    std::vector<to_tensor> stale = stale_after_write(n, dsts, psrc);
    // (only the sources and what must read the old contents first: other deferred values the host holds are not
    //  demanded by this call -- under a garbage collector "held" does not mean "wanted", see scope_end)
    if (!pending.empty() || !stale.empty()) flush(stale, pending);
  }
  // sources that already existed
  std::vector<std::unique_ptr<Holder>> keep;
  std::vector<const void*> sp;
  std::vector<void*> dp;
  std::vector<int64_t> dw;
  for (int i = 0; i < n; ++i) {
    bool was_pending = false;
    for (auto& c : pending) was_pending = was_pending || (c.first == dsts[i] && c.second == srcs[i]);
    if (was_pending || dsts[i]->total() == 0) continue;
    ensure(srcs[i]);
    keep.emplace_back(new Holder(contiguous(srcs[i])));
    sp.push_back(keep.back()->t->ptr);
    dp.push_back(dsts[i]->ptr);
    dw.push_back(dsts[i]->total() * (int64_t)dsts[i]->esize() / 4);
  }
  for (size_t b = 0; b < sp.size(); b += 16) {
    const int m = (int)std::min<size_t>(16, sp.size() - b);
    describe_other();
    launch_multi_copy(m, sp.data() + b, dp.data() + b, dw.data() + b, S());
  }
}

}  // namespace to
