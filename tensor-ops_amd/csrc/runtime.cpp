// Runtime: device selection, the single stream, a size-class pool allocator
// (so every op can return a fresh immutable value without hipMalloc on the hot
// path -- and none at all during graph capture), and ref-counted handles.
#include <cstring>

#include "common.hpp"

namespace to {

Runtime& rt() {
  static Runtime r;
  return r;
}

static std::vector<std::unique_ptr<LaunchRec>>* g_recorder = nullptr;
std::vector<std::unique_ptr<LaunchRec>>* launch_recorder() { return g_recorder; }
void set_launch_recorder(std::vector<std::unique_ptr<LaunchRec>>* r) { g_recorder = r; }

std::recursive_mutex& lock() {
  static std::recursive_mutex m;
  return m;
}

static int size_class(size_t bytes) {
  int c = 8;  // 256 B minimum
  while (((size_t)1 << c) < bytes) ++c;
  return c;
}

Buffer* pool_alloc(size_t bytes) {
  Runtime& r = rt();
  TO_CHECK(r.inited, TO_ERR_STATE, "to_init has not been called");
  const int c = size_class(bytes ? bytes : 1);
  if (r.free_lists.size() <= (size_t)c) r.free_lists.resize(c + 1);
  auto* b = new Buffer();
  b->bytes = (size_t)1 << c;
  auto& fl = r.free_lists[c];
  if (!fl.empty()) {
    b->ptr = fl.back();
    fl.pop_back();
  } else {
    // hipMalloc is not a stream operation; with relaxed capture mode it is legal while
    // capturing, and a warm-up run normally populates the pool first anyway.
    hipError_t e = hipMalloc(&b->ptr, b->bytes);
    if (e != hipSuccess) {
      delete b;
      fail(TO_ERR_HIP, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
    }
    r.pool_bytes += (int64_t)b->bytes;
  }
  return b;
}

void buffer_release(Buffer* b) {
  if (!b) return;
  if (b->refs.fetch_sub(1) != 1) return;
  if (b->owned && b->ptr) {
    Runtime& r = rt();
    const int c = size_class(b->bytes);
    if (r.free_lists.size() <= (size_t)c) r.free_lists.resize(c + 1);
    // Single-stream discipline: every consumer of this memory was enqueued on rt().stream
    // before this point, and every future producer will be enqueued after it, so
    // immediate reuse is stream-ordered and safe.
    r.free_lists[c].push_back(b->ptr);
  }
  delete b;
}

static std::atomic<uint64_t> g_next_id{1};
uint64_t fresh_id() { return g_next_id++; }

static void keep_for_capture(to_tensor t);

to_tensor new_tensor(int rank, const int64_t* dims, int64_t batch, int dtype) {
  TO_CHECK(rank >= 0 && rank <= TO_MAX_RANK, TO_ERR_ARG, "rank must be 0..8");
  TO_CHECK(batch >= 0, TO_ERR_ARG, "negative batch");
  TO_CHECK(dtype == TO_F32 || dtype == TO_F64, TO_ERR_ARG, "unknown dtype");
  auto* t = new to_tensor_s();
  t->rank = rank;
  t->dtype = dtype;
  int64_t n = 1;
  for (int i = 0; i < rank; ++i) {
    if (dims[i] < 0 || dims[i] > 2147483647LL) {
      delete t;
      fail(TO_ERR_SHAPE, "dimension out of range (0..2^31-1)");
    }
    t->dims[i] = dims[i];
    n *= dims[i];
  }
  int64_t s = 1;
  for (int i = rank - 1; i >= 0; --i) {
    t->strides[i] = s;
    s *= t->dims[i];
  }
  t->batch = batch;
  t->bstride = n;
  try {
    t->buf = pool_alloc((size_t)(n * (batch > 0 ? batch : 1)) * t->esize());
  } catch (...) {
    delete t;
    throw;
  }
  t->ptr = t->buf->ptr;
  t->id = g_next_id++;
  rt().live_handles++;
  keep_for_capture(t);
  return t;
}

// A handle with a shape and no storage: the result of a recorded op (lazy.cpp attaches the node).
to_tensor new_deferred(int rank, const int64_t* dims, int64_t batch, int dtype) {
  TO_CHECK(rank >= 0 && rank <= TO_MAX_RANK, TO_ERR_ARG, "rank must be 0..8");
  TO_CHECK(batch >= 0, TO_ERR_ARG, "negative batch");
  TO_CHECK(dtype == TO_F32 || dtype == TO_F64, TO_ERR_ARG, "unknown dtype");
  auto* t = new to_tensor_s();
  t->rank = rank;
  t->dtype = dtype;
  int64_t n = 1;
  for (int i = 0; i < rank; ++i) {
    if (dims[i] < 0 || dims[i] > 2147483647LL) {
      delete t;
      fail(TO_ERR_SHAPE, "dimension out of range (0..2^31-1)");
    }
    t->dims[i] = dims[i];
    n *= dims[i];
  }
  int64_t s = 1;
  for (int i = rank - 1; i >= 0; --i) {
    t->strides[i] = s;
    s *= t->dims[i];
  }
  t->batch = batch;
  t->bstride = n;
  t->id = g_next_id++;
  rt().live_handles++;
  return t;
}

static void keep_for_capture(to_tensor t) {
  if (rt().capturing) {
    t->refs.fetch_add(1);
    rt().capture_kept.push_back(t);
  }
}

void alloc_storage(to_tensor t) {
  TO_CHECK(t->ptr == nullptr && t->buf == nullptr, TO_ERR_STATE, "handle already has storage");
  t->buf = pool_alloc((size_t)t->total() * t->esize());
  t->ptr = t->buf->ptr;
  keep_for_capture(t);
}

void adopt_storage(to_tensor t, to_tensor from) {
  TO_CHECK(t->ptr == nullptr && t->buf == nullptr, TO_ERR_STATE, "handle already has storage");
  TO_CHECK(from->ptr != nullptr && from->contiguous() && same_shape(t, from) && t->batch == from->batch &&
               t->dtype == from->dtype,
           TO_ERR_STATE, "adopt_storage: layouts differ: " + shape_str(t) + " vs " + shape_str(from));
  t->buf = from->buf;
  if (t->buf) t->buf->refs.fetch_add(1);
  t->ptr = from->ptr;
  keep_for_capture(t);
}

to_tensor new_view(to_tensor base, int rank, const int64_t* dims, const int64_t* strides,
                   int64_t batch, int64_t bstride, int64_t offset) {
  auto* t = new to_tensor_s();
  t->rank = rank;
  t->dtype = base->dtype;
  for (int i = 0; i < rank; ++i) {
    t->dims[i] = dims[i];
    t->strides[i] = strides[i];
  }
  t->batch = batch;
  t->bstride = bstride;
  if (base->pending()) {
    // a view of a value that does not exist yet: remember where it will be (resolved by to::ensure)
    to_tensor_s* root = base->view_base ? base->view_base : base;
    t->view_base = root;
    t->view_off = (base->view_base ? base->view_off : 0) + offset;
    root->refs.fetch_add(1);
    root->int_refs++;
    root->dviews.push_back(t);
  } else {
    t->buf = base->buf;
    if (t->buf) t->buf->refs.fetch_add(1);
    t->ptr = base->at(offset);
  }
  t->id = g_next_id++;
  rt().live_handles++;
  if (!t->pending()) keep_for_capture(t);
  return t;
}

void retain(to_tensor t) { t->refs.fetch_add(1); }

void lazy_drop_node(to_tensor t);  // lazy.cpp: unlinks and frees t->node (releases its inputs)

void release(to_tensor t) {
  if (!t) return;
  if (t->refs.fetch_sub(1) != 1) return;
  if (t->node) lazy_drop_node(t);
  if (t->view_base) {
    to_tensor_s* b = t->view_base;
    t->view_base = nullptr;
    b->int_refs--;
    for (size_t i = 0; i < b->dviews.size(); ++i)
      if (b->dviews[i] == t) {
        b->dviews[i] = b->dviews.back();
        b->dviews.pop_back();
        break;
      }
    release(b);
  }
  buffer_release(t->buf);
  rt().live_handles--;
  delete t;
}

bool same_shape(to_tensor a, to_tensor b) {
  if (a->rank != b->rank) return false;
  for (int i = 0; i < a->rank; ++i)
    if (a->dims[i] != b->dims[i]) return false;
  return true;
}

std::string shape_str(to_tensor t) {
  std::string s = t->batch > 0 ? "[B=" + std::to_string(t->batch) + ";" : "[";
  for (int i = 0; i < t->rank; ++i) s += (i ? "," : "") + std::to_string(t->dims[i]);
  return s + (t->dtype == TO_F64 ? "]:f64" : "]");
}

to_tensor contiguous(to_tensor x) {
  if (x->contiguous()) {
    retain(x);
    return x;
  }
  to_tensor out = new_tensor(x->rank, x->dims, x->batch, x->dtype);
  int64_t d[TO_MAX_RANK + 1], st[TO_MAX_RANK + 1];
  int r = 0;
  if (x->batch > 0) {
    d[0] = x->batch;
    st[0] = x->bstride;
    r = 1;
  }
  for (int i = 0; i < x->rank; ++i, ++r) {
    d[r] = x->dims[i];
    st[r] = x->strides[i];
  }
  try {
    launch_copy_strided(x->dtype, x->ptr, out->ptr, r, d, st, rt().stream);
  } catch (...) {
    release(out);
    throw;
  }
  return out;
}

}  // namespace to
