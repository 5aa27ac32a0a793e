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

// ---- caller memory <-> device ------------------------------------------------------------------------------------------
// Why the library stages: a downloaded result once carried HOST heap bytes -- the freed fp64 temporary of the numpy
// reference, in ~8 KiB pieces 128 KiB apart -- after hipMemcpyAsync(pageable, device) + hipStreamSynchronize had returned,
// on boxes where eight processes shared the GPU and the host's memory manager was busy (DESIGN_HISTORY.md 11.1, profiles/r05_stress/).
// With pageable memory the runtime lets the copy engine write the caller's pages through a user-pointer mapping it makes
// on the fly; a page the kernel moves meanwhile takes the bytes with it or does not.  Memory from hipHostMalloc is locked
// when it is allocated, so: the engine only ever sees the library's two pinned chunks, and the CPU moves the bytes between
// them and the caller's memory (the next chunk's DMA overlaps the copy).
namespace {
constexpr size_t STAGE_CHUNK = 4u << 20;
struct Staging {
  char* buf[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
  int on = -1;
  TransferStats st{0, 0, 0, 0};
} g_stage;

bool staging_on() {
  if (g_stage.on < 0) {
    const char* e = std::getenv("TOPS_PINNED_STAGING");
    g_stage.on = !(e && e[0] == '0');
  }
  return g_stage.on != 0;
}

void staging_init() {
  if (g_stage.buf[0]) return;
  for (int k = 0; k < 2; ++k) {
    TO_HIP(hipHostMalloc(reinterpret_cast<void**>(&g_stage.buf[k]), STAGE_CHUNK, hipHostMallocDefault));
    TO_HIP(hipEventCreateWithFlags(&g_stage.ev[k], hipEventDisableTiming));
  }
}

// memory the caller pinned itself is as safe as ours: no second copy.  The WHOLE range has to be pinned: a registered
// prefix of a larger pageable array (hipHostRegister over its first pages) answers "host memory" for its first byte and
// would hand the pageable rest to the copy engine.  So: both ends must be host memory the runtime knows, and
// the allocation / registration that holds the first byte (hipMemGetAddressRange: base and size of the runtime's memory
// object) must reach past the last one.  Any "don't know" is answered with the staged path.
bool caller_pinned(const void* p, size_t nbytes) {
  if (nbytes < (64u << 10)) return false;  // (not worth the queries)
  auto pinned_at = [](const void* q) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, q) != hipSuccess) {
      (void)hipGetLastError();  // "not a registered pointer" is the answer, not an error
      return false;
    }
    return a.type == hipMemoryTypeHost;
  };
  const char* first = static_cast<const char*>(p);
  const char* last = first + (nbytes - 1);
  if (!pinned_at(first) || !pinned_at(last)) return false;
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange(&base, &size, const_cast<void*>(p)) != hipSuccess || !base) {
    (void)hipGetLastError();
    return false;
  }
  const char* b0 = static_cast<const char*>(base);
  return b0 <= first && last < b0 + size;
}
}  // namespace

void host_to_device(void* dst, const void* host, size_t nbytes, hipStream_t s) {
  if (!nbytes) return;
  if (!staging_on() || caller_pinned(host, nbytes)) {
    g_stage.st.direct_calls++;
    g_stage.st.direct_bytes += (long long)nbytes;
    TO_HIP(hipMemcpyAsync(dst, host, nbytes, hipMemcpyHostToDevice, s));
    TO_HIP(hipStreamSynchronize(s));
    return;
  }
  staging_init();
  g_stage.st.staged_calls++;
  g_stage.st.staged_bytes += (long long)nbytes;
  bool busy[2] = {false, false};
  int k = 0;
  for (size_t off = 0; off < nbytes; k ^= 1) {
    const size_t c = nbytes - off < STAGE_CHUNK ? nbytes - off : STAGE_CHUNK;
    if (busy[k]) TO_HIP(hipEventSynchronize(g_stage.ev[k]));  // the engine has read this chunk's previous contents
    std::memcpy(g_stage.buf[k], static_cast<const char*>(host) + off, c);
    TO_HIP(hipMemcpyAsync(static_cast<char*>(dst) + off, g_stage.buf[k], c, hipMemcpyHostToDevice, s));
    TO_HIP(hipEventRecord(g_stage.ev[k], s));
    busy[k] = true;
    off += c;
  }
  TO_HIP(hipStreamSynchronize(s));
}

void device_to_host(void* host, const void* src, size_t nbytes, hipStream_t s) {
  if (!nbytes) return;
  if (!staging_on() || caller_pinned(host, nbytes)) {
    g_stage.st.direct_calls++;
    g_stage.st.direct_bytes += (long long)nbytes;
    TO_HIP(hipMemcpyAsync(host, src, nbytes, hipMemcpyDeviceToHost, s));
    TO_HIP(hipStreamSynchronize(s));
    return;
  }
  staging_init();
  g_stage.st.staged_calls++;
  g_stage.st.staged_bytes += (long long)nbytes;
  size_t p_off[2] = {0, 0}, p_len[2] = {0, 0};
  int k = 0;
  auto drain = [&](int j) {  // chunk j has landed in pinned memory: hand it to the caller
    if (!p_len[j]) return;
    TO_HIP(hipEventSynchronize(g_stage.ev[j]));
    std::memcpy(static_cast<char*>(host) + p_off[j], g_stage.buf[j], p_len[j]);
    p_len[j] = 0;
  };
  for (size_t off = 0; off < nbytes; k ^= 1) {
    const size_t c = nbytes - off < STAGE_CHUNK ? nbytes - off : STAGE_CHUNK;
    drain(k);
    TO_HIP(hipMemcpyAsync(g_stage.buf[k], static_cast<const char*>(src) + off, c, hipMemcpyDeviceToHost, s));
    TO_HIP(hipEventRecord(g_stage.ev[k], s));
    p_off[k] = off;
    p_len[k] = c;
    off += c;
  }
  drain(k);      // the older of the two
  drain(k ^ 1);
}

// ---- small tables for batched launches -----------------------------------------------------------------------------------
namespace {
constexpr int TAB_SLOTS = 4;
constexpr size_t TAB_BYTES = 64u << 10;
struct Tables {
  char* host[TAB_SLOTS] = {nullptr};
  char* dev[TAB_SLOTS] = {nullptr};
  hipEvent_t ev[TAB_SLOTS] = {nullptr};
  bool busy[TAB_SLOTS] = {false};
  int next = 0;
} g_tab;
}  // namespace

const void* table_upload(const void* host, size_t bytes, hipStream_t s) {
  TO_CHECK(bytes <= TAB_BYTES, TO_ERR_UNSUPPORTED, "internal: pointer table larger than its slot");
  const int k = g_tab.next;
  g_tab.next = (k + 1) % TAB_SLOTS;
  if (!g_tab.host[k]) {
    TO_HIP(hipHostMalloc(reinterpret_cast<void**>(&g_tab.host[k]), TAB_BYTES, hipHostMallocDefault));
    TO_HIP(hipMalloc(reinterpret_cast<void**>(&g_tab.dev[k]), TAB_BYTES));
    TO_HIP(hipEventCreateWithFlags(&g_tab.ev[k], hipEventDisableTiming));
  }
  if (g_tab.busy[k]) TO_HIP(hipEventSynchronize(g_tab.ev[k]));   // the copy engine has read this slot's previous contents
  std::memcpy(g_tab.host[k], host, bytes);
  TO_HIP(hipMemcpyAsync(g_tab.dev[k], g_tab.host[k], bytes, hipMemcpyHostToDevice, s));
  TO_HIP(hipEventRecord(g_tab.ev[k], s));
  g_tab.busy[k] = true;
  return g_tab.dev[k];
}

static void tables_shutdown() {
  for (int k = 0; k < TAB_SLOTS; ++k) {
    if (g_tab.host[k]) (void)hipHostFree(g_tab.host[k]);
    if (g_tab.dev[k]) (void)hipFree(g_tab.dev[k]);
    if (g_tab.ev[k]) (void)hipEventDestroy(g_tab.ev[k]);
    g_tab.host[k] = g_tab.dev[k] = nullptr;
    g_tab.ev[k] = nullptr;
    g_tab.busy[k] = false;
  }
  g_tab.next = 0;
}

void staging_shutdown() {
  tables_shutdown();
  for (int k = 0; k < 2; ++k) {
    if (g_stage.buf[k]) (void)hipHostFree(g_stage.buf[k]);
    if (g_stage.ev[k]) (void)hipEventDestroy(g_stage.ev[k]);
    g_stage.buf[k] = nullptr;
    g_stage.ev[k] = nullptr;
  }
  g_stage.on = -1;
}

TransferStats transfer_stats() { return g_stage.st; }

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

void alloc_storage_shared(int n, const to_tensor* ts) {
  size_t total = 0;
  for (int i = 0; i < n; ++i) {
    TO_CHECK(ts[i]->ptr == nullptr && ts[i]->buf == nullptr, TO_ERR_STATE, "handle already has storage");
    const size_t b = (size_t)ts[i]->total() * ts[i]->esize();
    TO_CHECK(b % 16 == 0, TO_ERR_STATE, "internal: shared storage needs slices of whole 16-byte pieces");
    total += b;
  }
  if (n == 0) return;
  Buffer* buf = pool_alloc(total);   // (refs = 1: the first handle's)
  size_t off = 0;
  for (int i = 0; i < n; ++i) {
    if (i > 0) buf->refs.fetch_add(1);
    ts[i]->buf = buf;
    ts[i]->ptr = static_cast<char*>(buf->ptr) + off;
    off += (size_t)ts[i]->total() * ts[i]->esize();
    keep_for_capture(ts[i]);
  }
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
