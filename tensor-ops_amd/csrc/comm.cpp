// Data-parallel exchange on the C ABI (SURVEY.md 8(e), 8(b) "to_comm_init / to_allreduce_sum"): ONE
// all-reduce(sum) of the flat weight-gradient buffer per step, RCCL over xGMI, enqueued on the library
// stream.  A host that is not Python (the Haskell shim) gets the collective without torch.distributed;
// the Python harness keeps torch.distributed as its default transport and can switch to this one.
//
// RCCL is loaded with dlopen on first use: a single-GPU user never needs it, and a process that also
// uses torch's bundled RCCL does not get a second copy mapped unless it asks for this path.
#include <dlfcn.h>

#include <cstring>

#include "common.hpp"

namespace to {

namespace {
typedef struct ncclComm* comm_t;
struct unique_id { char internal[128]; };   // NCCL_UNIQUE_ID_BYTES
enum { kFloat32 = 7, kFloat64 = 8, kSum = 0 };  // ncclFloat32 / ncclFloat64 / ncclSum

struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(unique_id*) = nullptr;
  int (*CommInitRank)(comm_t*, int, unique_id, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  int (*CommDestroy)(comm_t) = nullptr;
  int (*CommCount)(comm_t, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  comm_t comm = nullptr;
  int rank = 0, world = 0;
};
Rccl g_rccl;

// All symbols are resolved into a local copy and committed together: a library that lacks one leaves g_rccl untouched
// (and is closed again), so the next call reports the same error instead of finding `lib` set and calling a null pointer
// (ADVICE r5).  ncclCommCount is optional: without it comm_world() answers with what to_comm_init was given.
void load() {
  if (g_rccl.lib) return;
  const char* env = getenv("TOPS_RCCL_LIB");
  const char* names[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* lib = nullptr;
  for (const char* n : names) {
    if (!n) continue;
    lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (lib) break;
  }
  TO_CHECK(lib != nullptr, TO_ERR_STATE, std::string("cannot load librccl.so: ") + dlerror());
  Rccl r;
  const char* missing = nullptr;
  auto sym = [&](const char* s, bool required = true) {
    void* p = dlsym(lib, s);
    if (!p && required && !missing) missing = s;
    return p;
  };
  r.GetUniqueId = reinterpret_cast<int (*)(unique_id*)>(sym("ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<int (*)(comm_t*, int, unique_id, int)>(sym("ncclCommInitRank"));
  r.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, comm_t, hipStream_t)>(sym("ncclAllReduce"));
  r.CommDestroy = reinterpret_cast<int (*)(comm_t)>(sym("ncclCommDestroy"));
  r.GetErrorString = reinterpret_cast<const char* (*)(int)>(sym("ncclGetErrorString"));
  r.CommCount = reinterpret_cast<int (*)(comm_t, int*)>(sym("ncclCommCount", false));
  if (missing) {
    const std::string what = std::string("librccl.so lacks ") + missing;
    dlclose(lib);
    fail(TO_ERR_STATE, what);
  }
  r.lib = lib;
  g_rccl = r;
}

void ok(int r, const char* what) {
  if (r != 0) fail(TO_ERR_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"));
}
}  // namespace

void comm_unique_id(void* out128) {
  load();
  unique_id id;
  ok(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(out128, &id, sizeof(id));
}

void comm_init(int rank, int world, const void* id128) {
  load();
  TO_CHECK(world >= 1 && rank >= 0 && rank < world, TO_ERR_ARG, "comm_init: bad rank / world size");
  TO_CHECK(g_rccl.comm == nullptr, TO_ERR_STATE, "comm_init: a communicator already exists");
  unique_id id;
  std::memcpy(&id, id128, sizeof(id));
  ok(g_rccl.CommInitRank(&g_rccl.comm, world, id, rank), "ncclCommInitRank");
  g_rccl.rank = rank;
  g_rccl.world = world;
}

void comm_allreduce_sum(to_tensor t, hipStream_t s) {
  TO_CHECK(g_rccl.comm != nullptr, TO_ERR_STATE, "comm_allreduce_sum: call to_comm_init first");
  TO_CHECK(t->contiguous(), TO_ERR_ARG, "comm_allreduce_sum needs a contiguous buffer");
  ok(g_rccl.AllReduce(t->ptr, t->ptr, (size_t)t->total(), t->dtype == TO_F64 ? kFloat64 : kFloat32, kSum,
                      g_rccl.comm, s),
     "ncclAllReduce");
  count_launch();
}

void comm_shutdown() {
  if (g_rccl.comm) {
    (void)g_rccl.CommDestroy(g_rccl.comm);
    g_rccl.comm = nullptr;
  }
}

// what RCCL itself says the communicator spans (ncclCommCount), not what the host passed to comm_init
int comm_world() {
  if (!g_rccl.comm) return 0;
  if (!g_rccl.CommCount) return g_rccl.world;   // (an RCCL without ncclCommCount: the host's own number)
  int n = 0;
  ok(g_rccl.CommCount(g_rccl.comm, &n), "ncclCommCount");
  TO_CHECK(n == g_rccl.world, TO_ERR_STATE, "RCCL reports " + std::to_string(n) + " ranks, to_comm_init was given " + std::to_string(g_rccl.world));
  return n;
}

}  // namespace to
