// One-shot two-phase all-reduce over peer-mapped buffers ("each GPU pushes S/R slices to R-1 peers
// over all links, reduces, then all-gathers") -- the xGMI-native exchange for the data-parallel batched gradTOp:
// 203,532 floats (814 KB) per step, latency-bound, and on a ring bound by ONE of the seven links per hop.  Here
// every rank talks to every peer at once:
//   phase 1  rank r writes slice j of its gradient into peer j's inbox slot r        (R-1 links busy, 1/R of the data each)
//   phase 2  rank j sums the R versions of slice j IN RANK ORDER (so the result does not depend on arrival
//            order and every replica gets the same bits) and writes the sum into every rank's outbox
//   phase 3  every rank holds the whole reduced gradient; the consumer runs in the same kernel:
//            g <- sum, and/or p <- p - rate * sum   (the SGD update of FeedForward.hs:141-147, no extra launch)
// One process per GPU; buffers are exchanged as hipIpc handles over whatever transport the host has (the 64
// bytes per rank travel like the RCCL unique id).  Synchronisation is by epoch-stamped flags in the peers'
// fine-grained memory (system-scope release/acquire); a wait that exceeds the watchdog makes the kernel give up and
// report it instead of hanging the device.
//
// The reference has no distributed code at all; this is new capability next to the RCCL path
// (comm.cpp), selected with `bench.py --collective p2p`.
#include <cstring>

#include "common.hpp"

namespace to {

namespace {
constexpr int P2P_MAX_WORLD = 8;
constexpr int P2P_WGS = 128;           // co-resident by a wide margin (256 CUs); enough requests in flight for
                                       // fine-grained (uncached) memory: 128 x 256 lanes x 16 bytes
constexpr int P2P_FLAG_STRIDE = 16;    // one flag per 64-byte line

struct P2PLayout {          // offsets in bytes inside every rank's exchange buffer
  size_t flags1, flags2;    // [world] epoch stamps: "rank r's phase-1 / phase-2 writes into me are complete"
  size_t inbox;             // [world][slice_cap] elements
  size_t outbox;            // [world * slice_cap] elements: the reduced vector
  size_t total;
};

struct P2PState {
  bool ready = false;
  int rank = 0, world = 1;
  int dtype = TO_F32;
  int64_t max_elems = 0, slice_cap = 0;
  P2PLayout lay{};
  void* local = nullptr;
  void* peer[P2P_MAX_WORLD] = {nullptr};
  bool opened[P2P_MAX_WORLD] = {false};
  unsigned epoch = 0;
  unsigned* arrive = nullptr;   // local (device) counters for the in-kernel grid barriers
  int* status = nullptr;        // host-mapped: 0 ok, else which wait timed out
  int* status_dev = nullptr;
};
P2PState g_p2p;

P2PLayout layout_for(int64_t slice_cap, int world, size_t es) {
  P2PLayout l{};
  size_t off = 0;
  l.flags1 = off; off += (size_t)P2P_MAX_WORLD * P2P_FLAG_STRIDE * 4;
  l.flags2 = off; off += (size_t)P2P_MAX_WORLD * P2P_FLAG_STRIDE * 4;
  off = (off + 255) / 256 * 256;
  l.inbox = off; off += (size_t)world * slice_cap * es;
  off = (off + 255) / 256 * 256;
  l.outbox = off; off += (size_t)world * slice_cap * es;
  l.total = (off + 4095) / 4096 * 4096;
  return l;
}

struct P2PArgs {
  char* peer[P2P_MAX_WORLD];   // peer[rank] is the local buffer
  int rank, world;
  long n, slice;               // elements in all, per slice (the last slice may be short)
  P2PLayout lay;
  unsigned epoch;              // filled in by the kernel from arrive[8] (device-resident: a replayed launch
                               // record must see a fresh epoch every time)
  unsigned* arrive;
  int* status;
  long long timeout_ticks;
  void* g;                     // the local gradient (read; overwritten with the sum when write_g)
  void* p;                     // optional parameters: p <- p - rate * sum
  double rate;
  int write_g;
};

__device__ __forceinline__ unsigned ld_sys(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// all workgroups of this launch have arrived (and their earlier writes are visible system-wide)
__device__ bool grid_arrive(const P2PArgs& a, int which, bool* last) {
  __shared__ unsigned ticket;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0)
    ticket = __hip_atomic_fetch_add(a.arrive + which, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  *last = ticket == (unsigned)(gridDim.x * a.epoch - 1);   // counters only ever grow: epoch * WGs arrivals so far
  return true;
}

__device__ bool wait_flags(const P2PArgs& a, size_t flags_off, int code) {
  // one thread per peer polls; the workgroup learns the outcome through LDS
  __shared__ int ok;
  if (threadIdx.x == 0) ok = 1;
  __syncthreads();
  if ((int)threadIdx.x < a.world && (int)threadIdx.x != a.rank) {
    const unsigned* f = reinterpret_cast<const unsigned*>(a.peer[a.rank] + flags_off) + threadIdx.x * P2P_FLAG_STRIDE;
    const long long t0 = wall_clock64();
    while (ld_sys(f) < a.epoch) {
      if (wall_clock64() - t0 > a.timeout_ticks) {
        ok = 0;
        *a.status = code + (int)threadIdx.x;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  return ok != 0;
}

// S: the unit moved per lane -- one element, or four (16 bytes) when length and addresses allow.  a.n / a.slice
// count units.
template <class S, class E>
__global__ __launch_bounds__(256) void p2p_allreduce_kernel(P2PArgs a) {
  a.epoch = __hip_atomic_load(a.arrive + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;  // bumped by the last workgroup out
  const int W = a.world, me = a.rank;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nthr = (long)gridDim.x * blockDim.x;
  const S* g = static_cast<const S*>(a.g);
  S* gw = static_cast<S*>(a.g);
  S* pw = static_cast<S*>(a.p);
  const E r = (E)a.rate;
  bool last;
  if (W == 1) {  // nothing to exchange: the consumer alone
    for (long i = tid; i < a.n; i += nthr)
      if (pw) pw[i] = pw[i] - r * g[i];
    return;
  }
  // ---- phase 1: my version of slice j -> peer j's inbox[me] ---------------------------------------------------
  for (int d = 1; d < W; ++d) {
    const int j = (me + d) % W;   // start with a different peer on every rank: all links at once
    const long lo = j * a.slice, hi = lo + a.slice < a.n ? lo + a.slice : a.n;
    S* dst = reinterpret_cast<S*>(a.peer[j] + a.lay.inbox) + (long)me * a.slice;
    for (long i = lo + tid; i < hi; i += nthr) dst[i - lo] = g[i];
  }
  grid_arrive(a, 0, &last);
  if (last && threadIdx.x < (unsigned)W && (int)threadIdx.x != me)
    st_sys(reinterpret_cast<unsigned*>(a.peer[threadIdx.x] + a.lay.flags1) + me * P2P_FLAG_STRIDE, a.epoch);
  // ---- phase 2: reduce my slice in rank order, broadcast it ------------------------------------------------------
  if (!wait_flags(a, a.lay.flags1, 100)) return;
  {
    const long lo = me * a.slice, hi = lo + a.slice < a.n ? lo + a.slice : a.n;
    const S* inbox = reinterpret_cast<const S*>(a.peer[me] + a.lay.inbox);
    for (long i = lo + tid; i < hi; i += nthr) {
      S acc = g[i] * E(0);
      for (int q = 0; q < W; ++q) acc += (q == me) ? g[i] : __builtin_nontemporal_load(inbox + (long)q * a.slice + (i - lo));
      for (int d = 0; d < W; ++d) {
        const int j = (me + d) % W;
        reinterpret_cast<S*>(a.peer[j] + a.lay.outbox)[i] = acc;
      }
    }
  }
  grid_arrive(a, 1, &last);
  if (last && threadIdx.x < (unsigned)W && (int)threadIdx.x != me)
    st_sys(reinterpret_cast<unsigned*>(a.peer[threadIdx.x] + a.lay.flags2) + me * P2P_FLAG_STRIDE, a.epoch);
  // ---- phase 3: the whole reduced vector is here: consume it --------------------------------------------------------
  if (!wait_flags(a, a.lay.flags2, 200)) return;
  {
    // my own slice was written by other workgroups of this launch: make sure they are done
    __shared__ int spin_ok;
    if (threadIdx.x == 0) {
      spin_ok = 1;
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(a.arrive + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x * a.epoch) {
        if (wall_clock64() - t0 > a.timeout_ticks) {
          spin_ok = 0;
          *a.status = 300;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (!spin_ok) return;
  }
  const S* sum = reinterpret_cast<const S*>(a.peer[me] + a.lay.outbox);
  for (long i = tid; i < a.n; i += nthr) {
    const S v = __builtin_nontemporal_load(sum + i);
    if (a.write_g) gw[i] = v;
    if (pw) pw[i] = pw[i] - r * v;
  }
  grid_arrive(a, 2, &last);
  if (last && threadIdx.x == 0) __hip_atomic_store(a.arrive + 8, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace

void p2p_create(int64_t max_elems, int dtype, int world, void* out_handle64) {
  TO_CHECK(!g_p2p.local, TO_ERR_STATE, "p2p: an exchange buffer already exists (to_p2p_shutdown first)");
  TO_CHECK(world >= 1 && world <= P2P_MAX_WORLD, TO_ERR_ARG, "p2p: world size must be 1..8");
  TO_CHECK(max_elems >= 1, TO_ERR_ARG, "p2p: empty buffer");
  const size_t es = dtype == TO_F64 ? 8 : 4;
  g_p2p.world = world;
  g_p2p.dtype = dtype;
  g_p2p.max_elems = max_elems;
  g_p2p.slice_cap = ((max_elems + world - 1) / world + 63) / 64 * 64;
  g_p2p.lay = layout_for(g_p2p.slice_cap, world, es);
  // peers write into this memory and this GPU polls it while they do, mid-kernel: that needs fine-grained (coherent)
  // device memory.  No silent fallback: plain device memory gives no such guarantee (stale sums, watchdog timeouts),
  // so it is used only when the caller asks for it (TOPS_P2P_FINEGRAINED=0), and a failure here leaves no state behind.
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  hipIpcMemHandle_t h;
  std::memset(&h, 0, sizeof(h));
  static const int fine = [] { const char* e = ab_getenv("TOPS_P2P_FINEGRAINED"); return e ? atoi(e) : 1; }();
  try {
    if (fine) {
      hipError_t e = hipExtMallocWithFlags(&g_p2p.local, g_p2p.lay.total, hipDeviceMallocFinegrained);
      if (e == hipSuccess && world > 1) e = hipIpcGetMemHandle(&h, g_p2p.local);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        fail(TO_ERR_HIP, std::string("p2p: fine-grained exchange memory cannot be allocated or exported (") +
                             hipGetErrorString(e) + "); TOPS_P2P_FINEGRAINED=0 accepts plain device memory, without the "
                             "visibility guarantee the protocol relies on");
      }
    } else {
      TO_HIP(hipMalloc(&g_p2p.local, g_p2p.lay.total));
      if (world > 1) TO_HIP(hipIpcGetMemHandle(&h, g_p2p.local));
    }
    TO_HIP(hipMemset(g_p2p.local, 0, g_p2p.lay.total));
    TO_HIP(hipMalloc(&g_p2p.arrive, 64));
    TO_HIP(hipMemset(g_p2p.arrive, 0, 64));
    TO_HIP(hipHostMalloc(&g_p2p.status, sizeof(int), hipHostMallocMapped));
    *g_p2p.status = 0;
    TO_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&g_p2p.status_dev), g_p2p.status, 0));
    TO_HIP(hipDeviceSynchronize());
  } catch (...) {
    if (g_p2p.local) (void)hipFree(g_p2p.local);
    if (g_p2p.arrive) (void)hipFree(g_p2p.arrive);
    if (g_p2p.status) (void)hipHostFree(g_p2p.status);
    g_p2p = P2PState();
    throw;
  }
  std::memcpy(out_handle64, &h, 64);
}

void p2p_connect(int rank, const void* handles) {
  TO_CHECK(g_p2p.local != nullptr, TO_ERR_STATE, "p2p: call to_p2p_create first");
  TO_CHECK(rank >= 0 && rank < g_p2p.world, TO_ERR_ARG, "p2p: bad rank");
  g_p2p.rank = rank;
  for (int r = 0; r < g_p2p.world; ++r) {
    if (r == rank) {
      g_p2p.peer[r] = g_p2p.local;
      continue;
    }
    hipIpcMemHandle_t h;
    std::memcpy(&h, static_cast<const char*>(handles) + 64 * r, 64);
    TO_HIP(hipIpcOpenMemHandle(&g_p2p.peer[r], h, hipIpcMemLazyEnablePeerAccess));
    g_p2p.opened[r] = true;
  }
  g_p2p.epoch = 0;
  g_p2p.ready = true;
}

void p2p_allreduce(to_tensor g, to_tensor p, double rate, bool write_g, hipStream_t s) {
  TO_CHECK(g_p2p.ready, TO_ERR_STATE, "p2p: call to_p2p_create / to_p2p_connect first");
  TO_CHECK(g->contiguous() && g->dtype == g_p2p.dtype && g->total() <= g_p2p.max_elems && g->total() >= 1, TO_ERR_ARG,
           "p2p: the buffer must be contiguous, of the exchange's dtype and no longer than it was created for");
  if (p) TO_CHECK(p->contiguous() && p->dtype == g->dtype && p->total() == g->total(), TO_ERR_ARG,
                  "p2p: parameters and gradients must match");
  TO_CHECK(*g_p2p.status == 0, TO_ERR_STATE, "p2p: an earlier exchange timed out (code " + std::to_string(*g_p2p.status) + ")");
  P2PArgs a{};
  for (int r = 0; r < g_p2p.world; ++r) a.peer[r] = static_cast<char*>(g_p2p.peer[r]);
  a.rank = g_p2p.rank;
  a.world = g_p2p.world;
  // 16 bytes per lane when the length and the addresses allow (the flat training buffers always do)
  const int64_t per16 = g->dtype == TO_F64 ? 2 : 4;
  // (decided by the LENGTH alone: the slice layout of the inbox / outbox must be the same on every rank, and a local
  //  address is not something the ranks agree on)
  const bool wide = g->total() % per16 == 0;
  TO_CHECK(!wide || ((reinterpret_cast<uintptr_t>(g->ptr) & 15u) == 0 && (!p || (reinterpret_cast<uintptr_t>(p->ptr) & 15u) == 0)),
           TO_ERR_ARG, "p2p: buffers whose length is a multiple of 16 bytes must be 16-byte aligned");
  const int64_t unit = wide ? per16 : 1;
  a.n = g->total() / unit;
  a.slice = (a.n + a.world - 1) / a.world;
  TO_CHECK(a.slice * unit <= g_p2p.slice_cap, TO_ERR_ARG, "p2p: slice larger than the exchange buffer");
  a.lay = g_p2p.lay;
  // the inbox / outbox are indexed with the per-call slice length: ranks must pass vectors of one length
  a.epoch = 0;
  a.arrive = g_p2p.arrive;
  a.status = g_p2p.status_dev;
  static const double timeout_s = [] { const char* e = getenv("TOPS_P2P_TIMEOUT_S"); return e ? atof(e) : 5.0; }();
  a.timeout_ticks = (long long)(timeout_s * 100e6);  // wall_clock64 ticks at 100 MHz
  a.g = g->ptr;
  a.p = p ? p->ptr : nullptr;
  a.rate = rate;
  a.write_g = write_g ? 1 : 0;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  if (g->dtype == TO_F64) {
    if (wide) launch_k(p2p_allreduce_kernel<f64x2, double>, dim3(P2P_WGS), dim3(256), 0, s, a);
    else launch_k(p2p_allreduce_kernel<double, double>, dim3(P2P_WGS), dim3(256), 0, s, a);
  } else {
    if (wide) launch_k(p2p_allreduce_kernel<f32x4, float>, dim3(P2P_WGS), dim3(256), 0, s, a);
    else launch_k(p2p_allreduce_kernel<float, float>, dim3(P2P_WGS), dim3(256), 0, s, a);
  }
  TO_HIP(hipGetLastError());
  count_launch();
}

int p2p_status() { return g_p2p.status ? *g_p2p.status : 0; }
int p2p_world() { return g_p2p.ready ? g_p2p.world : 0; }

void p2p_shutdown() {
  if (!g_p2p.local) return;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < P2P_MAX_WORLD; ++r)
    if (g_p2p.opened[r]) (void)hipIpcCloseMemHandle(g_p2p.peer[r]);
  (void)hipFree(g_p2p.local);
  (void)hipFree(g_p2p.arrive);
  (void)hipHostFree(g_p2p.status);
  g_p2p = P2PState();
}

}  // namespace to
