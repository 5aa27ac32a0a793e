// Per-sample online SGD of an ffLayer stack as ONE persistent launch on ONE XCD.
//
// The reference's training loop is `foldl' (\nt (i,o) -> trainNetwork crossEntropy rate i o nt)` over the samples
// (app/MNIST.hs:390-396, app/Dots.hs:74-80): every sample's step reads the parameters the previous step wrote, so the
// loop is a chain of tiny dependent matrix-vector products -- six dependent launches per sample on the generic path,
// ~31 us per sample for the app's 784 -> 300 -> 100 -> 10 stack, all of it launch latency.  Here the whole chain lives
// in one kernel whose workgroups keep the parameters in LDS between samples:
//
//   * layer 1 (the big one: 300 x 784) is split by ROWS over G <= 32 workgroups; workgroup g owns rows R_g of W1, b1
//     and computes its slice of h1 = logistic(W1 x + b1);
//   * layer 2 is split by COLUMNS along the same index set: workgroup g owns W2[:, R_g] and contributes the partial
//     product W2[:, R_g] h1[R_g] -- o2 numbers -- to an exchange buffer.  One barrier per sample; afterwards every
//     workgroup sums the G partials (in workgroup order: every workgroup gets the same bits) and has z2;
//   * layers 3..L and the loss head are small and replicated: every workgroup computes them and applies the identical
//     update to its own copy, so the copies stay bit-identical and no further exchange is needed;
//   * backward: dz_L from the head, back through the replicated layers, dz2 -> the workgroup's slice of
//     dz1 = (W2[:, R_g]^T dz2) h1 (1 - h1) needs only what the workgroup owns; then every parameter is updated in LDS
//     (`p - r * g`, FeedForward.hs:145-147).  The next sample's input is fetched while this one is computed.
//
// All participating workgroups sit on one XCD (the grid is 8 G workgroups, workgroup b runs on XCD b % 8, those with
// b % 8 != 0 leave at once): the exchange goes through that XCD's L2 with L1-bypassing loads and stores and a counter
// in the same L2 -- no device-scope cache write-back or invalidate, which is what made a cross-XCD grid barrier cost
// 180 us in round 2.  A workgroup that waits longer than the watchdog gives up and reports.
#include <cstring>

#include "common.hpp"

namespace to {

namespace {

constexpr int ON_MAX_LAYERS = 6;
constexpr int ON_THREADS = 512;
// the commit's decision word: counter[ON_DECISION] (behind the 32 vote slots), decided once by compare-and-swap
constexpr int ON_DECISION = 32;
constexpr unsigned long long ON_COMMIT = 1ull, ON_ABORT = 2ull;
constexpr int ON_XREGS = 4;  // input elements prefetched per thread: i0 <= 2048

template <class S>
struct OnlineArgs {
  int L;
  int dims[ON_MAX_LAYERS + 1];  // i0, o1 .. oL
  S* W[ON_MAX_LAYERS];
  S* b[ON_MAX_LAYERS];
  const S* X;
  const S* Y;
  const long long* idx;  // sample order (device), or null: 0 .. n-1
  long n;
  S rate;
  int head;          // 1: softmax >>> crossEntropy, 2: logistic >>> squaredError
  int G, rpw;        // workgroups, rows of layer 1 per workgroup
  unsigned long long* exch;  // [2][G][o2][words of S]: {tag, 32 bits of the value} (zero at launch)
  unsigned* counter;         // 256 bytes, zero at launch: the commit verdicts, one 64-bit word per workgroup
  int* status;       // host-visible: nonzero = a barrier timed out at that sample + 1
  long long timeout; // wall_clock64 ticks
  long long* dbg;    // development (TOPS_ONLINE_STAMPS): phase time stamps of workgroup 0 at sample 64
};

__device__ __forceinline__ float logistic_f(float z) { return 1.0f / (1.0f + __expf(-z)); }
__device__ __forceinline__ double logistic_f(double z) { return 1.0 / (1.0 + exp(-z)); }
__device__ __forceinline__ float exp_f(float z) { return __expf(z); }
__device__ __forceinline__ double exp_f(double z) { return exp(z); }
__device__ __forceinline__ float fma_f(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double fma_f(double a, double b, double c) { return fma(a, b, c); }

// Sum over the 64 lanes of a wave, result in every lane.  Row operations on the VALU's data path (DPP) instead of six
// trips through the LDS crossbar (ds_bpermute, what __shfl_xor compiles to): quads, half rows, rows, then the two row
// broadcasts; lane 63 ends up with the total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += dpp_f<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
  v += dpp_f<0x141, 0xf>(v);   // row_half_mirror
  v += dpp_f<0x140, 0xf>(v);   // row_mirror: every lane of a row holds the row's sum
  v += dpp_f<0x142, 0xa>(v);   // row_bcast15 into rows 1 and 3
  v += dpp_f<0x143, 0xc>(v);   // row_bcast31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// L1-bypassing accesses to the exchange buffer (all readers and writers share one L2)
template <class S> __device__ __forceinline__ void st_l2(S* p, S v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class S> __device__ __forceinline__ S ld_l2(const S* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// (S = float, or double: the reference's own element type, `HMat Double`, BLAS/HMat.hs:35)
template <class S>
__global__ __launch_bounds__(ON_THREADS) void online_sgd_kernel(OnlineArgs<S> a) {
  if (blockIdx.x & 7) return;  // XCD 0 only
  const int g = blockIdx.x >> 3, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  S* lds = reinterpret_cast<S*>(lds_raw);
  const int L = a.L, i0 = a.dims[0], o1 = a.dims[1], o2 = a.dims[2], oL = a.dims[L];
  const int r0 = g * a.rpw, nr = min(a.rpw, o1 - r0) > 0 ? min(a.rpw, o1 - r0) : 0;
  // ---- LDS layout -------------------------------------------------------------------------------------------------
  S* p = lds;
  S* W1s = p; p += (long)a.rpw * i0;          // [rpw][i0]
  S* b1s = p; p += a.rpw;
  S* W2s = p; p += (long)o2 * a.rpw;          // [o2][rpw]: columns R_g of W2
  S* bb[ON_MAX_LAYERS];                       // biases of layers 2..L (replicated)
  S* Wr[ON_MAX_LAYERS];                       // weights of layers 3..L (replicated), rows padded by one S
  bb[1] = p; p += o2;
  for (int l = 2; l < L; ++l) {
    Wr[l] = p; p += (long)a.dims[l + 1] * (a.dims[l] + 1);
    bb[l] = p; p += a.dims[l + 1];
  }
  S* xs = p; p += i0;
  S* ys = p; p += oL;
  S* h1s = p; p += a.rpw;
  S* dz1s = p; p += a.rpw;
  S* act[ON_MAX_LAYERS + 1];                  // act[l]: output of layer l (l >= 2), full
  S* dz[ON_MAX_LAYERS + 1];
  for (int l = 2; l <= L; ++l) {
    act[l] = p; p += a.dims[l];
    dz[l] = p; p += a.dims[l];
  }
  S* red = p; p += 8;
  S* part = p; p += a.G * o2;                 // the partial sums of z2 from every workgroup
  // ---- parameters -> LDS ----------------------------------------------------------------------------------------------
  for (long e = tid; e < (long)nr * i0; e += ON_THREADS) W1s[e] = a.W[0][(long)r0 * i0 + e];
  for (int e = tid; e < nr; e += ON_THREADS) b1s[e] = a.b[0][r0 + e];
  for (int e = tid; e < o2 * a.rpw; e += ON_THREADS) {
    const int j = e / a.rpw, r = e - j * a.rpw;
    W2s[e] = r < nr ? a.W[1][(long)j * o1 + r0 + r] : S(0.);
  }
  for (int e = tid; e < o2; e += ON_THREADS) bb[1][e] = a.b[1][e];
  for (int l = 2; l < L; ++l) {
    const int K = a.dims[l], O = a.dims[l + 1];
    for (int e = tid; e < O * K; e += ON_THREADS) Wr[l][(e / K) * (K + 1) + e % K] = a.W[l][e];
    for (int e = tid; e < O; e += ON_THREADS) bb[l][e] = a.b[l][e];
  }
  auto row_of = [&](long t) { return a.idx ? (long)a.idx[t] : t; };
  // the first sample's input
  S xr[ON_XREGS], yr = S(0.);
  {
    const long s = a.n > 0 ? row_of(0) : 0;
#pragma unroll
    for (int q = 0; q < ON_XREGS; ++q) {
      const int k = tid + q * ON_THREADS;
      xr[q] = (a.n > 0 && k < i0) ? a.X[s * i0 + k] : S(0.);
    }
    if (a.n > 0 && tid < oL) yr = a.Y[s * oL + tid];
  }
  long s_next = a.n > 1 ? row_of(1) : 0;  // the row index is fetched one sample further ahead than the row
  if (tid == 0) red[7] = S(0.);
  __syncthreads();
  const S rate = a.rate;
  int stamp_i = 0;
#define ON_STAMP()                                                                       \
  do {                                                                                   \
    if (a.dbg && t == 64 && g == 0 && tid == 0) a.dbg[stamp_i++] = wall_clock64();       \
  } while (0)
  for (long t = 0; t < a.n; ++t) {
    ON_STAMP();
    // ---- this sample's input into LDS, the next one's on its way -----------------------------------------------------
#pragma unroll
    for (int q = 0; q < ON_XREGS; ++q) {
      const int k = tid + q * ON_THREADS;
      if (k < i0) xs[k] = xr[q];
    }
    if (tid < oL) ys[tid] = yr;
    __syncthreads();
    ON_STAMP();
    // ---- layer 1, this workgroup's rows: one wave per row ---------------------------------------------------------------
    // (round 6: a wave whose turn comes twice -- 10 rows on 8 waves -- runs its two rows TOGETHER: their LDS reads are in flight
    //  at once and share the reads of x; each row's own chain of sums is what it was, so the bits are)
    constexpr int NWV = ON_THREADS / 64;
    for (int r = wave; r < nr; r += 2 * NWV) {
      const S* __restrict__ w = W1s + r * i0;
      S acc0 = S(0.), acc1 = S(0.), acc2 = S(0.), acc3 = S(0.);
      int k = lane;
      if (r + NWV < nr) {                                 // (uniform) this wave has a second row
        const S* __restrict__ v = w + NWV * i0;
        S bcc0 = S(0.), bcc1 = S(0.), bcc2 = S(0.), bcc3 = S(0.);
        for (; k + 192 < i0; k += 256) {   // four independent chains per row: the LDS reads of one pass are all in flight together
          const S x0 = xs[k], x1 = xs[k + 64], x2 = xs[k + 128], x3 = xs[k + 192];
          acc0 = fma_f(w[k], x0, acc0);
          acc1 = fma_f(w[k + 64], x1, acc1);
          acc2 = fma_f(w[k + 128], x2, acc2);
          acc3 = fma_f(w[k + 192], x3, acc3);
          bcc0 = fma_f(v[k], x0, bcc0);
          bcc1 = fma_f(v[k + 64], x1, bcc1);
          bcc2 = fma_f(v[k + 128], x2, bcc2);
          bcc3 = fma_f(v[k + 192], x3, bcc3);
        }
        for (; k < i0; k += 64) {
          const S x0 = xs[k];
          acc0 = fma_f(w[k], x0, acc0);
          bcc0 = fma_f(v[k], x0, bcc0);
        }
        S bcc = (bcc0 + bcc1) + (bcc2 + bcc3);
        bcc = wave_sum(bcc);
        if (lane == 1) h1s[r + NWV] = logistic_f(bcc + b1s[r + NWV]);
      } else {
        for (; k + 192 < i0; k += 256) {
          acc0 = fma_f(w[k], xs[k], acc0);
          acc1 = fma_f(w[k + 64], xs[k + 64], acc1);
          acc2 = fma_f(w[k + 128], xs[k + 128], acc2);
          acc3 = fma_f(w[k + 192], xs[k + 192], acc3);
        }
        for (; k < i0; k += 64) acc0 = fma_f(w[k], xs[k], acc0);
      }
      S acc = (acc0 + acc1) + (acc2 + acc3);
      acc = wave_sum(acc);
      if (lane == 0) h1s[r] = logistic_f(acc + b1s[r]);
    }
    __syncthreads();
    ON_STAMP();
    // ---- partial z2 = W2[:, R_g] h1[R_g] -> exchange ----------------------------------------------------------------------
    // Every 32-bit word of a partial travels with the sample's tag in ONE 64-bit store; a reader polls the words it
    // needs until their tags say "this sample".  Data and "it is there" are the same memory transaction: one L2 round
    // trip per sample (store -> L2 -> load), where a counter barrier needs three (store ack, arrive, poll, then load).
    // A slot is written again two samples later, which its writer can only reach after every workgroup has written the
    // sample in between -- i.e. after all of them have read this one.
    constexpr int WORDS = sizeof(S) / 4;
    const unsigned tag = (unsigned)(t + 1);
    unsigned long long* ex = a.exch + (long)(t & 1) * a.G * o2 * WORDS;
    for (int j = tid; j < o2; j += ON_THREADS) {
      const S* w = W2s + j * a.rpw;
      S acc = S(0.);
      for (int r = 0; r < nr; ++r) acc = fma_f(w[r], h1s[r], acc);
      unsigned wd[WORDS];
      __builtin_memcpy(wd, &acc, sizeof(S));
#pragma unroll
      for (int u = 0; u < WORDS; ++u)
        __hip_atomic_store(ex + ((long)g * o2 + j) * WORDS + u, ((unsigned long long)tag << 32) | wd[u], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- z2 = b2 + sum_g partial_g ; layer 2's activation -----------------------------------------------------------------
    {
      const int total = a.G * o2 * WORDS;
      unsigned* pw = reinterpret_cast<unsigned*>(part);
      bool ok = true;
      const long long c0 = wall_clock64();
      for (int e0 = tid; e0 < total; e0 += 4 * ON_THREADS) {
        unsigned long long v[4];
        bool have[4] = {false, false, false, false};
        while (true) {
          bool all = true;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * ON_THREADS;
            if (e < total && !have[u]) v[u] = __hip_atomic_load(ex + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * ON_THREADS;
            if (e < total && !have[u]) {
              have[u] = (unsigned)(v[u] >> 32) == tag;
              all = all && have[u];
            }
          }
          if (all) break;
          if (wall_clock64() - c0 > a.timeout) {
            ok = false;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = e0 + u * ON_THREADS;
          if (e < total) pw[e] = (unsigned)v[u];
        }
      }
      if (!ok) {
        *a.status = (int)(t + 1);
        red[7] = S(1.);   // (any thread: the flag is cleared before the loop and only ever set)
      }
    }
    // The next sample's row is requested HERE, behind the last wait on memory of this iteration: loads return in order, so
    // issued any earlier the exchange above would sit behind a row that comes from HBM.  It has the rest of the sample
    // (the replicated layers, the head, the backward pass, the updates) to arrive.
    if (t + 1 < a.n) {
      const long s = s_next;
#pragma unroll
      for (int q = 0; q < ON_XREGS; ++q) {
        const int k = tid + q * ON_THREADS;
        if (k < i0) xr[q] = a.X[s * i0 + k];
      }
      if (tid < oL) yr = a.Y[s * oL + tid];
      if (t + 2 < a.n) s_next = row_of(t + 2);
    }
    __syncthreads();
    ON_STAMP();
    if (red[7] != S(0.)) {        // a peer never showed up (uniform): this workgroup votes "failed", decides ABORT for
      if (tid == 0) {             // everybody (unless COMMIT was decided -- impossible: that needs this vote to be "ok") and leaves
        unsigned long long* cmv = reinterpret_cast<unsigned long long*>(a.counter);
        __hip_atomic_store(cmv + g, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long expect = 0ull;
        __hip_atomic_compare_exchange_strong(cmv + ON_DECISION, &expect, ON_ABORT, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    for (int j = tid; j < o2; j += ON_THREADS) {
      // (a fixed order -- four interleaved chains, then their sum -- so that every workgroup gets the same bits; four
      //  chains keep four LDS reads in flight)
      S z0 = S(0.), z1 = S(0.), z2 = S(0.), z3 = S(0.);
      int q = 0;
      for (; q + 4 <= a.G; q += 4) {
        z0 += part[(q + 0) * o2 + j];
        z1 += part[(q + 1) * o2 + j];
        z2 += part[(q + 2) * o2 + j];
        z3 += part[(q + 3) * o2 + j];
      }
      for (; q < a.G; ++q) z0 += part[q * o2 + j];
      const S z = bb[1][j] + ((z0 + z1) + (z2 + z3));
      act[2][j] = L == 2 ? z : logistic_f(z);
    }
    __syncthreads();
    ON_STAMP();
    // ---- replicated layers 3..L ---------------------------------------------------------------------------------------------
    for (int l = 2; l < L; ++l) {
      const int K = a.dims[l], O = a.dims[l + 1];
      for (int j = wave; j < O; j += ON_THREADS / 64) {   // one wave per output: a 64-lane dot product
        const S* w = Wr[l] + j * (K + 1);
        S z = S(0.);
        for (int k = lane; k < K; k += 64) z = fma_f(w[k], act[l][k], z);
        z = wave_sum(z);
        z += bb[l][j];
        if (lane == 0) act[l + 1][j] = l + 1 == L ? z : logistic_f(z);
      }
      __syncthreads();
    ON_STAMP();
    }
    // ---- loss head on z_L (oL <= 64: wave 0) ------------------------------------------------------------------------------
    if (wave == 0) {
      const S z = lane < oL ? act[L][lane] : S(-3.0e38), y = lane < oL ? ys[lane] : S(0.);
      S d;
      if (a.head == 1) {
        S mx = z, sy = y;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const S other = __shfl_xor(mx, off);   // (one shuffle, outside the select: a shuffle under divergence reads dead lanes)
          mx = mx > other ? mx : other;
          sy += __shfl_xor(sy, off);
        }
        const S e = lane < oL ? exp_f(z - mx) : S(0.);
        S se = e;
        se = wave_sum(se);
        d = e / se * sy - y;              // softmax(z) * sum(y) - y
      } else {
        const S s = logistic_f(z), e = y - s;
        d = -S(2.0) * e * s * (S(1.0) - s);   // logistic >>> squaredError
      }
      if (lane < oL) dz[L][lane] = d;
    }
    __syncthreads();
    ON_STAMP();
    // ---- back through the replicated layers (OLD weights) -------------------------------------------------------------------
    for (int l = L - 1; l >= 2; --l) {
      const int K = a.dims[l], O = a.dims[l + 1];
      for (int k = tid; k < K; k += ON_THREADS) {
        S s = S(0.);
        for (int j = 0; j < O; ++j) s = fma_f(Wr[l][j * (K + 1) + k], dz[l + 1][j], s);
        const S h = act[l][k];
        dz[l][k] = s * h * (S(1.0) - h);
      }
      __syncthreads();
    ON_STAMP();
    }
    // ---- dz1 on this workgroup's rows ---------------------------------------------------------------------------------------
    for (int r = wave; r < nr; r += 2 * NWV) {   // (two rows of a wave together, as in layer 1)
      S s = S(0.);
      if (r + NWV < nr) {
        const int r2 = r + NWV;
        S s2 = S(0.);
        for (int j = lane; j < o2; j += 64) {
          const S d = dz[2][j];
          s = fma_f(W2s[j * a.rpw + r], d, s);
          s2 = fma_f(W2s[j * a.rpw + r2], d, s2);
        }
        s2 = wave_sum(s2);
        if (lane == 1) {
          const S h = h1s[r2];
          dz1s[r2] = s2 * h * (S(1.0) - h);
        }
      } else {
        for (int j = lane; j < o2; j += 64) s = fma_f(W2s[j * a.rpw + r], dz[2][j], s);
      }
      s = wave_sum(s);
      if (lane == 0) {
        const S h = h1s[r];
        dz1s[r] = s * h * (S(1.0) - h);
      }
    }
    __syncthreads();
    ON_STAMP();
    // ---- p <- p - rate * g, everything this workgroup holds ------------------------------------------------------------------
    // (a thread owns its columns k of every row: the input element is read once, four rows are read, updated and written
    //  as a group so that their LDS round trips overlap.  Round 6 tried all of a column's rows in one guarded batch of
    //  sixteen: 1.9 -> 2.5 us for this phase -- the uniform guards cost more than the round trips they merge.  Kept: W2's
    //  outputs handed out from the top of the workgroup (below), and the replicated
    //  layers walked flat over all threads instead of a row per wave)
    for (int k = tid; k < i0; k += ON_THREADS) {
      const S x = -rate * xs[k];
      S* __restrict__ w = W1s + k;
      int r = 0;
      for (; r + 4 <= nr; r += 4) {
        S w0 = w[(r + 0) * i0], w1 = w[(r + 1) * i0], w2 = w[(r + 2) * i0], w3 = w[(r + 3) * i0];
        w0 = fma_f(dz1s[r + 0], x, w0);
        w1 = fma_f(dz1s[r + 1], x, w1);
        w2 = fma_f(dz1s[r + 2], x, w2);
        w3 = fma_f(dz1s[r + 3], x, w3);
        w[(r + 0) * i0] = w0; w[(r + 1) * i0] = w1; w[(r + 2) * i0] = w2; w[(r + 3) * i0] = w3;
      }
      for (; r < nr; ++r) w[r * i0] = fma_f(dz1s[r], x, w[r * i0]);
    }
    for (int e = tid; e < nr; e += ON_THREADS) b1s[e] -= rate * dz1s[e];
    int j_top = ON_THREADS - 1 - tid;   // W2's outputs from the TOP of the workgroup: the threads without a second column of W1
    // (opaque to the compiler: with `511 - tid` visible it addresses dz[2][j] as (dz[2] - tid) + 511 elements -- a FLAT load
    //  whose BASE lies below the LDS aperture for a small stack, which the hardware takes for a global address and faults on
    //  (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION on the 2 -> 16 -> 1 stack); these arrays are reached through generic pointers)
    asm volatile("" : "+v"(j_top));
    for (int j = j_top; j < o2; j += ON_THREADS) {
      const S c = -rate * dz[2][j];
      S* __restrict__ w = W2s + j * a.rpw;
      int r = 0;
      for (; r + 4 <= nr; r += 4) {
        S w0 = w[r], w1 = w[r + 1], w2 = w[r + 2], w3 = w[r + 3];
        w0 = fma_f(c, h1s[r], w0); w1 = fma_f(c, h1s[r + 1], w1); w2 = fma_f(c, h1s[r + 2], w2); w3 = fma_f(c, h1s[r + 3], w3);
        w[r] = w0; w[r + 1] = w1; w[r + 2] = w2; w[r + 3] = w3;
      }
      for (; r < nr; ++r) w[r] = fma_f(c, h1s[r], w[r]);
    }
    for (int e = tid; e < o2; e += ON_THREADS) bb[1][e] -= rate * dz[2][e];
    for (int l = 2; l < L; ++l) {
      const int K = a.dims[l], O = a.dims[l + 1];
      for (int e = tid; e < O * K; e += ON_THREADS) {
        const int j = e / K, k = e - j * K;
        S* w = Wr[l] + j * (K + 1) + k;
        *w = fma_f(-rate * dz[l + 1][j], act[l][k], *w);
      }
      for (int e = tid; e < O; e += ON_THREADS) bb[l][e] -= rate * dz[l + 1][e];
    }
    __syncthreads();
    ON_STAMP();
  }
  // ---- commit: ONE decision for all workgroups -------------------------------------------------------------------------
  // A workgroup that got through its stream votes "ok" and waits for the other votes.  Votes alone cannot make the write-back
  // all-or-nothing (round 3's form): a peer that saw every "ok" in time would write its slice while one whose poll timed out
  // a moment earlier would not -- parameters half updated.  So the outcome is a single word that is decided
  // exactly once, by compare-and-swap: whoever has seen all G "ok" votes proposes COMMIT, whoever has seen a "failed" vote or
  // run out of time proposes ABORT, the first proposal wins and EVERYBODY obeys the word, not their own view.  A workgroup
  // that timed out but reads COMMIT writes its slice after all: COMMIT can only have been proposed by a peer that saw all G
  // votes "ok", this one's included, so every slice is complete.  Same L1-bypassing accesses, same watchdog.
  {
    unsigned long long* cm = reinterpret_cast<unsigned long long*>(a.counter);
    // (the decision travels to the other threads in red[6]: the kernel's dynamic LDS may be all 160 KiB, a static
    //  __shared__ word on top of it would make the launch fail)
    if (tid == 0) __hip_atomic_store(cm + g, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < a.G) {
      const long long c0 = wall_clock64();
      unsigned long long v = 0;
      while (true) {
        v = __hip_atomic_load(cm + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != 0) break;
        if (__hip_atomic_load(cm + ON_DECISION, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;   // (decided meanwhile)
        if (wall_clock64() - c0 > a.timeout) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (v != 1ull) red[7] = S(1.);   // (any thread: a vote that is missing or "failed")
    }
    __syncthreads();
    if (tid == 0) {
      unsigned long long expect = 0ull;
      const unsigned long long mine = red[7] != S(0.) ? ON_ABORT : ON_COMMIT;
      const bool won = __hip_atomic_compare_exchange_strong(cm + ON_DECISION, &expect, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long decision = won ? mine : expect;   // (expect holds what was there)
      red[6] = decision == ON_COMMIT ? S(1.) : S(0.);
      if (decision == ON_ABORT && *a.status == 0) *a.status = (int)(a.n + 1);   // (no sample to name: reported as "sample n")
    }
    __syncthreads();
    if (red[6] != S(1.)) return;
  }
  // ---- parameters back to memory (replicated ones from workgroup 0: all copies are the same bits) -----------------------
  for (long e = tid; e < (long)nr * i0; e += ON_THREADS) a.W[0][(long)r0 * i0 + e] = W1s[e];
  for (int e = tid; e < nr; e += ON_THREADS) a.b[0][r0 + e] = b1s[e];
  for (int e = tid; e < o2 * a.rpw; e += ON_THREADS) {
    const int j = e / a.rpw, r = e - j * a.rpw;
    if (r < nr) a.W[1][(long)j * o1 + r0 + r] = W2s[e];
  }
  if (g == 0) {
    for (int e = tid; e < o2; e += ON_THREADS) a.b[1][e] = bb[1][e];
    for (int l = 2; l < L; ++l) {
      const int K = a.dims[l], O = a.dims[l + 1];
      for (int e = tid; e < O * K; e += ON_THREADS) a.W[l][e] = Wr[l][(e / K) * (K + 1) + e % K];
      for (int e = tid; e < O; e += ON_THREADS) a.b[l][e] = bb[l][e];
    }
  }
}

static long long*& dbg_ptr() {
  static long long* p = nullptr;
  return p;
}
struct OnlineState {
  void* exch = nullptr;
  size_t exch_bytes = 0;
  unsigned* counter = nullptr;
  int* status = nullptr;
  int* status_dev = nullptr;
};
OnlineState g_on;

}  // namespace

// Can the persistent kernel take this stack?  (fp32, 2..6 layers, logistic hidden layers, a head of at most 64 outputs,
// an input of at most 2048 elements, everything a workgroup holds within 160 KiB of LDS)
bool online_sgd_plan(int dtype, int L, const int64_t* dims, int* G_out, int* rpw_out, size_t* lds_out) {
  const int64_t es = dtype == TO_F64 ? 8 : 4;
  if (L < 2 || L > ON_MAX_LAYERS) return false;
  for (int l = 0; l <= L; ++l)
    if (dims[l] < 1 || dims[l] > 65535) return false;
  if (dims[0] > ON_THREADS * ON_XREGS || dims[L] > 64) return false;
  const int64_t o1 = dims[1];
  for (int G = (int)std::min<int64_t>(32, o1); G >= 1; --G) {
    const int64_t rpw = (o1 + G - 1) / G;
    if ((o1 + rpw - 1) / rpw != G) continue;  // every workgroup owns at least one row
    int64_t f = rpw * dims[0] + rpw + dims[2] * rpw + dims[2];
    for (int l = 2; l < L; ++l) f += dims[l + 1] * (dims[l] + 1) + dims[l + 1];
    f += dims[0] + dims[L] + 2 * rpw + 8 + G * dims[2];
    for (int l = 2; l <= L; ++l) f += 2 * dims[l];
    if (f * es <= 160 * 1024) {
      *G_out = G;
      *rpw_out = (int)rpw;
      *lds_out = (size_t)(f * es);
      return true;
    }
    break;  // fewer workgroups only make the slices larger
  }
  return false;
}

template <class S>
static void launch_online_t(int L, const int64_t* dims, void* const* W, void* const* b, const void* X, const void* Y,
                            const long long* idx_dev, int64_t n, double rate, int head, int G, int rpw, size_t lds, hipStream_t s) {
  OnlineArgs<S> a{};
  a.L = L;
  for (int l = 0; l <= L; ++l) a.dims[l] = (int)dims[l];
  for (int l = 0; l < L; ++l) {
    a.W[l] = static_cast<S*>(W[l]);
    a.b[l] = static_cast<S*>(b[l]);
  }
  a.X = static_cast<const S*>(X);
  a.Y = static_cast<const S*>(Y);
  a.idx = idx_dev;
  a.n = n;
  a.rate = (S)rate;
  a.head = head;
  a.G = G;
  a.rpw = rpw;
  a.exch = static_cast<unsigned long long*>(g_on.exch);
  a.counter = g_on.counter;
  a.status = g_on.status_dev;
  static const double timeout_s = [] { const char* e = getenv("TOPS_ONLINE_TIMEOUT_S"); return e ? atof(e) : 2.0; }();
  a.timeout = (long long)(timeout_s * 100e6);
  static long long* dbg = [] {
    long long* p = nullptr;
    if (ab_getenv("TOPS_ONLINE_STAMPS") && hipHostMalloc(&p, 64 * sizeof(long long), hipHostMallocMapped) == hipSuccess) {
      std::memset(p, 0, 64 * sizeof(long long));
      return p;
    }
    return (long long*)nullptr;
  }();
  a.dbg = dbg;
  if (dbg) {
    static bool reg = false;
    if (!reg) {
      reg = true;
      atexit([] {
        long long* p = dbg_ptr();
        if (!p) return;
        std::fprintf(stderr, "[online] phase stamps (us since the sample began):");
        for (int i = 1; i < 64 && p[i]; ++i) std::fprintf(stderr, " %.2f", (p[i] - p[0]) * 0.01);
        std::fprintf(stderr, "\n");
      });
    }
    dbg_ptr() = dbg;
  }
  static bool attr = false;
  if (!attr) {
    TO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(online_sgd_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
    attr = true;
  }
  launch_k(online_sgd_kernel<S>, dim3(8 * G), dim3(ON_THREADS), lds, s, a);
}

// The exchange needs every participating workgroup on ONE XCD (one L2).  The kernel gets that from the dispatcher's
// round-robin -- workgroup b of a grid lands on XCD b % 8 in SPX mode with every CU enabled -- which is observed
// behaviour, not a documented contract: probed once per process with a grid of the same shape (every participating
// workgroup reports the XCC it runs on) before the kernel is ever trusted with parameters.  Other partition modes or a CU
// mask fail the probe and the caller keeps the generic per-sample path.
__global__ void online_xcc_probe_kernel(int* out) {
  if (blockIdx.x % 8 != 0 || threadIdx.x != 0) return;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  out[blockIdx.x / 8] = (int)(xcc & 0xf);
}

bool online_sgd_placement_ok(hipStream_t s) {
  static int verdict = -1;
  if (verdict >= 0) return verdict == 1;
  int* host = nullptr;
  if (hipHostMalloc(&host, 32 * sizeof(int), hipHostMallocMapped) != hipSuccess) return false;
  int* dev = nullptr;
  bool ok = hipHostGetDevicePointer(reinterpret_cast<void**>(&dev), host, 0) == hipSuccess;
  for (int rep = 0; ok && rep < 4; ++rep) {   // (four launches: a placement that only sometimes holds is no placement)
    for (int i = 0; i < 32; ++i) host[i] = -1;
    launch_k(online_xcc_probe_kernel, dim3(8 * 32), dim3(64), 0, s, dev);
    ok = hipStreamSynchronize(s) == hipSuccess;
    for (int i = 0; ok && i < 32; ++i) ok = host[i] >= 0 && host[i] == host[0];
  }
  (void)hipHostFree(host);
  verdict = ok ? 1 : 0;
  return ok;
}

void launch_online_sgd(int dtype, int L, const int64_t* dims, void* const* W, void* const* b, const void* X, const void* Y,
                       const long long* idx_dev, int64_t n, double rate, int head, hipStream_t s) {
  int G = 0, rpw = 0;
  size_t lds = 0;
  TO_CHECK(online_sgd_plan(dtype, L, dims, &G, &rpw, &lds), TO_ERR_UNSUPPORTED, "online SGD kernel: stack outside its range");
  TO_CHECK(online_sgd_placement_ok(s), TO_ERR_UNSUPPORTED,
           "online SGD kernel: workgroups b, b+8, b+16 ... of a grid do not share an XCD on this device (partition mode / CU mask)");
  const size_t need = (size_t)2 * G * dims[2] * (dtype == TO_F64 ? 2 : 1) * 8;
  if (!g_on.counter) {
    TO_HIP(hipMalloc(&g_on.counter, 512));   // 32 vote slots + the decision word
    TO_HIP(hipHostMalloc(&g_on.status, sizeof(int), hipHostMallocMapped));
    TO_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&g_on.status_dev), g_on.status, 0));
    *g_on.status = 0;
  }
  if (g_on.exch_bytes < need) {
    if (g_on.exch) (void)hipFree(g_on.exch);
    g_on.exch = nullptr;
    TO_HIP(hipMalloc(&g_on.exch, need));
    g_on.exch_bytes = need;
  }
  TO_HIP(hipMemsetAsync(g_on.counter, 0, 512, s));
  TO_HIP(hipMemsetAsync(g_on.exch, 0, need, s));  // (no tag of an earlier launch may pass for one of this launch)
  if (dtype == TO_F64) launch_online_t<double>(L, dims, W, b, X, Y, idx_dev, n, rate, head, G, rpw, lds, s);
  else launch_online_t<float>(L, dims, W, b, X, Y, idx_dev, n, rate, head, G, rpw, lds, s);
  TO_HIP(hipGetLastError());
  count_launch();
}

int online_sgd_status() { return g_on.status ? *g_on.status : 0; }
void online_sgd_reset_status() {
  if (g_on.status) *g_on.status = 0;
}

}  // namespace to
