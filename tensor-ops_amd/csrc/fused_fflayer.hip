// Pre-fused pieces of the batched ffLayer-stack gradient (the benchmarked network of
// app/MNIST.hs:264-265 / app/Dots.hs:72-73).  Same mathematics as the generic TOp path
// (src/TensorOps/Learn/NeuralNet.hs:42-77, FeedForward.hs:178-214), collapsed by hand:
//   softmax >>> crossEntropy backward:  dz = softmax(z) * sum(y) - y
//   logistic >>> squaredError backward: dz = -2 (y - s) s (1 - s),  s = logistic(z)
// One wave per sample row; the row lives in registers (n <= 64 per pass, looped beyond).
#include "common.hpp"

namespace to {

template <class S>
__device__ __forceinline__ S wsum(S v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float exp_s(float x) { return expf(x); }
__device__ __forceinline__ double exp_s(double x) { return exp(x); }
__device__ __forceinline__ float log_s(float x) { return logf(x); }
__device__ __forceinline__ double log_s(double x) { return log(x); }

// kind 0: softmax + crossEntropy ; kind 1: logistic + squaredError.
// z: [B, n] pre-activations of the last layer; y: [B, n]; dz: [B, n]; loss (optional): [B]
template <class S>
__global__ __launch_bounds__(256) void loss_grad_rows_kernel(const S* __restrict__ z, const S* __restrict__ y,
                                                             S* __restrict__ dz, S* __restrict__ loss, long B,
                                                             int n, int kind) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const int lane = threadIdx.x & 63;
  const S* zr = z + row * n;
  const S* yr = y + row * n;
  S* dr = dz + row * n;
  if (kind == 0) {
    // the reference computes exp z / sum exp z without max-subtraction (NeuralNet.hs:52-59);
    // subtracting the row max is the same value in exact arithmetic and avoids overflow
    S mx = S(-INFINITY);
    for (int j = lane; j < n; j += 64) mx = zr[j] > mx ? zr[j] : mx;
    mx = wmax(mx);
    S se = S(0), sy = S(0);
    for (int j = lane; j < n; j += 64) {
      se += exp_s(zr[j] - mx);
      sy += yr[j];
    }
    se = wsum(se);
    sy = wsum(sy);
    const S inv = S(1) / se;
    S l = S(0);
    for (int j = lane; j < n; j += 64) {
      const S p = exp_s(zr[j] - mx) * inv;
      dr[j] = p * sy - yr[j];
      l -= yr[j] * log_s(p);
    }
    if (loss) {
      l = wsum(l);
      if (lane == 0) loss[row] = l;
    }
  } else {
    S l = S(0);
    for (int j = lane; j < n; j += 64) {
      const S s = S(1) / (S(1) + exp_s(-zr[j]));
      const S e = yr[j] - s;
      dr[j] = S(-2) * e * s * (S(1) - s);
      l += e * e;
    }
    if (loss) {
      l = wsum(l);
      if (lane == 0) loss[row] = l;
    }
  }
}

void launch_loss_grad_rows(int dtype, const void* z, const void* y, void* dz, void* loss, int64_t B, int64_t n,
                           int kind, hipStream_t s) {
  if (B == 0 || n == 0) return;
  if (dtype == TO_F64)
    launch_k(loss_grad_rows_kernel<double>, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s,
                       (const double*)z, (const double*)y, (double*)dz, (double*)loss, (long)B, (int)n, kind);
  else
    launch_k(loss_grad_rows_kernel<float>, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s,
                       (const float*)z, (const float*)y, (float*)dz, (float*)loss, (long)B, (int)n, kind);
  TO_HIP(hipGetLastError());
  count_launch();
}

// ---- one-sample step: every layer's weight gradient is an outer product -----------------------------
// gW_l = dz_l (x) a_{l-1}, gb_l = dz_l for ALL layers in one launch (online SGD, app/MNIST.hs:390-396: the
// reference trains sample by sample).  acc: P <- P + alpha * gradient in place (alpha = -rate) instead.
// A row of W_l and its bias element form one run of cols + 1 items, so that one index space covers both.
template <class S>
struct Rank1Many {
  int n;
  const S* dz[RANK1_MAX_LAYERS];
  const S* a[RANK1_MAX_LAYERS];
  S* w[RANK1_MAX_LAYERS];
  S* b[RANK1_MAX_LAYERS];           // may be null: no bias output for that layer
  const S* w_in[RANK1_MAX_LAYERS];  // null: W = alpha * dz (x) a ; else W = w_in + alpha * dz (x) a  (w_in may be w: in place)
  const S* b_in[RANK1_MAX_LAYERS];
  S alpha[RANK1_MAX_LAYERS];
  int cols[RANK1_MAX_LAYERS];
  long start[RANK1_MAX_LAYERS + 1];  // first item of layer l; start[n] = total
};

template <class S>
__global__ __launch_bounds__(256) void rank1_many_kernel(Rank1Many<S> g) {
  const long total = g.start[g.n];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int l = 0;
#pragma unroll
    for (int q = 1; q < RANK1_MAX_LAYERS; ++q) l += (q < g.n && i >= g.start[q]) ? 1 : 0;
    const long loc = i - g.start[l];
    const int cols = g.cols[l];
    const long r = loc / (cols + 1);
    const int c = (int)(loc - r * (cols + 1));
    const S d = g.dz[l][r];
    if (c < cols) {
      const S v = g.alpha[l] * (d * g.a[l][c]);
      g.w[l][r * cols + c] = g.w_in[l] ? g.w_in[l][r * cols + c] + v : v;
    } else if (g.b[l]) {
      const S v = g.alpha[l] * d;
      g.b[l][r] = g.b_in[l] ? g.b_in[l][r] + v : v;
    }
  }
}

template <class S>
static void launch_rank1_t(int n, const void* const* dz, const void* const* a, void* const* w, void* const* b,
                           const void* const* w_in, const void* const* b_in, const double* alpha,
                           const int64_t* rows, const int64_t* cols, hipStream_t s) {
  Rank1Many<S> g{};
  g.n = n;
  long total = 0;
  for (int l = 0; l < n; ++l) {
    g.dz[l] = (const S*)dz[l]; g.a[l] = (const S*)a[l]; g.w[l] = (S*)w[l]; g.b[l] = (S*)b[l];
    g.w_in[l] = (const S*)w_in[l]; g.b_in[l] = (const S*)b_in[l];
    g.alpha[l] = (S)alpha[l];
    g.cols[l] = (int)cols[l];
    g.start[l] = total;
    total += rows[l] * (cols[l] + 1);
  }
  for (int l = n; l <= RANK1_MAX_LAYERS; ++l) g.start[l] = total;
  if (total == 0) return;
  const unsigned blocks = (unsigned)std::min<long>((total + 255) / 256, 4096);
  launch_k(rank1_many_kernel<S>, dim3(blocks), dim3(256), 0, s, g);
  TO_HIP(hipGetLastError());
  count_launch();
}

void launch_rank1_general(int dtype, int n, const void* const* dz, const void* const* a, void* const* w, void* const* b,
                          const void* const* w_in, const void* const* b_in, const double* alpha, const int64_t* rows,
                          const int64_t* cols, hipStream_t s) {
  TO_CHECK(n >= 1 && n <= RANK1_MAX_LAYERS, TO_ERR_ARG, "rank-1 update: 1..8 layers");
  if (dtype == TO_F64) launch_rank1_t<double>(n, dz, a, w, b, w_in, b_in, alpha, rows, cols, s);
  else launch_rank1_t<float>(n, dz, a, w, b, w_in, b_in, alpha, rows, cols, s);
}

void launch_rank1_many(int dtype, int n, const void* const* dz, const void* const* a, void* const* w, void* const* b,
                       const int64_t* rows, const int64_t* cols, double alpha, bool acc, hipStream_t s) {
  TO_CHECK(n >= 1 && n <= RANK1_MAX_LAYERS, TO_ERR_ARG, "rank-1 update: 1..8 layers");
  const void* wi[RANK1_MAX_LAYERS];
  const void* bi[RANK1_MAX_LAYERS];
  double al[RANK1_MAX_LAYERS];
  for (int l = 0; l < n; ++l) {
    wi[l] = acc ? w[l] : nullptr;
    bi[l] = acc ? b[l] : nullptr;
    al[l] = acc ? alpha : 1.0;
  }
  launch_rank1_general(dtype, n, dz, a, w, b, wi, bi, al, rows, cols, s);
}

}  // namespace to
