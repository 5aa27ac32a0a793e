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
    hipLaunchKernelGGL(loss_grad_rows_kernel<double>, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s,
                       (const double*)z, (const double*)y, (double*)dz, (double*)loss, (long)B, (int)n, kind);
  else
    hipLaunchKernelGGL(loss_grad_rows_kernel<float>, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s,
                       (const float*)z, (const float*)y, (float*)dz, (float*)loss, (long)B, (int)n, kind);
  TO_HIP(hipGetLastError());
  count_launch();
}

}  // namespace to
