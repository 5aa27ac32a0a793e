// Pre-fused pieces of the batched ffLayer-stack gradient (the benchmarked network of
// app/MNIST.hs:264-265 / app/Dots.hs:72-73).  Same mathematics as the generic TOp path
// (src/TensorOps/Learn/NeuralNet.hs:42-77, FeedForward.hs:178-214), collapsed by hand:
//   softmax >>> crossEntropy backward:  dz = softmax(z) * sum(y) - y
//   logistic >>> squaredError backward: dz = -2 (y - s) s (1 - s),  s = logistic(z)
// One wave per sample row; the row lives in registers (n <= 64 per pass, looped beyond).
#include "common.hpp"

namespace to {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// kind 0: softmax + crossEntropy ; kind 1: logistic + squaredError.
// z: [B, n] pre-activations of the last layer; y: [B, n]; dz: [B, n]; loss (optional): [B]
__global__ __launch_bounds__(256) void loss_grad_rows_kernel(const float* __restrict__ z,
                                                             const float* __restrict__ y,
                                                             float* __restrict__ dz,
                                                             float* __restrict__ loss, long B, int n,
                                                             int kind) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const int lane = threadIdx.x & 63;
  const float* zr = z + row * n;
  const float* yr = y + row * n;
  float* dr = dz + row * n;
  if (kind == 0) {
    // the reference computes exp z / sum exp z without max-subtraction (NeuralNet.hs:52-59);
    // subtracting the row max is the same value in exact arithmetic and avoids overflow
    float mx = -INFINITY;
    for (int j = lane; j < n; j += 64) mx = fmaxf(mx, zr[j]);
    mx = wmax(mx);
    float se = 0.f, sy = 0.f;
    for (int j = lane; j < n; j += 64) {
      se += expf(zr[j] - mx);
      sy += yr[j];
    }
    se = wsum(se);
    sy = wsum(sy);
    const float inv = 1.0f / se;
    float l = 0.f;
    for (int j = lane; j < n; j += 64) {
      const float p = expf(zr[j] - mx) * inv;
      dr[j] = p * sy - yr[j];
      l -= yr[j] * logf(p);
    }
    if (loss) {
      l = wsum(l);
      if (lane == 0) loss[row] = l;
    }
  } else {
    float l = 0.f;
    for (int j = lane; j < n; j += 64) {
      const float s = 1.0f / (1.0f + expf(-zr[j]));
      const float e = yr[j] - s;
      dr[j] = -2.0f * e * s * (1.0f - s);
      l += e * e;
    }
    if (loss) {
      l = wsum(l);
      if (lane == 0) loss[row] = l;
    }
  }
}

void launch_loss_grad_rows(const float* z, const float* y, float* dz, float* loss, int64_t B,
                           int64_t n, int kind, hipStream_t s) {
  if (B == 0 || n == 0) return;
  hipLaunchKernelGGL(loss_grad_rows_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, z, y, dz,
                     loss, (long)B, (int)n, kind);
  TO_HIP(hipGetLastError());
  count_launch();
}

}  // namespace to
