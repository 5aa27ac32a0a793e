// Does a short pure-MFMA kernel run slower per instruction than a long one (clock ramp / launch cost)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = 1.0f + lane * 0.001f, b = 0.5f;
  for (int t = 0; t < iters; ++t)
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; (void)hipMalloc(&out, 1024 * 1024 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int iters : {32, 64, 128, 256, 512, 1024, 4096}) {
    for (int w = 0; w < 3; ++w) k<<<256, 1024>>>(out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) k<<<256, 1024>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double flops = 256.0 * 16 * iters * 32 * 4096.0;
    printf("iters=%5d  %.4f ms/launch  %.1f TF  (ideal at 157.3: %.4f ms)\n", iters, ms, flops / ms / 1e9, flops / 157.3e12 * 1e3);
  }
  return 0;
}
