// What does the matrix pipe sustain on this box?  (1) pure v_mfma_f32_32x32x2_f32 stream,
// (2) the same with the GEMM's LDS fragment reads, at 4 / 2 / 1 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_ceiling mfma_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// LDS_READS: 0 none, 1 fragment reads, 2 + barrier per k-tile, 3 + the GEMM's LDS stores (4 x b32
// transposing + 1 x b128) before the barrier, 4 + two 16-byte global loads per thread per k-tile
// feeding those stores, 5 = like 4 but the LDS stores sit in the middle of the MFMA sequence
template <int LDS_READS>
__global__ __launch_bounds__(1024) void k(float* out, int iters, const float4* __restrict__ src = nullptr) {
  __shared__ float sm[2 * 16 * 520];
  for (int i = threadIdx.x; i < 2 * 16 * 520; i += blockDim.x) sm[i] = 1.0f + (i & 7) * 0.125f;
  __syncthreads();
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float a[2] = {1.0f + lane * 0.001f, 0.5f}, b[2] = {0.25f, 2.0f};
  const float* Ar0 = sm + (wave & 3) * 64 + l31;
  const float* Br0 = sm + 16 * 260 + (wave >> 2) * 64 + l31;
  float4 ra = {1, 2, 3, 4}, rb = {1, 2, 3, 4};
  const int tid = threadIdx.x;
  const float4* gp = src ? src + (size_t)blockIdx.x * 4096 * 2 + tid : nullptr;
  for (int t = 0; t < iters; ++t) {
    float4 na = ra, nb = rb;
    if (LDS_READS == 6) {
      // direct global -> LDS (no VGPR staging, no ds_write): every wave fills one 1 KiB row of each operand
      typedef __attribute__((address_space(1))) const void* gptr_t;
      typedef __attribute__((address_space(3))) void* lptr_t;
      float* Ad = sm + ((t + 1) & 1) * 16 * 520;
      __builtin_amdgcn_global_load_lds((gptr_t)(gp + (size_t)(t & 255) * 8192 * 256), (lptr_t)(Ad + wave * 260), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(gp + (size_t)(t & 255) * 8192 * 256 + 1024),
                                       (lptr_t)(Ad + 16 * 260 + wave * 260), 16, 0, 0);
    } else if (LDS_READS >= 4) {
      na = gp[(size_t)(t & 255) * 8192 * 256];
      nb = gp[(size_t)(t & 255) * 8192 * 256 + 1024];
    }
    const float* Ar = Ar0 + (LDS_READS >= 3 ? (t & 1) * 16 * 520 : 0);
    const float* Br = Br0 + (LDS_READS >= 3 ? (t & 1) * 16 * 520 : 0);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      if (LDS_READS == 5 && kk == 4) {
        float* Ad = sm + ((t + 1) & 1) * 16 * 520;
        const int m = tid / 4, kq = (tid % 4) * 4;
        Ad[(kq + 0) * 260 + m] = na.x; Ad[(kq + 1) * 260 + m] = na.y;
        Ad[(kq + 2) * 260 + m] = na.z; Ad[(kq + 3) * 260 + m] = na.w;
        *reinterpret_cast<float4*>(Ad + 16 * 260 + (tid / 64) * 260 + (tid % 64) * 4) = nb;
      }
      if (LDS_READS) {
        a[0] = Ar[(kk * 2 + half) * 260];
        a[1] = Ar[(kk * 2 + half) * 260 + 32];
        b[0] = Br[(kk * 2 + half) * 260];
        b[1] = Br[(kk * 2 + half) * 260 + 32];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (LDS_READS == 3 || LDS_READS == 4) {  // (6: nothing to store)
      float* Ad = sm + ((t + 1) & 1) * 16 * 520;
      const int m = tid / 4, kq = (tid % 4) * 4;
      Ad[(kq + 0) * 260 + m] = na.x; Ad[(kq + 1) * 260 + m] = na.y;
      Ad[(kq + 2) * 260 + m] = na.z; Ad[(kq + 3) * 260 + m] = na.w;
      *reinterpret_cast<float4*>(Ad + 16 * 260 + (tid / 64) * 260 + (tid % 64) * 4) = nb;
    }
    if (LDS_READS >= 2) __syncthreads();
    ra = na; rb = nb;
  }
  float s = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int L>
void run(const char* name, int threads, int blocks) {
  float* out;
  hipMalloc(&out, 4096 * 1024 * 4);
  static float4* src = nullptr;
  if (!src) { hipMalloc(&src, (size_t)256 * 8192 * 256 * 16 + (1 << 20)); hipMemset(src, 0, (size_t)256 * 8192 * 256 * 16); }
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<L><<<blocks, threads>>>(out, 100, src);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<L><<<blocks, threads>>>(out, iters, src);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * (threads / 64) * iters * 32.0 * 4096.0;
  printf("%-34s threads=%4d blocks=%4d  %.3f ms  %.1f TF\n", name, threads, blocks, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  run<0>("pure MFMA", 1024, 256);
  run<0>("pure MFMA", 512, 256);
  run<0>("pure MFMA", 256, 256);
  run<0>("pure MFMA 2 blocks/CU", 512, 512);
  run<1>("MFMA + LDS fragment reads", 1024, 256);
  run<1>("MFMA + LDS fragment reads", 512, 256);
  run<2>("MFMA + LDS reads + barrier/k-tile", 1024, 256);
  run<2>("MFMA + LDS reads + barrier/k-tile", 512, 512);
  run<3>("+ LDS stores before barrier", 1024, 256);
  run<4>("+ global loads", 1024, 256);
  run<5>("+ global loads, stores mid-tile", 1024, 256);
  run<6>("global_load_lds direct (no ds_write)", 1024, 256);
  return 0;
}
