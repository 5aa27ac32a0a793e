// What an epilogue instruction costs on gfx950, alone and next to a stream of fp32 MFMAs (one wave per SIMD, the
// situation of the short-K GEMM's fused `map logistic`): shader cycles per instruction per wave for v_fma_f32,
// v_pk_fma_f32, v_exp_f32, v_rcp_f32, v_accvgpr_read_b32, ds_write_b128 -- (a) 64 independent instructions back to back,
// (b) the same 64 spread one behind each of 64 v_mfma_f32_32x32x2_f32 (1024 cycles of MFMAs on their own).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { OP_NONE, OP_FMA, OP_PKFMA, OP_EXP, OP_RCP, OP_ACCREAD, OP_DSW128, OP_CHAIN };   // CHAIN: accread, fma, exp, add, rcp

template <int OP, bool WITH_MFMA, int BURST = 1>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, float* sink, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[256 * 4 * 4];
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = seed * (r + j);
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = seed + 0.001f * (lane + i);
  float a = seed + lane * 0.01f, b = 1.0f - seed;
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < 16; ++it) {
#pragma unroll
    for (int n = 0; n < 64; ++n) {
      if (WITH_MFMA) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[n & 3]) : "v"(a), "v"(b));
      if (BURST > 1) {   // BURST independent instructions behind every BURST-th MFMA (same totals as one per MFMA)
        if (n % BURST == BURST - 1) {
#pragma unroll
          for (int u = 0; u < BURST; ++u) {
            float& w = x[u & 15];
            if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(w) : "v"(a), "v"(b));
            if (OP == OP_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(w));
            if (OP == OP_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(w));
            if (OP == OP_PKFMA) {
              f32x2& p = *reinterpret_cast<f32x2*>(&x[(u & 7) * 2]);
              const f32x2 aa = {a, a}, bb = {b, b};
              asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(aa), "v"(bb));
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      float& v = x[n & 15];
      if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
      if (OP == OP_PKFMA) {
        f32x2& p = *reinterpret_cast<f32x2*>(&x[(n & 7) * 2]);
        const f32x2 aa = {a, a}, bb = {b, b};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(aa), "v"(bb));
      }
      if (OP == OP_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
      if (OP == OP_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(v));
      if (OP == OP_ACCREAD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[(n + 2) & 3][n & 15]));
      if (OP == OP_DSW128) {
        f32x4 q = {x[0], x[1], x[2], x[3]};
        asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(threadIdx.x * 16)), "v"(q) : "memory");
      }
      if (OP == OP_CHAIN) {   // one element of the fused logistic: read, scale + bias, exp2, 1 +, 1/
        float t;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(acc[(n + 2) & 3][n & 15]));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(a), "v"(b));
        asm volatile("v_exp_f32 %0, %0" : "+v"(t));
        asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(t));
        asm volatile("v_rcp_f32 %0, %0" : "+v"(t));
        v = t;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long c1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += x[i];
  for (int j = 0; j < 4; ++j) s += acc[j][lane & 15];
  sink[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
  if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
}

template <int OP, bool M, int BURST = 1>
static double run(unsigned long long* d_out, float* d_sink) {
  probe<OP, M, BURST><<<256, 256>>>(d_out, d_sink, 0.5f);   // one workgroup per CU, one wave per SIMD
  hipDeviceSynchronize();
  probe<OP, M, BURST><<<256, 256>>>(d_out, d_sink, 0.5f);
  hipDeviceSynchronize();
  unsigned long long h[256];
  hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < 256; ++i) s += (double)h[i];
  return s / 256 / (16.0 * 64);   // cycles per slot
}

int main() {
  unsigned long long* d_out;
  float* d_sink;
  hipMalloc(&d_out, 256 * 8);
  hipMalloc(&d_sink, 256 * 256 * 4);
  const double base = run<OP_NONE, true>(d_out, d_sink);
  printf("v_mfma_f32_32x32x2_f32 alone: %.1f cycles per MFMA\n", base);
  printf("%-22s %12s %22s\n", "instruction", "alone", "added to one MFMA slot");
#define ROW(NAME, OP) printf("%-22s %12.1f %22.1f\n", NAME, run<OP, false>(d_out, d_sink), run<OP, true>(d_out, d_sink) - base)
  ROW("v_fma_f32", OP_FMA);
  ROW("v_pk_fma_f32", OP_PKFMA);
  ROW("v_exp_f32", OP_EXP);
  ROW("v_rcp_f32", OP_RCP);
  ROW("v_accvgpr_read_b32", OP_ACCREAD);
  ROW("ds_write_b128", OP_DSW128);
  ROW("logistic element (5)", OP_CHAIN);
  printf("\nthe same instructions in bursts behind every n-th MFMA (cycles added per instruction):\n%-14s %8s %8s %8s %8s %8s\n", "", "1", "2", "4", "8", "16");
#define BROW(NAME, OP) printf("%-14s %8.1f %8.1f %8.1f %8.1f %8.1f\n", NAME, run<OP, true, 1>(d_out, d_sink) - base, \
    run<OP, true, 2>(d_out, d_sink) - base, run<OP, true, 4>(d_out, d_sink) - base, run<OP, true, 8>(d_out, d_sink) - base, run<OP, true, 16>(d_out, d_sink) - base)
  BROW("v_fma_f32", OP_FMA);
  BROW("v_pk_fma_f32", OP_PKFMA);
  BROW("v_exp_f32", OP_EXP);
  BROW("v_rcp_f32", OP_RCP);
  return 0;
}
