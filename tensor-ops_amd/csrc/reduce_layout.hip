// Reductions, broadcasts and layout kernels:
//   sum_axis   -> `sumRows` (src/TensorOps/Types.hs:82-84; BTensor.hs:754-773),
//                 `sumB`/`dot`/`traceB` tails, and the sum over the hidden batch
//   bcast_axis -> `mapRows (\_ -> dtdz)` (the `TO.sumRows` gradient, TOp.hs:155-158)
//   copy_strided -> materialise a `transp` view (Types.hs:71-73; Nested.hs:520-528)
//   fill / rand / diag / get_diag -> `konst`, `genRand`, `diag`, `getDiag`
// All HBM/latency-bound: coalesced accesses, wave64 shuffles, LDS cross-wave step.
#include "common.hpp"

namespace to {

template <class S>
__device__ __forceinline__ S wave_sum(S v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
#define TO_DISPATCH(dtype, CALL)                       \
  do {                                                 \
    if ((dtype) == TO_F64) { using S = double; CALL; } \
    else { using S = float; CALL; }                    \
  } while (0)

// J small: one workgroup per (o, j) output, lanes stride over i.
template <class S>
__global__ __launch_bounds__(256) void sum_axis_rows_kernel(const S* __restrict__ x,
                                                            S* __restrict__ out, long R, long J,
                                                            long so, long si, long sj) {
  __shared__ S part[4];
  const long oj = blockIdx.x;
  const long o = oj / J, j = oj - o * J;
  const S* p = x + o * so + j * sj;
  S acc = S(0);
  for (long i = threadIdx.x; i < R; i += 256) acc += p[i * si];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[oj] = (part[0] + part[1]) + (part[2] + part[3]);
}

// one wave per output (many small rows, e.g. the softmax denominator of every sample)
template <class S>
__global__ __launch_bounds__(256) void sum_axis_wave_kernel(const S* __restrict__ x,
                                                            S* __restrict__ out, long OJ, long R,
                                                            long J, long so, long si, long sj) {
  const long oj = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (oj >= OJ) return;
  const long o = oj / J, j = oj - o * J;
  const S* p = x + o * so + j * sj;
  S acc = S(0);
  for (long i = threadIdx.x & 63; i < R; i += 64) acc += p[i * si];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) out[oj] = acc;
}

// J >= 64 with sj == 1: lanes across j (coalesced), 4 row groups per workgroup over i.
template <class S>
__global__ __launch_bounds__(256) void sum_axis_cols_kernel(const S* __restrict__ x,
                                                            S* __restrict__ out, long R, long J,
                                                            long so, long si, long sj, long r_total) {
  // r_total: rows that really exist when `o` indexes R-chunks of one tall matrix (o*R + i < r_total)
  __shared__ S part[4][64];
  const long o = blockIdx.y;
  const long j = (long)blockIdx.x * 64 + (threadIdx.x & 63);
  const int g = threadIdx.x >> 6;
  S acc = S(0);
  if (j < J) {
    const S* p = x + o * so + j * sj;
    long rmax = R;
    if (r_total >= 0 && o * R + R > r_total) rmax = r_total - o * R;
    for (long i = g; i < rmax; i += 4) acc += p[i * si];
  }
  part[g][threadIdx.x & 63] = acc;
  __syncthreads();
  if (g == 0 && j < J) {
    const int l = threadIdx.x & 63;
    out[o * J + j] = (part[0][l] + part[1][l]) + (part[2][l] + part[3][l]);
  }
}

// huge single reduction (sumB / dot tail): two stages through a partials buffer
template <class S>
__global__ __launch_bounds__(256) void sum_partial_kernel(const S* __restrict__ x,
                                                          S* __restrict__ partial, long n,
                                                          long si) {
  __shared__ S part[4];
  S acc = S(0);
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) acc += x[i * si];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

template <class S>
static void sum_axis_t(int dtype, const S* x, S* out, int64_t O, int64_t R, int64_t J, int64_t so,
                       int64_t si, int64_t sj, hipStream_t s) {
  const int64_t OJ = O * J;
  if (OJ == 0) return;
  if (R == 0) {
    launch_fill(dtype, out, OJ, 0.0, s);
    return;
  }
  // a tall matrix's column sums (sumRows, a batch's bias gradient) from ~1M elements on: gemv.hip's column kernel with no vector
  if (O == 1 && sj == 1 && launch_column_sum(dtype, x, out, R, J, si, s)) return;
  if (OJ == 1 && R >= (1 << 16)) {
    const int64_t nb = 1024;
    Holder tmp(new_tensor(1, &nb, 0, dtype));  // a tracked temporary: stays reserved if a graph is capturing
    launch_k(sum_partial_kernel<S>, dim3((unsigned)nb), dim3(256), 0, s, x, (S*)tmp.t->ptr, (long)R,
                       (long)si);
    launch_k(sum_axis_rows_kernel<S>, dim3(1), dim3(256), 0, s, (const S*)tmp.t->ptr, out,
                       (long)nb, 1L, 0L, 1L, 0L);
    TO_HIP(hipGetLastError());
    count_launch();
    count_launch();
    return;
  }
  // tall reduction with few outputs (e.g. the batch sum of bias gradients, [1024,256] -> [256]):
  // two deterministic passes, R split over `rs` workgroup rows so the whole chip takes part
  if (O == 1 && J >= 64 && sj == 1 && R >= 256 && (J + 63) / 64 < 128) {
    int64_t rs = 256 / ((J + 63) / 64);
    if (rs > R / 16) rs = R / 16;
    if (rs >= 2) {
      const int64_t chunk = (R + rs - 1) / rs;
      const int64_t pd[2] = {rs, J};
      Holder tmp(new_tensor(2, pd, 0, dtype));
      dim3 grid((unsigned)((J + 63) / 64), (unsigned)rs);
      launch_k(sum_axis_cols_kernel<S>, grid, dim3(256), 0, s, x, (S*)tmp.t->ptr, (long)chunk, (long)J,
                         (long)(chunk * si), (long)si, (long)sj, (long)R);
      launch_k(sum_axis_cols_kernel<S>, dim3((unsigned)((J + 63) / 64), 1), dim3(256), 0, s,
                         (const S*)tmp.t->ptr, out, (long)rs, (long)J, 0L, (long)J, 1L, (long)rs);
      TO_HIP(hipGetLastError());
      count_launch();
      count_launch();
      return;
    }
  }
  if (J >= 64 && sj == 1) {
    dim3 grid((unsigned)((J + 63) / 64), (unsigned)O);
    launch_k(sum_axis_cols_kernel<S>, grid, dim3(256), 0, s, x, out, (long)R, (long)J,
                       (long)so, (long)si, (long)sj, -1L);
  } else if (R <= 256 && OJ >= 64) {
    launch_k(sum_axis_wave_kernel<S>, dim3((unsigned)((OJ + 3) / 4)), dim3(256), 0, s, x, out,
                       (long)OJ, (long)R, (long)J, (long)so, (long)si, (long)sj);
  } else {
    launch_k(sum_axis_rows_kernel<S>, dim3((unsigned)OJ), dim3(256), 0, s, x, out, (long)R,
                       (long)J, (long)so, (long)si, (long)sj);
  }
  TO_HIP(hipGetLastError());
  count_launch();
}

void launch_sum_axis(int dtype, const void* x, void* out, int64_t O, int64_t R, int64_t J, int64_t so,
                     int64_t si, int64_t sj, hipStream_t s) {
  TO_DISPATCH(dtype, sum_axis_t<S>(dtype, (const S*)x, (S*)out, O, R, J, so, si, sj, s));
}

// C[m, n] (row stride c_sm) = sum over the splits of a [splits][M][N] workspace: the second pass of a split-K GEMM
// whose output is a block of a larger matrix (four consecutive columns per lane; N % 4 == 0)
__global__ __launch_bounds__(256) void sum_splits_strided_kernel(const float* __restrict__ w, float* __restrict__ C, int splits,
                                                                  long M, long N, long c_sm) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const long nq = N / 4, total = M * nq, stride = (long)gridDim.x * blockDim.x;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const long m = e / nq, q = e - m * nq;
    f32x4 acc = *reinterpret_cast<const f32x4*>(w + m * N + 4 * q);
    for (int sp = 1; sp < splits; ++sp) acc += *reinterpret_cast<const f32x4*>(w + ((long)sp * M + m) * N + 4 * q);
    *reinterpret_cast<f32x4*>(C + m * c_sm + 4 * q) = acc;   // (callers: C 16-byte aligned, c_sm a multiple of 4)
  }
}

void launch_sum_splits_strided(const void* work, void* C, int splits, int64_t M, int64_t N, int64_t c_sm, hipStream_t s) {
  const long total = M * (N / 4);
  if (total == 0) return;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  launch_k(sum_splits_strided_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)work, (float*)C, splits, (long)M,
           (long)N, (long)c_sm);
  TO_HIP(hipGetLastError());
  count_launch();
}

template <class S>
__global__ void bcast_axis_kernel(const S* __restrict__ d, S* __restrict__ out, long total,
                                  long R, long J, long dso) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const long j = e % J;
    const long o = e / (R * J);
    out[e] = d[o * dso + j];
  }
}

void launch_bcast_axis(int dtype, const void* d, void* out, int64_t O, int64_t R, int64_t J, int64_t dso,
                       hipStream_t s) {
  const long total = O * R * J;
  if (total == 0) return;
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  TO_DISPATCH(dtype, launch_k(bcast_axis_kernel<S>, dim3((unsigned)blocks), dim3(256), 0, s, (const S*)d,
                                        (S*)out, total, (long)R, (long)J, (long)dso));
  TO_HIP(hipGetLastError());
  count_launch();
}

struct CopyDims {
  long dims[TO_MAX_RANK + 1];
  long strides[TO_MAX_RANK + 1];
  int rank;
};

// general gather: consecutive threads write consecutive packed elements
template <class S>
__global__ void copy_strided_kernel(const S* __restrict__ src, S* __restrict__ dst,
                                    CopyDims c, long total) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    long rem = e, off = 0;
#pragma unroll
    for (int d = TO_MAX_RANK; d >= 0; --d) {
      if (d < c.rank) {
        const long i = rem % c.dims[d];
        rem /= c.dims[d];
        off += i * c.strides[d];
      }
    }
    dst[e] = src[off];
  }
}

// 2-D transpose through a 64x65 LDS tile: both global sides coalesced
template <class S>
__global__ __launch_bounds__(256) void transpose2d_kernel(const S* __restrict__ src,
                                                          S* __restrict__ dst, long rows,
                                                          long cols, long s_row, long s_col,
                                                          long batch_src, long batch_dst) {
  // dst[b][r][c] (packed rows x cols) = src[b*batch_src + r*s_row + c*s_col], s_row == 1
  __shared__ S tile[64][65];
  const long b = blockIdx.z;
  const long r0 = (long)blockIdx.y * 64, c0 = (long)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int k = ty; k < 64; k += 4) {  // read: consecutive lanes along r (stride 1 in src)
    const long r = r0 + tx, c = c0 + k;
    if (r < rows && c < cols) tile[k][tx] = src[b * batch_src + r * s_row + c * s_col];
  }
  __syncthreads();
  for (int k = ty; k < 64; k += 4) {  // write: consecutive lanes along c (stride 1 in dst)
    const long r = r0 + k, c = c0 + tx;
    if (r < rows && c < cols) dst[b * batch_dst + r * cols + c] = tile[tx][k];
  }
}

template <class S>
static void copy_strided_t(const S* src, S* dst, int rank, const int64_t* dims, const int64_t* strides,
                           hipStream_t s) {
  long total = 1;
  for (int i = 0; i < rank; ++i) total *= dims[i];
  if (total == 0) return;
  // drop size-1 dims, merge mergeable neighbours
  long d[TO_MAX_RANK + 1], st[TO_MAX_RANK + 1];
  int r = 0;
  for (int i = 0; i < rank; ++i) {
    if (dims[i] == 1) continue;
    if (r > 0 && st[r - 1] == strides[i] * dims[i]) {
      d[r - 1] *= dims[i];
      st[r - 1] = strides[i];
    } else {
      d[r] = dims[i];
      st[r] = strides[i];
      ++r;
    }
  }
  if (r == 0) { d[0] = 1; st[0] = 1; r = 1; }
  if (r == 1 && st[0] == 1) {
    TO_HIP(hipMemcpyAsync(dst, src, total * sizeof(S), hipMemcpyDeviceToDevice, s));
    count_launch();
    return;
  }
  // [rows, cols] with the row index contiguous in src (a transposed matrix), optional batch
  if ((r == 2 && st[0] == 1) || (r == 3 && st[1] == 1)) {
    const int o = (r == 3) ? 1 : 0;
    const long nb = (r == 3) ? d[0] : 1, bs = (r == 3) ? st[0] : 0;
    const long rows = d[o], cols = d[o + 1];
    if (nb <= 65535 && (rows + 63) / 64 <= 65535) {
      dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64), (unsigned)nb);
      launch_k(transpose2d_kernel<S>, grid, dim3(256), 0, s, src, dst, rows, cols, st[o],
                         st[o + 1], bs, rows * cols);
      TO_HIP(hipGetLastError());
      count_launch();
      return;
    }
  }
  CopyDims c{};
  c.rank = r;
  for (int i = 0; i < r; ++i) { c.dims[i] = d[i]; c.strides[i] = st[i]; }
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  launch_k(copy_strided_kernel<S>, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, c,
                     total);
  TO_HIP(hipGetLastError());
  count_launch();
}

void launch_copy_strided(int dtype, const void* src, void* dst, int rank, const int64_t* dims,
                         const int64_t* strides, hipStream_t s) {
  TO_DISPATCH(dtype, copy_strided_t<S>((const S*)src, (S*)dst, rank, dims, strides, s));
}

template <class S>
__global__ void fill_kernel(S* __restrict__ dst, long n, S v) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = v;
}

void launch_fill(int dtype, void* dst, int64_t n, double v, hipStream_t s) {
  if (n == 0) return;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  TO_DISPATCH(dtype, launch_k(fill_kernel<S>, dim3((unsigned)blocks), dim3(256), 0, s, (S*)dst, (long)n, (S)v));
  TO_HIP(hipGetLastError());
  count_launch();
}

// counter-based generator: splitmix64(seed + golden*index); the same integer recipe is
// restated on the host in tests so uniform draws are bit-identical CPU/GPU.
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

template <class S>
__global__ void rand_kernel(S* __restrict__ dst, long n, int dist, S a, S b, uint64_t seed) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t h = splitmix64(seed + 0x9e3779b97f4a7c15ull * (uint64_t)i);
    if constexpr (sizeof(S) == 4) {
      const float u1 = (float)(h >> 40) * (1.0f / 16777216.0f);  // [0,1), 24 bits
      if (dist == 0) {
        dst[i] = a + (b - a) * u1;
      } else if (dist >= 2) {   // inverse CDF on one uniform draw
        dst[i] = dist == 2 ? -logf(1.0f - u1) / a
               : dist == 3 ? a + b * tanf(3.14159265358979323846f * (u1 - 0.5f))
                           : a - b * copysignf(logf(1.0f - 2.0f * fabsf(u1 - 0.5f)), u1 - 0.5f);
      } else {
        const float u2 = (float)((h >> 16) & 0xffffff) * (1.0f / 16777216.0f);
        const float rr = sqrtf(-2.0f * logf(1.0f - u1));  // 1-u1 in (0,1]
        dst[i] = a + b * rr * cosf(6.28318530717958647692f * u2);
      }
    } else {
      const double u1 = (double)(h >> 11) * (1.0 / 9007199254740992.0);  // [0,1), 53 bits
      if (dist == 0) {
        dst[i] = a + (b - a) * u1;
      } else if (dist >= 2) {
        dst[i] = dist == 2 ? -log(1.0 - u1) / a
               : dist == 3 ? a + b * tan(3.14159265358979323846 * (u1 - 0.5))
                           : a - b * copysign(log(1.0 - 2.0 * fabs(u1 - 0.5)), u1 - 0.5);
      } else {
        const uint64_t h2 = splitmix64(h);
        const double u2 = (double)(h2 >> 11) * (1.0 / 9007199254740992.0);
        const double rr = sqrt(-2.0 * log(1.0 - u1));
        dst[i] = a + b * rr * cos(6.28318530717958647692 * u2);
      }
    }
  }
}

void launch_rand(int dtype, void* dst, int64_t n, int dist, double a, double b, uint64_t seed, hipStream_t s) {
  if (n == 0) return;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  TO_DISPATCH(dtype, launch_k(rand_kernel<S>, dim3((unsigned)blocks), dim3(256), 0, s, (S*)dst, (long)n,
                                        dist, (S)a, (S)b, seed));
  TO_HIP(hipGetLastError());
  count_launch();
}

template <class S>
__global__ void diag_kernel(const S* __restrict__ x, S* __restrict__ out, long n, long step) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i * step] = x[i];
}

void launch_diag(int dtype, const void* x, void* out, int64_t n, int rank, hipStream_t s) {
  if (n == 0) return;
  long step = 0, p = 1;
  for (int d = 0; d < rank; ++d) { step += p; p *= n; }  // 1 + n + n^2 + ...
  TO_DISPATCH(dtype, launch_k(diag_kernel<S>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                                        (const S*)x, (S*)out, (long)n, step));
  TO_HIP(hipGetLastError());
  count_launch();
}

// argMax per row, one wave per row; ties -> earliest index (see include/tensorops_hip.h)
template <class S>
__global__ __launch_bounds__(256) void arg_max_rows_kernel(const S* __restrict__ x,
                                                           long long* __restrict__ out, long B, long n,
                                                           long bstride, long stride, S sgn) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const int lane = threadIdx.x & 63;
  const S* p = x + row * bstride;
  S best = S(0);
  long bi = -1;
  for (long j = lane; j < n; j += 64) {
    const S v = sgn * p[j * stride];  // sgn = -1: argMin (Min/Arg keeps the left element on ties, like Max/Arg)
    if (bi < 0 || !(best >= v)) { best = v; bi = j; }   // same comparison chain as the Max/Arg fold
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const S ob = __shfl_xor(best, off, 64);
    const long oi = __shfl_xor((long long)bi, off, 64);
    // combine two partial winners: the left (smaller index) one wins unless the other is strictly greater
    if (oi >= 0 && (bi < 0 || (oi < bi ? !(ob < best) : !(best >= ob)))) { best = ob; bi = oi; }
  }
  if (lane == 0) out[row] = bi;
}

void launch_arg_max_rows(int dtype, const void* x, long long* out, int64_t B, int64_t n, int64_t bstride,
                         int64_t stride, hipStream_t s, bool minimum) {
  if (B == 0) return;
  TO_DISPATCH(dtype, launch_k(arg_max_rows_kernel<S>, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s,
                                        (const S*)x, out, (long)B, (long)n, (long)bstride, (long)stride, minimum ? S(-1) : S(1)));
  TO_HIP(hipGetLastError());
  count_launch();
}

template <class S>
__global__ void one_hot_kernel(S* __restrict__ out, const long long* __restrict__ idx, long B, long n,
                               S hot, S cold) {
  const long stride = (long)gridDim.x * blockDim.x, total = B * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride)
    out[e] = ((e % n) == idx[e / n]) ? hot : cold;
}

void launch_one_hot(int dtype, void* out, const long long* idx, int64_t B, int64_t n, double hot, double cold,
                    hipStream_t s) {
  const long total = B * n;
  if (total == 0) return;
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  TO_DISPATCH(dtype, launch_k(one_hot_kernel<S>, dim3((unsigned)blocks), dim3(256), 0, s, (S*)out, idx,
                                        (long)B, (long)n, (S)hot, (S)cold));
  TO_HIP(hipGetLastError());
  count_launch();
}

// n contiguous copies in one launch: segment s covers dwords [off[s], off[s+1]) of the concatenation
struct MultiCopyArgs {
  const unsigned* src[16];
  unsigned* dst[16];
  long off[17];
  int n;
};
__global__ void multi_copy_kernel(MultiCopyArgs a) {
  const long stride = (long)gridDim.x * blockDim.x, total = a.off[a.n];
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    int s = 0;
#pragma unroll
    for (int k = 1; k < 16; ++k) s += (k < a.n && e >= a.off[k]) ? 1 : 0;
    a.dst[s][e - a.off[s]] = a.src[s][e - a.off[s]];
  }
}

// ... 16 bytes a thread when every piece is 16-byte aligned and a whole number of quads (a 64 MB parameter matrix: 75 us a dword at a
// time, 1.7 TB/s; tools/step_scan.py)
__global__ void multi_copy4_kernel(MultiCopyArgs a) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const long stride = (long)gridDim.x * blockDim.x, total = a.off[a.n];   // (offsets in quads)
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    int s = 0;
#pragma unroll
    for (int k = 1; k < 16; ++k) s += (k < a.n && e >= a.off[k]) ? 1 : 0;
    reinterpret_cast<u32x4*>(a.dst[s])[e - a.off[s]] = reinterpret_cast<const u32x4*>(a.src[s])[e - a.off[s]];
  }
}

void launch_multi_copy(int n, const void* const* srcs, void* const* dsts, const int64_t* dwords, hipStream_t s) {
  MultiCopyArgs a{};
  a.n = n;
  bool quads = true;
  for (int i = 0; i < n; ++i)
    quads = quads && dwords[i] % 4 == 0 && ((reinterpret_cast<uintptr_t>(srcs[i]) | reinterpret_cast<uintptr_t>(dsts[i])) & 15u) == 0;
  long total = 0;
  for (int i = 0; i < n; ++i) {
    a.src[i] = (const unsigned*)srcs[i];
    a.dst[i] = (unsigned*)dsts[i];
    a.off[i] = total;
    total += quads ? dwords[i] / 4 : dwords[i];
  }
  for (int i = n; i <= 16; ++i) a.off[i] = total;
  if (total == 0) return;
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (quads) launch_k(multi_copy4_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
  else launch_k(multi_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
  TO_HIP(hipGetLastError());
  count_launch();
}

// `mapRows` / `ixRows` results (src/TensorOps/Types.hs:77-81,100-106): out[b][r0 + k][:] = row_k[b][:] for up to 16 rows
// per launch (dword granularity: an fp64 element is two)
struct StackArgs {
  const unsigned* src[16];
  long sb[16];  // dwords between the samples of row k (0: the row is shared by every sample)
  int n;
};
__global__ void stack_rows_kernel(StackArgs a, unsigned* __restrict__ out, long B, long row_dw, long out_sb) {
  const long stride = (long)gridDim.x * blockDim.x, per = (long)a.n * row_dw, total = B * per;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const long b = e / per, rem = e - b * per;
    const int k = (int)(rem / row_dw);
    const long j = rem - (long)k * row_dw;
    const unsigned* s = a.src[0];
    long sb = a.sb[0];
#pragma unroll
    for (int q = 1; q < 16; ++q)
      if (q == k) {
        s = a.src[q];
        sb = a.sb[q];
      }
    out[b * out_sb + rem] = s[b * sb + j];
  }
}

void launch_stack_rows(int n, const void* const* rows, const int64_t* row_sb_dwords, void* out, int64_t B,
                       int64_t row_dwords, int64_t out_sb_dwords, hipStream_t s) {
  if (n == 0 || B == 0 || row_dwords == 0) return;
  StackArgs a{};
  a.n = n;
  for (int i = 0; i < n; ++i) {
    a.src[i] = (const unsigned*)rows[i];
    a.sb[i] = (long)row_sb_dwords[i];
  }
  const long total = (long)B * n * row_dwords;
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  launch_k(stack_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, (unsigned*)out, (long)B, (long)row_dwords,
           (long)out_sb_dwords);
  TO_HIP(hipGetLastError());
  count_launch();
}

// out[k][:] = x[idx[k]][:]  (row gather over the hidden batch; 16-byte chunks when rows allow)
template <class V>
__global__ void gather_rows_kernel(const V* __restrict__ x, V* __restrict__ out,
                                   const long long* __restrict__ idx, long n_rows, long row_v) {
  const long stride = (long)gridDim.x * blockDim.x, total = n_rows * row_v;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const long k = e / row_v, j = e - k * row_v;
    out[e] = x[idx[k] * row_v + j];
  }
}

void launch_gather_rows(const void* x, void* out, const long long* idx, int64_t n_rows, int64_t row_bytes,
                        hipStream_t s) {
  if (n_rows == 0 || row_bytes == 0) return;
  const bool v16 = row_bytes % 16 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0;
  const long row_v = v16 ? row_bytes / 16 : row_bytes / 4;
  long blocks = (n_rows * row_v + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (v16)
    launch_k(gather_rows_kernel<float4>, dim3((unsigned)blocks), dim3(256), 0, s, (const float4*)x,
                       (float4*)out, idx, (long)n_rows, row_v);
  else
    launch_k(gather_rows_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)x,
                       (float*)out, idx, (long)n_rows, row_v);
  TO_HIP(hipGetLastError());
  count_launch();
}

template <class S>
__global__ void get_diag_kernel(const S* __restrict__ x, S* __restrict__ out, long n,
                                long step) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = x[i * step];
}

void launch_get_diag(int dtype, const void* x, void* out, int64_t n, int64_t step, hipStream_t s) {
  if (n == 0) return;
  TO_DISPATCH(dtype, launch_k(get_diag_kernel<S>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                                        (const S*)x, (S*)out, (long)n, (long)step));
  TO_HIP(hipGetLastError());
  count_launch();
}

}  // namespace to
