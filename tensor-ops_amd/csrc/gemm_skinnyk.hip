// Short-K streaming GEMM for gfx950: C[M,N] = act(alpha * A[M,K] . B[K,N] + bias), K in {16, 32, 64}, N a multiple
// of 256, M in the hundreds of thousands -- BASELINE config 5, `gmul '[512,512,64] x '[64,512]` (+ mapped
// logistic): M = 262144, K = 64, N = 512, 17.18 GFLOP against 604 MB (537 MB of it the store of C).  The
// reference maps 512 boxed per-slice GEMMs (src/TensorOps/Backend/BTensor.hs:695-713) and then `cmap`s the
// closure over the result (src/TensorOps/Learn/NeuralNet.hs:42-44).
//
// Why a kernel of its own (version 1, kept selectable with TOPS_SKINNYK_V=1 for comparison).  With K = 64 a 256x256 tile is four k-steps of MFMAs and then 256 KB of stores; the
// tiled kernels alternate the two phases behind workgroup barriers and the matrix pipe idles while C drains
// (0.19-0.20 ms, 57 % of the MFMA bound).  Here nothing is shared between waves after the prologue, so nothing
// has to be waited for collectively:
//   * a workgroup owns one 256-column panel: its slice of B (<= 64 KiB) is transposed into LDS once ([n][k],
//     16-byte groups XOR-swizzled) and stays;
//   * every wave owns a stream of 32-row blocks of that panel: its A rows come straight from global memory into
//     the MFMA fragment layout (eight 16-byte loads per lane per block, issued a block ahead and BEFORE the
//     previous block's stores, so the in-order memory counter returns them first), B fragments by ds_read_b128;
//   * 256 v_mfma_f32_32x32x2_f32 per block into 128 accumulators;
//   * the block leaves through a WAVE-PRIVATE 8-row LDS strip, four passes: the memory system wants whole
//     1 KiB rows per store instruction (measured on this shape: dword stores straight from the MFMA layout, two
//     128-byte lines per instruction, 2.3 TB/s; 16-byte stores of 32-byte row pieces worse; whole rows 5.5 TB/s);
//   * no barrier in the loop: the eight waves of a workgroup (two per SIMD) run out of phase, so one wave's
//     stores -- and the `logistic` of a fused map, VALU/transcendental work -- sit under the other's MFMAs.
// Bound: max(MFMA 109 us, HBM 76 us at spec).
#include <algorithm>

#include "common.hpp"

namespace to {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct SkinnyArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  long M;
  int N, K;
  long a_sm;        // A row stride (elements); A is k-contiguous
  long b_sk, b_sn;  // B element strides
  long c_sm;
  float alpha;
  int npanels;      // N / 256
  int nrb;          // M / 32
  int stagger;
  int xcd_pairs;    // 1: the workgroups that stream the SAME rows through different panels sit on one XCD (see wg_map)
  unsigned long long* dbg;   // development build, TOPS_SKINNYK_DBG=1: eight 100 MHz stamps per wave (tools/c5_stamps.py)
  // sibling products in one launch (GemmProblem::a_table): row block b belongs to matrix b / bpm, whose rows start at a_tab[.]
  const float* const* a_tab;
  int bpm;                   // 32-row blocks per matrix
};

#ifdef TOPS_AB_KNOBS
#define SK_STAMP(i) do { if (g.dbg && lane == 0) g.dbg[((long)blockIdx.x * 4 + wave) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define SK_STAMP(i) do { } while (0)
#endif

// Workgroup -> (panel, stream).  Workgroup b runs on XCD b % 8.  The npanels workgroups that walk the same row blocks
// (one per 256-column panel) read the same A rows at the same pace; placed on ONE XCD the second reader finds them in
// that XCD's L2 instead of fetching them from HBM again (config 5: A fetched 134.9 MB for 67.1 MB; the kernel runs at
// the socket power cap, so traffic is time).
__device__ __forceinline__ void wg_map(const SkinnyArgs& g, int b, int nwg, int* panel, int* wg_in_panel, int* wgs_per_panel) {
  *wgs_per_panel = nwg / g.npanels;
  if (g.xcd_pairs && nwg % (8 * g.npanels) == 0) {
    *panel = (b >> 3) % g.npanels;
    *wg_in_panel = (b / (8 * g.npanels)) * 8 + (b & 7);
  } else {
    *panel = b % g.npanels;
    *wg_in_panel = b / g.npanels;
  }
}

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// Two packed adds whose inputs come out of transcendental instructions.  gfx940+ needs one wait state between a
// transcendental and a VALU instruction that reads its result, and the compiler's hazard recognizer does not look inside
// inline asm: the s_nop covers the first add whatever the compiler placed last in front of it, the first add covers the
// second.  (Early-clobber outputs: the first result must not land on the second add's input.)
__device__ __forceinline__ void pk_add2_after_trans(f32x2& d0, f32x2& d1, f32x2 a0, f32x2 a1, f32x2 b) {
  asm("s_nop 0\n\tv_pk_add_f32 %0, %2, %4\n\tv_pk_add_f32 %1, %3, %4" : "=&v"(d0), "=&v"(d1) : "v"(a0), "v"(a1), "v"(b));
}

constexpr int SK_ROW = 264;  // strip row stride in floats: the two half-waves land on disjoint bank halves

// KQ = K / 8 (k-slots come in groups of 8: four for each half-wave); ACT, NT compile-time: the epilogue is
// straight-line code
template <int KQ, int ACT, int NT>
__global__ __launch_bounds__(512) void gemm_skinnyk_kernel(SkinnyArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int K = KQ * 8, GROUPS = KQ * 2;
  float* Bs = smem;                        // [256][K], 16-byte group q of column n at slot q ^ (n & (GROUPS-1))
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* strip = smem + 256 * K + wave * (8 * SK_ROW);
  const int l31 = lane & 31, half = lane >> 5;
  const int panel = blockIdx.x % g.npanels, wg_in_panel = blockIdx.x / g.npanels;
  const int wgs_per_panel = gridDim.x / g.npanels;
  const int n0 = panel * 256;
  // ---- prologue: this panel of B -> LDS, transposed ---------------------------------------------------------------
  // one unit = four consecutive k of one column: four loads (lanes walk n, coalesced when B is n-contiguous), one
  // conflict-free 16-byte LDS write.  ALL of a thread's loads are issued before the first write (a loop that waits
  // for each load in turn spends K/2 memory latencies here).
  {
    f32x4 v[KQ];
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
      const int u = tid + 512 * i, n = u & 255, kq = u >> 8;
      const float* bp = g.B + (long)(4 * kq) * g.b_sk + (long)(n0 + n) * g.b_sn;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[i][c] = bp[(long)c * g.b_sk];
    }
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
      const int u = tid + 512 * i, n = u & 255, kq = u >> 8;
      *reinterpret_cast<f32x4*>(Bs + n * K + ((kq ^ (n & (GROUPS - 1))) * 4)) = v[i];
    }
  }
  __syncthreads();
  // ---- the wave's stream of 32-row blocks --------------------------------------------------------------------------
  const int stride = wgs_per_panel * 8;
  int rb = wg_in_panel * 8 + wave;
  if (wg_in_panel >= wgs_per_panel || rb >= g.nrb) return;
  // the two waves of a SIMD (w and w + 4) start together and do identical work: put them in antiphase
  if (wave >= 4 && g.stagger) __builtin_amdgcn_s_sleep(127);
  auto a_ptr = [&](int b) { return g.A + ((long)b * 32 + l31) * g.a_sm + 4 * half; };
  f32x4 a_cur[KQ], a_nxt[KQ];
  {
    const float* ap = a_ptr(rb);
#pragma unroll
    for (int q = 0; q < KQ; ++q) a_cur[q] = *reinterpret_cast<const f32x4*>(ap + 8 * q);
  }
  float bj[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bj[j] = 0.f;
  if (g.bias) {
#pragma unroll
    for (int j = 0; j < 8; ++j) bj[j] = g.bias[n0 + j * 32 + l31];
  }
  // Everything loaded so far has landed before the loop starts: otherwise the compiler's wait-count pass merges the
  // first iteration (a_cur in flight) into the loop header and every iteration waits, before its first MFMA, for the
  // previous block's stores to drain and for the loads it has just issued.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  while (true) {
    const int nxt = rb + stride;
    const bool more = nxt < g.nrb;
    if (more) {
      const float* ap = a_ptr(nxt);
#pragma unroll
      for (int q = 0; q < KQ; ++q) a_nxt[q] = *reinterpret_cast<const f32x4*>(ap + 8 * q);
    }
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // B fragments one pair of column tiles ahead of the MFMAs that consume them (the LDS round trip of a pair
    // hides behind the eight MFMAs of the pair before it)
    auto b_frag = [&](int q, int j) {
      const int n = j * 32 + l31;
      const int grp = (2 * q + half) ^ (n & (GROUPS - 1));
      return *reinterpret_cast<const f32x4*>(Bs + n * K + grp * 4);
    };
    // four column tiles per group: four independent accumulator chains keep the matrix pipe issuing back to back
    // (two chains leave it a few passes idle between dependent MFMAs)
    f32x4 bc[4], bn[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bc[t] = b_frag(0, t);
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
#pragma unroll
      for (int jg = 0; jg < 2; ++jg) {
        const int nq = (jg == 1) ? q + 1 : q, nj = (jg == 1) ? 0 : 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) bn[t] = (nq < KQ) ? b_frag(nq, nj + t) : bc[t];
        __builtin_amdgcn_sched_barrier(0);  // (the scheduler would sink the reads to just before their use)
#pragma unroll
        for (int ss = 0; ss < 4; ++ss)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc[4 * jg + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q][ss], bc[t][ss], acc[4 * jg + t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) bc[t] = bn[t];
      }
    }
    // register r of lane (l31, half) is row (r&3) + 8*(r>>2) + 4*half, column j*32 + l31 of the block.
    // Pass p moves rows 8p .. 8p+7 through the strip and out as eight whole 1 KiB rows.
    float* crow = g.C + ((long)rb * 32) * g.c_sm + n0 + 4 * lane;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          float v = g.alpha * acc[j][4 * p + rr] + bj[j];
          if (ACT == 1) v = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
          strip[(rr + 4 * half) * SK_ROW + j * 32 + l31] = v;
        }
#pragma unroll
      for (int row = 0; row < 8; ++row) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(strip + row * SK_ROW + 4 * lane);
        f32x4* dst = reinterpret_cast<f32x4*>(crow + (long)(8 * p + row) * g.c_sm);
        if (NT) __builtin_nontemporal_store(v, dst);
        else *dst = v;
      }
    }
    if (!more) break;
    rb = nxt;
#pragma unroll
    for (int q = 0; q < KQ; ++q) a_cur[q] = a_nxt[q];
  }
}

// ---- version 3 (the default): one wave per SIMD, block i leaves inside the MFMAs of block i+1 -------------------------
// What the cycle counter says about version 1 (s_memtime stamps per wave and phase, one workgroup):
//   * a wave's MFMA phase runs at full rate when it has the SIMD's matrix pipe to itself (16.5 k cycles per 256 MFMAs),
//     but while it runs, the OTHER wave's epilogue hardly advances (20-28 k cycles for a chain that takes ~4 k alone):
//     the two waves of a SIMD do not overlap, they take turns -- MFMA, MFMA, then both epilogues with the pipe idle.
//     s_setprio either way and s_nop gaps between the MFMAs change nothing;
//   * the same inside ONE wave (an earlier form of this kernel, 64 drain pieces interleaved with the MFMAs): a group of
//     16 MFMAs takes 1024 cycles alone, ~1390 with 16 v_fma + 16 ds_write_b32 among them, ~1190 with 8 global stores,
//     1024-1044 with 8 ds_read_b128.  An fp32 MFMA keeps the register file's read ports busy: whatever READS vector
//     registers (VALU, LDS writes, stores) adds its own cycles to the block; LDS reads (one address register) are free.
// So the cheapest way out is the one that reads the fewest registers, and nothing may ever wait:
//   * four waves per workgroup, 512 registers each, TWO accumulator sets (both in AccVGPRs): while the 256 MFMAs of block
//     i+1 run into one, block i leaves the other in 76 small pieces, one or two after every four MFMAs;
//   * the MFMA operands are swapped (weights as the A operand, rows as the B operand): the tile comes out transposed, a
//     lane holds row l31 and FOUR CONSECUTIVE COLUMNS per register quad, so the LDS transposition works in 16-byte
//     units: 32 ds_write_b128 per block, straight from the AccVGPRs, instead of 128 ds_write_b32;
//   * a plain gmul (alpha = 1, no bias, no map) has no VALU work at all; with an epilogue the bias quad comes from LDS
//     one piece ahead and logistic is v_fma, v_exp_f32, v_add, v_rcp_f32 (alpha and bias pre-multiplied by -log2 e);
//     `1/x` as IEEE division is ten VALU instructions, which is what made the fused map cost 0.03 ms;
//   * the block leaves in column passes of CP columns: 32 rows x CP columns through the wave-private strip (rows
//     padded by 16 bytes: conflict-free b128 writes, one address register each way), then out as row pieces of CP*4
//     bytes, 1 KiB per store instruction, each read back from the strip three pieces before its store.
// Per block 17.2 k cycles against 16.4 k of MFMAs.  Config 5a (rocprofv3 kernel time): 146 us against 165 us for
// version 1; with the fused logistic 169 us (version 1 with the cheap reciprocal: 165 us, with the division 206 us).
template <int KQ, int ACT, int NT, int CP, bool PLAIN, bool COMPUTE, bool DRAIN>
__device__ __forceinline__ void skinny3_step(f32x16 (&ac)[8], const f32x16 (&ad)[8], const f32x4 (&a)[KQ],
                                             const float* __restrict__ Bs, float* strip, const f32x4 (&bq)[256 / CP],
                                             float* cbase, long c_sm, float alpha, int lane) {
  constexpr int K = KQ * 8, GROUPS = KQ * 2;
  constexpr int NPASS = 256 / CP, TPP = CP / 32;       // column passes per block, column tiles per pass
  constexpr int WR = TPP * 4, RD = CP / 8;             // b128 writes / reads (= stores) per lane per pass
  constexpr int SLOTS = CP / 4;                        // 16-byte slots per strip row
  constexpr int SROW = CP + 4;                         // strip row stride (floats): b128 writes of eight consecutive rows
                                                       // conflict-free, one address register + immediates on both sides
  constexpr int RPI = 256 / CP;                        // rows per read/store instruction (64 lanes x 16 B = 1 KiB)
  constexpr int RDD = 3;                               // row pieces read this many pieces before their store
  constexpr int PPP = WR + RD + RDD;                   // pieces per pass: writes, then reads with the stores RDD behind
  constexpr int PIECES = NPASS * PPP;                  // K = 64, CP = 64: 4 * 19 = 76
  constexpr int NSLOT = GROUPS * 4;                    // MFMA slots (four MFMAs each) per block
  const int l31 = lane & 31, half = lane >> 5;
  const int rrow = lane / SLOTS, rslot = lane % SLOTS;  // this lane's row (within an instruction) and slot when reading
  f32x4 rv[4];
  float* cp = cbase;
  f32x2 one2 = {1.0f, 1.0f}, alpha2 = {alpha, alpha};
  if (!PLAIN) asm volatile("" : "+v"(one2), "+v"(alpha2));   // (register pairs, not two literal moves per use)
  auto piece = [&](int pc) {
    const int pass = pc / PPP, w = pc % PPP;
    if (w < WR) {                                        // one register quad -> the strip, straight from the AccVGPRs
      const int j = pass * TPP + (w >> 2), q = w & 3;
      f32x4 v;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = ad[j][4 * q + c];
      const int slot = (w >> 2) * 8 + 2 * q + half;
      *reinterpret_cast<f32x4*>(strip + l31 * SROW + 4 * slot) = v;
    } else {
      const int tt = w - WR;
      if (tt < RD) {                                     // RPI rows back as row pieces
        const int row = tt * RPI + rrow;
        rv[tt % (RDD + 1)] = *reinterpret_cast<const f32x4*>(strip + row * SROW + 4 * rslot);
      }
      if (tt >= RDD) {
        const int i = tt - RDD;
        if (i == 0) cp = cbase + pass * CP;
        f32x4* dst = reinterpret_cast<f32x4*>(cp);
        if (!PLAIN) {
          // The map runs on the READ side of the strip (round 4): the row piece is in VGPRs already (no v_accvgpr_read),
          // its four columns are this lane's for the whole pass (bias quad bq[pass] lives in registers for the whole
          // kernel), so alpha * v + bias (both pre-multiplied by -log2 e for the logistic) is a PACKED fma, two elements
          // per instruction: v_pk_fma, v_exp_f32, v_pk_add (1 +), v_rcp_f32 -- 12 VALU instructions per four elements where
          // the write-side form had 20 (4 v_accvgpr_read, 4 v_fma, 4 v_exp, 4 v_add, 4 v_rcp).  An fp32 MFMA runs at the
          // VALU's own rate and does not overlap with it (tools/probes/valu_rates.hip): every instruction saved is a
          // slot given back to the MFMA stream.
          // (the packed instructions are written out: left to itself the compiler splits a third of them into two plain ones)
          f32x4& r = rv[i % (RDD + 1)];
          f32x2 lo = pk_fma(f32x2{r[0], r[1]}, alpha2, f32x2{bq[pass][0], bq[pass][1]});
          f32x2 hi = pk_fma(f32x2{r[2], r[3]}, alpha2, f32x2{bq[pass][2], bq[pass][3]});
          if (ACT == 1) {   // logistic: v_exp_f32 and v_rcp_f32 (1 ulp each)
            const f32x2 elo = {__builtin_amdgcn_exp2f(lo[0]), __builtin_amdgcn_exp2f(lo[1])};
            const f32x2 ehi = {__builtin_amdgcn_exp2f(hi[0]), __builtin_amdgcn_exp2f(hi[1])};
            pk_add2_after_trans(lo, hi, elo, ehi, one2);
            lo[0] = __builtin_amdgcn_rcpf(lo[0]); lo[1] = __builtin_amdgcn_rcpf(lo[1]);
            hi[0] = __builtin_amdgcn_rcpf(hi[0]); hi[1] = __builtin_amdgcn_rcpf(hi[1]);
          }
          r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
        }
        if (NT) __builtin_nontemporal_store(rv[i % (RDD + 1)], dst);
        else *dst = rv[i % (RDD + 1)];
        cp += RPI * c_sm;
      }
    }
  };
  if (!COMPUTE) {
#pragma unroll
    for (int pc = 0; pc < PIECES; ++pc) piece(pc);
    return;
  }
  auto b_frag = [&](int q, int j) {
    const int n = j * 32 + l31;
    const int grp = (2 * q + half) ^ (n & (GROUPS - 1));
    return *reinterpret_cast<const f32x4*>(Bs + n * K + grp * 4);
  };
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) ac[j][r] = 0.f;
  f32x4 bc[4], bn[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) bc[t] = b_frag(0, t);
#pragma unroll
  for (int gi = 0; gi < GROUPS; ++gi) {
    const int q = gi >> 1, jg = gi & 1;
    const int nq = (jg == 1) ? q + 1 : q, nj = (jg == 1) ? 0 : 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) bn[t] = (nq < KQ) ? b_frag(nq, nj + t) : bc[t];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
#pragma unroll
      for (int t = 0; t < 4; ++t)   // weights first: the tile comes out transposed (lane = row, registers = columns)
        ac[4 * jg + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bc[t][ss], a[q][ss], ac[4 * jg + t], 0, 0, 0);
      if (DRAIN) {
        const int sl = gi * 4 + ss;
#pragma unroll
        for (int pc = sl * PIECES / NSLOT; pc < (sl + 1) * PIECES / NSLOT; ++pc) piece(pc);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) bc[t] = bn[t];
  }
}

template <int KQ, int ACT, int NT, int CP, bool PLAIN>
__global__ __launch_bounds__(256) void gemm_skinnyk3_kernel(SkinnyArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int K = KQ * 8, GROUPS = KQ * 2, NW = 4, NTH = NW * 64, UPT = GROUPS * 256 / NTH;
  float* Bs = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* strip = smem + 256 * K + wave * (32 * (CP + 4));
  float* bias_s = smem + 256 * K + NW * 32 * (CP + 4);
  const int l31 = lane & 31, half = lane >> 5;
  int panel, wg_in_panel, wgs_per_panel;
  wg_map(g, blockIdx.x, gridDim.x, &panel, &wg_in_panel, &wgs_per_panel);
  const int n0 = panel * 256;
  SK_STAMP(0);
  // the first two row blocks' A rows are asked for BEFORE the weights are staged: their trip to HBM (~2 us) runs under
  // the staging instead of behind it
  const int stride = wgs_per_panel * NW;
  int rb = __builtin_amdgcn_readfirstlane(wg_in_panel * NW + wave);   // wave-uniform: the loop runs on the scalar unit
  const bool live = wg_in_panel < wgs_per_panel && rb < g.nrb;
  auto load_a = [&](f32x4 (&a)[KQ], int b) {
    const float* base = g.A;
    if (g.a_tab) {   // (uniform: b is a scalar, the table entry comes through the scalar cache)
      const int mi = b / g.bpm;
      base = g.a_tab[mi];
      b -= mi * g.bpm;
    }
    const float* ap = base + ((long)b * 32 + l31) * g.a_sm + 4 * half;
#pragma unroll
    for (int q = 0; q < KQ; ++q) a[q] = *reinterpret_cast<const f32x4*>(ap + 8 * q);
  };
  f32x4 aX[KQ], aY[KQ];
  int nxt = rb + stride;
  bool more = nxt < g.nrb;
  if (live) {
    load_a(aX, rb);
    if (more) load_a(aY, nxt);
  }
  {
    f32x4 v[UPT];
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
      const int u = tid + NTH * i, n = u & 255, kq = u >> 8;
      const float* bp = g.B + (long)(4 * kq) * g.b_sk + (long)(n0 + n) * g.b_sn;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[i][c] = bp[(long)c * g.b_sk];
    }
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
      const int u = tid + NTH * i, n = u & 255, kq = u >> 8;
      *reinterpret_cast<f32x4*>(Bs + n * K + ((kq ^ (n & (GROUPS - 1))) * 4)) = v[i];
    }
    if (!PLAIN) bias_s[tid] = (g.bias ? g.bias[n0 + tid] : 0.f) * (ACT == 1 ? -1.44269504088896340736f : 1.0f);
  }
  __syncthreads();
  SK_STAMP(1);
  if (!live) return;
  constexpr int SLOTS = CP / 4;
  auto c_base = [&](int b) { return g.C + ((long)b * 32 + lane / SLOTS) * g.c_sm + n0 + 4 * (lane % SLOTS); };
  f32x16 accA[8], accB[8];
  // the bias of this lane's four columns of every column pass (a lane reads the same slot of every strip row)
  const float alpha_e = g.alpha * (ACT == 1 ? -1.44269504088896340736f : 1.0f);
  f32x4 bq[256 / CP];
#pragma unroll
  for (int ps = 0; ps < 256 / CP; ++ps)
    bq[ps] = PLAIN ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(bias_s + ps * CP + 4 * (lane % SLOTS));
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the loop's wait counts start from a known state (see version 1)
  SK_STAMP(2);
#ifdef TOPS_AB_KNOBS
  const unsigned long long dbg_c2 = __builtin_readcyclecounter();
#endif
  skinny3_step<KQ, ACT, NT, CP, PLAIN, true, false>(accA, accB, aX, Bs, strip, bq, nullptr, g.c_sm, alpha_e, lane);
  SK_STAMP(3);
#ifdef TOPS_AB_KNOBS
  if (g.dbg && lane == 0) g.dbg[((long)blockIdx.x * 4 + wave) * 8 + 7] = __builtin_readcyclecounter() - dbg_c2;   // shader cycles of the MFMA-only block
#endif
  int prev = rb;
  while (true) {
    if (!more) {
      SK_STAMP(4);
      skinny3_step<KQ, ACT, NT, CP, PLAIN, false, true>(accB, accA, aX, Bs, strip, bq, c_base(prev), g.c_sm, alpha_e, lane);
      break;
    }
    rb = nxt; nxt = rb + stride; more = nxt < g.nrb;
    if (more) load_a(aX, nxt);
    skinny3_step<KQ, ACT, NT, CP, PLAIN, true, true>(accB, accA, aY, Bs, strip, bq, c_base(prev), g.c_sm, alpha_e, lane);
    prev = rb;
    if (!more) {
      SK_STAMP(4);
      skinny3_step<KQ, ACT, NT, CP, PLAIN, false, true>(accA, accB, aX, Bs, strip, bq, c_base(prev), g.c_sm, alpha_e, lane);
      break;
    }
    rb = nxt; nxt = rb + stride; more = nxt < g.nrb;
    if (more) load_a(aY, nxt);
    skinny3_step<KQ, ACT, NT, CP, PLAIN, true, true>(accA, accB, aX, Bs, strip, bq, c_base(prev), g.c_sm, alpha_e, lane);
    prev = rb;
  }
  SK_STAMP(5);
#ifdef TOPS_AB_KNOBS
  if (g.dbg) { __builtin_amdgcn_s_waitcnt(0); SK_STAMP(6); }   // ... and with the last stores acknowledged
#endif
}

bool gemm_skinnyk_applicable(const GemmProblem& p) {
  static const int enable = [] { const char* e = ab_getenv("TOPS_GEMM_SKINNYK"); return e ? atoi(e) : 1; }();
  if (!enable || p.dtype != TO_F32 || p.batch != 1 || p.reduce_batch) return false;
  if (p.K != 64 && p.K != 32 && p.K != 16) return false;
  if (p.N % 256 != 0 || p.N < 256 || p.N > 256 * 64) return false;
  if (p.M % 32 != 0 || p.M * p.N < (1LL << 24) || p.M / 32 < 2048) return false;  // a long stream of rows
  if (p.a_sk != 1 || p.a_sm % 4 != 0 || (reinterpret_cast<uintptr_t>(p.A) & 15u)) return false;
  if (p.c_sm < p.N || p.c_sm % 4 != 0 || (reinterpret_cast<uintptr_t>(p.C) & 15u)) return false;
  if (p.beta != 0.0 || p.dact || p.rowsum || p.loss_rows || p.act > 1) return false;
  if (p.M / 32 > 2147483647LL) return false;
  return true;
}

void launch_gemm_skinnyk(const GemmProblem& p, hipStream_t s) {
  static const int version = [] { const char* e = ab_getenv("TOPS_SKINNYK_V"); return e ? atoi(e) : 3; }();
  SkinnyArgs g{};
  g.A = (const float*)p.A; g.B = (const float*)p.B; g.C = (float*)p.C;
  g.bias = (const float*)p.bias;
  g.M = p.M; g.N = (int)p.N; g.K = (int)p.K;
  g.a_sm = p.a_sm; g.b_sk = p.b_sk; g.b_sn = p.b_sn; g.c_sm = p.c_sm;
  g.alpha = (float)p.alpha;
  g.npanels = (int)(p.N / 256);
  g.nrb = (int)(p.M / 32);
  if (p.a_table) {
    TO_CHECK(p.a_table_rows > 0 && p.a_table_rows % 32 == 0 && p.M % p.a_table_rows == 0, TO_ERR_STATE, "internal: bad A table");
    TO_CHECK(version == 3, TO_ERR_UNSUPPORTED, "the first short-K design (TOPS_SKINNYK_V=1) has no A table");
    g.a_tab = static_cast<const float* const*>(p.a_table);
    g.bpm = (int)(p.a_table_rows / 32);
  }
  static const int stagger = [] { const char* e = ab_getenv("TOPS_SKINNYK_STAGGER"); return e ? atoi(e) : 1; }();
  g.stagger = stagger;
  static const int pairs = [] { const char* e = ab_getenv("TOPS_SKINNYK_XCD_PAIRS"); return e ? atoi(e) : 1; }();
  g.xcd_pairs = pairs;
#ifdef TOPS_AB_KNOBS
  static const int dbg = [] { const char* e = ab_getenv("TOPS_SKINNYK_DBG"); return e ? atoi(e) : 0; }();
  static unsigned long long* dbg_buf = nullptr;
  if (dbg && !dbg_buf) TO_HIP(hipMalloc(&dbg_buf, 256 * 4 * 8 * sizeof(unsigned long long)));
  if (dbg) { TO_HIP(hipMemset(dbg_buf, 0, 256 * 4 * 8 * sizeof(unsigned long long))); g.dbg = dbg_buf; }
#endif
  bool nt = p.M * p.N * 4 > (256LL << 20);
  static const int nt_env = [] { const char* e = ab_getenv("TOPS_SKINNYK_NT"); return e ? atoi(e) : -1; }();
  if (nt_env >= 0) nt = nt_env != 0;
  const int cp = 64;  // columns per drain pass (128 would need four 16 KiB strips next to 64 KiB of weights at K = 64)
  const int nwaves = version == 3 ? 4 : 8;
  const size_t lds = version == 3 ? ((size_t)256 * p.K + 4 * 32 * (cp + 4) + 256) * 4
                                  : ((size_t)256 * p.K + nwaves * 8 * SK_ROW) * 4;
  const int grid = 256 / g.npanels * g.npanels;  // whole panels' worth of workgroups, one per CU
  static bool attr_set[64] = {false};
  auto launch = [&](auto kern, int which) {
    if (!attr_set[which]) {
      TO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set[which] = true;
    }
    launch_k(kern, dim3(grid), dim3(nwaves * 64), lds, s, g);
  };
  const int v = (p.act ? 2 : 0) + (nt ? 1 : 0);
  const bool plain = !p.act && !p.bias && p.alpha == 1.0;
#define TOPS_SKINNY3(KQ, CP, base)                                                              \
  if (plain) { if (nt) launch(gemm_skinnyk3_kernel<KQ, 0, 1, CP, true>, base + 0);              \
               else launch(gemm_skinnyk3_kernel<KQ, 0, 0, CP, true>, base + 1); }               \
  else switch (v) {                                                                             \
    case 0: launch(gemm_skinnyk3_kernel<KQ, 0, 0, CP, false>, base + 2); break;                 \
    case 1: launch(gemm_skinnyk3_kernel<KQ, 0, 1, CP, false>, base + 3); break;                 \
    case 2: launch(gemm_skinnyk3_kernel<KQ, 1, 0, CP, false>, base + 4); break;                 \
    default: launch(gemm_skinnyk3_kernel<KQ, 1, 1, CP, false>, base + 5); break;                \
  }
#define TOPS_SKINNY(KQ, base)                                                  \
  if (version == 3) {                                                          \
    TOPS_SKINNY3(KQ, 64, 24 + 3 * base)                                       \
  } else switch (v) {                                                          \
    case 0: launch(gemm_skinnyk_kernel<KQ, 0, 0>, base + 0); break;            \
    case 1: launch(gemm_skinnyk_kernel<KQ, 0, 1>, base + 1); break;            \
    case 2: launch(gemm_skinnyk_kernel<KQ, 1, 0>, base + 2); break;            \
    default: launch(gemm_skinnyk_kernel<KQ, 1, 1>, base + 3); break;           \
  }
  switch (p.K) {
    case 64: TOPS_SKINNY(8, 0) break;
    case 32: TOPS_SKINNY(4, 4) break;
    default: TOPS_SKINNY(2, 8) break;
  }
#undef TOPS_SKINNY
#undef TOPS_SKINNY3
  TO_HIP(hipGetLastError());
  count_launch();
#ifdef TOPS_AB_KNOBS
  if (dbg) {   // per stamp: min / median / max over the waves, microseconds since the first wave began
    static int printed = 0;
    if (printed++ == dbg) {   // (the dbg-th launch: warm)
      TO_HIP(hipStreamSynchronize(s));
      std::vector<unsigned long long> h(256 * 4 * 8);
      TO_HIP(hipMemcpy(h.data(), dbg_buf, h.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull;
      for (int w = 0; w < 1024; ++w) if (h[w * 8] && h[w * 8] < t0) t0 = h[w * 8];
      static const char* names[7] = {"entry", "B staged + barrier", "first A rows landed", "first block computed",
                                     "last drain begins", "last store issued", "last store acknowledged"};
      for (int i = 0; i < 7; ++i) {
        std::vector<double> v;
        for (int w = 0; w < 1024; ++w) if (h[w * 8 + i]) v.push_back((double)(h[w * 8 + i] - t0) * 0.01);
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        if (i == 3) {
          std::vector<double> c;
          for (int w = 0; w < 1024; ++w) if (h[w * 8 + 7]) c.push_back((double)h[w * 8 + 7]);
          std::sort(c.begin(), c.end());
          if (!c.empty()) fprintf(stderr, "skinnyk dbg first block (256 MFMAs, no drain): %.0f / %.0f / %.0f shader cycles (min / med / max)\n", c.front(), c[c.size() / 2], c.back());
        }
        fprintf(stderr, "skinnyk dbg %-24s min %7.2f  p10 %7.2f  med %7.2f  p90 %7.2f  max %7.2f us (%zu waves)\n", names[i], v.front(),
                v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back(), v.size());
      }
    }
  }
#endif
}

}  // namespace to
