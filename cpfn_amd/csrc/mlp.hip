// Per-point MLP stacks (1x1 conv + BatchNorm(batch statistics) + ReLU [+ max-pool]) for gfx950.
//
// Data are points-major bf16 rows [P, C] (P = every point / neighbour of the batch), so a
// 1x1 convolution is Y[P,N] = A[P,K]·W[N,K]ᵀ and both MFMA operands are K-contiguous 16-byte
// fragments.  bf16 operands, fp32 accumulation (v_mfma_f32_16x16x32_bf16); BatchNorm
// statistics come out of the fp32 accumulators in the GEMM epilogue, so a layer is
//     GEMM(+Σy, Σy²) -> [C]-sized finalize -> fused normalise+ReLU(+max-pool)
// instead of the reference's conv / batch_norm / relu / max chain over fp32 NCHW tensors
// (modules/pointset_abstraction.py:70-74, modules/pointset_feature_propagation.py:49-51).
//
// MFMA orientation: the WEIGHT tile is the A operand (rows = output channels) and the POINT
// tile the B operand (columns = points), so every lane ends up with 4 consecutive output
// channels of one point: channels are the contiguous axis of a row, stores are 8-byte
// pieces that tile a row, and per-channel statistics are per-register sums.
//
// All reductions (statistics, weight gradients) go through per-workgroup partial buffers that
// a second tiny kernel sums in a fixed order: bitwise reproducible, no float atomics.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
  return __builtin_bit_cast(unsigned short, (__bf16)f);  // v_cvt_pk_bf16_f32: RNE, NaN-preserving
}

// a' = relu(scale·a + shift) on eight consecutive channels of one point, rounded to bf16 exactly like
// bn_relu_apply_kernel: lets a GEMM / weight-gradient kernel consume the PREVIOUS layer's pre-BN output directly
// (the activated tensor is then never written or read).
__device__ __forceinline__ bf16x8 bn_relu_frag(bf16x8 v, const float *sc, const float *sh) {
  typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
  const u16x8 u = __builtin_bit_cast(u16x8, v);
  u16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f2bf(fmaxf(fmaf(sc[j], bf2f(u[j]), sh[j]), 0.f));
  return __builtin_bit_cast(bf16x8, o);
}

// ---- in-kernel timing probe of the GEMM family (bench.py's roofline leg) -----------------------------------------
// A replayed hipGraph cannot be bracketed per kernel with host-side events, so when a probe is installed
// (cpfn_mlp_gemm_set_probe) every launch of the family times ITSELF with the device's constant-rate wall clock
// (s_memrealtime, 100 MHz): every workgroup stores its own (start, end) pair into the launch's slot of the probe
// buffer — plain 16-byte stores, no atomics (a first version folded the times into one counter with two device-scope
// atomics per workgroup: ~18 us per launch of pure atomic latency, the step went from 2.4 to 3.0 ms) — and the host
// takes max(end) - min(start) per slot afterwards.  A launch captured into a graph keeps the slot it was given at
// capture time, so after a run of replays the buffer holds the LAST replay's launches.  No probe (the default): the
// argument is NULL and the cost is one uniform branch.
struct GemmProbeSlot {
  unsigned long long nwg, pad;                 // header: workgroups of the launch that wrote this slot
  unsigned long long t[1][2];                  // [workgroup][start, end] (max_wg entries)
};
struct GemmProbeState {
  unsigned long long *buf = nullptr;           // slots x (2 + 2 max_wg) u64
  int slots = 0, max_wg = 0;
  unsigned next = 0;                           // launches handed a slot so far
};
GemmProbeState g_probe_state;
// slot of the launch being issued (NULL: probe off or grid too large for a slot)
static inline unsigned long long *probe_slot(dim3 grid) {
  GemmProbeState &p = g_probe_state;
  if (!p.buf || (long long)grid.x * grid.y * grid.z > p.max_wg) return nullptr;
  return p.buf + (size_t)(p.next++ % (unsigned)p.slots) * (2 + 2 * (size_t)p.max_wg);
}
__device__ __forceinline__ unsigned long long probe_begin(const unsigned long long *slot) {
  return slot ? (unsigned long long)wall_clock64() : 0ull;
}
__device__ __forceinline__ void probe_end(unsigned long long *slot, unsigned long long t0, unsigned kind = 1) {
  if (!slot) return;
  __syncthreads();                // every wave of the workgroup has issued its last store
  if (threadIdx.x == 0) {
    const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    typedef __attribute__((ext_vector_type(2))) unsigned long long u64x2;
    *(u64x2 *)&slot[2 + 2 * (size_t)wg] = (u64x2){t0, (unsigned long long)wall_clock64()};
    if (wg == 0) {
      slot[0] = (unsigned long long)gridDim.x * gridDim.y * gridDim.z;
      slot[1] = kind;            // 1 streaming GEMM, 2 generic GEMM, 3 small-P GEMM, 4 weight gradient, 5 one-pass backward, 6 small layer: weight + data gradient
    }
  }
}
// (the weight-gradient and one-pass backward kernels take probe slots too — kinds 4 and 5 — so that bench.py can name the
//  family with the most time in the step and tools/dbg/probe_timeline.py has ~50 anchors inside a replayed step)
static inline unsigned long long *probe_slot_all(dim3 grid) { return probe_slot(grid); }

constexpr int G_THREADS = 256;
constexpr int G_ROWS = 128;   // points per workgroup tile (4 waves x 32)
constexpr int G_KC = 128;     // K chunk staged in LDS
constexpr int G_LDW = G_KC + 8;

// Y[P,N] (bf16 or fp32) = A[P,K] (bf16, optional row gather) · W[N,K]ᵀ (bf16) (+ bias)
// stats_partial[gridDim.x][2][N]: per-workgroup Σy, Σy² over its valid rows (fp32 accumulators).
//
// Two kernels share the MFMA tile shape (128 points x BN channels per workgroup, 32 points per wave):
//  * mlp_gemm_stream_kernel<BN,KS,STATS>: K = 32·KS in {64,128}, bf16 output, N % BN == 0.  These are
//    the layers that carry the bytes (P >= 131072 rows).  HBM-bound streaming structure: W panel
//    loaded into LDS once per workgroup; A fragments of tile t+1 are requested BEFORE the MFMAs of
//    tile t (two register buffers, straight-line loop body so the compiler emits counted
//    s_waitcnt vmcnt(N) instead of vmcnt(0)); the output tile leaves through a wave-private LDS patch
//    as 16-byte row-contiguous stores (4 rows x 256 B per wave instruction).
//  * mlp_gemm_kernel<BN,STATS>: any K % 32 == 0 (chunked through LDS), optional row gather, bias,
//    fp32 / ragged-N output: the small-P layers (sa3, sfp1, sfp2), K > 128, and the fc2 heads.
constexpr int G_LDO = 128 + 8;  // output staging row stride (elements): 272 B

// eight 16-bit elements rotated left by r positions (result[i] = v[(i + r) & 7]) without register indexing
__device__ __forceinline__ uint4 rot_u16x8(uint4 v, int r) {
  if (r & 4) v = (uint4){v.z, v.w, v.x, v.y};
  if (r & 2) v = (uint4){v.y, v.z, v.w, v.x};
  if (r & 1) v = (uint4){(v.x >> 16) | (v.y << 16), (v.y >> 16) | (v.z << 16), (v.z >> 16) | (v.w << 16), (v.w >> 16) | (v.x << 16)};
  return v;
}

// Fill the LDS weight panel s_w[r][k] (r = output channel n0+r, k in [kc, kc+kcn)).
// w_trans == 0: W is [N,K] row-major (16-byte loads along k).
// w_trans == 1: W is [K,N] row-major (the FORWARD layer's weight used for the data gradient): 16-byte
//               loads along n, transposed on the way into LDS (8-byte pieces, see below), so no transposed
//               weight copy has to be materialised per step.
template <int BN, int LDW = G_LDW, int NT = G_THREADS>
__device__ __forceinline__ void fill_w_panel(unsigned short *s_w, const unsigned short *__restrict__ W, int K,
                                             int N, int n0, int kc, int kcn, int w_trans, int t) {
  if (!w_trans) {
    const int cpr = kcn / 8;  // 16-byte chunks per row
    for (int e = t; e < BN * cpr; e += NT) {
      const int r = e / cpr, c = e - r * cpr;
      *(uint4 *)&s_w[r * LDW + c * 8] = *(const uint4 *)&W[(size_t)(n0 + r) * K + kc + c * 8];
    }
  } else {
    // W is [K, N] (a forward weight used for the data gradient): transpose on the way into LDS.  A lane takes
    // FOUR consecutive k rows of one 8-column chunk and writes 8-byte pieces (4 k values of one column); the
    // column order is rotated by the chunk index, otherwise the 16 lanes of a k-row hit one LDS bank (a panel row
    // is 272 B, so 8 rows apart = 2176 B = 17 x 128 B).  The first version wrote single bf16s with that 16-way
    // conflict: the transposed launches ran 25 us against 17.5 us for the plain ones.
    constexpr int cpn = BN / 8;  // 16-byte chunks along n
    for (int e = t; e < (kcn / 4) * cpn; e += NT) {
      const int k4 = e / cpn, c = e - k4 * cpn;
      uint4 v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = *(const uint4 *)&W[(size_t)(kc + 4 * k4 + r) * N + n0 + c * 8];
      // rotate the eight columns of every row vector by (c & 7) positions with whole-register selects — indexing
      // the registers with a lane-dependent j would push them to scratch memory
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = rot_u16x8(v[r], c & 7);
      const unsigned short *h0 = (const unsigned short *)&v[0], *h1 = (const unsigned short *)&v[1],
                           *h2 = (const unsigned short *)&v[2], *h3 = (const unsigned short *)&v[3];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int j = (jj + c) & 7;      // rotated position jj holds column j
        uint2 o;
        o.x = (unsigned)h0[jj] | ((unsigned)h1[jj] << 16);
        o.y = (unsigned)h2[jj] | ((unsigned)h3[jj] << 16);
        *(uint2 *)&s_w[(c * 8 + j) * LDW + 4 * k4] = o;
      }
    }
  }
}

// Buffer addressing for the streaming kernel: SGPR base, ONE lane offset per operand row computed once per kernel,
// the tile position added as one 32-bit value (in the lane offset: the scalar offset of a buffer instruction is not
// bounds-checked).  Rows past P are out of range for the hardware bounds check (loads return zeros, stores are
// dropped), so there is no per-tile 64-bit address arithmetic, clamping or exec masking: ~140 of
// the ~750 vector instructions per tile of the first version were address arithmetic (a wave64 VALU instruction costs
// 4 cycles; with BatchNorm statistics and operand transform these kernels are VALU-limited, not MFMA-limited).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
template <int KS>
__device__ __forceinline__ void stream_load_a(bf16x8 (&af)[2][KS], __amdgpu_buffer_rsrc_t rs_a, const unsigned (&aoff)[2],
                                              unsigned tile_off /* bytes, wave-uniform */) {
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      af[tt][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_a, aoff[tt] + tile_off + ks * 64, 0, 0));
}

// Sum over the 16 lanes of a DPP row (lanes that share lane>>4): every lane ends up with the total.
// quad_perm xor-1, quad_perm xor-2, row_half_mirror, row_mirror — four VALU adds, no LDS traffic.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}

struct StreamBufs {
  __amdgpu_buffer_rsrc_t a, y, yb;       // operand rows, output rows, (BST) pre-BN output of the layer below
  unsigned aoff[2];                      // lane byte offset of its two operand rows inside a tile
  unsigned yoff;                         // lane byte offset of its first output piece inside a tile
  unsigned a_tile, y_tile, y_step;       // bytes per 128-row tile of A / Y, bytes between a lane's output pieces
};

// XT ("xyz tail", sa2's first layer: 128 gathered feature channels + the 3 centred coordinates of the neighbour): the
// coordinates do not travel as three bf16 columns of a K = 192 operand (50 MB instead of 33.5 per launch, > 256
// registers and 90 KB of LDS: one workgroup per CU) but as an fp32 [P,3] tensor; every lane builds ONE more k-step from the
// two points it owns — x = hi + lo in bf16, columns [x_hi y_hi z_hi x_lo y_lo z_lo x_hi y_hi | z_hi 0 ...] against the
// weight columns [w_hi w_hi w_lo] — so the three products are accurate to ~2^-16 and the layer is the K = 128 kernel plus
// NT x 2 MFMAs.  xt_frag: the lane's 8 k-values of that step (k 0-7 in the lanes lq = 0, k 8-15 in lq = 1, zeros above).
__device__ __forceinline__ bf16x8 xt_frag(float x, float y, float z, int lq) {
  const unsigned xh = f2bf(x), yh = f2bf(y), zh = f2bf(z);
  const unsigned xl = f2bf(x - bf2f((unsigned short)xh)), yl = f2bf(y - bf2f((unsigned short)yh)), zl = f2bf(z - bf2f((unsigned short)zh));
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4w;
  u32x4w v = {0u, 0u, 0u, 0u};
  if (lq == 0) v = (u32x4w){xh | (yh << 16), zh | (xl << 16), yl | (zl << 16), xh | (yh << 16)};
  if (lq == 1) v = (u32x4w){zh, 0u, 0u, 0u};
  return __builtin_bit_cast(bf16x8, v);
}

template <int BN, int KS, bool STATS, bool ATR, bool BST, bool XT = false>
__device__ __forceinline__ void stream_tile(bf16x8 (&af)[2][KS], const unsigned short *s_w,
                                            unsigned short *s_o, const StreamBufs &sb, int P, int row0, int next_tile,
                                            int wave, int lane, float (&st_s)[8], float (&st_q)[8],
                                            const float *s_ss /*[2][32*KS]: scale, shift*/,
                                            const float *s_bs /*[2][BN]: scale, shift of the layer below (BST)*/,
                                            const unsigned short *s_wx /*XT: [BN][16] bf16*/,
                                            float (&xz)[2][3] /*XT: xyz of the lane's two points, reloaded for the next tile*/,
                                            const float *xyz) {
  constexpr int NT = BN / 16;
  constexpr int CPR = BN / 8;  // 16-byte chunks per row
  // the 64-wide variants have the registers to request the pieces of Yb before the MFMAs (the two sa1 data gradients,
  // 524288 rows: 47 / 36 us with the loads issued at the epilogue, where their latency is exposed); the 128-wide ones
  // request them at the start of the epilogue
  constexpr bool YB_EARLY = BST && BN == 64;
  u32x4_t yb[BST ? 32 * CPR / 64 : 1];   // (a plain vector type: arrays of HIP's uint4 struct end up in scratch memory)
  const int lr = lane & 15, lq = lane >> 4;
  const unsigned y_tile_off = (unsigned)(row0 / G_ROWS) * sb.y_tile;
  if (YB_EARLY) {
#pragma unroll
    for (int i = 0; i < 32 * CPR / 64; ++i) yb[i] = __builtin_amdgcn_raw_buffer_load_b128(sb.yb, sb.yoff + y_tile_off + i * sb.y_step, 0, 0);
  }
  f32x4 acc[NT][2];
#pragma unroll
  for (int i = 0; i < NT; ++i) { acc[i][0] = (f32x4){0, 0, 0, 0}; acc[i][1] = (f32x4){0, 0, 0, 0}; }
  if (XT) {
    const bf16x8 x0 = xt_frag(xz[0][0], xz[0][1], xz[0][2], lq), x1 = xt_frag(xz[1][0], xz[1][1], xz[1][2], lq);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      bf16x8 wf = *(const bf16x8 *)&s_wx[(nt * 16 + lr) * 16 + 8 * (lq & 1)];
      if (lq >= 2) wf = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      acc[nt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, x0, acc[nt][0], 0, 0, 0);
      acc[nt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, x1, acc[nt][1], 0, 0, 0);
    }
    // the next tile's coordinates (clamped rows: a tile past the end is never used)
    const int pn = (next_tile * G_ROWS) + wave * 32 + lr, pa = min(pn, P - 1), pb = min(pn + 16, P - 1);
#pragma unroll
    for (int q = 0; q < 3; ++q) { xz[0][q] = xyz[(size_t)pa * 3 + q]; xz[1][q] = xyz[(size_t)pb * 3 + q]; }
    // (the barrier keeps the K loop's weight-fragment reads from being hoisted up here — with them the kernel spills; placed
    //  behind the K loop instead, the coordinate k-step measured 32.7 us against 29.0 here; the plain K = 128 kernel: 19.3)
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ATR) {   // BatchNorm + ReLU of the previous layer applied to the operand in place, one k-step at a time
      float sc[8], sh[8];
      *(cpfn_f32x4 *)&sc[0] = *(const cpfn_f32x4 *)&s_ss[ks * 32 + 8 * lq];
      *(cpfn_f32x4 *)&sc[4] = *(const cpfn_f32x4 *)&s_ss[ks * 32 + 8 * lq + 4];
      *(cpfn_f32x4 *)&sh[0] = *(const cpfn_f32x4 *)&s_ss[32 * KS + ks * 32 + 8 * lq];
      *(cpfn_f32x4 *)&sh[4] = *(const cpfn_f32x4 *)&s_ss[32 * KS + ks * 32 + 8 * lq + 4];
      af[0][ks] = bn_relu_frag(af[0][ks], sc, sh);
      af[1][ks] = bn_relu_frag(af[1][ks], sc, sh);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const bf16x8 wf = *(const bf16x8 *)&s_w[(nt * 16 + lr) * (32 * KS + 8) + ks * 32 + 8 * lq];
      acc[nt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[0][ks], acc[nt][0], 0, 0, 0);
      acc[nt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[1][ks], acc[nt][1], 0, 0, 0);
    }
    // keep the k-steps apart: left alone the scheduler hoists all 32 weight-fragment reads (and the operand
    // transform of every k-step) in front of the first MFMA, which costs >100 registers
    if (ATR) __builtin_amdgcn_sched_barrier(0);
  }
  // ONE operand buffer: the next tile's rows are requested as soon as the MFMAs have consumed this one's, and fly
  // during the epilogue below (statistics, LDS staging, stores).  The first version kept two buffers (tile t+1
  // requested before the MFMAs of tile t): 268 registers for the plain 128-wide variant and 454-490 with the
  // statistics — one wave per SIMD, so a CU held ONE workgroup and its load / compute / store phases overlapped
  // with nothing.  Occupancy, not a deeper per-wave pipeline, is what hides the latency here.
  stream_load_a<KS>(af, sb.a, sb.aoff, (unsigned)next_tile * sb.a_tile);
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f32x4 v = acc[nt][tt];
      bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
      *(bf16x4 *)&s_o[(tt * 16 + lr) * G_LDO + nt * 16 + 4 * lq] = o;
    }
  }
  // The tile leaves through the wave's LDS patch as 16-byte row-contiguous pieces; a lane always carries the SAME
  // 8-channel chunk (c = lane % CPR), so the BatchNorm statistics are 16 per-lane running sums over the pieces it
  // stores (Σy, Σy² of the bf16 values BatchNorm will actually normalise), reduced across lanes once after the
  // tile loop.  The first version summed the fp32 accumulators: 64 running sums per lane (all channels of the
  // lane's MFMA rows) — with them the 128-wide variants needed 380-490 registers and ran one workgroup per CU.
  // BST (data-gradient launches): the output IS the gradient g_a of the layer below, so BatchNorm-backward pass 1 of
  // that layer — Σ g_z and Σ g_z·y with g_z = g_a·[scale·y + shift > 0] — is taken here from the pieces being
  // stored and the matching pieces of that layer's pre-BN output Yb (same rows, same 8-channel chunk, loaded just
  // ahead of the store loop).  The separate bn_relu_bwd pass (which re-read g_a and y) is then not launched.
  if (BST && !YB_EARLY) {
#pragma unroll
    for (int i = 0; i < 32 * CPR / 64; ++i) yb[i] = __builtin_amdgcn_raw_buffer_load_b128(sb.yb, sb.yoff + y_tile_off + i * sb.y_step, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 32 * CPR / 64; ++i) {
    const int e = i * 64 + lane;
    const int r = e / CPR, c = e - r * CPR;
    const int p = row0 + wave * 32 + r;
    const uint4 vv = *(const uint4 *)&s_o[r * G_LDO + c * 8];
    __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){vv.x, vv.y, vv.z, vv.w}, sb.y, sb.yoff + y_tile_off + i * sb.y_step, 0, 0);
    if (p < P) {      // (the store needs no guard: rows past P are out of the buffer's range; the sums do)
      if (BST) {
        // (scale / shift of the lane's chunk come from LDS for every piece: as 16 more live registers they pushed
        //  the 128-wide K = 128 variant over 256 and back to one workgroup per CU)
        float bsc[8], bsh[8];
        if (BN == 64 && KS <= 4) {   // (these variants have the registers: read once per tile, the compiler hoists it)
          *(cpfn_f32x4 *)&bsc[0] = *(const cpfn_f32x4 *)&s_bs[c * 8];
          *(cpfn_f32x4 *)&bsc[4] = *(const cpfn_f32x4 *)&s_bs[c * 8 + 4];
          *(cpfn_f32x4 *)&bsh[0] = *(const cpfn_f32x4 *)&s_bs[BN + c * 8];
          *(cpfn_f32x4 *)&bsh[4] = *(const cpfn_f32x4 *)&s_bs[BN + c * 8 + 4];
        } else {
          *(cpfn_f32x4 *)&bsc[0] = cpfn_lds_read4(&s_bs[c * 8]);
          *(cpfn_f32x4 *)&bsc[4] = cpfn_lds_read4(&s_bs[c * 8 + 4]);
          *(cpfn_f32x4 *)&bsh[0] = cpfn_lds_read4(&s_bs[BN + c * 8]);
          *(cpfn_f32x4 *)&bsh[4] = cpfn_lds_read4(&s_bs[BN + c * 8 + 4]);
        }
        const unsigned g4[4] = {vv.x, vv.y, vv.z, vv.w};
        const unsigned y4[4] = {yb[i][0], yb[i][1], yb[i][2], yb[i][3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float g0 = __uint_as_float(g4[j] << 16), g1 = __uint_as_float(g4[j] & 0xffff0000u);
          const float y0 = __uint_as_float(y4[j] << 16), y1 = __uint_as_float(y4[j] & 0xffff0000u);
          const float z0 = fmaf(bsc[2 * j], y0, bsh[2 * j]) > 0.f ? g0 : 0.f;
          const float z1 = fmaf(bsc[2 * j + 1], y1, bsh[2 * j + 1]) > 0.f ? g1 : 0.f;
          st_s[2 * j] += z0;
          st_s[2 * j + 1] += z1;
          st_q[2 * j] = fmaf(z0, y0, st_q[2 * j]);
          st_q[2 * j + 1] = fmaf(z1, y1, st_q[2 * j + 1]);
        }
      }
      if (STATS) {
        const unsigned w4[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = __uint_as_float(w4[j] << 16), hi = __uint_as_float(w4[j] & 0xffff0000u);
          st_s[2 * j] += lo;
          st_s[2 * j + 1] += hi;
          st_q[2 * j] = fmaf(lo, lo, st_q[2 * j]);
          st_q[2 * j + 1] = fmaf(hi, hi, st_q[2 * j + 1]);
        }
      }
    }
  }
}

template <int BN, int KS, bool STATS, bool ATR = false, bool BST = false, bool XT = false>
__global__ __launch_bounds__(G_THREADS) __attribute__((amdgpu_waves_per_eu(2))) void mlp_gemm_stream_kernel(
    const unsigned short *__restrict__ A, int lda, const unsigned short *__restrict__ W, int w_trans, int P, int N,
    unsigned short *__restrict__ Y, int ldy, float *__restrict__ stats_partial, int tiles_per_wg,
    const float *__restrict__ a_scale = nullptr, const float *__restrict__ a_shift = nullptr,
    const unsigned short *__restrict__ Yb = nullptr /* BST: [P, ldy] like Y */, unsigned long long *probe = nullptr,
    const float *__restrict__ xyz = nullptr /* XT: [P,3] */, const float *__restrict__ wx = nullptr /* XT: [N,3] fp32 */) {
  constexpr int NT = BN / 16, K = 32 * KS, CPR = BN / 8;
  const unsigned long long probe_t0 = probe_begin(probe);
  __shared__ __attribute__((aligned(16))) unsigned short s_w[BN * (32 * KS + 8)];   // whole-K panel, rows padded by 16 B
  __shared__ __attribute__((aligned(16))) unsigned short s_o[4][32 * G_LDO];
  __shared__ __attribute__((aligned(16))) float s_red[4][2][BN];
  __shared__ __attribute__((aligned(16))) float s_ss[ATR ? 2 * K : 4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int n0 = blockIdx.y * BN;
  if (ATR) {   // visible after the W-panel barrier
    for (int e = t; e < K; e += G_THREADS) { s_ss[e] = a_scale[e]; s_ss[K + e] = a_shift[e]; }
  }
  __shared__ __attribute__((aligned(16))) float s_bs[BST ? 2 * BN : 4];
  __shared__ __attribute__((aligned(16))) unsigned short s_wx[XT ? BN * 16 : 8];
  float xz[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  if (XT) {   // weight columns of the extra k-step: [w_hi(3) w_hi(3) w_lo(3) 0 ...] per channel (visible after the W-panel barrier)
    for (int e = t; e < BN; e += G_THREADS) {
      const float w0 = wx[(size_t)(n0 + e) * 3], w1 = wx[(size_t)(n0 + e) * 3 + 1], w2 = wx[(size_t)(n0 + e) * 3 + 2];
      const unsigned short h0 = f2bf(w0), h1 = f2bf(w1), h2 = f2bf(w2);
      const unsigned short l0 = f2bf(w0 - bf2f(h0)), l1 = f2bf(w1 - bf2f(h1)), l2 = f2bf(w2 - bf2f(h2));
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4w;
      *(u32x4w *)&s_wx[e * 16] = (u32x4w){(unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h0 << 16),
                                          (unsigned)h1 | ((unsigned)h2 << 16), (unsigned)l0 | ((unsigned)l1 << 16)};
      *(u32x4w *)&s_wx[e * 16 + 8] = (u32x4w){(unsigned)l2, 0u, 0u, 0u};
    }
  }
  float st_s[8], st_q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { st_s[i] = 0.f; st_q[i] = 0.f; }
  if (BST) {   // scale / shift of the layer below for this column block (visible after the W-panel barrier)
    for (int e = t; e < BN; e += G_THREADS) { s_bs[e] = a_scale[n0 + e]; s_bs[BN + e] = a_shift[n0 + e]; }
  }
  // buffer descriptors: the host guarantees P * max(lda, ldy) * 2 < 2^32
  StreamBufs sb;
  const unsigned y_bytes = ((unsigned)(P - 1) * ldy + N) * 2u;
  sb.a = __builtin_amdgcn_make_buffer_rsrc((void *)A, 0, ((unsigned)(P - 1) * lda + K) * 2u, 0x00020000);
  sb.y = __builtin_amdgcn_make_buffer_rsrc((void *)Y, 0, y_bytes, 0x00020000);
  sb.yb = __builtin_amdgcn_make_buffer_rsrc(BST ? (void *)Yb : (void *)Y, 0, y_bytes, 0x00020000);
  sb.aoff[0] = ((unsigned)(wave * 32 + lr) * lda + 8 * lq) * 2u;
  sb.aoff[1] = sb.aoff[0] + 16u * lda * 2u;
  sb.yoff = ((unsigned)(wave * 32 + lane / CPR) * ldy + n0 + (lane % CPR) * 8) * 2u;
  sb.a_tile = (unsigned)G_ROWS * lda * 2u;
  sb.y_tile = (unsigned)G_ROWS * ldy * 2u;
  sb.y_step = (unsigned)(64 / CPR) * ldy * 2u;
  const int ntiles = (P + G_ROWS - 1) / G_ROWS;
  const int tile0 = blockIdx.x * tiles_per_wg;
  const int tile_end = min(tile0 + tiles_per_wg, ntiles);
  if (tile0 < tile_end) {
    bf16x8 a[2][KS];
    stream_load_a<KS>(a, sb.a, sb.aoff, (unsigned)tile0 * sb.a_tile);
    if (XT) {
      const int p0 = tile0 * G_ROWS + wave * 32 + lr, pa = min(p0, P - 1), pb = min(p0 + 16, P - 1);
#pragma unroll
      for (int q = 0; q < 3; ++q) { xz[0][q] = xyz[(size_t)pa * 3 + q]; xz[1][q] = xyz[(size_t)pb * 3 + q]; }
    }
    fill_w_panel<BN, 32 * KS + 8>(s_w, W, K, N, n0, 0, K, w_trans, t);
    __syncthreads();
    for (int tile = tile0; tile < tile_end; ++tile) {
      // the reload inside is unconditional (a tile past the end is out of the buffer's range: zeros, no traffic), so
      // the loop body is straight-line
      stream_tile<BN, KS, STATS, ATR, BST, XT>(a, s_w, s_o[wave], sb, P, tile * G_ROWS, tile + 1, wave, lane, st_s, st_q, s_ss, s_bs,
                                               s_wx, xz, xyz);
    }
  }
  if (STATS || BST) {
    // once per workgroup: lanes that carry the same 8-channel chunk (lane % CPR) are summed by xor-shuffles, the
    // first CPR lanes of every wave hand their 8 channels over through LDS
    constexpr int CPR = BN / 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int m = CPR; m < 64; m <<= 1) { st_s[j] += __shfl_xor(st_s[j], m, 64); st_q[j] += __shfl_xor(st_q[j], m, 64); }
    }
    if (lane < CPR) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { s_red[wave][0][lane * 8 + j] = st_s[j]; s_red[wave][1][lane * 8 + j] = st_q[j]; }
    }
    __syncthreads();
    for (int e = t; e < 2 * BN; e += G_THREADS) {
      const int which = e / BN, c = e - which * BN;
      const float s = s_red[0][which][c] + s_red[1][which][c] + s_red[2][which][c] + s_red[3][which][c];
      stats_partial[((size_t)blockIdx.x * 2 + which) * N + n0 + c] = s;
    }
  }
  probe_end(probe, probe_t0);
}

// ---- generic kernel: K chunks of 128 through a DOUBLE-BUFFERED LDS weight panel.
// The small-P layers (sa3, sfp1, sfp2: 2048-8192 rows, K up to 1280) are pure latency: 64-128 workgroups, each
// walking its K chunks one after the other.  The first version did load -> wait -> LDS store -> barrier -> MFMA
// per chunk (~3 us per chunk: 43 us for 2048 x 1280 -> 256).  Now chunk c+1's weight pieces and A fragments are
// requested into registers BEFORE the MFMAs of chunk c and stored to the other panel buffer after them, so a chunk
// costs one barrier and the global latency hides behind the MFMAs of the previous chunk; the (tile, chunk) sequence
// is flattened, so the first chunk of the next row tile is also in flight during the epilogue of the current one.
constexpr int G_SS_MAX = 512;   // operand-transform scale/shift staged in LDS up to this K

template <int BN>
__device__ __forceinline__ void w_chunk_load(uint4 (&v)[BN / 16], const unsigned short *__restrict__ W, int K, int N,
                                             int n0, int kc, int kcn, int w_trans, int t) {
  constexpr int NV = BN / 16;
  if (!w_trans) {
    // the chunk is addressed as 16 16-byte columns per row whatever kcn is: no runtime division, idle lanes load 0
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = t + i * G_THREADS, r = e >> 4, c = e & 15;
      v[i] = (c * 8 < kcn) ? *(const uint4 *)&W[(size_t)(n0 + r) * K + kc + c * 8] : (uint4){0, 0, 0, 0};
    }
  } else {
    constexpr int cpn = BN / 8;
#pragma unroll
    for (int j = 0; j < NV / 4; ++j) {
      const int e = t + j * G_THREADS, k4 = e / cpn, c = e - k4 * cpn;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        v[j * 4 + r] = (4 * k4 < kcn) ? *(const uint4 *)&W[(size_t)(kc + 4 * k4 + r) * N + n0 + c * 8] : (uint4){0, 0, 0, 0};
    }
  }
}

template <int BN>
__device__ __forceinline__ void w_chunk_store(unsigned short *s_w, uint4 (&v)[BN / 16], int w_trans, int t) {
  constexpr int NV = BN / 16;
  if (!w_trans) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = t + i * G_THREADS, r = e >> 4, c = e & 15;
      *(uint4 *)&s_w[r * G_LDW + c * 8] = v[i];
    }
  } else {
    // transposed on the way in: 8-byte pieces (4 k values of one column), column order rotated by the chunk index
    // (see fill_w_panel)
    constexpr int cpn = BN / 8;
#pragma unroll
    for (int j = 0; j < NV / 4; ++j) {
      const int e = t + j * G_THREADS, k4 = e / cpn, c = e - k4 * cpn;
      uint4 w4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) w4[r] = rot_u16x8(v[j * 4 + r], c & 7);
      const unsigned short *h0 = (const unsigned short *)&w4[0], *h1 = (const unsigned short *)&w4[1],
                           *h2 = (const unsigned short *)&w4[2], *h3 = (const unsigned short *)&w4[3];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int col = (jj + c) & 7;
        uint2 o;
        o.x = (unsigned)h0[jj] | ((unsigned)h1[jj] << 16);
        o.y = (unsigned)h2[jj] | ((unsigned)h3[jj] << 16);
        *(uint2 *)&s_w[(c * 8 + col) * G_LDW + 4 * k4] = o;
      }
    }
  }
}

__device__ __forceinline__ void a_chunk_load(bf16x8 (&af)[2][4], const unsigned short *__restrict__ A, int lda,
                                             const int *__restrict__ gidx, int P, int row0, int kc, int kcn, int wave,
                                             int lr, int lq) {
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    int p = row0 + wave * 32 + tt * 16 + lr;
    p = p < P ? p : P - 1;
    const unsigned short *src = A + (size_t)(gidx ? gidx[p] : p) * lda + kc + 8 * lq;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      if (ks * 32 < kcn) af[tt][ks] = *(const bf16x8 *)(src + ks * 32);
  }
}

template <int BN, bool STATS>
__global__ __launch_bounds__(G_THREADS) void mlp_gemm_kernel(
    const unsigned short *__restrict__ A, int lda, const int *__restrict__ gidx,
    const unsigned short *__restrict__ W, int w_trans, int P, int K, int N, void *__restrict__ Y, int ldy, int y_f32,
    int n_store, const float *__restrict__ bias, float *__restrict__ stats_partial, int tiles_per_wg,
    const float *__restrict__ a_scale, const float *__restrict__ a_shift, unsigned long long *probe) {
  constexpr int NT = BN / 16;
  const unsigned long long probe_t0 = probe_begin(probe);
  __shared__ __attribute__((aligned(16))) unsigned short s_w[2][BN * G_LDW];
  __shared__ float s_red[4][2][BN];
  __shared__ __attribute__((aligned(16))) float s_ss[2][G_SS_MAX];
  // DENSE fp32 output (the packed heads: ldy == n_store <= 64, one column block): the tile's rows are contiguous in memory, so
  // it leaves through LDS as 16-byte pieces of ONE contiguous block instead of 16-byte pieces at a 140-byte row stride
  __shared__ __attribute__((aligned(16))) float s_of[BN == 64 && !STATS ? G_ROWS * 64 : 4];
  const bool dense_out = BN == 64 && !STATS && y_f32 && n_store == ldy && n_store <= 64 && gridDim.y == 1 &&
                         ((unsigned long long)Y & 15ull) == 0;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int n0 = blockIdx.y * BN;

  f32x4 s1[NT], s2[NT];
  if (STATS) {
#pragma unroll
    for (int i = 0; i < NT; ++i) { s1[i] = (f32x4){0, 0, 0, 0}; s2[i] = (f32x4){0, 0, 0, 0}; }
  }
  const bool ss_lds = a_scale && K <= G_SS_MAX;
  if (ss_lds) {   // visible after the first panel barrier
    for (int e = t; e < K; e += G_THREADS) { s_ss[0][e] = a_scale[e]; s_ss[1][e] = a_shift[e]; }
  }
  const int nchunks = (K + G_KC - 1) / G_KC;
  const bool single = nchunks == 1;   // whole K in one panel: filled once per workgroup
  const int ntiles = (P + G_ROWS - 1) / G_ROWS;
  const int tile0 = blockIdx.x * tiles_per_wg;
  const int tile_end = min(tile0 + tiles_per_wg, ntiles);
  const int total = tile_end > tile0 ? (tile_end - tile0) * nchunks : 0;

  uint4 wv[BN / 16];
  bf16x8 an[2][4];
  if (total > 0) {
    const int kcn = min(G_KC, K);
    a_chunk_load(an, A, lda, gidx, P, tile0 * G_ROWS, 0, kcn, wave, lr, lq);
    w_chunk_load<BN>(wv, W, K, N, n0, 0, kcn, w_trans, t);
  }
  f32x4 acc[NT][2];
  int tile = tile0, c = 0;
  for (int q = 0; q < total; ++q) {
    const int kc = c * G_KC, kcn = min(G_KC, K - kc);   // multiple of 32
    const unsigned short *sw = s_w[single ? 0 : (q & 1)];
    if (!single || q == 0) {
      w_chunk_store<BN>(s_w[single ? 0 : (q & 1)], wv, w_trans, t);
      __syncthreads();   // the only barrier of a chunk: the other buffer was last read before the previous one
    }
    bf16x8 af[2][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { af[0][ks] = an[0][ks]; af[1][ks] = an[1][ks]; }
    int nc = c + 1, ntile = tile;
    if (nc == nchunks) { nc = 0; ++ntile; }
    if (q + 1 < total) {   // next chunk in flight during this chunk's MFMAs
      const int nkc = nc * G_KC, nkcn = min(G_KC, K - nkc);
      a_chunk_load(an, A, lda, gidx, P, ntile * G_ROWS, nkc, nkcn, wave, lr, lq);
      if (!single) w_chunk_load<BN>(wv, W, K, N, n0, nkc, nkcn, w_trans, t);
    }
    if (c == 0) {
#pragma unroll
      for (int i = 0; i < NT; ++i) { acc[i][0] = (f32x4){0, 0, 0, 0}; acc[i][1] = (f32x4){0, 0, 0, 0}; }
    }
    if (a_scale) {   // BatchNorm + ReLU of the previous layer applied to the operand on the fly
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        if (ks * 32 < kcn) {
          float sc[8], sh[8];
          const int k0 = kc + ks * 32 + 8 * lq;
          if (ss_lds) {
            *(cpfn_f32x4 *)&sc[0] = *(const cpfn_f32x4 *)&s_ss[0][k0]; *(cpfn_f32x4 *)&sc[4] = *(const cpfn_f32x4 *)&s_ss[0][k0 + 4];
            *(cpfn_f32x4 *)&sh[0] = *(const cpfn_f32x4 *)&s_ss[1][k0]; *(cpfn_f32x4 *)&sh[4] = *(const cpfn_f32x4 *)&s_ss[1][k0 + 4];
          } else {
            *(float4 *)&sc[0] = *(const float4 *)(a_scale + k0); *(float4 *)&sc[4] = *(const float4 *)(a_scale + k0 + 4);
            *(float4 *)&sh[0] = *(const float4 *)(a_shift + k0); *(float4 *)&sh[4] = *(const float4 *)(a_shift + k0 + 4);
          }
          af[0][ks] = bn_relu_frag(af[0][ks], sc, sh);
          af[1][ks] = bn_relu_frag(af[1][ks], sc, sh);
        }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks * 32 < kcn) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const bf16x8 wf = *(const bf16x8 *)&sw[(nt * 16 + lr) * G_LDW + ks * 32 + 8 * lq];
          acc[nt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[0][ks], acc[nt][0], 0, 0, 0);
          acc[nt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[1][ks], acc[nt][1], 0, 0, 0);
        }
      }
    }
    if (c == nchunks - 1) {
      // epilogue: lane holds channels n0 + nt*16 + 4*lq + r (r<4) of point row0 + wave*32 + tt*16 + lr
      const int row0 = tile * G_ROWS;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int p = row0 + wave * 32 + tt * 16 + lr;
        const bool valid = p < P;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          f32x4 v = acc[nt][tt];
          const int n = n0 + nt * 16 + 4 * lq;
          if (bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (n + r < N) ? bias[n + r] : 0.f;
          }
          if (STATS && valid) { s1[nt] += v; s2[nt] += v * v; }
          if (dense_out) {
            float *so = s_of + (wave * 32 + tt * 16 + lr) * n_store + n;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n + r < n_store) so[r] = v[r];
          } else if (valid) {
            if (y_f32) {
              float *o = (float *)Y + (size_t)p * ldy + n;
              if (n + 3 < n_store) {
                // one 16-byte store (dword-aligned: the row stride of the packed heads is 35 floats) instead of four
                // guarded dword stores — the fc2 launch issued 32 scalar stores per lane and tile
                typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
                *(f32x4_a4 *)o = (f32x4_a4){v[0], v[1], v[2], v[3]};
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  if (n + r < n_store) o[r] = v[r];
              }
            } else if (n + 3 < n_store) {
              bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
              *(bf16x4 *)((unsigned short *)Y + (size_t)p * ldy + n) = o;
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (n + r < n_store) ((unsigned short *)Y)[(size_t)p * ldy + n + r] = f2bf(v[r]);
            }
          }
        }
      }
      if (dense_out) {
        __syncthreads();
        const int nrows = min(G_ROWS, P - row0), nel = nrows * n_store;
        float *dst = (float *)Y + (size_t)row0 * n_store;            // (row0 * n_store * 4 B = a multiple of 16 B: 128-row tiles)
        for (int e = 4 * t; e + 3 < nel; e += 4 * G_THREADS) *(cpfn_f32x4 *)(dst + e) = *(const cpfn_f32x4 *)(s_of + e);
        if (t < (nel & 3)) dst[(nel & ~3) + t] = s_of[(nel & ~3) + t];
        __syncthreads();                                             // (the next tile's epilogue overwrites s_of)
      }
    }
    c = nc;
    tile = ntile;
  }
  if (STATS) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = s1[nt][r], b = s2[nt][r];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
        if (lr == 0) { s_red[wave][0][nt * 16 + 4 * lq + r] = a; s_red[wave][1][nt * 16 + 4 * lq + r] = b; }
      }
    }
    __syncthreads();
    for (int e = t; e < 2 * BN; e += G_THREADS) {
      const int which = e / BN, c2 = e - which * BN;
      const float s = s_red[0][which][c2] + s_red[1][which][c2] + s_red[2][which][c2] + s_red[3][which][c2];
      if (n0 + c2 < N) stats_partial[((size_t)blockIdx.x * 2 + which) * N + n0 + c2] = s;
    }
  }
  probe_end(probe, probe_t0, 2);
}

template <int LD>
__device__ __forceinline__ bf16x8 tr_frag(const unsigned short *tile, int col0, int lane) {
  // fragment F[x = lane&15][k = 8(lane>>4)+j] = tile[row k][col0 + x]  (tile rows = contraction index)
  const int grp = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const unsigned short *a0 = tile + (8 * grp + q) * LD + col0 + 4 * pp;
  const unsigned short *a1 = a0 + 4 * LD;
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)a1);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);  // one whole-vector cast: element-wise casts of the tr-read result miscompile
}

// ---- small-P kernel (P <= 16384 rows: sa3, sfp1, sfp2 and their data gradients).
// These layers move a few MB and are pure latency: with 128-row tiles they fill 16-128 workgroups and walk K
// (up to 1280) chunk by chunk, 2-3 us of exposed memory latency per chunk.  Here a workgroup owns RT (32|64) rows x
// 64 channels and its four waves SPLIT K (wave w takes the 32-wide k-steps w, w+4, ...): 4x-16x more workgroups,
// a 4x shorter serial chain per wave, no LDS panel and no barrier inside the K loop.  A and W fragments go straight
// from global memory (L2-resident after the first touch) into MFMA operand registers through a 2-slot register
// pipeline of bounds-checked buffer loads (counted vmcnt waits).  For the data gradient (W stored [K,N]) a wave
// bounces its 32 x 64 weight slice through a wave-private LDS tile and reads it back with ds_read_b64_tr_b16.
// The four K-partial accumulators are summed through LDS in a fixed order; wave w finishes channels 16w..16w+15:
// BatchNorm statistics by DPP row sums straight into the partial buffer, bf16 rows to Y.
// Measured (rocprofv3, tools/smallp_probe.py: the 14 small-P launches of one GlobalSPFN step, operands cold):
//   first version (128-row tiles, chunk-serial)   ~220 us     same with double-buffered panel   174 us
//   this kernel                                    112 us     (2048 x 1280 -> 256: 43 -> 9.8 us)
// What is left is launch + two memory round trips per workgroup: with stores, statistics and MFMAs removed the
// sum only drops to 98 us.
constexpr int SP_MAX_ROWS = 16384;
constexpr int SP_SS_MAX = 512;
constexpr int SP_DEPTH = 2;   // slots in flight per wave (4 measured slower on every shape: registers -> occupancy)
// 64-row tiles halve the weight re-reads; 32-row tiles when that would leave fewer than 256 workgroups
static inline int sp_rows(long long P, int N) { return ((P + 63) / 64) * (N / 64 > 0 ? N / 64 : 1) >= 256 ? 64 : 32; }

// Data gradient of a small layer with the reduction of the layer below riding on it (cpfn_mlp_dgrad_small):
//   BST:   pass 1 of the BatchNorm backward of the layer BELOW from the tile being stored (sum g_z, sum g_z y with the
//          ReLU mask from that layer's pre-BN output Yb): its stand-alone cpfn_bn_relu_bwd launch disappears.
// (Round 2 also formed g_y on the operand load here and in the 64 x 64 weight gradient: every 64-column block re-forms the
//  whole panel, the two kernels got 4-6 us slower each and cancelled the saved launch — removed in round 3.)
struct SmallpBwdArgs {
  const unsigned short *Yb;                 // BST: [P, ldy] like Y
  const float *b_scale, *b_shift;           // BST: [N]
};

template <int RT>
struct SmallpLds {
  static constexpr int TT = RT / 16, LDT = 64 + 8, RAW_TILE = 4 * 32 * LDT * 2, RAW_RED = 4 * 4 * TT * 64 * 16;
  __attribute__((aligned(16))) unsigned char raw[RAW_TILE > RAW_RED ? RAW_TILE : RAW_RED];
  __attribute__((aligned(16))) float ss[2][SP_SS_MAX];
};

// (body with the workgroup's position as arguments: mlp_gemm_smallp_kernel runs it on its own grid, mlp_bwd_small_kernel on
//  the tail of a grid whose head is a small layer's weight gradient)
template <int RT, bool STATS, bool WT, bool BST>
__device__ __forceinline__ void mlp_gemm_smallp_body(
    SmallpLds<RT> &lds, int bx, int by,
    const unsigned short *__restrict__ A, int lda, int a_bytes, const unsigned short *__restrict__ W,
    int P, int K, int N, unsigned short *__restrict__ Y, int ldy, float *__restrict__ stats_partial,
    const float *__restrict__ a_scale, const float *__restrict__ a_shift, const SmallpBwdArgs &bw) {
  constexpr int TT = RT / 16, D = SP_DEPTH, LDT = 64 + 8;
  static_assert(!(STATS && BST) && (!BST || WT), "the riding reduction belongs to the data gradient");
  unsigned char *s_raw = lds.raw;
  float (*s_ss)[SP_SS_MAX] = lds.ss;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int n0 = by * 64, row0 = bx * RT;
  if (a_scale) {
    for (int e = t; e < K; e += 256) { s_ss[0][e] = a_scale[e]; s_ss[1][e] = a_shift[e]; }
    __syncthreads();
  }
  const int S = K / 32;
  // Buffer loads (SGPR base + 32-bit lane offset, hardware bounds check): a pipeline slot past the end of K gets an
  // out-of-range offset, which returns zeros WITHOUT touching memory — the loop body stays straight-line (counted
  // vmcnt waits) and the tail slots cost nothing.  (Plain loads with clamped addresses re-read real data there:
  // with K = 256 that was 4x the traffic, and the kernel was slower than the one it replaces.)
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void *)A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, N * K * 2, 0x00020000);
  unsigned aoff[TT];   // byte offsets
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    int p = row0 + tt * 16 + lr;
    p = p < P ? p : P - 1;
    aoff[tt] = ((unsigned)p * lda + 8 * lq) * 2;
  }
  unsigned woff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = lane + 64 * i;   // WT: 16-byte piece e of the [32 k][64 n] slice: k row e>>3, columns 8(e&7)..
    woff[i] = WT ? ((unsigned)(e >> 3) * N + n0 + (e & 7) * 8) * 2 : ((unsigned)(n0 + i * 16 + lr) * K + 8 * lq) * 2;
  }
  const unsigned wstep = WT ? 32u * N * 2 : 64u;
  unsigned short *tile = (unsigned short *)s_raw + wave * 32 * LDT;   // this wave's [32 k][64 n] slice (WT only)
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // (a plain vector type: HIP's uint4 struct blocks SROA here)
  u32x4 ra[D][TT];
  u32x4 rw[D][4];
  auto issue = [&](int d, int s) __attribute__((always_inline)) {
    const unsigned oob = s < S ? 0u : 0x80000000u;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) ra[d][tt] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, (aoff[tt] + s * 64) | oob, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) rw[d][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (woff[i] + s * wstep) | oob, 0, 0);
  };
  f32x4 acc[4][TT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) acc[i][tt] = (f32x4){0, 0, 0, 0};
#pragma unroll
  for (int d = 0; d < D; ++d) issue(d, wave + 4 * d);
  const int cnt = (S + 3) / 4;
  // (one instantiation of the stage body per pipeline slot: the slot index must be a compile-time constant so that
  //  ra / rw stay in registers)
  auto stage = [&](auto slot, int i0) __attribute__((always_inline)) {
      constexpr int d = decltype(slot)::value;
      const int s = wave + 4 * (i0 + d);
      if (s < S) {
        bf16x8 af[TT], wf[4];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) af[tt] = __builtin_bit_cast(bf16x8, ra[d][tt]);
        if (WT) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int e = lane + 64 * i;
            *(u32x4 *)&tile[(e >> 3) * LDT + (e & 7) * 8] = rw[d][i];
          }
          // (wave-private tile: LDS executes one wave's instructions in order, and the compiler keeps the
          //  may-alias write -> transposing read -> next write order: no barrier of any kind is needed)
#pragma unroll
          for (int i = 0; i < 4; ++i) wf[i] = tr_frag<LDT>(tile, i * 16, lane);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) wf[i] = __builtin_bit_cast(bf16x8, rw[d][i]);
        }
        if (a_scale) {   // BatchNorm + ReLU of the previous layer applied to the operand on the fly
          float sc[8], sh[8];
          const int k0 = s * 32 + 8 * lq;
          *(cpfn_f32x4 *)&sc[0] = *(const cpfn_f32x4 *)&s_ss[0][k0]; *(cpfn_f32x4 *)&sc[4] = *(const cpfn_f32x4 *)&s_ss[0][k0 + 4];
          *(cpfn_f32x4 *)&sh[0] = *(const cpfn_f32x4 *)&s_ss[1][k0]; *(cpfn_f32x4 *)&sh[4] = *(const cpfn_f32x4 *)&s_ss[1][k0 + 4];
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) af[tt] = bn_relu_frag(af[tt], sc, sh);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int tt = 0; tt < TT; ++tt)
            acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[tt], acc[i][tt], 0, 0, 0);
      }
      // the slot is refilled AFTER its operands were consumed (no register copies: occupancy, not per-wave depth,
      // is what hides the latency here — measured: depth 2 beats depth 4 on every shape)
      issue(d, wave + 4 * (i0 + d + D));
  };
  static_assert(D == 2, "the pipeline slots are spelled out below");
  for (int i0 = 0; i0 < cnt; i0 += D) {
    stage(std::integral_constant<int, 0>{}, i0);
    stage(std::integral_constant<int, 1>{}, i0);
  }
  // K-partials of the four waves -> LDS in register layout [src wave][nt][tt][lane] (conflict-free 16-byte stores);
  // wave w then owns channel block nt = w and adds the four partials in wave order (fixed: reproducible)
  __syncthreads();   // the tiles are dead (s_raw is reused)
  f32x4 *s_part = (f32x4 *)s_raw;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) s_part[((wave * 4 + i) * TT + tt) * 64 + lane] = acc[i][tt];
  __syncthreads();
  const int n = n0 + wave * 16 + 4 * lq;
  f32x4 sm = {0, 0, 0, 0}, sq = {0, 0, 0, 0};
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    f32x4 v = s_part[((0 * 4 + wave) * TT + tt) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) v += s_part[((w * 4 + wave) * TT + tt) * 64 + lane];
    const int p = row0 + tt * 16 + lr;
    if (p < P) {
      if (STATS) { sm += v; sq += v * v; }
      bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
      *(bf16x4 *)(Y + (size_t)p * ldy + n) = o;
      if (BST) {     // (on the ROUNDED gradient, as the stand-alone pass would read it back)
        const bf16x4 yb = *(const bf16x4 *)(bw.Yb + (size_t)p * ldy + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float yv = (float)yb[r];
          const float z = fmaf(bw.b_scale[n + r], yv, bw.b_shift[n + r]) > 0.f ? (float)o[r] : 0.f;
          sm[r] += z;
          sq[r] = fmaf(z, yv, sq[r]);
        }
      }
    }
  }
  if (STATS || BST) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { sm[r] = row16_sum(sm[r]); sq[r] = row16_sum(sq[r]); }
    if (lr == 0) {
      *(f32x4 *)&stats_partial[((size_t)bx * 2 + 0) * N + n] = sm;
      *(f32x4 *)&stats_partial[((size_t)bx * 2 + 1) * N + n] = sq;
    }
  }
}

template <int RT, bool STATS, bool WT, bool BST = false>
__global__ __launch_bounds__(256) void mlp_gemm_smallp_kernel(
    const unsigned short *__restrict__ A, int lda, int a_bytes, const unsigned short *__restrict__ W,
    int P, int K, int N, unsigned short *__restrict__ Y, int ldy, float *__restrict__ stats_partial,
    const float *__restrict__ a_scale, const float *__restrict__ a_shift, unsigned long long *probe,
    const SmallpBwdArgs bw = SmallpBwdArgs()) {
  const unsigned long long probe_t0 = probe_begin(probe);
  __shared__ SmallpLds<RT> lds;
  mlp_gemm_smallp_body<RT, STATS, WT, BST>(lds, blockIdx.x, blockIdx.y, A, lda, a_bytes, W, P, K, N, Y, ldy, stats_partial, a_scale,
                                            a_shift, bw);
  probe_end(probe, probe_t0, 3);
}

// ---------------------------------------------------------------- BatchNorm finalize (forward)
// scale = γ·rstd, shift = β − mean·scale; running statistics updated like torch (unbiased var,
// the conv bias — dropped from the GEMM because batch-norm cancels it — re-enters the mean).
// Fixed-order two-level sum of per-block partials: 16 channels x RSUB block-subsets per workgroup
// (subset r adds blocks r, r+RSUB, ...; the RSUB subset sums are then added in order) — parallel,
// coalesced, and still bitwise reproducible.
constexpr int RSUB = 64;
constexpr int RTPB = 16 * RSUB;
__device__ __forceinline__ void partial_sums_16x16(const float *__restrict__ partial, int nblk, int N, int c,
                                                   int r, double (*s_acc)[16][2], double &s1, double &s2) {
  double a1 = 0.0, a2 = 0.0;
  if (c < N) {
#pragma unroll 4
    for (int i = r; i < nblk; i += RSUB) {
      a1 += (double)partial[((size_t)i * 2 + 0) * N + c];
      a2 += (double)partial[((size_t)i * 2 + 1) * N + c];
    }
  }
  s_acc[r][threadIdx.x & 15][0] = a1;
  s_acc[r][threadIdx.x & 15][1] = a2;
  __syncthreads();
  s1 = 0.0; s2 = 0.0;
  if (r == 0) {
    for (int q = 0; q < RSUB; ++q) { s1 += s_acc[q][threadIdx.x & 15][0]; s2 += s_acc[q][threadIdx.x & 15][1]; }
  }
}
// (Round 3 tried the RSUB subset sums as a shuffle / LDS tree with all of a thread's loads issued up front: the same 6 us under
//  rocprofv3, but +1.7 us per launch INSIDE the replayed step — in-kernel probe, gaps around all 34 finalize launches — i.e.
//  ~55 us per step slower.  Reverted: what this launch costs is its latency chain, and the plain loop has the shorter one.)

__global__ __launch_bounds__(RTPB) void bn_finalize_kernel(const float *__restrict__ partial, int nblk, int N, float count,
                                   const float *__restrict__ gamma, const float *__restrict__ beta,
                                   const float *__restrict__ conv_bias, float eps, float momentum,
                                   float *__restrict__ running_mean, float *__restrict__ running_var,
                                   float *__restrict__ scale, float *__restrict__ shift,
                                   float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                   long long *__restrict__ counter_a, long long *__restrict__ counter_b) {
  __shared__ double s_acc[RSUB][16][2];
  // step counters advanced by this launch (the layer's num_batches_tracked; the dropout step counter of a stack whose
  // output dropout reads it in the NEXT launch): was one multi-tensor add per forward pass
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (counter_a) ++*counter_a;
    if (counter_b) ++*counter_b;
  }
  const int c = blockIdx.x * 16 + (threadIdx.x & 15), r = threadIdx.x >> 4;
  double s1, s2;
  partial_sums_16x16(partial, nblk, N, c, r, s_acc, s1, s2);
  if (r != 0 || c >= N) return;
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * rstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  mean_out[c] = (float)mean;
  rstd_out[c] = rstd;
  if (running_mean) {
    const float b = conv_bias ? conv_bias[c] : 0.f;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * ((float)mean + b);
    const double unbiased = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// Evaluation mode (running statistics): z = gamma (y + b - rm) / sqrt(rv + eps) + beta as scale / shift of the
// bias-free GEMM output, in the 4 x C layout bn_finalize leaves (scale, shift, "mean" = rm - b, rstd) — one launch
// instead of the eight framework kernels per layer the expression costs as tensor ops (17 layers per forward pass).
__global__ void bn_eval_affine_kernel(const float *__restrict__ gamma, const float *__restrict__ beta,
                                      const float *__restrict__ conv_bias, const float *__restrict__ rm,
                                      const float *__restrict__ rv, float eps, int C, float *__restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float rstd = 1.0f / sqrtf(rv[c] + eps);
  const float b = conv_bias ? conv_bias[c] : 0.f, sc = gamma[c] * rstd;
  out[c] = sc;
  out[C + c] = beta[c] + (b - rm[c]) * sc;
  out[2 * C + c] = rm[c] - b;
  out[3 * C + c] = rstd;
}

// ---------------------------------------------------------------- normalise + ReLU (+ max-pool)
// out[p,c] = relu(scale[c]·y[p,c] + shift[c]); 8 channels (16 B) per lane.
// ---- dropout fused into the BatchNorm apply passes (the reference applies F.dropout(p = 0.5) to the fc1
// features in every mode, PointNet2/pn2_network.py:63; as PyTorch ops that is a mask-producing kernel in the forward
// pass and a masked-scale kernel in the backward pass over [B*N, 128]).  Counter-based: element chunk e (8 consecutive
// channels of one point) keeps element j iff the j-th 16-bit field of splitmix64(seed, e) is >= p * 65536, so the
// backward passes recompute the mask from the 8-byte seed instead of reading a stored one.
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// keep[j] for the 8 elements of chunk e: scale factor (1/(1-p)) or 0
__device__ __forceinline__ void dropout_factors(unsigned long long seed, unsigned long long e, unsigned thresh16,
                                                float inv_keep, float (&f)[8]) {
  const unsigned long long h0 = splitmix64(seed ^ (2 * e)), h1 = splitmix64(seed ^ (2 * e + 1));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[j] = ((unsigned)(h0 >> (16 * j)) & 0xffffu) >= thresh16 ? inv_keep : 0.f;
    f[4 + j] = ((unsigned)(h1 >> (16 * j)) & 0xffffu) >= thresh16 ? inv_keep : 0.f;
  }
}
__host__ __device__ inline unsigned dropout_thresh16(float p) {
  const float t = p * 65536.f + 0.5f;
  return t <= 0.f ? 0u : (t >= 65536.f ? 65536u : (unsigned)t);
}

__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const unsigned short *__restrict__ Yr,
                                                            const float *__restrict__ scale,
                                                            const float *__restrict__ shift, long long total8,
                                                            int C, unsigned short *__restrict__ out,
                                                            const long long *__restrict__ drop_counter,
                                                            unsigned long long drop_base, unsigned thresh16,
                                                            float inv_keep, unsigned long long *__restrict__ drop_seed_out) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  unsigned long long seed = 0;
  if (drop_counter) {
    seed = splitmix64(drop_base + 0xD1B54A32D192ED03ull * (unsigned long long)*drop_counter);
    if (e == 0) *drop_seed_out = seed;            // the backward passes of THIS forward pass read it from here
  }
  if (e >= total8) return;
  const int c0 = (int)((e * 8) % C);
  const uint4 raw = *(const uint4 *)(Yr + e * 8);
  const unsigned short *y = (const unsigned short *)&raw;
  float f[8];
  if (drop_counter) dropout_factors(seed, (unsigned long long)e, thresh16, inv_keep, f);
  unsigned short o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = fmaxf(fmaf(scale[c0 + j], bf2f(y[j]), shift[c0 + j]), 0.f);
    if (drop_counter) v *= f[j];
    o[j] = f2bf(v);
  }
  *(uint4 *)(out + e * 8) = *(const uint4 *)o;
}

// One workgroup per (group g of Kn consecutive rows, tile of 256 channels): out[g,c] = relu(max_k z),
// arg[g,c] = first k attaining it, yarg[g,c] = raw y at that k (needed by the backward pass).
// Lane layout: chunk ch = t % nch (8 channels, 16-byte loads), row sub-lane rs = t / nch.  The row sub-lanes
// of one wave are combined with shuffles, the four waves through 8 KB of LDS (the first version staged every
// sub-lane through 67 KB of LDS: two workgroups per CU, 2 TB/s).
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const unsigned short *__restrict__ Yr,
                                                              const float *__restrict__ scale,
                                                              const float *__restrict__ shift, int Kn, int C,
                                                              unsigned short *__restrict__ out,
                                                              unsigned char *__restrict__ arg,
                                                              unsigned short *__restrict__ yarg) {
  __shared__ float s_z[4][8 * 33];
  __shared__ int s_k[4][8 * 33];
  const int g = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int chunks = C / 8;                 // C in {64,128,256,...}: a power of two >= 8 chunks
  const int cb = blockIdx.y * 32;
  const int nch = min(32, chunks - cb);     // 8, 16 or 32
  const int rsub = 256 / nch;               // row sub-lanes
  const int ch = t % nch, rs = t / nch;
  float bz[8];
  int bk[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { bz[j] = -INFINITY; bk[j] = 0x7fffffff; }
  const int c0 = (cb + ch) * 8;
  {
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; }
    const unsigned short *base = Yr + (size_t)g * Kn * C + c0;
    for (int k = rs; k < Kn; k += 4 * rsub) {      // four rows in flight per lane (see bn_relu_bwd_kernel)
      uint4 raw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) raw[u] = *(const uint4 *)(base + (size_t)min(k + u * rsub, Kn - 1) * C);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kk = k + u * rsub;
        const unsigned short *y = (const unsigned short *)&raw[u];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float z = fmaf(sc[j], bf2f(y[j]), sh[j]);
          if (kk < Kn && z > bz[j]) { bz[j] = z; bk[j] = kk; }
        }
      }
    }
  }
  // combine the row sub-lanes that live in this wave (lowest k wins ties: "first k attaining the max")
  for (int off = nch; off < 64; off <<= 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float z2 = __shfl_xor(bz[j], off);
      const int k2 = __shfl_xor(bk[j], off);
      if (z2 > bz[j] || (z2 == bz[j] && k2 < bk[j])) { bz[j] = z2; bk[j] = k2; }
    }
  }
  if (lane < nch) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { s_z[wave][ch * 8 + j + (ch >> 2)] = bz[j]; s_k[wave][ch * 8 + j + (ch >> 2)] = bk[j]; }
  }
  __syncthreads();
  if (t < nch * 8) {
    const int ch2 = t / 8, j = t % 8;
    float z = -INFINITY;
    int kk = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float zz = s_z[r][ch2 * 8 + j + (ch2 >> 2)];
      const int k2 = s_k[r][ch2 * 8 + j + (ch2 >> 2)];
      if (zz > z || (zz == z && k2 < kk)) { z = zz; kk = k2; }
    }
    if (kk >= Kn) kk = 0;                   // all-NaN column: keep row 0 like the first version
    const int c = (cb + ch2) * 8 + j;
    out[(size_t)g * C + c] = f2bf(fmaxf(z, 0.f));
    arg[(size_t)g * C + c] = (unsigned char)kk;
    yarg[(size_t)g * C + c] = Yr[((size_t)g * Kn + kk) * C + c];
  }
}

// ---------------------------------------------------------------- BatchNorm backward, pass 1
// g_z = g_a·[z>0]; per-workgroup partial Σ g_z and Σ g_z·y per channel.  Gz may alias Ga.
// Rows per workgroup of the column-reduction passes: ~1024 workgroups when P allows (a 4-workgroup launch
// on the 2048-row sa4 layers was a 70 us latency chain), between 16 and 512 rows each.
static inline int bn_rows_per_block(long long P) {
  int r = 16;
  while (r < 512 && (P + r - 1) / r > 1024) r *= 2;
  return r;
}
template <bool DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DROP ? 3 : 4))) void bn_relu_bwd_kernel(const unsigned short *__restrict__ Ga,
                                                          const unsigned short *__restrict__ Yr,
                                                          const float *__restrict__ scale,
                                                          const float *__restrict__ shift, long long P, int C,
                                                          unsigned short *__restrict__ Gz,
                                                          float *__restrict__ partial, int rpb,
                                                          const unsigned long long *__restrict__ drop_seed,
                                                          unsigned thresh16, float inv_keep) {
  __shared__ float s_red[2][256][8 + 1];
  const unsigned long long seed = DROP ? *drop_seed : 0ull;
  const int t = threadIdx.x;
  const int chunks = C / 8;
  const long long row0 = (long long)blockIdx.x * rpb;
  float a1[8], a2[8];
  for (int cb = 0; cb < chunks; cb += 256) {
    const int nch = min(256, chunks - cb);
    const int rsub = 256 / nch;  // nch is a power of two <= 256 for every layer width used
    const int ch = t % nch, rs = t / nch;
    const int c0 = (cb + ch) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a1[j] = 0.f; a2[j] = 0.f; }
    if (rs < rsub) {
      float sc[8], sh[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; }
      // four rows per trip, all eight 16-byte loads issued before the first use (the compiler serialises a
      // plain row loop: load, wait, store, load, ...); rows past the end are clamped and masked out
      const long long rend = min(P, row0 + rpb);
      for (long long r = row0 + rs; r < rend; r += 4 * rsub) {
        uint4 rg[4], ry[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long rr = min(r + (long long)u * rsub, rend - 1);
          rg[u] = *(const uint4 *)(Ga + rr * C + c0);
          ry[u] = *(const uint4 *)(Yr + rr * C + c0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long rr = r + (long long)u * rsub;
          const bool live = rr < rend;
          const unsigned short *g = (const unsigned short *)&rg[u], *y = (const unsigned short *)&ry[u];
          unsigned short o[8];
          float f[8];
          if (DROP) dropout_factors(seed, (unsigned long long)((min(rr, rend - 1) * C + c0) >> 3), thresh16, inv_keep, f);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float yv = bf2f(y[j]);
            float ga = bf2f(g[j]);
            if (DROP) ga *= f[j];                   // incoming gradient is w.r.t. the dropped activation
            const float gz = (live && fmaf(sc[j], yv, sh[j]) > 0.f) ? ga : 0.f;
            o[j] = f2bf(gz);
            a1[j] += gz;
            a2[j] = fmaf(gz, yv, a2[j]);
          }
          if (Gz && live) *(uint4 *)(Gz + rr * C + c0) = *(const uint4 *)o;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) { s_red[0][t][j] = a1[j]; s_red[1][t][j] = a2[j]; }
    __syncthreads();
    // 16*nch outputs (2 sums x nch chunks x 8 channels) spread over all 256 lanes, each adding its rsub
    // row-subset values in a fixed order (the first version left this to nch lanes: a 7 us serial tail)
    for (int o = t; o < 16 * nch; o += 256) {
      const int which = o / (8 * nch), rem = o - which * 8 * nch, chn = rem >> 3, j = rem & 7;
      float s = 0.f;
      for (int r = 0; r < rsub; ++r) s += s_red[which][r * nch + chn][j];
      partial[((size_t)blockIdx.x * 2 + which) * C + cb * 8 + rem] = s;
    }
    __syncthreads();
  }
}

// dβ = Σg_z, dγ = rstd·(Σg_z·y − mean·Σg_z);  g_y = s·g_z + c2·y + c3 with
// s = γ·rstd, c2 = −s·dγ·rstd/count, c3 = −s·dβ/count − c2·mean  (training-mode batch norm).
__global__ __launch_bounds__(RTPB) void bn_bwd_finalize_kernel(const float *__restrict__ partial, int nblk, int C, float count,
                                       const float *__restrict__ gamma, const float *__restrict__ mean,
                                       const float *__restrict__ rstd, int training, float *__restrict__ dgamma,
                                       float *__restrict__ dbeta, float *__restrict__ coef /*[3][C]*/) {
  __shared__ double s_acc[RSUB][16][2];
  const int c = blockIdx.x * 16 + (threadIdx.x & 15), r = threadIdx.x >> 4;
  double s1, s2;
  partial_sums_16x16(partial, nblk, C, c, r, s_acc, s1, s2);
  if (r != 0 || c >= C) return;
  const double m = mean[c], rs = rstd[c];
  const double dg = rs * (s2 - m * s1);
  dgamma[c] = (float)dg;
  dbeta[c] = (float)s1;
  const double s = (double)gamma[c] * rs;
  const double c2 = training ? -s * dg * rs / count : 0.0;
  const double c3 = training ? -s * s1 / count - c2 * m : 0.0;
  coef[c] = (float)s;
  coef[C + c] = (float)c2;
  coef[2 * C + c] = (float)c3;
}

// g_y[p,c] = s·g_z + c2·y + c3   (dense).  Gy may alias Gz.
// With scale/shift given, Gz is really g_a and the ReLU mask [scale·y+shift > 0] is recomputed here, so the
// reduction pass (bn_relu_bwd) does not have to write the masked gradient at all.
template <bool MASK>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const unsigned short *__restrict__ Gz,
                                                           const unsigned short *__restrict__ Yr,
                                                           const float *__restrict__ coef,
                                                           const float *__restrict__ scale,
                                                           const float *__restrict__ shift, long long total8,
                                                           int C, unsigned short *__restrict__ Gy,
                                                           const unsigned long long *__restrict__ drop_seed,
                                                           unsigned thresh16, float inv_keep) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total8) return;
  const int c0 = (int)((e * 8) % C);
  float f[8];
  if (drop_seed) dropout_factors(*drop_seed, (unsigned long long)e, thresh16, inv_keep, f);
  const uint4 rg = *(const uint4 *)(Gz + e * 8);
  const uint4 ry = *(const uint4 *)(Yr + e * 8);
  const unsigned short *g = (const unsigned short *)&rg, *y = (const unsigned short *)&ry;
  unsigned short o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float yv = bf2f(y[j]);
    float gz = bf2f(g[j]);
    if (drop_seed) gz *= f[j];
    if (MASK) gz = fmaf(scale[c0 + j], yv, shift[c0 + j]) > 0.f ? gz : 0.f;
    o[j] = f2bf(fmaf(coef[c0 + j], gz, fmaf(coef[C + c0 + j], yv, coef[2 * C + c0 + j])));
  }
  *(uint4 *)(Gy + e * 8) = *(const uint4 *)o;
}

// pooled: g_z[g,k,c] = (k == arg[g,c]) ? g_pool[g,c]·[z_arg>0] : 0
// One workgroup per group: the [C]-sized per-group vectors (arg, masked pooled gradient) and the
// per-channel coefficients are read ONCE per lane (8 channels, 16-byte loads) and reused over the
// group's Kn rows, so the kernel streams Y -> Gy at one 16-byte load + store per 8 elements.
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(const unsigned short *__restrict__ Gp,
                                                                const unsigned char *__restrict__ arg,
                                                                const unsigned short *__restrict__ yarg,
                                                                const unsigned short *__restrict__ Yr,
                                                                const float *__restrict__ scale,
                                                                const float *__restrict__ shift,
                                                                const float *__restrict__ coef, int Kn, int C,
                                                                int kper, unsigned short *__restrict__ Gy) {
  const long long g = blockIdx.x;
  const int kbeg = blockIdx.y * kper, kend = min(Kn, kbeg + kper);   // row range of this workgroup
  const int t = threadIdx.x;
  const int chunks = C / 8;
  for (int cb = 0; cb < chunks; cb += 256) {
    const int nch = min(256, chunks - cb);       // power of two
    const int rsub = 256 / nch;
    const int ch = t % nch, rs = t / nch;
    const int c0 = (cb + ch) * 8;
    float gz[8], c0v[8], c1v[8], c2v[8];
    int ak[8];
    {
      const uint4 rgp = *(const uint4 *)(Gp + g * C + c0);
      const uint4 rya = *(const uint4 *)(yarg + g * C + c0);
      const uint2 rar = *(const uint2 *)(arg + g * C + c0);
      const unsigned short *gp = (const unsigned short *)&rgp, *ya = (const unsigned short *)&rya;
      const unsigned char *ar = (const unsigned char *)&rar;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float za = fmaf(scale[c0 + j], bf2f(ya[j]), shift[c0 + j]);
        gz[j] = za > 0.f ? bf2f(gp[j]) : 0.f;
        ak[j] = ar[j];
        c0v[j] = coef[c0 + j]; c1v[j] = coef[C + c0 + j]; c2v[j] = coef[2 * C + c0 + j];
      }
    }
    for (int k = kbeg + rs; k < kend; k += 4 * rsub) {   // four rows in flight per lane
      uint4 ry[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        ry[u] = *(const uint4 *)(Yr + ((size_t)g * Kn + min(k + u * rsub, kend - 1)) * C + c0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kk = k + u * rsub;
        const unsigned short *y = (const unsigned short *)&ry[u];
        unsigned short o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          o[j] = f2bf(fmaf(c0v[j], ak[j] == kk ? gz[j] : 0.f, fmaf(c1v[j], bf2f(y[j]), c2v[j])));
        if (kk < kend) *(uint4 *)(Gy + ((size_t)g * Kn + kk) * C + c0) = *(const uint4 *)o;
      }
    }
  }
}

// ---------------------------------------------------------------- weight gradient
// dW[n,k] = Σ_p Gy[p,n]·A[p,k]: the contraction runs over ROWS, so both MFMA operands are
// transposed tiles — staged row-major in LDS and read with ds_read_b64_tr_b16.
// grid (N/TN, ceil(K/TK), splits); partial[split][N][K] fp32.  The row loop is a 4-deep register pipeline:
// the 16-byte chunks of step i+4 are in flight while step i goes registers -> LDS -> transposed fragments ->
// MFMA, so a workgroup's time is its bytes, not (steps x memory latency) as in the first version (which
// had a 25 us floor on every layer).  128x128 tiles read each operand once for the 128-wide layers.
constexpr int WG_STEP = 32;       // rows per MFMA step
constexpr int WG_DEPTH = 4;       // steps in flight

template <int TN, int TK>
struct WgradLds {
  __attribute__((aligned(16))) unsigned short g[WG_STEP * (TN + 8)];
  __attribute__((aligned(16))) unsigned short a[WG_STEP * (TK + 8)];
};

// (body with the workgroup's position as arguments, like mlp_gemm_smallp_body)
template <int TN, int TK>
__device__ __forceinline__ void mlp_wgrad_body(WgradLds<TN, TK> &lds, int bx, int by, int bz,
                                               const unsigned short *__restrict__ Gy, int ldg,
                                               const unsigned short *__restrict__ A, int lda,
                                               const int *__restrict__ gidx, long long P, int N, int K,
                                               long long rows_per_split, float *__restrict__ partial,
                                               const float *__restrict__ a_scale, const float *__restrict__ a_shift) {
  constexpr int LDN = TN + 8, LDK = TK + 8;       // LDS row strides (elements)
  constexpr int CG = TN / 64, CA = TK / 64;       // 16-byte chunks per thread and step
  constexpr int MI = TN / 32, MJ = TK / 32;       // MFMA tiles per wave (wave sub-tile = TN/2 x TK/2)
  unsigned short *s_g = lds.g, *s_a = lds.a;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n0 = bx * TN, k0 = by * TK;
  const long long p0 = (long long)bz * rows_per_split, p1 = min(P, p0 + rows_per_split);
  if (p0 >= p1) {   // empty split: its partial slab must still be zero
    float *o = partial + (size_t)bz * N * K;
    for (int e = t; e < TN * TK; e += 256) {
      const int n = n0 + e / TK, k = k0 + e % TK;
      if (n < N && k < K) o[(size_t)n * K + k] = 0.f;
    }
    return;
  }
  const int wn = (wave >> 1) * (TN / 2), wk = (wave & 1) * (TK / 2);
  f32x4 acc[MI][MJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  // chunk c = t + 256 i of a step: row c / (T/8), column 8 (c % (T/8))
  uint4 vg[WG_DEPTH][CG], va[WG_DEPTH][CA];
  // optional BatchNorm + ReLU of the PREVIOUS layer on the A operand (a lane's chunk columns never change)
  float asc[CA][8], ash[CA][8];
  if (a_scale) {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int col = min(k0 + ((t + 256 * i) % (TK / 8)) * 8, K - 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { asc[i][j] = a_scale[col + j]; ash[i][j] = a_shift[col + j]; }
    }
  }
  auto issue = [&](int sidx, long long base) {
#pragma unroll
    for (int i = 0; i < CG; ++i) {
      const int c = t + 256 * i;
      const long long p = min(base + c / (TN / 8), p1 - 1);      // clamped: always a valid row, zeroed at store time
      vg[sidx][i] = *(const uint4 *)(Gy + p * ldg + n0 + (c % (TN / 8)) * 8);
    }
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int c = t + 256 * i;
      const long long p = min(base + c / (TK / 8), p1 - 1);
      const long long ar = gidx ? (long long)gidx[p] : p;
      const int col = min(k0 + (c % (TK / 8)) * 8, K - 8);
      va[sidx][i] = *(const uint4 *)(A + ar * lda + col);
    }
  };
  auto stage = [&](int sidx, long long base) {
#pragma unroll
    for (int i = 0; i < CG; ++i) {
      const int c = t + 256 * i, r = c / (TN / 8);
      uint4 v = vg[sidx][i];
      if (base + r >= p1) v = (uint4){0, 0, 0, 0};
      *(uint4 *)&s_g[r * LDN + (c % (TN / 8)) * 8] = v;
    }
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int c = t + 256 * i, r = c / (TK / 8), col = (c % (TK / 8)) * 8;
      uint4 v = va[sidx][i];
      if (a_scale) v = __builtin_bit_cast(uint4, bn_relu_frag(__builtin_bit_cast(bf16x8, v), asc[i], ash[i]));
      if (base + r >= p1 || k0 + col >= K) v = (uint4){0, 0, 0, 0};
      *(uint4 *)&s_a[r * LDK + col] = v;
    }
  };
#pragma unroll
  for (int d = 0; d < WG_DEPTH; ++d) issue(d, p0 + (long long)d * WG_STEP);
  for (long long base0 = p0; base0 < p1; base0 += WG_STEP * WG_DEPTH) {
#pragma unroll
    for (int d = 0; d < WG_DEPTH; ++d) {
      // no "if (base < p1)" here: a skipped stage would make the number of loads in flight path-dependent and
      // the compiler falls back to vmcnt(0) drains; steps past the end stage zeros instead (rows_per_split is
      // a multiple of WG_STEP*WG_DEPTH, so only the last split of a ragged P ever does that)
      const long long base = base0 + (long long)d * WG_STEP;
      __syncthreads();
      stage(d, base);
      __syncthreads();
      issue(d, base + WG_STEP * WG_DEPTH);
      bf16x8 fg[MI], fa[MJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) fg[i] = tr_frag<LDN>(s_g, wn + 16 * i, lane);
#pragma unroll
      for (int j = 0; j < MJ; ++j) fa[j] = tr_frag<LDK>(s_a, wk + 16 * j, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fg[i], fa[j], acc[i][j], 0, 0, 0);
    }
  }
  // D[row = n-local 4(lane>>4)+r][col = k-local lane&15]
  float *o = partial + (size_t)bz * N * K;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn + i * 16 + 4 * (lane >> 4) + r, k = k0 + wk + j * 16 + (lane & 15);
        if (n < N && k < K) o[(size_t)n * K + k] = acc[i][j][r];
      }
}

template <int TN, int TK>
__global__ __launch_bounds__(256) void mlp_wgrad_kernel(const unsigned short *__restrict__ Gy, int ldg,
                                                        const unsigned short *__restrict__ A, int lda,
                                                        const int *__restrict__ gidx, long long P, int N, int K,
                                                        long long rows_per_split, float *__restrict__ partial,
                                                        const float *__restrict__ a_scale,
                                                        const float *__restrict__ a_shift,
                                                        unsigned long long *probe = nullptr) {
  const unsigned long long probe_t0 = probe_begin(probe);
  __shared__ WgradLds<TN, TK> lds;
  mlp_wgrad_body<TN, TK>(lds, blockIdx.x, blockIdx.y, blockIdx.z, Gy, ldg, A, lda, gidx, P, N, K, rows_per_split, partial, a_scale, a_shift);
  probe_end(probe, probe_t0, 4);
}

// A SMALL layer's weight gradient and data gradient in ONE launch (cpfn_mlp_bwd_small): both read the same g_y and do not
// depend on each other, and as two launches of 3-10 us each on the step's chain they cost a kernel boundary (2.5-3 us) plus
// the shorter of the two durations more than they must — seven times per backward pass (sa3, sfp1, sfp2).  The grid's first
// nW workgroups are mlp_wgrad_kernel<64,64>'s, the rest mlp_gemm_smallp_kernel<32,false,true,BST>'s; the two bodies share
// the workgroup's LDS (a union) and — at 32-row tiles — have the same register footprint (116 / 128), so neither loses
// occupancy.  Same arithmetic, same partial layouts: bit-identical to the two launches.
struct BwdSmallGrid { int nW, wgx, wgy, dgx; };
template <bool BST>
__global__ __launch_bounds__(256) void mlp_bwd_small_kernel(
    BwdSmallGrid gr, const unsigned short *__restrict__ Gy, int ldg, const unsigned short *__restrict__ A, int lda, long long P,
    int N, int K, long long rows_per_split, float *__restrict__ partial, const float *__restrict__ a_scale,
    const float *__restrict__ a_shift, const unsigned short *__restrict__ W, int g_bytes, unsigned short *__restrict__ Gout, int ldo,
    float *__restrict__ stats_partial, const SmallpBwdArgs bw, unsigned long long *probe) {
  const unsigned long long probe_t0 = probe_begin(probe);
  __shared__ union U { WgradLds<64, 64> w; SmallpLds<32> d; __device__ U() {} } lds;
  const int b = blockIdx.x;
  if (b < gr.nW) {
    const int bx = b % gr.wgx, r = b / gr.wgx;
    mlp_wgrad_body<64, 64>(lds.w, bx, r % gr.wgy, r / gr.wgy, Gy, ldg, A, lda, nullptr, P, N, K, rows_per_split, partial, a_scale, a_shift);
  } else {
    const int d = b - gr.nW;
    // in the small-P kernel's terms: operand = g_y [P, N] (contraction over the layer's N output channels), outputs = K channels
    mlp_gemm_smallp_body<32, false, true, BST>(lds.d, d % gr.dgx, d / gr.dgx, Gy, ldg, g_bytes, W, (int)P, N, K, Gout, ldo, stats_partial,
                                               nullptr, nullptr, bw);
  }
  probe_end(probe, probe_t0, 6);
}

// ---------------------------------------------------------------- weight gradient + data gradient in one pass
// A dense layer's backward reads its BatchNorm-adjoint gradient g_y twice: mlp_wgrad (g_y^T . A) and the data-gradient
// GEMM (g_y . W).  Here ONE kernel does both from one read: the g_y tile that the weight-gradient loop stages in LDS
// anyway (row-major, so its rows are also MFMA point fragments) is multiplied with the transposed weight panel (filled
// once per workgroup) and the STEP x TK slab of the data gradient leaves through an LDS patch one step later as 16-byte
// row-contiguous stores.  BST: as in the streaming GEMM, pass 1 of the BatchNorm backward of the layer BELOW (sum g_z,
// sum g_z.y) is taken from the slab being stored and the matching rows of that layer's pre-BN output, requested one step
// ahead.  APPLY: Gy is the gradient with respect to the layer's ACTIVATED output and the BatchNorm-backward apply pass
// (g_y = c0 . [scale . y + shift > 0] . g + c1 . y + c2, rounded to bf16 exactly as cpfn_bn_bwd_apply stores it) runs on
// the staged chunks from the layer's own pre-BN output Yr: g_y is never written to or read from memory.
// Shapes <TN, TK, STEP>: <128,128,32>, <256,128,32>, <64,64,64>, <128,64,64> (layer N -> channels of g_y, K -> channels of its input;
// STEP rows per step, 128 rows in flight).  Grid (1, 1, splits), the split layout of mlp_wgrad_kernel: same partials,
// bit for bit.
// EIGHT waves: for <128,128> a wave's share of the dW tile is 32 x 64 (32 accumulator registers) and a thread stages one
// 16-byte chunk per tensor and step, which keeps the kernel under 224 registers.  That matters inside the replayed step:
// a wave of > 256 registers cannot be placed on a SIMD that hosts a wave of the geometry branch, and the first version
// (4 waves, 304 registers) then ran in two rounds — 52-61 us instead of 36 (in-kernel probe, tools/dbg/probe_timeline.py).
struct BwdApplyArgs {                       // APPLY != 0: what forms g_y on the staged chunks
  const unsigned short *Yr;                  // the layer's own pre-BN output [P, TN]
  const float *coef, *y_scale, *y_shift;     // finalize coefficients [3][TN], forward scale / shift (ReLU mask)
  const unsigned long long *drop_seed;       // APPLY == 3: dropout on the layer's output (cpfn_bn_bwd_apply's)
  unsigned thresh16;
  float inv_keep;
  const unsigned char *pool_arg;             // APPLY == 2 (max-pooled layer): arg-max row of every (group, channel) ...
  const unsigned short *pool_yarg;           // ... the pre-BN value there; Gy is then the POOLED gradient [P / pool_k, TN]
  int pool_k;
  long long groups;
  const float *xt_xyz;                       // XT: [P,3] fp32 coordinates that are three more input channels of the layer
  float *xt_partial;                         //     [splits][TN][3]: split partials of their weight-gradient columns
};

// (Round 2 also had an instantiation that RECOMPUTED sa1's first-layer output from the coordinates inside the 64 -> 64
//  shape instead of reading it: 134 MB fewer reads bought 2 us of 65 — the shape is bound by VALU + LDS issue — and it was
//  removed in round 3; the recompute lives on in cpfn_smallk_wgrad_apply_xyz, where it pays.)
// XT (sa2's first layer, see stream_tile): the layer has three more input channels, the fp32 coordinates xyz [P,3].  They need
// no data gradient (coordinates are inputs) and their weight-gradient columns dWx [TN,3] = g_y^T . xyz ride along: the
// 32 x 3 coordinate tile of a step is staged as bf16 beside the input tile and costs the four waves that own wk = 0 MI MFMAs
// more per step.
template <int TN, int TK, int STEP, bool BST, int APPLY, bool XT = false, bool BDROP = false>
__global__ __launch_bounds__(512) void mlp_bwd_fused_kernel(
    const unsigned short *__restrict__ Gy, int ldg, const unsigned short *__restrict__ A, int lda,
    const unsigned short *__restrict__ W /* forward weight panel [TN][TK] bf16 */, long long P, long long rows_per_split,
    float *__restrict__ partial, unsigned short *__restrict__ Gout, int ldo, const float *__restrict__ a_scale,
    const float *__restrict__ a_shift, const unsigned short *__restrict__ Yb, const float *__restrict__ b_scale,
    const float *__restrict__ b_shift, float *__restrict__ stats_partial, const BwdApplyArgs ap,
    unsigned long long *probe = nullptr) {
  const unsigned long long probe_t0 = probe_begin(probe);
  const unsigned short *__restrict__ Yr = ap.Yr;
  // BDROP (with BST, no apply pass: the fc2 heads): the layer BELOW ends in the fused dropout, i.e. the slab leaving here is the
  // gradient w.r.t. the DROPPED activation — the riding reduction scales it by the mask recomputed from the 8-byte seed, as
  // bn_relu_bwd_kernel<true> does (that launch, 15 us on the fc1 features, is then not made).
  static_assert(!BDROP || (BST && APPLY == 0), "the layer below's dropout belongs to the riding reduction of a linear layer");
  const unsigned long long bdrop_seed = BDROP ? *ap.drop_seed : 0ull;
  // (rows in flight: 128, and 64 for the 256-wide g_y — the same bytes, half the staging registers.  The 64-row-step shapes
  //  are NOT waiting for memory: taking 134 MB of the 64 -> 64 kernel's reads away (recomputing them) saved 2 us of 65, and 256 rows in
  //  flight instead of 128 made both of them 5 % slower (registers); their apply / statistics / conversion VALU work and
  //  LDS traffic per row are what a 64-channel row costs.)
  constexpr int NT = 512, LDN = TN + 8, LDK = TK + 8, DEPTH = (WG_STEP * WG_DEPTH) / STEP / (TN > 128 ? 2 : 1), KSTEPS = STEP / 32;
  constexpr int CPRG = TN / 8, CPRA = TK / 8;                 // 16-byte chunks per row of the TN- / TK-wide tensors
  constexpr int NG = STEP * CPRG / NT;                        // g_y chunks per thread and step
  // TK-wide tensors (input, data-gradient slab, y of the layer below): RPA rows per pass, TA of the 512 threads busy
  // (TK = 192: 24 chunks per row -> 16 rows x 24 = 384 threads, two passes per 32-row step)
  constexpr int RPA = (NT / CPRA >= 64 ? 64 : NT / CPRA >= 32 ? 32 : 16) < STEP ? (NT / CPRA >= 64 ? 64 : NT / CPRA >= 32 ? 32 : 16) : STEP;
  constexpr int TA = RPA * CPRA, NA = STEP / RPA;
  constexpr int MI = TN / 64, MJ = TK / 32;                   // dW tiles per wave (waves 4 x 2 over TN x TK)
  constexpr int CHB = TK / 16, TPW = (STEP / 16) * CHB / 8;   // 16-channel blocks of the data-gradient slab, its tiles per wave
  static_assert(NG >= 1 && NA >= 1 && TA <= NT && (WG_STEP * WG_DEPTH) % (STEP * DEPTH) == 0 && (STEP / 16) * CHB == 8 * TPW, "shape");
  static_assert(!BST || (64 % CPRA == 0), "the riding reduction needs a power-of-two chunk count per row");
  // DB (the 64-row-step shapes: sa1, bound by VALU + LDS issue rather than memory, and run after the geometry work, so
  // their LDS footprint is free): the three row tiles are double-buffered by step parity, which leaves ONE barrier per step
  // (stage -> barrier -> store previous slab / MFMAs) instead of two — the waves may drift a step apart and the VALU-heavy
  // staging of one overlaps the LDS / MFMA phase of another.
  constexpr bool DB = STEP == 64;
  static_assert(!DB || DEPTH % 2 == 0, "the buffer of a step is its pipeline slot's parity");
  __shared__ __attribute__((aligned(16))) unsigned short s_g2[DB ? 2 : 1][STEP * LDN];
  __shared__ __attribute__((aligned(16))) unsigned short s_a2[DB ? 2 : 1][STEP * LDK];
  __shared__ __attribute__((aligned(16))) unsigned short s_o2[DB ? 2 : 1][STEP * LDK];
  __shared__ __attribute__((aligned(16))) unsigned short s_wt[TK * LDN];
  constexpr int LDX = 16 + 8;
  __shared__ __attribute__((aligned(16))) unsigned short s_x2[XT ? (DB ? 2 : 1) * STEP * LDX : 8];
  static_assert(!XT || (STEP * 3 <= NT && MI >= 1), "xyz tail: one float per thread and step");
  f32x4 accx[XT ? MI : 1];
#pragma unroll
  for (int i = 0; i < (XT ? MI : 1); ++i) accx[i] = (f32x4){0, 0, 0, 0};
  float vxt[XT ? DEPTH : 1];
  static_assert(!BST || sizeof(float) * 8 * 2 * TK <= sizeof(unsigned short) * STEP * LDN, "the statistics reduction reuses s_g");
  float(*s_red)[2][TK] = (float(*)[2][TK])s_g2[0];  // cross-wave reduction of the statistics: after the last step only
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 15, lq = lane >> 4;
  const long long p0 = (long long)blockIdx.z * rows_per_split, p1 = min(P, p0 + rows_per_split);
  float *o = partial + (size_t)blockIdx.z * TN * TK;
  if (XT) {      // columns 3..15 of the coordinate tile stay zero
    for (int e = t; e < (DB ? 2 : 1) * STEP * LDX; e += NT) s_x2[e] = 0;
    __syncthreads();
  }
  if (p0 >= p1) {   // empty split: its partial slab (and its statistics row) must still be zero
    for (int e = t; e < TN * TK; e += NT) o[e] = 0.f;
    if (XT) for (int e = t; e < TN * 3; e += NT) ap.xt_partial[(size_t)blockIdx.z * TN * 3 + e] = 0.f;
    if (BST) for (int e = t; e < 2 * TK; e += NT) stats_partial[(size_t)blockIdx.z * 2 * TK + e] = 0.f;
    probe_end(probe, probe_t0, 5);
    return;
  }
  const int wn = (wave >> 1) * (TN / 4), wk = (wave & 1) * (TK / 2);
  f32x4 acc[MI][MJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  // chunk i of thread t: row t / CPR + i (NT / CPR), columns 8 (t % CPR) — a thread's columns never change
  const int grow = t / CPRG, gcol = (t % CPRG) * 8, arow = (TA == NT ? t : t % TA) / CPRA, acol = (t % CPRA) * 8;
  const bool a_live = TA == NT || t < TA;
  uint4 vg[APPLY == 2 ? 1 : DEPTH][NG], va[DEPTH][NA], vy[APPLY ? DEPTH : 1][NG];
  uint4 vgp[APPLY == 2 ? DEPTH : 1], vya[APPLY == 2 ? DEPTH : 1];       // pooled gradient / arg-max value of the step's group
  uint2 var_[APPLY == 2 ? DEPTH : 1];                                    // arg-max row (8 channels, one byte each)
  // (pooled: a step never straddles two groups — the host checks pool_k % STEP == 0)
  float asc[8], ash[8];
  if (a_scale) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { asc[j] = a_scale[acol + j]; ash[j] = a_shift[acol + j]; }
  }
  // APPLY: the five per-channel vectors wait in LDS and are re-read at every stage (40 registers otherwise: with them
  // the <128,128> kernel needs 237, and two such waves plus a 56-register wave of the geometry branch do not fit one SIMD)
  // (64-row-step shapes — sa1, after the geometry work has ended, bound by VALU + LDS issue rather than memory — keep them
  //  in registers: 10 LDS reads per stage less)
  constexpr bool COEF_REGS = APPLY != 0 && STEP == 64;
  __shared__ __attribute__((aligned(16))) float s_cf[APPLY && !COEF_REGS ? 5 * TN : 4];
  float cfr[COEF_REGS ? 5 : 1][8];
  if (COEF_REGS) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      cfr[0][j] = ap.coef[gcol + j]; cfr[1][j] = ap.coef[TN + gcol + j]; cfr[2][j] = ap.coef[2 * TN + gcol + j];
      cfr[COEF_REGS ? 3 : 0][j] = ap.y_scale[gcol + j]; cfr[COEF_REGS ? 4 : 0][j] = ap.y_shift[gcol + j];
    }
  } else if (APPLY) {
    for (int e = t; e < 3 * TN; e += NT) s_cf[e] = ap.coef[e];
    if (t < TN) { s_cf[3 * TN + t] = ap.y_scale[t]; s_cf[4 * TN + t] = ap.y_shift[t]; }
    // (visible after the first barrier of the step loop)
  }
  float bsc[8], bsh[8], st_s[8], st_q[8];
  if (BST) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { bsc[j] = b_scale[acol + j]; bsh[j] = b_shift[acol + j]; st_s[j] = 0.f; st_q[j] = 0.f; }
  }
  auto issue = [&](int sidx, long long base) {
    if (APPLY == 2) {
      const long long grp = min(base / ap.pool_k, ap.groups - 1);
      vgp[sidx] = *(const uint4 *)(Gy + grp * TN + gcol);
      vya[sidx] = *(const uint4 *)(ap.pool_yarg + grp * TN + gcol);
      var_[sidx] = *(const uint2 *)(ap.pool_arg + grp * TN + gcol);
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const long long p = min(base + grow + i * (NT / CPRG), p1 - 1);      // clamped: always a valid row, zeroed at store time
      if (APPLY != 2) vg[sidx][i] = *(const uint4 *)(Gy + p * ldg + gcol);
      if (APPLY) vy[sidx][i] = *(const uint4 *)(Yr + p * ldg + gcol);
    }
    if (a_live) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const long long p = min(base + arow + i * RPA, p1 - 1);
        va[sidx][i] = *(const uint4 *)(A + p * lda + acol);
      }
    }
    if (XT && t < STEP * 3) {              // float t of the step's contiguous 32 x 3 block (rows past the split's end: clamped, zeroed at stage time)
      const long long p = min(base + t / 3, p1 - 1);
      vxt[XT ? sidx : 0] = ap.xt_xyz[p * 3 + t % 3];
    }
  };
  auto stage = [&](int sidx, long long base, int buf) {
    unsigned short *s_g = s_g2[buf], *s_a = s_a2[buf];
    float cf0[8], cf1[8], cf2[8], ysc[8], ysh[8];
    if (APPLY && COEF_REGS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { cf0[j] = cfr[0][j]; cf1[j] = cfr[1][j]; cf2[j] = cfr[2][j]; ysc[j] = cfr[3][j]; ysh[j] = cfr[4][j]; }
    } else if (APPLY) {
      int zero;                                        // opaque 0: keeps these loop-invariant reads INSIDE the loop
      asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
      const float *cp = &s_cf[gcol + zero];
      *(float4 *)&cf0[0] = *(const float4 *)&cp[0];          *(float4 *)&cf0[4] = *(const float4 *)&cp[4];
      *(float4 *)&cf1[0] = *(const float4 *)&cp[TN];         *(float4 *)&cf1[4] = *(const float4 *)&cp[TN + 4];
      *(float4 *)&cf2[0] = *(const float4 *)&cp[2 * TN];     *(float4 *)&cf2[4] = *(const float4 *)&cp[2 * TN + 4];
      *(float4 *)&ysc[0] = *(const float4 *)&cp[3 * TN];     *(float4 *)&ysc[4] = *(const float4 *)&cp[3 * TN + 4];
      *(float4 *)&ysh[0] = *(const float4 *)&cp[4 * TN];     *(float4 *)&ysh[4] = *(const float4 *)&cp[4 * TN + 4];
    }
    float pz[8];          // APPLY == 2: masked pooled gradient and arg-max row of this step's group, per channel
    int pk[8];
    if (APPLY == 2) {
      const unsigned gw[4] = {vgp[sidx].x, vgp[sidx].y, vgp[sidx].z, vgp[sidx].w};
      const unsigned yw[4] = {vya[sidx].x, vya[sidx].y, vya[sidx].z, vya[sidx].w};
      const unsigned aw[2] = {var_[sidx].x, var_[sidx].y};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float ya = __uint_as_float((j & 1) ? (yw[j >> 1] & 0xffff0000u) : (yw[j >> 1] << 16));
        const float gp = __uint_as_float((j & 1) ? (gw[j >> 1] & 0xffff0000u) : (gw[j >> 1] << 16));
        pz[j] = fmaf(ysc[j], ya, ysh[j]) > 0.f ? gp : 0.f;
        pk[j] = (int)((aw[j >> 2] >> (8 * (j & 3))) & 0xffu);
      }
    }
    const int kbase = APPLY == 2 ? (int)(base % ap.pool_k) : 0;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int r = grow + i * (NT / CPRG);
      uint4 g4 = APPLY == 2 ? (uint4){0, 0, 0, 0} : vg[sidx][i];
      if (APPLY & 1) {     // (the arithmetic of bn_bwd_apply_kernel<true>, element for element; APPLY == 3: with dropout)
        const uint4 y4 = vy[sidx][i];
        const unsigned gw[4] = {g4.x, g4.y, g4.z, g4.w}, yw[4] = {y4.x, y4.y, y4.z, y4.w};
        unsigned ow[4];
        float f[8];
        if (APPLY == 3)
          dropout_factors(*ap.drop_seed, (unsigned long long)(min(base + r, p1 - 1) * CPRG + (gcol >> 3)), ap.thresh16, ap.inv_keep, f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float y0 = __uint_as_float(yw[j] << 16), y1 = __uint_as_float(yw[j] & 0xffff0000u);
          float z0 = __uint_as_float(gw[j] << 16), z1 = __uint_as_float(gw[j] & 0xffff0000u);
          if (APPLY == 3) { z0 *= f[2 * j]; z1 *= f[2 * j + 1]; }
          z0 = fmaf(ysc[2 * j], y0, ysh[2 * j]) > 0.f ? z0 : 0.f;
          z1 = fmaf(ysc[2 * j + 1], y1, ysh[2 * j + 1]) > 0.f ? z1 : 0.f;
          const unsigned lo = f2bf(fmaf(cf0[2 * j], z0, fmaf(cf1[2 * j], y0, cf2[2 * j])));
          const unsigned hi = f2bf(fmaf(cf0[2 * j + 1], z1, fmaf(cf1[2 * j + 1], y1, cf2[2 * j + 1])));
          ow[j] = lo | (hi << 16);
        }
        g4 = (uint4){ow[0], ow[1], ow[2], ow[3]};
      }
      if (APPLY == 2) {     // (the arithmetic of bn_pool_bwd_apply_kernel, element for element)
        const uint4 y4 = vy[sidx][i];
        const unsigned yw[4] = {y4.x, y4.y, y4.z, y4.w};
        const int k = kbase + r;
        unsigned ow[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float y0 = __uint_as_float(yw[j] << 16), y1 = __uint_as_float(yw[j] & 0xffff0000u);
          const unsigned lo = f2bf(fmaf(cf0[2 * j], pk[2 * j] == k ? pz[2 * j] : 0.f, fmaf(cf1[2 * j], y0, cf2[2 * j])));
          const unsigned hi = f2bf(fmaf(cf0[2 * j + 1], pk[2 * j + 1] == k ? pz[2 * j + 1] : 0.f, fmaf(cf1[2 * j + 1], y1, cf2[2 * j + 1])));
          ow[j] = lo | (hi << 16);
        }
        g4 = (uint4){ow[0], ow[1], ow[2], ow[3]};
      }
      if (base + r >= p1) g4 = (uint4){0, 0, 0, 0};
      *(uint4 *)&s_g[r * LDN + gcol] = g4;
    }
    if (a_live) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int r = arow + i * RPA;
        uint4 a4 = va[sidx][i];
        if (a_scale) a4 = __builtin_bit_cast(uint4, bn_relu_frag(__builtin_bit_cast(bf16x8, a4), asc, ash));
        if (base + r >= p1) a4 = (uint4){0, 0, 0, 0};
        *(uint4 *)&s_a[r * LDK + acol] = a4;
      }
    }
    if (XT && t < STEP * 3) {
      const int r = t / 3;
      s_x2[(DB ? buf : 0) * STEP * LDX + r * LDX + t % 3] = base + r >= p1 ? (unsigned short)0 : f2bf(vxt[XT ? sidx : 0]);
    }
  };
  // the STEP x TK data-gradient slab of the PREVIOUS step leaves here (its LDS patch was completed before this step's
  // first barrier): NA 16-byte pieces per thread
  uint4 yb[NA];
  auto store_prev = [&](long long pbase, int buf) {
    const unsigned short *s_o = s_o2[buf];
    if (!a_live) return;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int r = arow + i * RPA;
      const long long p = pbase + r;
      const uint4 v = *(const uint4 *)&s_o[r * LDK + acol];
      if (p < p1) {
        *(uint4 *)(Gout + p * ldo + acol) = v;
        if (BST) {
          const uint4 ybv = yb[i];
          const unsigned g4[4] = {v.x, v.y, v.z, v.w}, y4[4] = {ybv.x, ybv.y, ybv.z, ybv.w};
          float df[8];
          if (BDROP) dropout_factors(bdrop_seed, (unsigned long long)((p * ldo + acol) >> 3), ap.thresh16, ap.inv_keep, df);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float g0 = __uint_as_float(g4[j] << 16), g1 = __uint_as_float(g4[j] & 0xffff0000u);
            if (BDROP) { g0 *= df[2 * j]; g1 *= df[2 * j + 1]; }
            const float y0 = __uint_as_float(y4[j] << 16), y1 = __uint_as_float(y4[j] & 0xffff0000u);
            const float z0 = fmaf(bsc[2 * j], y0, bsh[2 * j]) > 0.f ? g0 : 0.f;
            const float z1 = fmaf(bsc[2 * j + 1], y1, bsh[2 * j + 1]) > 0.f ? g1 : 0.f;
            st_s[2 * j] += z0; st_s[2 * j + 1] += z1;
            st_q[2 * j] = fmaf(z0, y0, st_q[2 * j]); st_q[2 * j + 1] = fmaf(z1, y1, st_q[2 * j + 1]);
          }
        }
      }
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(d, p0 + (long long)d * STEP);
  fill_w_panel<TK, LDN, NT>(s_wt, W, TN, TK, 0, 0, TN, 1, t);      // s_wt[k_out][n]: the forward weight [n][k] transposed
  long long prev = -1;
  for (long long base0 = p0; base0 < p1; base0 += STEP * DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const long long base = base0 + (long long)d * STEP;
      constexpr int dummy_ = 0; (void)dummy_;
      const int buf = DB ? (d & 1) : 0;
      unsigned short *s_g = s_g2[buf], *s_a = s_a2[buf], *s_o = s_o2[buf];
      if (DB) {
        // ONE barrier per step.  Why that orders every LDS access (tests/test_gpu_concurrency.py replays these shapes
        // hundreds of times beside the geometry graph and compares bit for bit): step d writes the row tiles of parity
        // d & 1, the barrier, reads the slab patch of parity (d - 1) & 1 (completed by every wave BEFORE it arrived at this
        // barrier), then reads the row tiles of parity d & 1 and writes the patch of parity d & 1.  A wave can only be
        // one barrier ahead of the slowest one, i.e. staging step d + 1 into the OTHER parity while the slowest still
        // reads step d's tiles; the tiles / patch of parity d & 1 are written again at step d + 2, behind barrier d + 1,
        // which the slowest wave only reaches after all its reads of step d.
        stage(d, base, buf);
        __syncthreads();
        if (prev >= 0) store_prev(prev, buf ^ 1);
      } else {
        __syncthreads();
        if (prev >= 0) store_prev(prev, DB ? buf ^ 1 : 0);
        stage(d, base, buf);
        __syncthreads();
      }
      issue(d, base + STEP * DEPTH);
      if (BST) {       // this step's slab, used one step later
#pragma unroll
        for (int i = 0; i < NA; ++i)
          yb[i] = *(const uint4 *)(Yb + min(base + arow + i * RPA, p1 - 1) * ldo + acol);
      }
      // ---- weight gradient: transposed fragments of both tiles (as mlp_wgrad_kernel), 32 rows per MFMA
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        bf16x8 fg[MI], fa[MJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) fg[i] = tr_frag<LDN>(s_g + kk * 32 * LDN, wn + 16 * i, lane);
#pragma unroll
        for (int j = 0; j < MJ; ++j) fa[j] = tr_frag<LDK>(s_a + kk * 32 * LDK, wk + 16 * j, lane);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < MJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fg[i], fa[j], acc[i][j], 0, 0, 0);
        if (XT && wk == 0) {         // (wave-uniform) the coordinate columns of the weight gradient
          const bf16x8 fx = tr_frag<LDX>(s_x2 + (DB ? buf : 0) * STEP * LDX + kk * 32 * LDX, 0, lane);
#pragma unroll
          for (int i = 0; i < MI; ++i) accx[XT ? i : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fg[i], fx, accx[XT ? i : 0], 0, 0, 0);
        }
      }
      if constexpr (CHB == 8 || CHB == 4) {
        // ---- data gradient of the same rows: this wave's output channels 16 cb .. +15 of rows 16 rb0 .. +31 (two tiles
        //      that share the weight fragment)
        //      (the 8 waves cover CHB column blocks x 8 / CHB row-tile pairs per trip: one trip for 32 x 128 and 64 x 64,
        //       two for the heads' 64-row step of 128 channels)
        const int cb = wave % CHB;
#pragma unroll
        for (int rb0 = (wave / CHB) * 2; rb0 < STEP / 16; rb0 += (8 / CHB) * 2) {
          f32x4 ad[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
          for (int ks = 0; ks < TN / 32; ++ks) {
            const bf16x8 pf0 = *(const bf16x8 *)&s_g[(rb0 * 16 + lr) * LDN + ks * 32 + 8 * lq];
            const bf16x8 pf1 = *(const bf16x8 *)&s_g[(rb0 * 16 + 16 + lr) * LDN + ks * 32 + 8 * lq];
            const bf16x8 wf = *(const bf16x8 *)&s_wt[(cb * 16 + lr) * LDN + ks * 32 + 8 * lq];
            ad[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, pf0, ad[0], 0, 0, 0);
            ad[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, pf1, ad[1], 0, 0, 0);
          }
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const f32x4 v = ad[tt];
            const bf16x4 ov = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *(bf16x4 *)&s_o[((rb0 + tt) * 16 + lr) * LDK + cb * 16 + 4 * lq] = ov;
          }
        }
      } else {
        // ---- (TK = 192: 2 x 12 tiles of 16 x 16) tile q = wave + 8 i of the slab
        f32x4 ad[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) ad[i] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < TN / 32; ++ks) {
#pragma unroll
          for (int i = 0; i < TPW; ++i) {
            const int q = wave + 8 * i, cb = q % CHB, rb = q / CHB;
            const bf16x8 pf = *(const bf16x8 *)&s_g[(rb * 16 + lr) * LDN + ks * 32 + 8 * lq];
            const bf16x8 wf = *(const bf16x8 *)&s_wt[(cb * 16 + lr) * LDN + ks * 32 + 8 * lq];
            ad[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, pf, ad[i], 0, 0, 0);
          }
        }
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const int q = wave + 8 * i, cb = q % CHB, rb = q / CHB;
          const f32x4 v = ad[i];
          const bf16x4 ov = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
          *(bf16x4 *)&s_o[(rb * 16 + lr) * LDK + cb * 16 + 4 * lq] = ov;
        }
      }
      prev = base;
    }
  }
  __syncthreads();
  store_prev(prev, DB ? (DEPTH - 1) & 1 : 0);
  // D[row = n-local 4(lane>>4)+r][col = k-local lane&15]
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        o[(size_t)(wn + i * 16 + 4 * (lane >> 4) + r) * TK + wk + j * 16 + (lane & 15)] = acc[i][j][r];
  if (XT && wk == 0 && (lane & 15) < 3) {
    float *ox = ap.xt_partial + (size_t)blockIdx.z * TN * 3;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) ox[(size_t)(wn + i * 16 + 4 * (lane >> 4) + r) * 3 + (lane & 15)] = accx[XT ? i : 0][r];
  }
  if (BST) {   // threads that share a column chunk (t % CPRA): shuffles inside the wave, then the eight waves through LDS
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int m = CPRA; m < 64; m <<= 1) { st_s[j] += __shfl_xor(st_s[j], m, 64); st_q[j] += __shfl_xor(st_q[j], m, 64); }
    }
    if (lane < CPRA) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { s_red[wave][0][lane * 8 + j] = st_s[j]; s_red[wave][1][lane * 8 + j] = st_q[j]; }
    }
    __syncthreads();
    for (int e = t; e < 2 * TK; e += NT) {
      const int which = e / TK, c = e - which * TK;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += s_red[w][which][c];
      stats_partial[((size_t)blockIdx.z * 2 + which) * TK + c] = v;
    }
  }
  probe_end(probe, probe_t0, 5);
}

template <int RS>
__global__ __launch_bounds__(16 * RS) void split_reduce_kernel(const float *__restrict__ partial, int splits, long long n,
                                    float *__restrict__ out) {
  __shared__ float s_acc[RS][16];
  const long long e = (long long)blockIdx.x * 16 + (threadIdx.x & 15);
  const int r = threadIdx.x >> 4;
  float a = 0.f;
  if (e < n) {
#pragma unroll 4
    for (int i = r; i < splits; i += RS) a += partial[(size_t)i * n + e];
  }
  s_acc[r][threadIdx.x & 15] = a;
  __syncthreads();
  if (r == 0 && e < n) {
    float s = 0.f;
    for (int q = 0; q < RS; ++q) s += s_acc[q][threadIdx.x & 15];
    out[e] = s;
  }
}

// The same fixed-order reduction for up to MSR_MAX partial buffers in ONE launch: the weight-gradient kernels of a
// whole backward pass leave their split partials behind and are finished together at its end (19 launches of
// ~7 us each, all latency, become one).  blockIdx -> (buffer, 64-element group) through prefix sums.
constexpr int MSR_MAX = 32;
struct MsrArgs {
  const float *partial[MSR_MAX];
  float *out[MSR_MAX];
  long long n[MSR_MAX];
  int splits[MSR_MAX];
  int row_in[MSR_MAX], row_out[MSR_MAX];   // 0, 0: flat; else only the first row_out of every row_in elements are kept
  int out_ld[MSR_MAX];                     // ... at row stride out_ld of the output (>= row_out; a slice of a wider matrix)
  int deep[MSR_MAX];           // 1: few outputs, many splits: 16 elements x 16 split-subsets per workgroup instead of 64 x 4; 2: wide
  int block0[MSR_MAX + 1];     // first workgroup of buffer i
  int count;
};
__global__ __launch_bounds__(256) void multi_split_reduce_kernel(MsrArgs a) {
  // 64 consecutive elements x 4 split-subsets per workgroup: 256-byte coalesced rows of the partial buffers.  "deep"
  // buffers (the 192 outputs x 1024 partials of the fp32-xyz layer, the 35 x 512 of the heads' bias: three / one workgroup
  // walking 256 / 128 rows each was a 20 us serial tail of this launch, which is why they had their own launches): 16 x 16.
  // "wide" buffers (mode 2: n % 4 == 0, 16-byte aligned — every large weight matrix): 256 elements x 4 subsets, one
  // float4 per lane and row and eight rows in flight — the launch reads ~250 MB and was latency-bound with 4-byte loads
  // (3.7 TB/s).  The order of the additions per element is the same in all three modes' common case (subset r adds rows
  // r, r + 4, ...; then (s0 + s1) + (s2 + s3)), so mode 2 is bit-identical to mode 0.
  __shared__ __attribute__((aligned(16))) float s_acc[16][64];
  int d = 0;
  while (d + 1 < a.count && (int)blockIdx.x >= a.block0[d + 1]) ++d;
  const float *__restrict__ partial = a.partial[d];
  const long long n = a.n[d];
  const int splits = a.splits[d];
  if (a.deep[d] == 2) {
    float4 (*s4)[64] = (float4 (*)[64])s_acc;          // [4 subsets][64 lanes] float4 = 4 KB
    const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
    const long long e = ((long long)(blockIdx.x - a.block0[d]) * 64 + lane) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < n) {
      const float *src = partial + e;
      int i = r;
      for (; i + 28 < splits; i += 32) {               // eight rows of this subset in flight
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const float4 *)(src + (size_t)(i + 4 * u) * n);
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      }
      for (; i < splits; i += 4) {
        const float4 v = *(const float4 *)(src + (size_t)i * n);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    s4[r][lane] = acc;
    __syncthreads();
    if (r == 0 && e < n) {
      const float4 s0 = s4[0][lane], s1 = s4[1][lane], s2 = s4[2][lane], s3 = s4[3][lane];
      const float v[4] = {(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                          (s0.w + s1.w) + (s2.w + s3.w)};
      const int ri = a.row_in[d];
      if (ri == 0) {
        *(float4 *)(a.out[d] + e) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long long row = (e + j) / ri;
          const int col = (int)((e + j) - row * ri);
          if (col < a.row_out[d]) a.out[d][row * a.out_ld[d] + col] = v[j];
        }
      }
    }
    return;
  }
  const bool deep = a.deep[d] != 0;
  const int epw = deep ? 16 : 64, nsub = deep ? 16 : 4;
  const int lane = deep ? (threadIdx.x & 15) : (threadIdx.x & 63), r = deep ? (threadIdx.x >> 4) : (threadIdx.x >> 6);
  const long long e = (long long)(blockIdx.x - a.block0[d]) * epw + lane;
  float acc = 0.f;
  if (e < n) {
#pragma unroll 4
    for (int i = r; i < splits; i += nsub) acc += partial[(size_t)i * n + e];
  }
  s_acc[r][lane] = acc;
  __syncthreads();
  if (r == 0 && e < n) {
    float v;
    if (deep) {
      v = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) v += s_acc[q][lane];
    } else {
      v = (s_acc[0][lane] + s_acc[1][lane]) + (s_acc[2][lane] + s_acc[3][lane]);
    }
    const int ri = a.row_in[d];
    if (ri == 0) {
      a.out[d][e] = v;
    } else {           // zero-padded K: drop the padding columns (the caller gets a compact [N, row_out] matrix)
      const long long row = e / ri;
      const int col = (int)(e - row * ri);
      if (col < a.row_out[d]) a.out[d][row * a.out_ld[d] + col] = v;
    }
  }
}

// ---------------------------------------------------------------- fp32 small-K first layer (sa1: K = 3)
// Y[p,c] = Σ_{j<KS} W[c,j]·X[p,j]  (fp32 inputs: relative coordinates are NOT rounded to bf16),
// bf16 output + Σy, Σy² partials.  One lane per (row-sub, 8-channel chunk).
constexpr int KS_MAX = 4;
template <int KS>
__device__ __forceinline__ void smallk_fwd_body(int bx, const float *__restrict__ X, const float *__restrict__ W, long long P, int C,
                                                unsigned short *__restrict__ Y, float *__restrict__ partial, int rpb) {
  __shared__ float s_red[2][256][8 + 1];
  const int t = threadIdx.x;
  const int nch = C / 8, rsub = 256 / nch;  // C <= 2048, power of two
  const int ch = t % nch, rs = t / nch, c0 = ch * 8;
  const long long row0 = (long long)bx * rpb;
  float w[8][KS];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int q = 0; q < KS; ++q) w[j][q] = W[(c0 + j) * KS + q];
  float a1[8], a2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a1[j] = 0.f; a2[j] = 0.f; }
  if (rs < rsub) {
    const long long rend = min(P, row0 + rpb);
    for (long long r = row0 + rs; r < rend; r += 4 * rsub) {      // four rows in flight per lane
      float x[4][KS];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long rr = min(r + (long long)u * rsub, rend - 1);
#pragma unroll
        for (int q = 0; q < KS; ++q) x[u][q] = X[rr * KS + q];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long rr = r + (long long)u * rsub;
        if (rr < rend) {
          unsigned short o[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < KS; ++q) v = fmaf(w[j][q], x[u][q], v);
            o[j] = f2bf(v);
            a1[j] += v;
            a2[j] = fmaf(v, v, a2[j]);
          }
          *(uint4 *)(Y + rr * C + c0) = *(const uint4 *)o;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { s_red[0][t][j] = a1[j]; s_red[1][t][j] = a2[j]; }
  __syncthreads();
  for (int o = t; o < 16 * nch; o += 256) {
    const int which = o / (8 * nch), rem = o - which * 8 * nch, chn = rem >> 3, j = rem & 7;
    float s = 0.f;
    for (int r = 0; r < rsub; ++r) s += s_red[which][r * nch + chn][j];
    partial[((size_t)bx * 2 + which) * C + rem] = s;
  }
}
template <int KS>
__global__ __launch_bounds__(256) void smallk_fwd_kernel(const float *__restrict__ X,
                                                         const float *__restrict__ W, long long P, int C,
                                                         unsigned short *__restrict__ Y,
                                                         float *__restrict__ partial, int rpb) {
  smallk_fwd_body<KS>((int)blockIdx.x, X, W, P, C, Y, partial, rpb);
}
// The same launch with the step's bf16 weight-panel refresh (cpfn_multi_cast) as its first workgroups: sa1's first layer reads
// the fp32 weight itself, so the two are independent — and both sit at the very start of the step's chain, where the refresh
// alone was a ~7 us launch of pure latency (cpfn_smallk_fwd_cast).
#include "cast_body.h"
template <int KS>
__global__ __launch_bounds__(256) void smallk_fwd_cast_kernel(McvArgs cast, int cast_blocks, const float *__restrict__ X,
                                                              const float *__restrict__ W, long long P, int C,
                                                              unsigned short *__restrict__ Y, float *__restrict__ partial, int rpb) {
  if ((int)blockIdx.x < cast_blocks) multi_cast_body<256>(cast, (int)blockIdx.x);
  else smallk_fwd_body<KS>((int)blockIdx.x - cast_blocks, X, W, P, C, Y, partial, rpb);
}

// dW[c,j] = Σ_p Gy[p,c]·X[p,j]: partial[gridDim.x][C][KS]
// APPLY: Gy is the gradient w.r.t. the layer's ACTIVATED output; g_y is formed on the fly from the layer's pre-BN output
// Yr exactly as cpfn_bn_bwd_apply rounds it to bf16 (so the stand-alone apply pass and the g_y tensor disappear).
// XYZ (with APPLY): the layer's pre-BN output y is not read but recomputed from the row's coordinates and the layer's own
// weight W0 [C][KS] (smallk_fwd_kernel's arithmetic, rounded to bf16): 12 bytes instead of 2 C per row.
template <int KS, bool APPLY, bool XYZ = false>
__global__ __launch_bounds__(256) void smallk_wgrad_kernel(const unsigned short *__restrict__ Gy,
                                                           const float *__restrict__ X, long long P,
                                                           int C, float *__restrict__ partial, int rpb,
                                                           const unsigned short *__restrict__ Yr = nullptr,
                                                           const float *__restrict__ coef = nullptr,
                                                           const float *__restrict__ y_scale = nullptr,
                                                           const float *__restrict__ y_shift = nullptr,
                                                           const float *__restrict__ W0 = nullptr) {
  static_assert(!XYZ || APPLY, "XYZ is a variant of the folded apply pass");
  __shared__ float s_red[256][8 * KS + 1];
  const int t = threadIdx.x;
  const int nch = C / 8, rsub = 256 / nch;
  const int ch = t % nch, rs = t / nch, c0 = ch * 8;
  float cf0[8], cf1[8], cf2[8], ysc[8], ysh[8], w0r[XYZ ? 8 : 1][KS];
  if (APPLY) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      cf0[j] = coef[c0 + j]; cf1[j] = coef[C + c0 + j]; cf2[j] = coef[2 * C + c0 + j];
      ysc[j] = y_scale[c0 + j]; ysh[j] = y_shift[c0 + j];
      if (XYZ) {
#pragma unroll
        for (int q = 0; q < KS; ++q) w0r[j][q] = W0[(c0 + j) * KS + q];
      }
    }
  }
  const long long row0 = (long long)blockIdx.x * rpb;
  float a[8][KS];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int q = 0; q < KS; ++q) a[j][q] = 0.f;
  if (rs < rsub) {
    const long long rend = min(P, row0 + rpb);
    for (long long r = row0 + rs; r < rend; r += 4 * rsub) {      // four rows in flight per lane
      float x[4][KS];
      uint4 rg[4], ry[APPLY ? 4 : 1];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long rr = min(r + (long long)u * rsub, rend - 1);
        rg[u] = *(const uint4 *)(Gy + rr * C + c0);
        if (APPLY && !XYZ) ry[u] = *(const uint4 *)(Yr + rr * C + c0);
#pragma unroll
        for (int q = 0; q < KS; ++q) x[u][q] = X[rr * KS + q];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool live = r + (long long)u * rsub < rend;
        const unsigned short *g = (const unsigned short *)&rg[u];
        const unsigned short *y = (const unsigned short *)&ry[APPLY ? u : 0];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float gv = live ? bf2f(g[j]) : 0.f;
          if (APPLY) {
            float yv;
            if (XYZ) {
              float v = 0.f;
#pragma unroll
              for (int q = 0; q < KS; ++q) v = fmaf(w0r[XYZ ? j : 0][q], x[u][q], v);
              yv = bf2f(f2bf(v));
            } else {
              yv = bf2f(y[j]);
            }
            const float gz = fmaf(ysc[j], yv, ysh[j]) > 0.f ? bf2f(g[j]) : 0.f;
            gv = live ? bf2f(f2bf(fmaf(cf0[j], gz, fmaf(cf1[j], yv, cf2[j])))) : 0.f;
          }
#pragma unroll
          for (int q = 0; q < KS; ++q) a[j][q] = fmaf(gv, x[u][q], a[j][q]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int q = 0; q < KS; ++q) s_red[t][j * KS + q] = a[j][q];
  __syncthreads();
  for (int o = t; o < C * KS; o += 256) {          // output (channel c, tap q), all lanes busy
    const int c = o / KS, q = o - c * KS;
    float s = 0.f;
    for (int r = 0; r < rsub; ++r) s += s_red[r * nch + (c >> 3)][(c & 7) * KS + q];
    partial[(size_t)blockIdx.x * C * KS + o] = s;
  }
}

// column sums of a row-major fp32 matrix X[P,C] (C <= 64): partial[gridDim.x][C], then split_reduce.
// (bias gradient of the heads: torch's strided reduce takes 0.66 ms and rocBLAS gemv 0.8 ms for [131072,35].)
constexpr int CS_ROWS = 256;   // rows per workgroup
// pad_bf16 (optional): the same pass also writes the rows as bf16 with 64 columns (X | zeros) — the padded
// gradient operand of the heads' weight / data gradient GEMMs (was torch.zeros [P,64] + a strided slice copy).
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float *__restrict__ X, long long P, int C,
                                                         float *__restrict__ partial,
                                                         unsigned short *__restrict__ pad_bf16) {
  __shared__ float s_acc[4][64];
  const int t = threadIdx.x, c = t & 63, rs = t >> 6;
  const long long row0 = (long long)blockIdx.x * CS_ROWS, rend = min(P, row0 + CS_ROWS);
  const int cc = c < C ? c : C - 1;
  float a = 0.f;
  for (long long r = row0 + rs; r < rend; r += 32) {          // eight rows in flight per lane
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = X[min(r + 4 * u, rend - 1) * C + cc];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float x = (c < C && r + 4 * u < rend) ? v[u] : 0.f;
      a += x;
      if (pad_bf16 && r + 4 * u < rend) pad_bf16[(r + 4 * u) * 64 + c] = f2bf(x);
    }
  }
  s_acc[rs][c] = a;
  __syncthreads();
  if (t < C) partial[(size_t)blockIdx.x * C + t] = s_acc[0][t] + s_acc[1][t] + s_acc[2][t] + s_acc[3][t];
}

inline bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

}  // namespace

// ============================================================================ C ABI

extern "C" int cpfn_mlp_gemm_blocks(long long P, int N) {
  // number of row-blocks (gridDim.x) the GEMM will use == rows of its stats-partial buffer
  if (P > 0 && P <= SP_MAX_ROWS) return (int)((P + sp_rows(P, N) - 1) / sp_rows(P, N));   // small-P kernel: one per row tile
  const long long tiles = (P + G_ROWS - 1) / G_ROWS;
  const int ny = (N + 127) / 128 > 0 ? (N + 127) / 128 : 1;
  // Workgroups per launch: at most ~448 (CPFN_GEMM_WGS overrides it for experiments).  Stand-alone, 512 (one round of
  // two per CU) was the optimum; inside the step the next batch's FPS holds 16 of the 256 CUs for the whole forward
  // pass (its 96 KB of LDS leaves no room for a 75 KB GEMM workgroup next to it), so 512 workgroups run as a round of
  // 480 plus a straggler round.  Measured on the replayed step (same box, A/B): 512 -> 2.435 ms, 480 / 448 / 400 ->
  // 2.404-2.414, 342 -> 2.419, 256 -> 2.440, 1024 -> 2.516.
  constexpr int target = 448;       // (342 ... 480 measured: no signal)
  long long tpw = (tiles * ny + target - 1) / target;
  if (tpw < 1) tpw = 1;
  // (tiles per workgroup: capped at 64 — at 16 the 1M-row launches of the LocalSPFN step, 32 clouds, fell back to 512
  //  workgroups = two rounds beside a 32-CU FPS: 2.680 -> 2.650 ms per step)
  constexpr int tpw_cap = 64;
  if (tpw > tpw_cap) tpw = tpw_cap;
  return (int)((tiles + tpw - 1) / tpw);
}

extern "C" int cpfn_mlp_dgrad_small_ok(long long P, int N, int K) {
  return P > 0 && P <= SP_MAX_ROWS && N > 0 && (N & 31) == 0 && N <= SP_SS_MAX && K > 0 && (K & 63) == 0 &&
         P * N * 2 < (1LL << 31) && (long long)N * K * 2 < (1LL << 31);
}

// (declared after cpfn_mlp_gemm_blocks: rows of stats_partial = cpfn_mlp_gemm_blocks(P, K))
extern "C" int cpfn_mlp_dgrad_small(const void *Gy, const void *W, long long P, int N, int K, void *Gout, int ldo,
                                    const void *bwd_y, const float *b_scale, const float *b_shift, float *stats_partial,
                                    void *stream) {
  if (!cpfn_mlp_dgrad_small_ok(P, N, K) || !Gy || !W || !Gout || (ldo & 3) || ldo < K || !bwd_y || !b_scale || !b_shift ||
      !stats_partial)
    return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // in the kernel's terms: operand A = Gz [P, N] (contraction over the layer's N output channels), outputs = K channels
  const int gx = cpfn_mlp_gemm_blocks(P, K);
  dim3 grid(gx, K / 64);
  const int a_bytes = (int)(((P - 1) * N + N) * 2);
  SmallpBwdArgs bw;
  bw.Yb = (const unsigned short *)bwd_y; bw.b_scale = b_scale; bw.b_shift = b_shift;
  const unsigned short *a = (const unsigned short *)Gy, *w = (const unsigned short *)W;
  unsigned short *y = (unsigned short *)Gout;
#define CPFN_DGRAD_SMALL(RT_)                                                                                             \
  mlp_gemm_smallp_kernel<RT_, false, true, true><<<grid, 256, 0, st>>>(a, N, a_bytes, w, (int)P, N, K, y, ldo, stats_partial, nullptr, nullptr, probe_slot(grid), bw)
  if (sp_rows(P, K) == 32) CPFN_DGRAD_SMALL(32); else CPFN_DGRAD_SMALL(64);
#undef CPFN_DGRAD_SMALL
  return cpfn_launch_status();
}

extern "C" int cpfn_mlp_gemm_set_probe(void *buf, int slots, int max_wg) {
  // buf: slots * (2 + 2 * max_wg) u64 of device memory, zero-filled by the caller (or NULL: probe off).  Applies to
  // every cpfn_mlp_gemm launch issued (or captured into a graph) from now on; launch i gets slot i % slots.
  if (buf && (slots <= 0 || max_wg <= 0)) return CPFN_EINVAL;
  g_probe_state.buf = (unsigned long long *)buf;
  g_probe_state.slots = buf ? slots : 0;
  g_probe_state.max_wg = buf ? max_wg : 0;
  g_probe_state.next = 0;
  return 0;
}

static inline bool gemm_stream_k(long long P, int K) {
  // whole-K panel in LDS: K <= 256.  K = 192 / 256 only for the long layers: with few row tiles the 50-68 KB panel
  // (cold in a real step, unlike in a micro-benchmark loop) costs more than the generic kernel's 128-wide K chunks
  return K == 64 || K == 128 || ((K == 192 || K == 256) && P >= 32768);
}

extern "C" int cpfn_mlp_gemm_can_fuse_bwd_stats(long long P, int K, int N) {
  return P > SP_MAX_ROWS && (P + G_ROWS) * (long long)(K > N ? K : N) * 2 < (1LL << 32) && gemm_stream_k(P, K) && N > 0 &&
         (N & 63) == 0;   // (contiguous operands: lda = K, ldy = N; 32-bit buffer offsets)
}

extern "C" int cpfn_mlp_gemm(const void *A, int lda, const int *gidx, const void *W, int w_trans, long long P, int K,
                             int N, void *Y, int ldy, int y_f32, int n_store, const float *bias,
                             float *stats_partial, const float *a_scale, const float *a_shift, const void *bwd_y,
                             void *stream) {
  if (P < 0 || K <= 0 || (K & 31) || N <= 0 || (N & 63) || !A || !W || !Y || lda < K || (lda & 7) || (!a_scale != !a_shift))
    return CPFN_EINVAL;
  if (bwd_y && (!stats_partial || !a_scale || gidx || bias || y_f32 || n_store != N || (ldy & 7) ||
                !cpfn_mlp_gemm_can_fuse_bwd_stats(P, K, N)))
    return CPFN_EINVAL;
  if (P == 0) return 0;
  if (P > 2000000000LL) return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int gx = cpfn_mlp_gemm_blocks(P, N);
  const long long tiles = (P + G_ROWS - 1) / G_ROWS;
  const int tpw = (int)((tiles + gx - 1) / gx);   // (gx > tiles for small P: the surplus workgroups write zero statistics)
  const unsigned short *a = (const unsigned short *)A, *w = (const unsigned short *)W;
  if (P <= SP_MAX_ROWS && !gidx && !bias && !y_f32 && n_store == N && (ldy & 3) == 0 && (!a_scale || K <= SP_SS_MAX) &&
      P * lda * 2 < (1LL << 31) && (long long)N * K * 2 < (1LL << 31)) {
    unsigned short *y = (unsigned short *)Y;
    dim3 grid(gx, N / 64);
    const int a_bytes = (int)(((P - 1) * lda + K) * 2);
#define CPFN_SMALLP(RT_)                                                                                              \
  do {                                                                                                                \
    if (stats_partial && w_trans) mlp_gemm_smallp_kernel<RT_, true, true><<<grid, 256, 0, st>>>(a, lda, a_bytes, w, (int)P, K, N, y, ldy, stats_partial, a_scale, a_shift, probe_slot(grid));  \
    else if (stats_partial) mlp_gemm_smallp_kernel<RT_, true, false><<<grid, 256, 0, st>>>(a, lda, a_bytes, w, (int)P, K, N, y, ldy, stats_partial, a_scale, a_shift, probe_slot(grid));       \
    else if (w_trans) mlp_gemm_smallp_kernel<RT_, false, true><<<grid, 256, 0, st>>>(a, lda, a_bytes, w, (int)P, K, N, y, ldy, nullptr, a_scale, a_shift, probe_slot(grid));                  \
    else mlp_gemm_smallp_kernel<RT_, false, false><<<grid, 256, 0, st>>>(a, lda, a_bytes, w, (int)P, K, N, y, ldy, nullptr, a_scale, a_shift, probe_slot(grid));                             \
  } while (0)
    if (sp_rows(P, N) == 32) CPFN_SMALLP(32); else CPFN_SMALLP(64);
#undef CPFN_SMALLP
    return cpfn_launch_status();
  }
  const bool stream_k = gemm_stream_k(P, K);
  // (the operand transform exists in the stream kernel only next to the BN statistics: forward layers)
  const bool stream_ok = stream_k && !gidx && !bias && !y_f32 && n_store == N && (ldy & 7) == 0 &&
                         (bwd_y || !a_scale || (stats_partial && K <= 128)) &&
                         (P + G_ROWS) * (long long)(lda > ldy ? lda : ldy) * 2 < (1LL << 32);   // 32-bit buffer offsets
  if (stream_ok) {
    unsigned short *y = (unsigned short *)Y;
#define CPFN_STREAM(BN_, KS_)                                                                                        \
  do {                                                                                                               \
    dim3 grid(gx, N / BN_);                                                                                          \
    if (bwd_y)                                                                                                       \
      mlp_gemm_stream_kernel<BN_, KS_, false, false, true><<<grid, G_THREADS, 0, st>>>(a, lda, w, w_trans, (int)P, N, y, ldy, stats_partial, tpw, a_scale, a_shift, (const unsigned short *)bwd_y, probe_slot(grid)); \
    else if (stats_partial && a_scale)                                                                               \
      mlp_gemm_stream_kernel<BN_, (KS_ <= 4 ? KS_ : 4), true, true><<<grid, G_THREADS, 0, st>>>(a, lda, w, w_trans, (int)P, N, y, ldy, stats_partial, tpw, a_scale, a_shift, nullptr, probe_slot(grid)); \
    else if (stats_partial)                                                                                          \
      mlp_gemm_stream_kernel<BN_, KS_, true><<<grid, G_THREADS, 0, st>>>(a, lda, w, w_trans, (int)P, N, y, ldy, stats_partial, tpw, nullptr, nullptr, nullptr, probe_slot(grid)); \
    else                                                                                                             \
      mlp_gemm_stream_kernel<BN_, KS_, false><<<grid, G_THREADS, 0, st>>>(a, lda, w, w_trans, (int)P, N, y, ldy, nullptr, tpw, nullptr, nullptr, nullptr, probe_slot(grid));      \
  } while (0)
    if (N % 128 == 0) {
      switch (K) { case 64: CPFN_STREAM(128, 2); break; case 128: CPFN_STREAM(128, 4); break;
                   case 192: CPFN_STREAM(128, 6); break; default: CPFN_STREAM(128, 8); }
    } else {
      switch (K) { case 64: CPFN_STREAM(64, 2); break; case 128: CPFN_STREAM(64, 4); break;
                   case 192: CPFN_STREAM(64, 6); break; default: CPFN_STREAM(64, 8); }
    }
#undef CPFN_STREAM
    return cpfn_launch_status();
  }
  const long long row_tiles = (P + G_ROWS - 1) / G_ROWS;
  const bool wide = (N % 128 == 0) && row_tiles * (N / 128) >= 256;   // otherwise 64-wide blocks: 2x the workgroups
  if (wide) {
    dim3 grid(gx, N / 128);
    if (stats_partial)
      mlp_gemm_kernel<128, true><<<grid, G_THREADS, 0, st>>>(a, lda, gidx, w, w_trans, (int)P, K, N, Y, ldy, y_f32, n_store, bias, stats_partial, tpw, a_scale, a_shift, probe_slot(grid));
    else
      mlp_gemm_kernel<128, false><<<grid, G_THREADS, 0, st>>>(a, lda, gidx, w, w_trans, (int)P, K, N, Y, ldy, y_f32, n_store, bias, nullptr, tpw, a_scale, a_shift, probe_slot(grid));
  } else {
    dim3 grid(gx, N / 64);
    if (stats_partial)
      mlp_gemm_kernel<64, true><<<grid, G_THREADS, 0, st>>>(a, lda, gidx, w, w_trans, (int)P, K, N, Y, ldy, y_f32, n_store, bias, stats_partial, tpw, a_scale, a_shift, probe_slot(grid));
    else
      mlp_gemm_kernel<64, false><<<grid, G_THREADS, 0, st>>>(a, lda, gidx, w, w_trans, (int)P, K, N, Y, ldy, y_f32, n_store, bias, nullptr, tpw, a_scale, a_shift, probe_slot(grid));
  }
  return cpfn_launch_status();
}

// Forward layer whose input is [A (K = 128 bf16 channels) | xyz (3 fp32 coordinates)] (sa2's first layer): Y = A . W^T +
// xyz . Wx^T with the coordinate term as one more k-step built in registers (stream_tile, XT).  Statistics rows as cpfn_mlp_gemm.
extern "C" int cpfn_mlp_gemm_xyz_ok(long long P, int K, int N) {
  return K == 128 && N == 128 && P >= 32768 && gemm_stream_k(P, K) && (P + G_ROWS) * 128LL * 2 < (1LL << 32);
}
extern "C" int cpfn_mlp_gemm_xyz(const void *A, int lda, const void *W, const float *xyz, const float *Wx, long long P, int K,
                                 int N, void *Y, int ldy, float *stats_partial, void *stream) {
  if (!cpfn_mlp_gemm_xyz_ok(P, K, N) || !A || !W || !xyz || !Wx || !Y || lda != K || ldy != N) return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int gx = cpfn_mlp_gemm_blocks(P, N);
  const long long tiles = (P + G_ROWS - 1) / G_ROWS;
  const int tpw = (int)((tiles + gx - 1) / gx);
  const dim3 grid(gx, N / 128);
  const unsigned short *a = (const unsigned short *)A, *w = (const unsigned short *)W;
  unsigned short *y = (unsigned short *)Y;
  if (stats_partial)
    mlp_gemm_stream_kernel<128, 4, true, false, false, true><<<grid, G_THREADS, 0, st>>>(a, lda, w, 0, (int)P, N, y, ldy, stats_partial, tpw, nullptr, nullptr, nullptr, probe_slot(grid), xyz, Wx);
  else
    mlp_gemm_stream_kernel<128, 4, false, false, false, true><<<grid, G_THREADS, 0, st>>>(a, lda, w, 0, (int)P, N, y, ldy, nullptr, tpw, nullptr, nullptr, nullptr, probe_slot(grid), xyz, Wx);
  return cpfn_launch_status();
}

extern "C" int cpfn_bn_finalize(const float *partial, int nblk, int N, float count, const float *gamma,
                                const float *beta, const float *conv_bias, float eps, float momentum,
                                float *running_mean, float *running_var, float *scale, float *shift,
                                float *mean, float *rstd, int64_t *counter_a, int64_t *counter_b, void *stream) {
  if (nblk <= 0 || N <= 0 || !partial || !gamma || !beta || !scale || !shift || !mean || !rstd) return CPFN_EINVAL;
  bn_finalize_kernel<<<cpfn_cdiv(N, 16), RTPB, 0, (hipStream_t)stream>>>(partial, nblk, N, count, gamma, beta, conv_bias,
                                                                        eps, momentum, running_mean, running_var,
                                                                        scale, shift, mean, rstd, (long long *)counter_a,
                                                                        (long long *)counter_b);
  return cpfn_launch_status();
}

extern "C" int cpfn_bn_eval_affine(const float *gamma, const float *beta, const float *conv_bias, const float *running_mean,
                                   const float *running_var, float eps, int C, float *out4C, void *stream) {
  if (C <= 0 || !gamma || !beta || !running_mean || !running_var || !out4C) return CPFN_EINVAL;
  bn_eval_affine_kernel<<<cpfn_cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(gamma, beta, conv_bias, running_mean, running_var, eps, C,
                                                                             out4C);
  return cpfn_launch_status();
}

extern "C" int cpfn_bn_relu_apply(const void *Y, const float *scale, const float *shift, long long P, int C,
                                  void *out, const long long *drop_counter, unsigned long long drop_base, float drop_p,
                                  unsigned long long *drop_seed_out, void *stream) {
  if (P < 0 || C <= 0 || (C & 7) || !Y || !scale || !shift || !out) return CPFN_EINVAL;
  if (drop_counter && (!drop_seed_out || !(drop_p >= 0.f && drop_p < 1.f))) return CPFN_EINVAL;
  if (P == 0) return 0;
  const long long total8 = P * C / 8;
  bn_relu_apply_kernel<<<cpfn_cdiv(total8, 256), 256, 0, (hipStream_t)stream>>>(
      (const unsigned short *)Y, scale, shift, total8, C, (unsigned short *)out, drop_counter, drop_base,
      dropout_thresh16(drop_p), 1.f / (1.f - drop_p), drop_seed_out);
  return cpfn_launch_status();
}

extern "C" int cpfn_bn_relu_maxpool(const void *Y, const float *scale, const float *shift, int G, int Kn, int C,
                                    void *out, unsigned char *arg, void *yarg, void *stream) {
  if (G < 0 || Kn <= 0 || Kn > 256 || C < 64 || (C & 7) || !pow2(C / 8) || !Y || !scale || !shift || !out || !arg || !yarg)
    return CPFN_EINVAL;
  if (G == 0) return 0;
  bn_relu_maxpool_kernel<<<dim3(G, cpfn_cdiv(C / 8, 32)), 256, 0, (hipStream_t)stream>>>((const unsigned short *)Y, scale, shift, Kn, C,
                                                             (unsigned short *)out, arg, (unsigned short *)yarg);
  return cpfn_launch_status();
}

static inline void launch_split_reduce(const float *ws, int splits, long long n, float *out, hipStream_t st) {
  if (splits > 64)
    split_reduce_kernel<64><<<cpfn_cdiv(n, 16), 1024, 0, st>>>(ws, splits, n, out);
  else
    split_reduce_kernel<16><<<cpfn_cdiv(n, 16), 256, 0, st>>>(ws, splits, n, out);
}

extern "C" int cpfn_bn_bwd_blocks(long long P) {
  const int r = bn_rows_per_block(P);
  return (int)((P + r - 1) / r);
}

extern "C" int cpfn_bn_relu_bwd(const void *Ga, const void *Y, const float *scale, const float *shift, long long P,
                                int C, void *Gz, float *partial, const unsigned long long *drop_seed, float drop_p,
                                void *stream) {
  if (P <= 0 || C <= 0 || (C & 7) || !pow2(C / 8) || !Ga || !Y || !scale || !shift || !partial) return CPFN_EINVAL;
  if (drop_seed && !(drop_p >= 0.f && drop_p < 1.f)) return CPFN_EINVAL;
  if (drop_seed)
    bn_relu_bwd_kernel<true><<<cpfn_bn_bwd_blocks(P), 256, 0, (hipStream_t)stream>>>(
        (const unsigned short *)Ga, (const unsigned short *)Y, scale, shift, P, C, (unsigned short *)Gz, partial,
        bn_rows_per_block(P), drop_seed, dropout_thresh16(drop_p), 1.f / (1.f - drop_p));
  else
    bn_relu_bwd_kernel<false><<<cpfn_bn_bwd_blocks(P), 256, 0, (hipStream_t)stream>>>(
        (const unsigned short *)Ga, (const unsigned short *)Y, scale, shift, P, C, (unsigned short *)Gz, partial,
        bn_rows_per_block(P), nullptr, 0u, 1.f);
  return cpfn_launch_status();
}

extern "C" int cpfn_bn_bwd_finalize(const float *partial, int nblk, int C, float count, const float *gamma,
                                    const float *mean, const float *rstd, int training, float *dgamma,
                                    float *dbeta, float *coef, void *stream) {
  if (nblk <= 0 || C <= 0 || !partial || !gamma || !mean || !rstd || !dgamma || !dbeta || !coef) return CPFN_EINVAL;
  bn_bwd_finalize_kernel<<<cpfn_cdiv(C, 16), RTPB, 0, (hipStream_t)stream>>>(partial, nblk, C, count, gamma, mean, rstd,
                                                                            training, dgamma, dbeta, coef);
  return cpfn_launch_status();
}

extern "C" int cpfn_bn_bwd_apply(const void *Gz, const void *Y, const float *coef, const float *scale,
                                 const float *shift, long long P, int C, void *Gy,
                                 const unsigned long long *drop_seed, float drop_p, void *stream) {
  if (P <= 0 || C <= 0 || (C & 7) || !Gz || !Y || !coef || !Gy || (!scale != !shift)) return CPFN_EINVAL;
  if (drop_seed && !(drop_p >= 0.f && drop_p < 1.f)) return CPFN_EINVAL;
  const long long total8 = P * C / 8;
  const unsigned th = dropout_thresh16(drop_p);
  const float ik = 1.f / (1.f - drop_p);
  if (scale)
    bn_bwd_apply_kernel<true><<<cpfn_cdiv(total8, 256), 256, 0, (hipStream_t)stream>>>(
        (const unsigned short *)Gz, (const unsigned short *)Y, coef, scale, shift, total8, C, (unsigned short *)Gy, drop_seed, th, ik);
  else
    bn_bwd_apply_kernel<false><<<cpfn_cdiv(total8, 256), 256, 0, (hipStream_t)stream>>>(
        (const unsigned short *)Gz, (const unsigned short *)Y, coef, scale, shift, total8, C, (unsigned short *)Gy, drop_seed, th, ik);
  return cpfn_launch_status();
}

extern "C" int cpfn_bn_pool_bwd_apply(const void *Gp, const unsigned char *arg, const void *yarg, const void *Y,
                                      const float *scale, const float *shift, const float *coef, int G, int Kn,
                                      int C, void *Gy, void *stream) {
  if (G <= 0 || Kn <= 0 || C <= 0 || (C & 7) || !Gp || !arg || !yarg || !Y || !scale || !shift || !coef || !Gy)
    return CPFN_EINVAL;
  if (!pow2(C / 8)) return CPFN_EINVAL;
  // few groups (sa4: 16 groups of 128 rows x 1024 channels): split the rows of a group over workgroups
  const int rsub = 256 / (C / 8 < 256 ? C / 8 : 256);
  int ys = G >= 1024 ? 1 : (1024 + G - 1) / G;
  int kper = cpfn_cdiv(cpfn_cdiv(Kn, ys), rsub) * rsub;        // multiple of the row sub-lane count
  if (kper < rsub) kper = rsub;
  ys = cpfn_cdiv(Kn, kper);
  bn_pool_bwd_apply_kernel<<<dim3(G, ys), 256, 0, (hipStream_t)stream>>>(
      (const unsigned short *)Gp, arg, (const unsigned short *)yarg, (const unsigned short *)Y, scale, shift, coef,
      Kn, C, kper, (unsigned short *)Gy);
  return cpfn_launch_status();
}

// output tile for one (P, N, K): 128 x 128 where the layer is wide and long enough to fill the chip with 128-tiles,
// 128 x 64 for the long 64 -> 128 layer (sa1: both operands read once instead of the input twice), 64 x 64 otherwise
static inline void wgrad_tile(long long P, int N, int K, int *TN, int *TK) {
  if (P >= 32768 && N % 128 == 0 && K >= 128) { *TN = 128; *TK = 128; }
  else if (P >= 32768 && N % 128 == 0 && K == 64) { *TN = 128; *TK = 64; }
  else { *TN = 64; *TK = 64; }
}

extern "C" int cpfn_mlp_wgrad_splits(long long P, int N, int K) {
  int TN, TK;
  wgrad_tile(P, N, K, &TN, &TK);
  const long long tiles = (long long)((N + TN - 1) / TN) * ((K + TK - 1) / TK);
  const long long target = TN == 128 ? 512 : 1024;     // workgroups (256 / 512 / 2048 for the 64-tiles: no measurable difference)
  long long s = (target + tiles - 1) / tiles;
  if (s > 256) s = 256;   // bound the partial buffer / reduce depth (128 and 512 measured: +40 us per step each)
  const long long max_s = (P + 127) / 128;            // at least 128 rows (one pipeline depth) per split
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return (int)s;
}

extern "C" int cpfn_mlp_wgrad(const void *Gy, int ldg, const void *A, int lda, const int *gidx, long long P, int N,
                              int K, const float *a_scale, const float *a_shift, float *workspace, float *dW,
                              void *stream) {
  if (P <= 0 || N <= 0 || K < 8 || (N & 63) || (K & 31) || !Gy || !A || !workspace || (ldg & 7) || (lda & 7) ||
      (!a_scale != !a_shift))
    return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int splits = cpfn_mlp_wgrad_splits(P, N, K);
  long long rps = (P + splits - 1) / splits;
  rps = ((rps + WG_STEP * WG_DEPTH - 1) / (WG_STEP * WG_DEPTH)) * (WG_STEP * WG_DEPTH);
  const unsigned short *g = (const unsigned short *)Gy, *a = (const unsigned short *)A;
  int TN, TK;
  wgrad_tile(P, N, K, &TN, &TK);
  if (TN == 128 && TK == 128) {
    dim3 grid(N / 128, (K + 127) / 128, splits);
    mlp_wgrad_kernel<128, 128><<<grid, 256, 0, st>>>(g, ldg, a, lda, gidx, P, N, K, rps, workspace, a_scale, a_shift, probe_slot_all(grid));
  } else if (TN == 128) {
    dim3 grid(N / 128, (K + 63) / 64, splits);
    mlp_wgrad_kernel<128, 64><<<grid, 256, 0, st>>>(g, ldg, a, lda, gidx, P, N, K, rps, workspace, a_scale, a_shift, probe_slot_all(grid));
  } else {
    dim3 grid(N / 64, (K + 63) / 64, splits);
    mlp_wgrad_kernel<64, 64><<<grid, 256, 0, st>>>(g, ldg, a, lda, gidx, P, N, K, rps, workspace, a_scale, a_shift, probe_slot_all(grid));
  }
  const long long n = (long long)N * K;
  if (dW) launch_split_reduce(workspace, splits, n, dW, st);    // NULL: the caller batches it (cpfn_multi_split_reduce)
  return cpfn_launch_status();
}

// Weight gradient + data gradient of a small layer as ONE launch (mlp_bwd_small_kernel): workspace as cpfn_mlp_wgrad leaves it
// (cpfn_mlp_wgrad_splits(P,N,K) slabs), Gout [P,K] = Gy . W (W: the FORWARD panel [N][K]); bwd_y (optional, + b_scale / b_shift +
// stats_partial [cpfn_mlp_bwd_small_blocks(P)][2][K]): pass 1 of the BatchNorm backward of the layer below on the stored tile.
extern "C" int cpfn_mlp_bwd_small_ok(long long P, int N, int K) {
  return cpfn_mlp_wgrad_apply_ok(P, N, K) && cpfn_mlp_dgrad_small_ok(P, N, K);
}
extern "C" int cpfn_mlp_bwd_small_blocks(long long P) { return (int)((P + 31) / 32); }
extern "C" int cpfn_mlp_bwd_small(const void *Gy, int ldg, const void *A, int lda, const void *W, long long P, int N, int K,
                                  const float *a_scale, const float *a_shift, float *workspace, void *Gout, int ldo,
                                  const void *bwd_y, const float *b_scale, const float *b_shift, float *stats_partial,
                                  void *stream) {
  if (!cpfn_mlp_bwd_small_ok(P, N, K) || !Gy || !A || !W || !workspace || !Gout || ldg != N || (lda & 7) || lda < K || (ldo & 3) ||
      ldo < K || (!a_scale != !a_shift) || (bwd_y && (!b_scale || !b_shift || !stats_partial)))
    return CPFN_EINVAL;
  const int splits = cpfn_mlp_wgrad_splits(P, N, K);
  long long rps = (P + splits - 1) / splits;
  rps = ((rps + WG_STEP * WG_DEPTH - 1) / (WG_STEP * WG_DEPTH)) * (WG_STEP * WG_DEPTH);
  BwdSmallGrid gr;
  gr.wgx = N / 64; gr.wgy = (K + 63) / 64; gr.nW = gr.wgx * gr.wgy * splits;
  gr.dgx = cpfn_mlp_bwd_small_blocks(P);
  const dim3 grid(gr.nW + gr.dgx * (K / 64));
  SmallpBwdArgs bw;
  bw.Yb = (const unsigned short *)bwd_y; bw.b_scale = b_scale; bw.b_shift = b_shift;
  const int g_bytes = (int)(((P - 1) * N + N) * 2);
  hipStream_t st = (hipStream_t)stream;
  if (bwd_y)
    mlp_bwd_small_kernel<true><<<grid, 256, 0, st>>>(gr, (const unsigned short *)Gy, ldg, (const unsigned short *)A, lda, P, N, K, rps,
                                                     workspace, a_scale, a_shift, (const unsigned short *)W, g_bytes,
                                                     (unsigned short *)Gout, ldo, stats_partial, bw, probe_slot_all(grid));
  else
    mlp_bwd_small_kernel<false><<<grid, 256, 0, st>>>(gr, (const unsigned short *)Gy, ldg, (const unsigned short *)A, lda, P, N, K, rps,
                                                      workspace, a_scale, a_shift, (const unsigned short *)W, g_bytes,
                                                      (unsigned short *)Gout, ldo, nullptr, bw, probe_slot_all(grid));
  return cpfn_launch_status();
}

// the 64 x 64-tile weight gradient with cpfn_bn_bwd_apply folded in (the layers whose data gradient is cpfn_mlp_dgrad_small)
extern "C" int cpfn_mlp_wgrad_apply_ok(long long P, int N, int K) {
  int TN, TK;
  if (P <= 0 || N <= 0 || K < 8 || (N & 63) || (K & 31)) return 0;
  wgrad_tile(P, N, K, &TN, &TK);
  return TN == 64 && TK == 64;
}

extern "C" int cpfn_mlp_bwd_fused_ok(long long P, int N, int K) {
  const bool shape = (N == 128 && K == 128) || (N == 64 && K == 64) || (N == 128 && K == 64) || (N == 256 && K == 128) ||
                     (N == 64 && K == 128);      // (64 <- 128: the fc2 heads, padded to 64 outputs; linear: no apply pass)
  return shape && P > SP_MAX_ROWS && P >= 32768;
}

extern "C" int cpfn_mlp_bwd_fused(const void *Gy, int ldg, const void *A, int lda, const void *W, long long P, int N, int K,
                                  const float *a_scale, const float *a_shift, float *workspace, void *Gout, int ldo,
                                  const void *bwd_y, const float *b_scale, const float *b_shift, float *stats_partial,
                                  const void *apply_y, const float *apply_coef, const float *y_scale, const float *y_shift,
                                  const unsigned long long *drop_seed, float drop_p, const unsigned char *pool_arg,
                                  const void *pool_yarg, int pool_k, const float *xt_xyz,
                                  float *xt_partial, void *stream) {
  if (!cpfn_mlp_bwd_fused_ok(P, N, K) || !Gy || !A || !W || !workspace || !Gout || (ldg & 7) || (lda & 7) || (ldo & 7) ||
      ldg < N || lda < K || ldo < K || (!a_scale != !a_shift) || (bwd_y && (!b_scale || !b_shift || !stats_partial)) ||
      (apply_y && (!apply_coef || !y_scale || !y_shift)))
    return CPFN_EINVAL;
  if (apply_y && ldg != N) return CPFN_EINVAL;     // (the kernel walks apply_y with the gradient's row stride)
  // drop_seed: with apply_y the layer's OWN fused dropout (fc1); without it — the 64 <- 128 heads shape only, with bwd_y — the
  // dropout of the layer BELOW, whose output feeds this linear layer (its mask scales the riding reduction)
  const bool below_drop = drop_seed && !apply_y && bwd_y && N == 64 && K == 128;
  if ((drop_seed && !below_drop && (!apply_y || pool_k > 0)) || (drop_seed && !(drop_p >= 0.f && drop_p < 1.f)) || pool_k < 0)
    return CPFN_EINVAL;
  const int step = (K >= 128 && N != 64) ? 32 : 64;
  // xyz tail (three fp32 coordinate channels beside the K bf16 ones; sa2's first layer): the 128 -> 128 shape with the dense
  // apply pass and no layer below
  if ((!xt_xyz) != (!xt_partial)) return CPFN_EINVAL;
  if (xt_xyz && !(N == 128 && K == 128 && apply_y && !bwd_y && !drop_seed && pool_k == 0)) return CPFN_EINVAL;
  if (pool_k > 0 && (!apply_y || !pool_arg || !pool_yarg || pool_k > 255 || pool_k % step || P % pool_k || ldg != N))
    return CPFN_EINVAL;
  const int splits = cpfn_mlp_wgrad_splits(P, N, K);
  long long rps = (P + splits - 1) / splits;
  rps = ((rps + WG_STEP * WG_DEPTH - 1) / (WG_STEP * WG_DEPTH)) * (WG_STEP * WG_DEPTH);
  const dim3 grid(1, 1, splits);
  hipStream_t st = (hipStream_t)stream;
  const unsigned short *g = (const unsigned short *)Gy, *a = (const unsigned short *)A, *w = (const unsigned short *)W,
                       *yb = (const unsigned short *)bwd_y;
  unsigned short *go = (unsigned short *)Gout;
  BwdApplyArgs ap;
  ap.Yr = (const unsigned short *)apply_y; ap.coef = apply_coef; ap.y_scale = y_scale; ap.y_shift = y_shift;
  ap.drop_seed = drop_seed; ap.thresh16 = dropout_thresh16(drop_seed ? drop_p : 0.f); ap.inv_keep = drop_seed ? 1.f / (1.f - drop_p) : 1.f;
  ap.pool_arg = pool_arg; ap.pool_yarg = (const unsigned short *)pool_yarg; ap.pool_k = pool_k > 0 ? pool_k : 1;
  ap.groups = pool_k > 0 ? P / pool_k : 1;
  ap.xt_xyz = xt_xyz; ap.xt_partial = xt_partial;
  const int mode = !apply_y ? 0 : (pool_k > 0 ? 2 : (drop_seed ? 3 : 1));
#define CPFN_BWD_FUSED(TN_, TK_, STEP_, BST_, APPLY_)                                                                      \
  mlp_bwd_fused_kernel<TN_, TK_, STEP_, BST_, APPLY_><<<grid, 512, 0, st>>>(g, ldg, a, lda, w, P, rps, workspace, go, ldo, \
                                                                            a_scale, a_shift, yb, b_scale, b_shift,       \
                                                                            stats_partial, ap, probe_slot_all(grid))
#define CPFN_BWD_FUSED_SHAPE(TN_, TK_, STEP_)                                  \
  do {                                                                         \
    if (bwd_y) {                                                               \
      if (mode == 2) CPFN_BWD_FUSED(TN_, TK_, STEP_, true, 2);                 \
      else if (mode == 1) CPFN_BWD_FUSED(TN_, TK_, STEP_, true, 1);            \
      else if (mode == 3) CPFN_BWD_FUSED(TN_, TK_, STEP_, true, 3);            \
      else CPFN_BWD_FUSED(TN_, TK_, STEP_, true, 0);                           \
    } else {                                                                   \
      if (mode == 2) CPFN_BWD_FUSED(TN_, TK_, STEP_, false, 2);                \
      else if (mode == 1) CPFN_BWD_FUSED(TN_, TK_, STEP_, false, 1);           \
      else if (mode == 3) CPFN_BWD_FUSED(TN_, TK_, STEP_, false, 3);           \
      else CPFN_BWD_FUSED(TN_, TK_, STEP_, false, 0);                          \
    }                                                                          \
  } while (0)
  if (xt_xyz)
    mlp_bwd_fused_kernel<128, 128, 32, false, 1, true><<<grid, 512, 0, st>>>(g, ldg, a, lda, w, P, rps, workspace, go, ldo, a_scale,
                                                                             a_shift, yb, b_scale, b_shift, stats_partial, ap,
                                                                             probe_slot_all(grid));
  else if (N == 128 && K == 128) CPFN_BWD_FUSED_SHAPE(128, 128, 32);
  else if (N == 256) CPFN_BWD_FUSED_SHAPE(256, 128, 32);
  else if (N == 64 && K == 128) {
    if (mode != 0 || (bwd_y && !below_drop)) return CPFN_EINVAL;     // (heads: linear; what may ride is fc1's dropped reduction)
    if (bwd_y)
      mlp_bwd_fused_kernel<64, 128, 64, true, 0, false, true><<<grid, 512, 0, st>>>(g, ldg, a, lda, w, P, rps, workspace, go, ldo,
                                                                                a_scale, a_shift, yb, b_scale, b_shift,
                                                                                stats_partial, ap, probe_slot_all(grid));
    else
      CPFN_BWD_FUSED(64, 128, 64, false, 0);
  } else if (N == 64 && K == 64) CPFN_BWD_FUSED_SHAPE(64, 64, 64);
  else CPFN_BWD_FUSED_SHAPE(128, 64, 64);
#undef CPFN_BWD_FUSED_SHAPE
#undef CPFN_BWD_FUSED
  return cpfn_launch_status();
}

extern "C" int cpfn_smallk_fwd_cast(const cpfn_cast_desc *casts, int n_casts, const float *X, int KS, const float *W, long long P,
                                    int C, void *Y, float *partial, void *stream) {
  if (P <= 0 || KS != 3 || C <= 0 || (C & 7) || !pow2(C / 8) || C / 8 > 256 || !X || !W || !Y || !partial || n_casts <= 0 ||
      n_casts > MCV_MAX || !casts)
    return CPFN_EINVAL;
  McvArgs a;
  int cast_blocks = 0;
  const int rc = mcv_fill(casts, n_casts, a, &cast_blocks);
  if (rc) return rc;
  const int nblk = cpfn_bn_bwd_blocks(P), rpb = bn_rows_per_block(P);
  smallk_fwd_cast_kernel<3><<<cast_blocks + nblk, 256, 0, (hipStream_t)stream>>>(a, cast_blocks, X, W, P, C, (unsigned short *)Y,
                                                                                 partial, rpb);
  return cpfn_launch_status();
}

extern "C" int cpfn_smallk_fwd(const float *X, int KS, const float *W, long long P, int C, void *Y, float *partial,
                               void *stream) {
  if (P <= 0 || KS <= 0 || KS > KS_MAX || C <= 0 || (C & 7) || !pow2(C / 8) || C / 8 > 256 || !X || !W || !Y || !partial)
    return CPFN_EINVAL;
  const int nblk = cpfn_bn_bwd_blocks(P), rpb = bn_rows_per_block(P);
  hipStream_t st = (hipStream_t)stream;
  unsigned short *y = (unsigned short *)Y;
  switch (KS) {
    case 1: smallk_fwd_kernel<1><<<nblk, 256, 0, st>>>(X, W, P, C, y, partial, rpb); break;
    case 2: smallk_fwd_kernel<2><<<nblk, 256, 0, st>>>(X, W, P, C, y, partial, rpb); break;
    case 3: smallk_fwd_kernel<3><<<nblk, 256, 0, st>>>(X, W, P, C, y, partial, rpb); break;
    default: smallk_fwd_kernel<4><<<nblk, 256, 0, st>>>(X, W, P, C, y, partial, rpb); break;
  }
  return cpfn_launch_status();
}

static int smallk_wgrad_launch(const void *Gy, const float *X, int KS, long long P, int C, float *workspace, float *dW,
                               const void *apply_y, const float *coef, const float *y_scale, const float *y_shift,
                               void *stream, const float *W0 = nullptr) {
  if (P <= 0 || KS <= 0 || KS > KS_MAX || C <= 0 || (C & 7) || !pow2(C / 8) || C / 8 > 256 || !Gy || !X || !workspace)
    return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = cpfn_bn_bwd_blocks(P);
  const unsigned short *g = (const unsigned short *)Gy, *yr = (const unsigned short *)apply_y;
  const int rpb = bn_rows_per_block(P);
#define CPFN_SMALLK_WGRAD(KS_)                                                                                              \
  do {                                                                                                                      \
    if (W0) smallk_wgrad_kernel<KS_, true, true><<<nblk, 256, 0, st>>>(g, X, P, C, workspace, rpb, nullptr, coef, y_scale, y_shift, W0); \
    else if (yr) smallk_wgrad_kernel<KS_, true><<<nblk, 256, 0, st>>>(g, X, P, C, workspace, rpb, yr, coef, y_scale, y_shift);   \
    else smallk_wgrad_kernel<KS_, false><<<nblk, 256, 0, st>>>(g, X, P, C, workspace, rpb);                                 \
  } while (0)
  switch (KS) {
    case 1: CPFN_SMALLK_WGRAD(1); break;
    case 2: CPFN_SMALLK_WGRAD(2); break;
    case 3: CPFN_SMALLK_WGRAD(3); break;
    default: CPFN_SMALLK_WGRAD(4); break;
  }
#undef CPFN_SMALLK_WGRAD
  const long long n = (long long)C * KS;
  if (dW) launch_split_reduce(workspace, nblk, n, dW, st);    // NULL: the caller batches it (cpfn_multi_split_reduce)
  return cpfn_launch_status();
}

extern "C" int cpfn_smallk_wgrad(const void *Gy, const float *X, int KS, long long P, int C, float *workspace,
                                 float *dW, void *stream) {
  return smallk_wgrad_launch(Gy, X, KS, P, C, workspace, dW, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int cpfn_smallk_wgrad_apply(const void *Gz, const void *Y, const float *coef, const float *y_scale,
                                       const float *y_shift, const float *X, int KS, long long P, int C, float *workspace,
                                       float *dW, void *stream) {
  if (!Y || !coef || !y_scale || !y_shift) return CPFN_EINVAL;
  return smallk_wgrad_launch(Gz, X, KS, P, C, workspace, dW, Y, coef, y_scale, y_shift, stream);
}

// ... with y recomputed from X and the layer's own fp32 weight W0 [C][KS] instead of read (see smallk_wgrad_kernel, XYZ)
extern "C" int cpfn_smallk_wgrad_apply_xyz(const void *Gz, const float *W0, const float *coef, const float *y_scale,
                                           const float *y_shift, const float *X, int KS, long long P, int C,
                                           float *workspace, float *dW, void *stream) {
  if (!W0 || !coef || !y_scale || !y_shift) return CPFN_EINVAL;
  return smallk_wgrad_launch(Gz, X, KS, P, C, workspace, dW, nullptr, coef, y_scale, y_shift, stream, W0);
}

extern "C" int cpfn_colsum_f32(const float *X, long long P, int C, float *workspace, float *out, void *pad_bf16,
                               void *stream) {
  if (P <= 0 || C <= 0 || C > 64 || !X || !workspace) return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (int)((P + CS_ROWS - 1) / CS_ROWS);
  colsum_f32_kernel<<<nblk, 256, 0, st>>>(X, P, C, workspace, (unsigned short *)pad_bf16);
  if (out) launch_split_reduce(workspace, nblk, C, out, st);  // NULL: the caller batches it (cpfn_multi_split_reduce)
  return cpfn_launch_status();
}

extern "C" int cpfn_multi_split_reduce(const cpfn_reduce_desc *descs, int count, void *stream) {
  if (count < 0 || (count > 0 && !descs)) return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < count; base += MSR_MAX) {
    MsrArgs a;
    a.count = count - base < MSR_MAX ? count - base : MSR_MAX;
    int blocks = 0;
    for (int i = 0; i < a.count; ++i) {
      const cpfn_reduce_desc &d = descs[base + i];
      if (!d.partial || !d.out || d.splits <= 0 || d.n <= 0 || d.row_in < 0 || d.row_out < 0 || d.row_out > d.row_in || (d.out_ld != 0 && (d.row_in == 0 || d.out_ld < d.row_out)) ||
          (d.row_in > 0 && (d.row_out == 0 || d.n % d.row_in))) return CPFN_EINVAL;
      a.partial[i] = d.partial; a.out[i] = d.out; a.n[i] = d.n; a.splits[i] = d.splits;
      a.row_in[i] = d.row_in; a.row_out[i] = d.row_out;
      a.out_ld[i] = d.out_ld > 0 ? d.out_ld : d.row_out;
      a.deep[i] = d.n <= 1024 && d.splits >= 128;
      if (!a.deep[i] && d.n >= 4096 && d.n % 4 == 0 && (((uintptr_t)d.partial | (d.row_in == 0 ? (uintptr_t)d.out : 0)) & 15) == 0)
        a.deep[i] = 2;          // wide: float4 per lane (same order of additions as the 64 x 4 layout)
      a.block0[i] = blocks;
      blocks += cpfn_cdiv(d.n, a.deep[i] == 2 ? 256 : a.deep[i] ? 16 : 64);
    }
    a.block0[a.count] = blocks;
    if (blocks) multi_split_reduce_kernel<<<blocks, 256, 0, st>>>(a);
  }
  return cpfn_launch_status();
}
