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
//
// This file: the forward / data-gradient GEMM kernels of the large layers (streaming kernel, generic kernel) and cpfn_mlp_gemm.
#include "mlp_common.h"
#include "seam.h"

GemmProbeState g_probe_state;

namespace {

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


#ifndef CPFN_STREAM_TWOBUF
#define CPFN_STREAM_TWOBUF 1
#endif
// XT + gather (round 6): the rows of the operand are not a [P, K] tensor but a GATHER of a small per-cloud table — sa2's
// grouped input rows feats[b, idx[b, s, k], :] — taken while loading: cpfn_group_concat_bf16's [P, K] copy (33.5 MB written, then
// read here and again by the backward kernel) never exists.  rows_per_cloud = S * K rows of the operand per cloud, n_src rows of
// the table per cloud; a 128-row tile never straddles two clouds (rows_per_cloud % 128 == 0).
struct GatherIn {
  const int *idx;              // [P] table row of every operand row (inside its cloud); nullptr: off
  int rows_per_cloud, n_src;
};

struct StreamBufs {
  __amdgpu_buffer_rsrc_t a, y, yb;       // operand rows, output rows, (BST) pre-BN output of the layer below
  unsigned aoff[2];                      // lane byte offset of its two operand rows inside a tile
  unsigned yoff;                         // lane byte offset of its first output piece inside a tile
  unsigned a_tile, y_tile, y_step;       // bytes per 128-row tile of A / Y, bytes between a lane's output pieces
  unsigned a_oob;                        // an offset past the end of A (lane offsets added to it do not wrap: the host's 2^32 bound)
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

// POOL (the pooled last layer of a set-abstraction stack, round 6): the max over the K neighbours starts in THIS kernel's
// epilogue.  z = scale*y + shift is monotone in y with the sign of scale = the sign of gamma (rstd > 0) — a parameter, known
// before the batch statistics are — so max_k z = affine(max_k (+-y)) and the maximum of +-y over a wave's 32 rows can be taken
// from the tile on its way out, BEFORE anybody knows scale / shift: every wave leaves (raw y of its winner, its row k inside the
// group) per channel, [P / 32][N], and cpfn_bn_pool_finish combines the pool_k / 32 wave results of a group and applies the
// affine map to the winner only.  The stand-alone pooling pass (cpfn_bn_relu_maxpool: a second read of the whole [P, N] output,
// 36 + 24 us per step for sa1 + sa2) is then a [G, N]-sized launch.
struct PoolOut {
  unsigned short *pmax;        // [ceil(P / 32)][N] bf16: raw y of the wave's winner
  unsigned char *pidx;         // [ceil(P / 32)][N]: its row inside the group (0 .. pool_k - 1)
  const float *gamma;          // [N]: the sign decides between max and min of y
  int pool_k;                  // 32 | 64 | 128
};

template <int BN, int KS, bool STATS, bool ATR, bool BST, bool XT = false, bool POOL = false>
__device__ __forceinline__ void stream_tile(bf16x8 (&af)[2][KS], const unsigned short *s_w,
                                            unsigned short *s_o, const StreamBufs &sb, int P, int row0, int next_tile,
                                            int wave, int lane, float (&st_s)[8], float (&st_q)[8],
                                            const float *s_ss /*[2][32*KS]: scale, shift*/,
                                            const float *s_bs /*[2][BN]: scale, shift of the layer below (BST)*/,
                                            const unsigned short *s_wx /*XT: [BN][16] bf16*/,
                                            float (&xz)[2][3] /*XT: xyz of the lane's two points, reloaded for the next tile*/,
                                            const float *xyz, const PoolOut &po = PoolOut(), const unsigned (&smask)[4] = {0u, 0u, 0u, 0u},
                                            int n0 = 0, int N = 0, const GatherIn &gin = GatherIn(), int lda = 0) {
  constexpr int NT = BN / 16;
  int gnext[2] = {0, 0};          // XT + gather: table rows of the lane's two operand rows in the NEXT tile
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
    const int pn = (max(next_tile, 0) * G_ROWS) + wave * 32 + lr, pa = min(pn, P - 1), pb = min(pn + 16, P - 1);   // (no next tile: any valid rows)
#pragma unroll
    for (int q = 0; q < 3; ++q) { xz[0][q] = xyz[(size_t)pa * 3 + q]; xz[1][q] = xyz[(size_t)pb * 3 + q]; }
    if (gin.idx) { gnext[0] = gin.idx[pa]; gnext[1] = gin.idx[pb]; }      // (requested a whole MFMA phase before they are used)
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
    // (round 4: for every instantiation — the plain ones sat at 254 registers, two of their workgroups filled a CU's register
    //  file and nothing of the next batch's geometry could share a SIMD with them; fenced: 94-170.  GlobalSPFN step +-0, LocalSPFN -6 us)
    __builtin_amdgcn_sched_barrier(0);
  }
  // ONE operand buffer: the next tile's rows are requested as soon as the MFMAs have consumed this one's, and fly
  // during the epilogue below (statistics, LDS staging, stores).  The first version kept two buffers (tile t+1
  // requested before the MFMAs of tile t): 268 registers for the plain 128-wide variant and 454-490 with the
  // statistics — one wave per SIMD, so a CU held ONE workgroup and its load / compute / store phases overlapped
  // with nothing.  Occupancy, not a deeper per-wave pipeline, is what hides the latency here.
  // (a next tile that is not this workgroup's — next_tile < 0 — is requested past the end of the buffer: zeros, no traffic;
  //  until round 4 every workgroup also fetched the first tile of its neighbour: 1.17 x the launch's bytes in the counters)
  if (XT && gin.idx) {
    // gathered rows of the next tile: (cloud * n_src + table row) * lda — the tile's cloud is wave-uniform
    const unsigned cbase = (unsigned)((max(next_tile, 0) * G_ROWS) / gin.rows_per_cloud) * (unsigned)gin.n_src;
    const unsigned goff[2] = {((cbase + (unsigned)gnext[0]) * (unsigned)lda + 8u * lq) * 2u, ((cbase + (unsigned)gnext[1]) * (unsigned)lda + 8u * lq) * 2u};
    stream_load_a<KS>(af, sb.a, goff, next_tile < 0 ? sb.a_oob : 0u);
  } else
  stream_load_a<KS>(af, sb.a, sb.aoff, next_tile < 0 ? sb.a_oob : (unsigned)next_tile * sb.a_tile);
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
  float pb[POOL ? 8 : 1];        // POOL: the lane's running maximum of +-y over the rows it stores (8 channels of one chunk) ...
  int pk[POOL ? 8 : 1];          // ... and the row (inside its group) that holds it: rows ascend with i, `>` keeps the first
  if (POOL) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { pb[j] = -INFINITY; pk[j] = 0x7fffffff; }
  }
#pragma unroll
  for (int i = 0; i < 32 * CPR / 64; ++i) {
    const int e = i * 64 + lane;
    const int r = e / CPR, c = e - r * CPR;
    const int p = row0 + wave * 32 + r;
    const uint4 vv = *(const uint4 *)&s_o[r * G_LDO + c * 8];
    if (POOL && p < P) {
      const int kg = (row0 + wave * 32 + r) & (po.pool_k - 1);
      const unsigned w4[4] = {vv.x ^ smask[0], vv.y ^ smask[1], vv.z ^ smask[2], vv.w ^ smask[3]};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v0 = __uint_as_float(w4[j] << 16), v1 = __uint_as_float(w4[j] & 0xffff0000u);
        if (v0 > pb[2 * j]) { pb[2 * j] = v0; pk[2 * j] = kg; }
        if (v1 > pb[2 * j + 1]) { pb[2 * j + 1] = v1; pk[2 * j + 1] = kg; }
      }
    }
    // (aux = 2: `nt` — the tile is not read again by this kernel; fewer dirty lines for the kernel boundary to write back:
    //  -4 us per step with the one-pass backward kernel's stores, same-box A/B .tnt, NOTEBOOK round 5)
    __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){vv.x, vv.y, vv.z, vv.w}, sb.y, sb.yoff + y_tile_off + i * sb.y_step, 0, 2);
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
  if (POOL) {
    // the lanes that carry the same chunk (lane % CPR; lane / CPR = row subset: higher subset = later rows) -> the wave's winner;
    // equal values keep the smaller row
#pragma unroll
    for (int m = CPR; m < 64; m <<= 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v2 = __shfl_xor(pb[j], m, 64);
        const int k2 = __shfl_xor(pk[j], m, 64);
        if (v2 > pb[j] || (v2 == pb[j] && k2 < pk[j])) { pb[j] = v2; pk[j] = k2; }
      }
    }
    const int prow = row0 / 32 + wave;
    if (lane < CPR && row0 + wave * 32 < P) {
      // raw y = the winner with its sign restored (an exact bit flip); no winner (every row NaN): k = 255, any value
      unsigned o[4];
      unsigned long long kk = 0ull;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned lo = (__float_as_uint(pb[2 * j]) >> 16), hi = (__float_as_uint(pb[2 * j + 1]) & 0xffff0000u);
        o[j] = (lo | hi) ^ smask[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) kk |= (unsigned long long)(pk[j] > 255 ? 255 : pk[j]) << (8 * j);
      *(uint4 *)&po.pmax[(size_t)prow * N + n0 + lane * 8] = (uint4){o[0], o[1], o[2], o[3]};
      *(unsigned long long *)&po.pidx[(size_t)prow * N + n0 + lane * 8] = kk;
    }
  }
}

template <int BN, int KS, bool STATS, bool ATR = false, bool BST = false, bool XT = false, bool POOL = false>
__global__ __launch_bounds__(G_THREADS) __attribute__((amdgpu_waves_per_eu(2))) void mlp_gemm_stream_kernel(
    const unsigned short *__restrict__ A, int lda, const unsigned short *__restrict__ W, int w_trans, int P, int N,
    unsigned short *__restrict__ Y, int ldy, float *__restrict__ stats_partial, int tiles_per_wg,
    const float *__restrict__ a_scale = nullptr, const float *__restrict__ a_shift = nullptr,
    const unsigned short *__restrict__ Yb = nullptr /* BST: [P, ldy] like Y */, unsigned long long *probe = nullptr,
    const float *__restrict__ xyz = nullptr /* XT: [P,3] */, const float *__restrict__ wx = nullptr /* XT: [N,3] fp32 */,
    const SeamOut so = SeamOut() /* STATS: the sums leave as fixed-point atomics instead of partial rows */,
    const SeamIn si = SeamIn() /* ATR: scale / shift folded from the previous layer's sums (seam.h) */,
    const PoolOut po = PoolOut() /* POOL: per-wave winners of the max over neighbours */,
    const GatherIn gin = GatherIn() /* XT: the operand rows are gathered from a per-cloud table A [clouds * n_src, lda] */) {
  constexpr int NT = BN / 16, K = 32 * KS, CPR = BN / 8;
  const unsigned long long probe_t0 = probe_begin(probe);
  __shared__ __attribute__((aligned(16))) unsigned short s_w[BN * (32 * KS + 8)];   // whole-K panel, rows padded by 16 B
  __shared__ __attribute__((aligned(16))) unsigned short s_o[4][32 * G_LDO];
  __shared__ __attribute__((aligned(16))) float s_red[4][2][BN];
  __shared__ __attribute__((aligned(16))) float s_ss[ATR ? 2 * K : 4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int n0 = blockIdx.y * BN;
  if (ATR && !si.acc) {   // visible after the W-panel barrier
    for (int e = t; e < K; e += G_THREADS) { s_ss[e] = a_scale[e]; s_ss[K + e] = a_shift[e]; }
  }
  if (STATS && so.acc && t == 0 && blockIdx.x == 0 && blockIdx.y == 0) seam_counters(so);
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
  unsigned smask[4] = {0u, 0u, 0u, 0u};      // POOL: bf16 sign bits of gamma for the channel pairs of the lane's output chunk
  if (POOL) {
    const int cc = n0 + (lane % CPR) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      smask[j] = (po.gamma[cc + 2 * j] < 0.f ? 0x8000u : 0u) | (po.gamma[cc + 2 * j + 1] < 0.f ? 0x80000000u : 0u);
  }
  if (BST) {   // scale / shift of the layer below for this column block (visible after the W-panel barrier)
    for (int e = t; e < BN; e += G_THREADS) { s_bs[e] = a_scale[n0 + e]; s_bs[BN + e] = a_shift[n0 + e]; }
  }
  // buffer descriptors: the host guarantees P * max(lda, ldy) * 2 < 2^32
  StreamBufs sb;
  const unsigned y_bytes = ((unsigned)(P - 1) * ldy + N) * 2u;
  const unsigned a_rows = (XT && gin.idx) ? (unsigned)((P + gin.rows_per_cloud - 1) / gin.rows_per_cloud) * (unsigned)gin.n_src : (unsigned)P;
  sb.a = __builtin_amdgcn_make_buffer_rsrc((void *)A, 0, ((a_rows - 1) * lda + K) * 2u, 0x00020000);
  sb.y = __builtin_amdgcn_make_buffer_rsrc((void *)Y, 0, y_bytes, 0x00020000);
  sb.yb = __builtin_amdgcn_make_buffer_rsrc(BST ? (void *)Yb : (void *)Y, 0, y_bytes, 0x00020000);
  sb.aoff[0] = ((unsigned)(wave * 32 + lr) * lda + 8 * lq) * 2u;
  sb.aoff[1] = sb.aoff[0] + 16u * lda * 2u;
  sb.yoff = ((unsigned)(wave * 32 + lane / CPR) * ldy + n0 + (lane % CPR) * 8) * 2u;
  sb.a_tile = (unsigned)G_ROWS * lda * 2u;
  sb.a_oob = ((a_rows - 1) * lda + K) * 2u;
  sb.y_tile = (unsigned)G_ROWS * ldy * 2u;
  sb.y_step = (unsigned)(64 / CPR) * ldy * 2u;
  const int ntiles = (P + G_ROWS - 1) / G_ROWS;
  const int tile0 = blockIdx.x * tiles_per_wg;
  const int tile_end = min(tile0 + tiles_per_wg, ntiles);
  if (tile0 < tile_end) {
    bf16x8 a[2][KS];
    // (a seam's accumulator words are requested FIRST — K <= 128: one channel per lane of the first two waves — then this
    //  workgroup's first rows and the weight panel; the fold's arithmetic runs behind the panel fill, when all of it has arrived)
    SeamFoldRegs fq;
    const bool folds = ATR && si.acc && t < K;
    if (folds) seam_fold_issue(si, t, fq);
    if (XT && gin.idx) {       // (the first tile's table rows: one exposed index round trip per workgroup)
      const int p0g = tile0 * G_ROWS + wave * 32 + lr;
      const unsigned cbase = (unsigned)((tile0 * G_ROWS) / gin.rows_per_cloud) * (unsigned)gin.n_src;
      const unsigned g0 = (unsigned)gin.idx[min(p0g, P - 1)], g1 = (unsigned)gin.idx[min(p0g + 16, P - 1)];
      const unsigned goff[2] = {((cbase + g0) * (unsigned)lda + 8u * lq) * 2u, ((cbase + g1) * (unsigned)lda + 8u * lq) * 2u};
      stream_load_a<KS>(a, sb.a, goff, 0u);
    } else
    stream_load_a<KS>(a, sb.a, sb.aoff, (unsigned)tile0 * sb.a_tile);
    if (XT) {
      const int p0 = tile0 * G_ROWS + wave * 32 + lr, pa = min(p0, P - 1), pb = min(p0 + 16, P - 1);
#pragma unroll
      for (int q = 0; q < 3; ++q) { xz[0][q] = xyz[(size_t)pa * 3 + q]; xz[1][q] = xyz[(size_t)pb * 3 + q]; }
    }
    fill_w_panel<BN, 32 * KS + 8>(s_w, W, K, N, n0, 0, K, w_trans, t);
    if (folds) {
      float sc, sh;
      seam_fold_finish(si, t, blockIdx.x == 0 && blockIdx.y == 0, fq, sc, sh);
      s_ss[t] = sc; s_ss[K + t] = sh;
    }
    __syncthreads();
    // TWO operand buffers for the operand-transform instantiations (every hidden layer's forward launch; round 4): with one,
    // the rows of tile t + 1 are requested after the MFMAs of tile t and have its epilogue (~1.5 us) to arrive — less than a
    // loaded memory round trip, so each of a workgroup's ~3 tiles exposed part of its latency (the plain kernel without its
    // stores: 10.8 us for 33.5 MB of loads).  With two, tile t + 2 is requested after the MFMAs of tile t and has a whole tile.
    // Round 1 had two buffers at 268-490 registers (64 running sums of the statistics); now the pair costs 32 of ~180.
    constexpr bool TWOBUF = CPFN_STREAM_TWOBUF && ATR && !XT && !BST;
    if (TWOBUF) {
      bf16x8 b[2][KS];
      stream_load_a<KS>(b, sb.a, sb.aoff, tile0 + 1 < tile_end ? (unsigned)(tile0 + 1) * sb.a_tile : sb.a_oob);
      for (int tile = tile0; tile < tile_end; tile += 2) {
        stream_tile<BN, KS, STATS, ATR, BST, XT, POOL>(a, s_w, s_o[wave], sb, P, tile * G_ROWS, tile + 2 < tile_end ? tile + 2 : -1, wave, lane,
                                                       st_s, st_q, s_ss, s_bs, s_wx, xz, xyz, po, smask, n0, N, gin, lda);
        if (tile + 1 < tile_end)
          stream_tile<BN, KS, STATS, ATR, BST, XT, POOL>(b, s_w, s_o[wave], sb, P, (tile + 1) * G_ROWS, tile + 3 < tile_end ? tile + 3 : -1, wave,
                                                         lane, st_s, st_q, s_ss, s_bs, s_wx, xz, xyz, po, smask, n0, N, gin, lda);
      }
    } else
    for (int tile = tile0; tile < tile_end; ++tile) {
      // the reload inside is unconditional (a tile past the end is out of the buffer's range: zeros, no traffic), so
      // the loop body is straight-line
      stream_tile<BN, KS, STATS, ATR, BST, XT, POOL>(a, s_w, s_o[wave], sb, P, tile * G_ROWS, tile + 1 < tile_end ? tile + 1 : -1, wave, lane,
                                                     st_s, st_q, s_ss, s_bs, s_wx, xz, xyz, po, smask, n0, N, gin, lda);
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
      if (STATS && so.acc) seam_add(so, N, which, n0 + c, s, blockIdx.x);
      else stats_partial[((size_t)blockIdx.x * 2 + which) * N + n0 + c] = s;
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
    return cpfn_smallp_gemm_launch(a, lda, w, w_trans, P, K, N, (unsigned short *)Y, ldy, stats_partial, a_scale, a_shift, gx, st);
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
static int gemm_xyz_launch(const void *A, int lda, const void *W, const float *xyz, const float *Wx, long long P, int K,
                           int N, void *Y, int ldy, float *stats_partial, const cpfn_seam_out *seam_out, void *stream,
                           const int *gidx = nullptr, int rows_per_cloud = 0, int n_src = 0) {
  if (!cpfn_mlp_gemm_xyz_ok(P, K, N) || !A || !W || !xyz || !Wx || !Y || lda != K || ldy != N || !seam_out_valid(seam_out) ||
      (seam_out && stats_partial))
    return CPFN_EINVAL;
  if (gidx && (rows_per_cloud <= 0 || (rows_per_cloud % G_ROWS) || n_src <= 0 || P % rows_per_cloud ||
               (P / rows_per_cloud) * (long long)n_src * lda * 2 >= (1LL << 32)))
    return CPFN_EINVAL;
  GatherIn gin;
  gin.idx = gidx; gin.rows_per_cloud = rows_per_cloud; gin.n_src = n_src;
  hipStream_t st = (hipStream_t)stream;
  const int gx = cpfn_mlp_gemm_blocks(P, N);
  const long long tiles = (P + G_ROWS - 1) / G_ROWS;
  const int tpw = (int)((tiles + gx - 1) / gx);
  const dim3 grid(gx, N / 128);
  const unsigned short *a = (const unsigned short *)A, *w = (const unsigned short *)W;
  unsigned short *y = (unsigned short *)Y;
  if (stats_partial || seam_out)
    mlp_gemm_stream_kernel<128, 4, true, false, false, true><<<grid, G_THREADS, 0, st>>>(a, lda, w, 0, (int)P, N, y, ldy, stats_partial, tpw, nullptr, nullptr, nullptr, probe_slot(grid), xyz, Wx, seam_out_arg(seam_out), SeamIn(), PoolOut(), gin);
  else
    mlp_gemm_stream_kernel<128, 4, false, false, false, true><<<grid, G_THREADS, 0, st>>>(a, lda, w, 0, (int)P, N, y, ldy, nullptr, tpw, nullptr, nullptr, nullptr, probe_slot(grid), xyz, Wx, SeamOut(), SeamIn(), PoolOut(), gin);
  return cpfn_launch_status();
}
extern "C" int cpfn_mlp_gemm_xyz(const void *A, int lda, const void *W, const float *xyz, const float *Wx, long long P, int K,
                                 int N, void *Y, int ldy, float *stats_partial, void *stream) {
  return gemm_xyz_launch(A, lda, W, xyz, Wx, P, K, N, Y, ldy, stats_partial, nullptr, stream);
}
extern "C" int cpfn_mlp_gemm_xyz_seam(const void *A, const void *W, const float *xyz, const float *Wx, long long P, int K, int N,
                                      void *Y, const cpfn_seam_out *out, void *stream) {
  if (!out) return CPFN_EINVAL;
  return gemm_xyz_launch(A, K, W, xyz, Wx, P, K, N, Y, N, nullptr, out, stream);
}

// ---- forward layers with their BatchNorm seams spelled out (seam.h) -------------------------------------------------------
extern "C" int cpfn_seam_words(int replicas, int C) {
  return replicas >= 1 && replicas <= 8 && C > 0 ? replicas * 2 * C + 2 : -1;      // sums + the poison word (+ one of padding)
}

// which of cpfn_mlp_gemm's kernels a forward layer (bf16 rows, lda = K, ldy = N, no gather / bias) takes: 1 small-P, 2 streaming, 0 other
static int gemm_fwd_route(long long P, int K, int N, bool transform) {
  if (P <= 0 || K <= 0 || (K & 31) || N <= 0 || (N & 63) || P > 2000000000LL) return 0;
  if (P <= SP_MAX_ROWS && (N & 3) == 0 && (!transform || K <= SP_SS_MAX) && P * K * 2 < (1LL << 31) && (long long)N * K * 2 < (1LL << 31))
    return 1;
  if (gemm_stream_k(P, K) && (K & 7) == 0 && (!transform || K <= 128) && (P + G_ROWS) * (long long)(K > N ? K : N) * 2 < (1LL << 32))
    return 2;
  return 0;
}

extern "C" int cpfn_mlp_gemm_seam_ok(long long P, int K, int N) {
  return (gemm_fwd_route(P, K, N, false) ? 1 : 0) | (gemm_fwd_route(P, K, N, true) ? 2 : 0);
}

extern "C" int cpfn_mlp_gemm_seam(const void *A, const void *W, long long P, int K, int N, void *Y, float *stats_partial,
                                  const cpfn_seam_out *out, const cpfn_seam_in *in, const float *a_scale, const float *a_shift,
                                  void *stream) {
  if (!A || !W || !Y || (!stats_partial == !out) || (!a_scale != !a_shift) || (in && a_scale) || !seam_out_valid(out) ||
      !seam_in_valid(in, K))
    return CPFN_EINVAL;
  const bool transform = in || a_scale;
  const int route = gemm_fwd_route(P, K, N, transform);
  if (!route) return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int gx = cpfn_mlp_gemm_blocks(P, N);
  const unsigned short *a = (const unsigned short *)A, *w = (const unsigned short *)W;
  unsigned short *y = (unsigned short *)Y;
  if (route == 1) {
    if (in && in->replicas > 4) return CPFN_EINVAL;       // (the small-P kernel holds at most four replicas' words per lane)
    return cpfn_smallp_gemm_launch(a, K, w, 0, P, K, N, y, N, stats_partial, a_scale, a_shift, gx, st, out, in);
  }
  const long long tiles = (P + G_ROWS - 1) / G_ROWS;
  const int tpw = (int)((tiles + gx - 1) / gx);
  const SeamOut so = seam_out_arg(out);
  const SeamIn si = seam_in_arg(in);
#define CPFN_STREAM_SEAM(BN_, KS_)                                                                                     \
  do {                                                                                                                 \
    dim3 grid(gx, N / BN_);                                                                                            \
    if (transform)                                                                                                     \
      mlp_gemm_stream_kernel<BN_, (KS_ <= 4 ? KS_ : 4), true, true><<<grid, G_THREADS, 0, st>>>(a, K, w, 0, (int)P, N, y, N, stats_partial, tpw, a_scale, a_shift, nullptr, probe_slot(grid), nullptr, nullptr, so, si); \
    else                                                                                                               \
      mlp_gemm_stream_kernel<BN_, KS_, true><<<grid, G_THREADS, 0, st>>>(a, K, w, 0, (int)P, N, y, N, stats_partial, tpw, nullptr, nullptr, nullptr, probe_slot(grid), nullptr, nullptr, so, si); \
  } while (0)
  if (N % 128 == 0) {
    switch (K) { case 64: CPFN_STREAM_SEAM(128, 2); break; case 128: CPFN_STREAM_SEAM(128, 4); break;
                 case 192: CPFN_STREAM_SEAM(128, 6); break; default: CPFN_STREAM_SEAM(128, 8); }
  } else {
    switch (K) { case 64: CPFN_STREAM_SEAM(64, 2); break; case 128: CPFN_STREAM_SEAM(64, 4); break;
                 case 192: CPFN_STREAM_SEAM(64, 6); break; default: CPFN_STREAM_SEAM(64, 8); }
  }
#undef CPFN_STREAM_SEAM
  return cpfn_launch_status();
}

// ---- the pooled last layer of a set-abstraction stack: the same forward layer with the max over neighbours started in its
//      epilogue (stream_tile, POOL).  Streaming kernel with the operand transform only (the layer's input is a hidden layer's
//      pre-BN output), N % 128 == 0, pool_k in {32, 64, 128} dividing P.
extern "C" int cpfn_mlp_gemm_pool_ok(long long P, int K, int N, int pool_k) {
  return gemm_fwd_route(P, K, N, true) == 2 && (N % 128) == 0 && (K == 64 || K == 128) &&
         (pool_k == 32 || pool_k == 64 || pool_k == 128) && P % pool_k == 0;
}
extern "C" int cpfn_mlp_gemm_pool(const void *A, const void *W, long long P, int K, int N, void *Y, float *stats_partial,
                                  const cpfn_seam_out *out, const cpfn_seam_in *in, const float *a_scale, const float *a_shift,
                                  int pool_k, const float *gamma, void *pmax, unsigned char *pidx, void *stream) {
  if (!A || !W || !Y || (!stats_partial == !out) || (!a_scale != !a_shift) || (!in == !a_scale) || !seam_out_valid(out) ||
      !seam_in_valid(in, K) || !gamma || !pmax || !pidx || !cpfn_mlp_gemm_pool_ok(P, K, N, pool_k))
    return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int gx = cpfn_mlp_gemm_blocks(P, N);
  const long long tiles = (P + G_ROWS - 1) / G_ROWS;
  const int tpw = (int)((tiles + gx - 1) / gx);
  const unsigned short *a = (const unsigned short *)A, *w = (const unsigned short *)W;
  unsigned short *y = (unsigned short *)Y;
  PoolOut po;
  po.pmax = (unsigned short *)pmax; po.pidx = pidx; po.gamma = gamma; po.pool_k = pool_k;
  const SeamOut so = seam_out_arg(out);
  const SeamIn si = seam_in_arg(in);
  dim3 grid(gx, N / 128);
  if (K == 64)
    mlp_gemm_stream_kernel<128, 2, true, true, false, false, true><<<grid, G_THREADS, 0, st>>>(a, K, w, 0, (int)P, N, y, N, stats_partial, tpw, a_scale, a_shift, nullptr, probe_slot(grid), nullptr, nullptr, so, si, po);
  else
    mlp_gemm_stream_kernel<128, 4, true, true, false, false, true><<<grid, G_THREADS, 0, st>>>(a, K, w, 0, (int)P, N, y, N, stats_partial, tpw, a_scale, a_shift, nullptr, probe_slot(grid), nullptr, nullptr, so, si, po);
  return cpfn_launch_status();
}

// cpfn_mlp_gemm_xyz(_seam) with the [P, K] operand GATHERED from a per-cloud table while loading (stream_tile, GatherIn):
// table [P / rows_per_cloud][n_src][K] bf16, gidx [P] int32 (row inside the cloud's table).  Exactly one of stats_partial / out.
extern "C" int cpfn_mlp_gemm_xyz_gather(const void *table, const int *gidx, int rows_per_cloud, int n_src, const void *W,
                                        const float *xyz, const float *Wx, long long P, int K, int N, void *Y, float *stats_partial,
                                        const cpfn_seam_out *out, void *stream) {
  if (!gidx || (!stats_partial == !out)) return CPFN_EINVAL;
  return gemm_xyz_launch(table, K, W, xyz, Wx, P, K, N, Y, N, stats_partial, out, stream, gidx, rows_per_cloud, n_src);
}
