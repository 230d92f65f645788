// Fused loss-side kernels of the SPFN training step for gfx950 (the callers
// on the far side of the fitters).  The reference evaluates these as ~500 small framework ops per
// step (SPFN/losses_implementation.py); each group below is one streaming pass.
//
//  head_post   : heads [P,3+4+K] -> unit normals, soft-max memberships, and the per-cloud normal and
//                type losses (Utils/training_utils.py:141-142, losses_implementation.py:152-159, 195-210)
//  seg_stats   : label-segmented sums of the memberships, the one quantity behind both the Hungarian
//                cost matrix and the relaxed-IoU loss (losses_implementation.py:19-24, 77-90)
//  residue     : mean residue of the GT points of every instance against the matched prediction of the
//                instance's GT type, plus the axis-agreement loss (losses_implementation.py:351-387,
//                480-497; SPFN/*_fitter.compute_residue_single)
//
// All reductions use per-block partials summed in a fixed order (no float atomics in forward passes).
#include "common.h"
#include "lsap.h"
#include "residue.h"

namespace {

constexpr int MAXK = LSAP_MAXK;      // instances per cloud supported by the fused kernels (28 global / 21 local)
constexpr int LP_THREADS = 256;

// ------------------------------------------------------------------------------------ head_post
// Y[P, 7+K] fp32: cols 0-2 normal, 3-6 type logits, 7.. membership logits.
// partial[b][chunk][3] = Σ (1-|x̂·x_gt|), Σ CE·[I!=-1], Σ [I!=-1]
//
// One lane per point, but the lanes never touch global memory row by row (35 floats at a 140-byte stride: 64
// cache lines per load instruction, ~1 TB/s): the workgroup's 256 rows are contiguous, so they are copied
// into LDS with fully coalesced loads, each lane then owns LDS row t (odd row stride: conflict-free), and the
// result rows go back the same way.
constexpr int LP_LD = 7 + MAXK;   // LDS row stride for the [*, 7+K] rows: 39 floats, odd
#define lp_stage_in cpfn_rows_to_lds<LP_THREADS>
#define lp_stage_out cpfn_rows_from_lds<LP_THREADS>

__global__ __launch_bounds__(LP_THREADS) void head_post_fwd_kernel(
    const float *__restrict__ Y, const float *__restrict__ Xgt, const long long *__restrict__ Igt,
    const long long *__restrict__ Tgt, int N, int K, float *__restrict__ Xn, float *__restrict__ Wsm,
    float *__restrict__ partial, float *__restrict__ seg_partial /* optional: [B][chunks][(K+2)*K] */,
    int *__restrict__ lab_partial /* optional: [B][chunks] largest label of the chunk (-1: none) */) {
  __shared__ float s_row[LP_THREADS * LP_LD];
  __shared__ float s_x[LP_THREADS * 3];
  __shared__ float s_red[LP_THREADS / 64][3];
  __shared__ int s_lmax[LP_THREADS / 64];
  __shared__ float s_seg[LP_THREADS / 64][MAXK][MAXK + 1];     // per-wave segmented sums (seg_partial only)
  __shared__ int s_cnt[MAXK];
  const int b = blockIdx.y, t = threadIdx.x, C = 7 + K;
  const int n0 = blockIdx.x * LP_THREADS;
  const int rows = min(LP_THREADS, N - n0);
  const size_t p0 = (size_t)b * N + n0;
  lp_stage_in(s_row, LP_LD, Y + p0 * C, rows, C, t);
  lp_stage_in(s_x, 3, Xgt + p0 * 3, rows, 3, t);
  __syncthreads();
  float l_n = 0.f, l_t = 0.f, l_c = 0.f;
  float e[MAXK], u0 = 0.f, u1 = 0.f, u2 = 0.f, is = 0.f;
  int seg_lab = -2;                       // this lane's label (-1 unlabelled, -2 past the end of the cloud)
  if (t < MAXK) s_cnt[t] = 0;
  if (t < rows) {
    const float *y = s_row + t * LP_LD;
    const float x0 = y[0], x1 = y[1], x2 = y[2];
    const float inv = 1.0f / fmaxf(sqrtf(x0 * x0 + x1 * x1 + x2 * x2), 1e-12f);   // F.normalize(eps=1e-12)
    u0 = x0 * inv; u1 = x1 * inv; u2 = x2 * inv;
    l_n = 1.0f - fabsf(u0 * s_x[t * 3] + u1 * s_x[t * 3 + 1] + u2 * s_x[t * 3 + 2]);
    // soft-max over the K membership logits
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, y[7 + k]);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) { e[k] = k < K ? __expf(y[7 + (k < K ? k : 0)] - m) : 0.f; s += e[k]; }
    is = 1.0f / s;
    // per-point type cross-entropy against the type of the point's GT instance
    const long long lab = Igt[p0 + t];
    seg_lab = (int)lab;
    if (lab != -1) {
      const long long tgt = Tgt[(size_t)b * K + (lab < 0 ? 0 : lab)];
      const float t0 = y[3], t1 = y[4], t2 = y[5], t3 = y[6];
      const float tm = fmaxf(fmaxf(t0, t1), fmaxf(t2, t3));
      const float lse = tm + logf(expf(t0 - tm) + expf(t1 - tm) + expf(t2 - tm) + expf(t3 - tm));
      l_t = lse - y[3 + tgt];
      l_c = 1.f;
    }
  }
  __syncthreads();                       // every lane has read its row: reuse the slab for the outputs
  if (t < rows) {
    float *w = s_row + t * LP_LD;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < K) w[k] = e[k] * is;
    s_x[t * 3] = u0; s_x[t * 3 + 1] = u1; s_x[t * 3 + 2] = u2;
  }
  // block reduce (wave shuffles, then 4 waves)
  for (int msk = 32; msk >= 1; msk >>= 1) {
    l_n += __shfl_xor(l_n, msk, 64); l_t += __shfl_xor(l_t, msk, 64); l_c += __shfl_xor(l_c, msk, 64);
  }
  int lmax = seg_lab < 0 ? -1 : seg_lab;
  for (int msk = 32; msk >= 1; msk >>= 1) lmax = max(lmax, __shfl_xor(lmax, msk, 64));
  if ((t & 63) == 0) { s_red[t >> 6][0] = l_n; s_red[t >> 6][1] = l_t; s_red[t >> 6][2] = l_c; s_lmax[t >> 6] = lmax; }
  if (seg_partial && seg_lab >= 0 && seg_lab < MAXK) atomicAdd(&s_cnt[seg_lab], 1);   // integer: exact, order-free
  __syncthreads();
  if (seg_partial) {
    // Label-segmented sums of the memberships (what seg_stats_fwd_kernel computes: rows l < K: Σ_{I=l} w, row K: Σ w,
    // row K+1: label counts) taken from the soft-max rows while they are in LDS — the separate pass re-read all of W
    // and cost 25-34 us.  Per wave a [32 labels+1] x [32 columns] fp32 MFMA contraction over its 64 points, two points
    // per v_mfma_f32_32x32x2_f32: A = one-hot(label) with an all-ones row K, B = the membership row.
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    const int lane = t & 63, wave = t >> 6, x = lane & 31, hk = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll 8
    for (int tt = 0; tt < 32; ++tt) {
      const int r = 2 * tt + hk, row = wave * 64 + r;
      const int lab_r = __shfl(seg_lab, r, 64);                         // lane r of this wave owns that row
      const float a = (x < K) ? (lab_r == x ? 1.f : 0.f) : ((x == K && lab_r != -2) ? 1.f : 0.f);
      const float bv = (x < K && lab_r != -2) ? s_row[row * LP_LD + x] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
    }
    // D[i][j]: j = lane & 31, i = 8 (v / 4) + 4 (lane >> 5) + (v % 4)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int i = 8 * (v >> 2) + 4 * hk + (v & 3);
      s_seg[wave][i][x] = acc[v];
    }
    __syncthreads();
    float *o = seg_partial + ((size_t)b * gridDim.x + blockIdx.x) * (K + 2) * K;
    for (int e2 = t; e2 < (K + 1) * K; e2 += LP_THREADS) {
      const int l = e2 / K, k = e2 - l * K;
      o[e2] = ((s_seg[0][l][k] + s_seg[1][l][k]) + s_seg[2][l][k]) + s_seg[3][l][k];
    }
    if (t < K) o[(K + 1) * K + t] = (float)s_cnt[t];
  }
  lp_stage_out(s_row, LP_LD, Wsm + p0 * K, rows, K, t);
  lp_stage_out(s_x, 3, Xn + p0 * 3, rows, 3, t);
  if (t < 3) {
    float s = 0.f;
    for (int w = 0; w < LP_THREADS / 64; ++w) s += s_red[w][t];
    partial[((size_t)b * gridDim.x + blockIdx.x) * 3 + t] = s;
  }
  if (lab_partial && t == 0) {
    int m = s_lmax[0];
    for (int w = 1; w < LP_THREADS / 64; ++w) m = max(m, s_lmax[w]);
    lab_partial[(size_t)b * gridDim.x + blockIdx.x] = m;
  }
}

// out[b] = (normal_loss, type_loss, count): fixed-order sum over chunks — and in the same launch the chunk sums of the
// segmented membership sums (chunk_sum_f32_kernel's job) and the
// number of GT instances per cloud (largest label + 1: cpfn_count_labels' job): three [B]- / [B,K+2,K]-sized launches
// after the heads pass became one.  Lanes e < total: S[e]; lanes total .. total + B - 1: one cloud's losses and n_gt.
__global__ void head_post_finish_kernel(const float *__restrict__ partial, const float *__restrict__ seg_partial,
                                        const int *__restrict__ lab_partial, int chunks, int N, int B, int per_b,
                                        long long total, float *__restrict__ out, float *__restrict__ S,
                                        long long *__restrict__ n_gt) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < total) {
    const long long b = e / per_b, r = e % per_b;
    float s = 0.f;
#pragma unroll 32
    for (int c = 0; c < chunks; ++c) s += seg_partial[((size_t)b * chunks + c) * per_b + r];
    S[e] = s;
    return;
  }
  const int b = (int)(e - total);
  if (b >= B) return;
  double a = 0, c = 0, d = 0;
  int lm = -1;
#pragma unroll 8
  for (int i = 0; i < chunks; ++i) {
    a += partial[((size_t)b * chunks + i) * 3];
    c += partial[((size_t)b * chunks + i) * 3 + 1];
    d += partial[((size_t)b * chunks + i) * 3 + 2];
    if (lab_partial) lm = max(lm, lab_partial[(size_t)b * chunks + i]);
  }
  out[b * 3] = (float)(a / N);
  out[b * 3 + 1] = (float)(c / d);   // 0/0 -> NaN for a cloud with no labelled point, like the reference
  out[b * 3 + 2] = (float)d;
  if (n_gt) n_gt[b] = (long long)lm + 1;
}

// gY[P,7+K] from: gXn[P,3] (may be null), gW[P,K] (may be null), gloss[B,2] = dL/d(normal_loss, type_loss)
// Same LDS-staged row movement as the forward kernel; the slab is reused for Y, W, gW in turn and for gY.
__global__ __launch_bounds__(LP_THREADS) void head_post_bwd_kernel(
    const float *__restrict__ Y, const float *__restrict__ Xgt, const long long *__restrict__ Igt,
    const long long *__restrict__ Tgt, const float *__restrict__ Wsm, const float *__restrict__ stats,
    const float *__restrict__ gXn, const float *__restrict__ gW, const float *__restrict__ gloss, int gl_planar, int N,
    int K, float *__restrict__ gY, const float *__restrict__ gS, unsigned short *__restrict__ pad_bf16 = nullptr,
    float *__restrict__ colsum_partial = nullptr) {
  // pad_bf16 [B*N, 64] + colsum_partial [B*N/256][7+K] (optional; N % 256 == 0): what cpfn_colsum_f32 makes of gY for the fc2 heads'
  // backward — the rows converted to bf16 and zero-padded to 64 columns (the gradient operand of their two GEMMs) and the
  // per-256-row column sums (their bias gradient, finished by the batched split reduction) — taken from the tile while it is
  // in LDS: that launch (11 us on the step's chain, a second pass over gY) is then not made.  Same order of additions.
  // gS (optional) [B, K+2, K]: gradient w.r.t. the label-segmented membership sums the forward launch produced; its
  // adjoint dW[n,k] = gS[K,k] + gS[label(n),k] (seg_stats_bwd_kernel) is added to gW here, so that neither that kernel
  // nor the framework's gradient-accumulation add of the two [B,N,K] tensors is launched.
  __shared__ float s_row[LP_THREADS * LP_LD];
  __shared__ float s_x[LP_THREADS * 3], s_gx[LP_THREADS * 3];
  __shared__ float s_gs[(MAXK + 1) * MAXK];
  if (gS)
    for (int e = threadIdx.x; e < (K + 1) * K; e += LP_THREADS) s_gs[e] = gS[(size_t)blockIdx.y * (K + 2) * K + e];
  const int b = blockIdx.y, t = threadIdx.x, C = 7 + K;
  const int n0 = blockIdx.x * LP_THREADS;
  const int rows = min(LP_THREADS, N - n0);
  const size_t p0 = (size_t)b * N + n0;
  const bool live = t < rows;
  lp_stage_in(s_row, LP_LD, Y + p0 * C, rows, C, t);
  lp_stage_in(s_x, 3, Xgt + p0 * 3, rows, 3, t);
  if (gXn) lp_stage_in(s_gx, 3, gXn + p0 * 3, rows, 3, t);
  __syncthreads();
  float y7[7] = {0, 0, 0, 0, 0, 0, 0}, sm[MAXK], o7[7] = {0, 0, 0, 0, 0, 0, 0};
  if (live)
    for (int j = 0; j < 7; ++j) y7[j] = s_row[t * LP_LD + j];
  __syncthreads();
  if (gW || gS) {                                        // soft-max adjoint needs W and gW rows
    lp_stage_in(s_row, LP_LD, Wsm + p0 * K, rows, K, t);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXK; ++k) sm[k] = (live && k < K) ? s_row[t * LP_LD + k] : 0.f;
    __syncthreads();
    if (gW) {
      lp_stage_in(s_row, LP_LD, gW + p0 * K, rows, K, t);
      __syncthreads();
    }
  }
  if (live) {
    // normals
    const float x0 = y7[0], x1 = y7[1], x2 = y7[2];
    const float nrm = sqrtf(x0 * x0 + x1 * x1 + x2 * x2);
    const float inv = 1.0f / fmaxf(nrm, 1e-12f);
    const float u0 = x0 * inv, u1 = x1 * inv, u2 = x2 * inv;
    const float g0 = s_x[t * 3], g1 = s_x[t * 3 + 1], g2 = s_x[t * 3 + 2];
    const float d = u0 * g0 + u1 * g1 + u2 * g2;
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    const float cn = -gloss[gl_planar ? b : b * 2] * sg / (float)N;
    float h0 = cn * g0, h1 = cn * g1, h2 = cn * g2;
    if (gXn) { h0 += s_gx[t * 3]; h1 += s_gx[t * 3 + 1]; h2 += s_gx[t * 3 + 2]; }
    if (nrm >= 1e-12f) {
      const float pr = u0 * h0 + u1 * h1 + u2 * h2;
      o7[0] = (h0 - u0 * pr) * inv; o7[1] = (h1 - u1 * pr) * inv; o7[2] = (h2 - u2 * pr) * inv;
    } else {
      o7[0] = h0 * inv; o7[1] = h1 * inv; o7[2] = h2 * inv;
    }
    // type logits
    const long long lab = Igt[p0 + t];
    if (lab != -1) {
      const long long tgt = Tgt[(size_t)b * K + (lab < 0 ? 0 : lab)];
      const float t0 = y7[3], t1 = y7[4], t2 = y7[5], t3 = y7[6];
      const float tm = fmaxf(fmaxf(t0, t1), fmaxf(t2, t3));
      const float e0 = expf(t0 - tm), e1 = expf(t1 - tm), e2 = expf(t2 - tm), e3 = expf(t3 - tm);
      const float is = 1.0f / (e0 + e1 + e2 + e3);
      const float c = gloss[gl_planar ? gridDim.y + b : b * 2 + 1] / stats[b * 3 + 2];
      o7[3] = c * (e0 * is - (tgt == 0 ? 1.f : 0.f));
      o7[4] = c * (e1 * is - (tgt == 1 ? 1.f : 0.f));
      o7[5] = c * (e2 * is - (tgt == 2 ? 1.f : 0.f));
      o7[6] = c * (e3 * is - (tgt == 3 ? 1.f : 0.f));
    }
  }
  // memberships: soft-max adjoint, written over the gW row in place (columns shift by 7)
  float om[MAXK];
  if (gW || gS) {
    float dot = 0.f;
    const long long labw = live ? Igt[p0 + t] : -1;
    const bool hasl = labw >= 0 && labw < K;
    const float *gl = s_gs + (hasl ? (int)labw : 0) * K, *ga = s_gs + K * K;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      om[k] = (gW && live && k < K) ? s_row[t * LP_LD + k] : 0.f;
      if (gS && live && k < K) {
        const float dseg = ga[k] + (hasl ? gl[k] : 0.f);       // (seg_stats_bwd_kernel's value, then the accumulation add)
        om[k] = gW ? om[k] + dseg : dseg;
      }
      dot = fmaf(om[k], sm[k], dot);
    }
#pragma unroll
    for (int k = 0; k < MAXK; ++k) om[k] = sm[k] * (om[k] - dot);
  } else {
#pragma unroll
    for (int k = 0; k < MAXK; ++k) om[k] = 0.f;
  }
  __syncthreads();
  if (live) {
    float *o = s_row + t * LP_LD;
    for (int j = 0; j < 7; ++j) o[j] = o7[j];
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < K) o[7 + k] = om[k];
  }
  __syncthreads();
  lp_stage_out(s_row, LP_LD, gY + p0 * C, rows, C, t);
  if (pad_bf16) {
    // 16-byte chunk e of the tile's rows x 8 chunks: row e / 8, columns 8 (e % 8) .. + 7 (zeros beyond C): coalesced
    for (int e = t; e < rows * 8; e += LP_THREADS) {
      const int r = e >> 3, c0 = (e & 7) * 8;
      unsigned w4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = c0 + 2 * j < C ? s_row[r * LP_LD + c0 + 2 * j] : 0.f;
        const float hi = c0 + 2 * j + 1 < C ? s_row[r * LP_LD + c0 + 2 * j + 1] : 0.f;
        w4[j] = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)lo) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)hi) << 16);
      }
      *(uint4 *)(pad_bf16 + (p0 + r) * 64 + c0) = (uint4){w4[0], w4[1], w4[2], w4[3]};
    }
  }
  if (colsum_partial) {
    // colsum_f32_kernel's order: lane (column c, subset rs) adds the rows = rs (mod 4) in ascending order, then ((s0 + s1) + s2) + s3
    __shared__ float s_cs[4][64];
    const int c = t & 63, rs = t >> 6;
    float a = 0.f;
    if (c < C)
      for (int r = rs; r < rows; r += 4) a += s_row[r * LP_LD + c];
    s_cs[rs][c] = a;
    __syncthreads();
    if (t < C) colsum_partial[((size_t)b * gridDim.x + blockIdx.x) * C + t] = s_cs[0][t] + s_cs[1][t] + s_cs[2][t] + s_cs[3][t];
  }
}

// ------------------------------------------------------------------------------------ seg_stats
// S[b][K+2][K]: rows l<K: Σ_{n: I=l} W[n,:];  row K: Σ_n W[n,:];  row K+1: #points with label l (as float).
// thread = (column kk, point subset); 32 label accumulators per thread, selected by compare (no atomics).
constexpr int SS_TILE = 256, SS_SUB = LP_THREADS / MAXK;   // 8 subsets; a tile = one latency: the whole 256-point chunk
// WIDE (K > 32, evaluation only: merged label sets of evaluation_localSPFN.py:129-131 have no upper bound): blockIdx.z
// enumerates (label tile, column tile) pairs of 32 x 32 and the same code accumulates one tile of S per workgroup.
template <bool WIDE>
__global__ __launch_bounds__(LP_THREADS) void seg_stats_fwd_kernel(const float *__restrict__ W,
                                                                   const long long *__restrict__ Igt, int N, int K,
                                                                   int pts_per_block, float *__restrict__ partial) {
  __shared__ float s_w[SS_TILE][MAXK];
  __shared__ int s_lab[SS_TILE];
  // accumulators [point sub-lane][label row (row MAXK = all points)][column], rows padded to 33 floats.  A lane owns
  // column kk of its sub-lane's slab, so `slab[label][kk] += w` is a private read-modify-write in LDS (DS ops of a
  // wave execute in order) — one ds_read + add + ds_write per point instead of 32 compare-select-adds into 32
  // register accumulators (26 us -> a few us: that loop was the kernel).
  __shared__ float s_acc[SS_SUB][MAXK + 1][MAXK + 1];
  __shared__ int s_cnt[MAXK];                       // points per label: integer LDS atomics (exact, order-free)
  const int b = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
  const int kt_n = WIDE ? (K + MAXK - 1) / MAXK : 1;
  const int l0 = WIDE ? (int)(blockIdx.z / kt_n) * MAXK : 0, k0 = WIDE ? (int)(blockIdx.z % kt_n) * MAXK : 0;
  const int kk = t % MAXK, sub = t / MAXK;
  const int n0 = chunk * pts_per_block, n1 = min(N, n0 + pts_per_block);
  float all = 0.f;
  for (int l = 0; l <= MAXK; ++l) s_acc[sub][l][kk] = 0.f;
  if (t < MAXK) s_cnt[t] = 0;
  for (int base = n0; base < n1; base += SS_TILE) {
    __syncthreads();
    for (int e = t; e < SS_TILE * MAXK; e += LP_THREADS) {
      const int i = e / MAXK, k = e % MAXK;
      s_w[i][k] = (base + i < n1 && k0 + k < K) ? W[((size_t)b * N + base + i) * K + k0 + k] : 0.f;
    }
    if (t < SS_TILE) {
      const int lab = (base + t < n1) ? (int)Igt[(size_t)b * N + base + t] - l0 : -2;
      s_lab[t] = lab;
      if (lab >= 0 && lab < MAXK) atomicAdd(&s_cnt[lab], 1);
    }
    __syncthreads();
    for (int i = sub; i < SS_TILE; i += SS_SUB) {
      const int lab = s_lab[i];
      const float w = s_w[i][kk];
      all += w;
      if (lab >= 0 && lab < MAXK) s_acc[sub][lab][kk] += w;
    }
  }
  s_acc[sub][MAXK][kk] = all;
  __syncthreads();
  float *o = partial + ((size_t)b * gridDim.x + chunk) * (K + 2) * K;
  if (!WIDE) {
    for (int e = t; e < (K + 1) * K; e += LP_THREADS) {
      const int l = e / K, k = e % K;
      const int row = l < K ? l : MAXK;
      float s = 0.f;
      for (int q = 0; q < SS_SUB; ++q) s += s_acc[q][row][k];
      o[l * K + k] = s;
    }
    if (t < K) o[(K + 1) * K + t] = (float)s_cnt[t];
  } else {
    const int kw = min(MAXK, K - k0), lw = min(MAXK, K - l0);
    for (int e = t; e < (MAXK + 1) * MAXK; e += LP_THREADS) {
      const int l = e / MAXK, k = e % MAXK;          // l == MAXK: the all-points row (written by label tile 0 only)
      if (k >= kw || (l < MAXK ? l >= lw : l0 != 0)) continue;
      float s = 0.f;
      for (int q = 0; q < SS_SUB; ++q) s += s_acc[q][l][k];
      o[(l < MAXK ? l0 + l : K) * K + k0 + k] = s;
    }
    if (k0 == 0 && t < lw) o[(K + 1) * K + l0 + t] = (float)s_cnt[t];
  }
}

__global__ void chunk_sum_f32_kernel(const float *__restrict__ partial, int chunks, int per_b, long long total,
                                     float *__restrict__ out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const long long b = e / per_b, r = e % per_b;
  float s = 0.f;
#pragma unroll 8
  for (int c = 0; c < chunks; ++c) s += partial[((size_t)b * chunks + c) * per_b + r];
  out[e] = s;
}

// dW[b,n,k] = gS[b, I[b,n], k] (labelled points) + gS[b, K, k]
// One lane per point builds its row in LDS (odd stride), the tile leaves with coalesced stores
// (the first version: one lane per element with two integer divisions by K and a redundant label load, 0.7 TB/s).
__global__ __launch_bounds__(LP_THREADS) void seg_stats_bwd_kernel(const float *__restrict__ gS,
                                                                   const long long *__restrict__ Igt, int N, int K,
                                                                   float *__restrict__ dW) {
  __shared__ float s_g[(MAXK + 2) * MAXK];
  __shared__ float s_row[LP_THREADS * (MAXK + 1)];
  const int b = blockIdx.y, t = threadIdx.x;
  for (int e = t; e < (K + 1) * K; e += LP_THREADS) s_g[e] = gS[(size_t)b * (K + 2) * K + e];
  __syncthreads();
  const int n0 = blockIdx.x * LP_THREADS, rows = min(LP_THREADS, N - n0);
  if (t < rows) {
    const long long lab = Igt[(size_t)b * N + n0 + t];
    const bool has = lab >= 0 && lab < K;
    const float *gl = s_g + (has ? (int)lab : 0) * K, *ga = s_g + K * K;
    float *o = s_row + t * (MAXK + 1);
    for (int k = 0; k < K; ++k) o[k] = ga[k] + (has ? gl[k] : 0.f);
  }
  __syncthreads();
  cpfn_rows_from_lds<LP_THREADS>(s_row, MAXK + 1, dW + ((size_t)b * N + n0) * K, rows, K, t);
}

// ------------------------------------------------------------------------------------ residue
// params[b][k][22]: plane n(0-2) c(3) | sphere c(4-6) r2(7) | cylinder a(8-10) c(11-13) r2(14) |
// cone apex(15-17) axis(18-20) half(21).   type ids: tid[4] = id of (plane, sphere, cylinder, cone).
struct D7 {   // value + 7 tangents (one per parameter of the selected primitive type)
  float v, d[7];
};
__device__ inline D7 mk(float v) { D7 r; r.v = v; for (int i = 0; i < 7; ++i) r.d[i] = 0.f; return r; }
__device__ inline D7 var(float v, int i) { D7 r = mk(v); r.d[i] = 1.f; return r; }
__device__ inline D7 operator+(D7 a, D7 b) { D7 r; r.v = a.v + b.v; for (int i = 0; i < 7; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ inline D7 operator-(D7 a, D7 b) { D7 r; r.v = a.v - b.v; for (int i = 0; i < 7; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ inline D7 operator*(D7 a, D7 b) { D7 r; r.v = a.v * b.v; for (int i = 0; i < 7; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ inline D7 scale(D7 a, float s) { D7 r; r.v = a.v * s; for (int i = 0; i < 7; ++i) r.d[i] = a.d[i] * s; return r; }
__device__ inline D7 chain(D7 a, float f, float df) { D7 r; r.v = f; for (int i = 0; i < 7; ++i) r.d[i] = a.d[i] * df; return r; }
__device__ inline D7 sqrt_safe(D7 a) {   // sqrt(|x| + 1e-10)
  const float s = sqrtf(fabsf(a.v) + 1e-10f);
  const float sg = a.v > 0.f ? 1.f : (a.v < 0.f ? -1.f : 0.f);
  return chain(a, s, sg * 0.5f / s);
}

__device__ inline D7 residue_of(int kind, const float *q, float px, float py, float pz) {
  if (kind == 0) {          // plane: (p·n − c)²            params n(0-2), c(3)
    D7 e = scale(var(q[0], 0), px) + scale(var(q[1], 1), py) + scale(var(q[2], 2), pz) - var(q[3], 3);
    return e * e;
  } else if (kind == 1) {   // sphere: (‖p − c‖ − r)²       params c(0-2), r2(3)
    D7 dx = mk(px) - var(q[0], 0), dy = mk(py) - var(q[1], 1), dz = mk(pz) - var(q[2], 2);
    D7 e = sqrt_safe(dx * dx + dy * dy + dz * dz) - sqrt_safe(var(q[3], 3));
    return e * e;
  } else if (kind == 2) {   // cylinder: (dist to axis − r)² params a(0-2), c(3-5), r2(6)
    D7 dx = mk(px) - var(q[3], 3), dy = mk(py) - var(q[4], 4), dz = mk(pz) - var(q[5], 5);
    D7 al = dx * var(q[0], 0) + dy * var(q[1], 1) + dz * var(q[2], 2);
    D7 e = sqrt_safe(dx * dx + dy * dy + dz * dz - al * al) - sqrt_safe(var(q[6], 6));
    return e * e;
  } else {                  // cone                          params apex(0-2), axis(3-5), half(6)
    D7 vx = mk(px) - var(q[0], 0), vy = mk(py) - var(q[1], 1), vz = mk(pz) - var(q[2], 2);
    D7 n2 = vx * vx + vy * vy + vz * vz;
    const float nrm = sqrtf(n2.v);
    D7 inv;                                    // 1 / max(‖v‖, 1e-12)  (F.normalize)
    if (nrm >= 1e-12f) inv = chain(n2, 1.f / nrm, -0.5f / (nrm * n2.v));
    else inv = mk(1e12f);
    D7 c = (vx * var(q[3], 3) + vy * var(q[4], 4) + vz * var(q[5], 5)) * inv;
    const float lim = 1.0f - 1e-6f;
    D7 alpha;                                  // acos_safe: clamp to [-1+1e-6, 1-1e-6] (zero slope when clamped)
    if (c.v > lim) alpha = mk(acosf(lim));
    else if (c.v < -lim) alpha = mk(acosf(-lim));
    else alpha = chain(c, acosf(c.v), -rsqrtf(1.f - c.v * c.v));
    D7 diff = alpha - var(q[6], 6);
    const float sg = diff.v > 0.f ? 1.f : (diff.v < 0.f ? -1.f : 0.f);
    D7 ad = chain(diff, fabsf(diff.v), sg);
    D7 sn;                                     // sin(min(|·|, π/2))
    if (ad.v > 1.57079632679f) sn = mk(1.f);
    else sn = chain(ad, sinf(ad.v), cosf(ad.v));
    return sn * sn * n2;
  }
}

// one workgroup per GT instance (b,k).  out[b][k][0] = mean residue, [1] = axis loss;
// dout[b][k][0..6] = d residue / d(params of the type), [7..9] = d axis-loss / d(axis or normal)
__global__ __launch_bounds__(128) void residue_fwd_kernel(const float *__restrict__ params,
                                                          const long long *__restrict__ match,
                                                          const long long *__restrict__ Tgt,
                                                          const float *__restrict__ pts, const float *__restrict__ gt_axes,
                                                          int K, int NP, int tid_plane, int tid_sphere, int tid_cyl,
                                                          int tid_cone, float *__restrict__ out, float *__restrict__ dout) {
  __shared__ __attribute__((aligned(16))) float s_red[2][8];
  const int bk = blockIdx.x, b = bk / K, t = threadIdx.x;
  const long long m = match[bk], ty = Tgt[bk];
  const int kind = ty == tid_plane ? 0 : (ty == tid_sphere ? 1 : (ty == tid_cyl ? 2 : 3));
  const float *P22 = params + ((size_t)b * K + m) * 22;
  float q[7] = {0, 0, 0, 0, 0, 0, 0};
  const int off = kind == 0 ? 0 : (kind == 1 ? 4 : (kind == 2 ? 8 : 15));
  const int nq = kind == 0 ? 4 : (kind == 1 ? 4 : 7);
  for (int i = 0; i < nq; ++i) q[i] = P22[off + i];
  D7 acc = mk(0.f);
  const float *pp = pts + (size_t)bk * NP * 3;
  for (int i = t; i < NP; i += 128) acc = acc + residue_of(kind, q, pp[3 * i], pp[3 * i + 1], pp[3 * i + 2]);
  float vals[8] = {acc.v, acc.d[0], acc.d[1], acc.d[2], acc.d[3], acc.d[4], acc.d[5], acc.d[6]};
  for (int j = 0; j < 8; ++j)
    for (int msk = 32; msk >= 1; msk >>= 1) vals[j] += __shfl_xor(vals[j], msk, 64);
  if ((t & 63) == 0)
    for (int j = 0; j < 8; ++j) s_red[t >> 6][j] = vals[j];
  __syncthreads();
  if (t == 0) {
    const float invn = 1.0f / (float)NP;
    float r[8];   // both rows through un-narrowable 16-byte reads (cpfn_lds_read4: no ds_read_b96)
    for (int h = 0; h < 2; ++h) {
      const cpfn_f32x4 u = cpfn_lds_read4(&s_red[0][4 * h]), v = cpfn_lds_read4(&s_red[1][4 * h]);
      r[4 * h] = u.x + v.x; r[4 * h + 1] = u.y + v.y; r[4 * h + 2] = u.z + v.z; r[4 * h + 3] = u.w + v.w;
    }
    out[bk * 2] = r[0] * invn;
    for (int j = 0; j < 7; ++j) dout[bk * 10 + j] = r[1 + j] * invn;
    // axis agreement 1 − |a_pred·a_gt| (plane normal / cylinder axis / cone axis; 0 for spheres)
    float pl = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f;
    if (kind != 1) {
      const int ao = kind == 0 ? 0 : (kind == 2 ? 8 : 18);
      const float *ga = gt_axes + ((size_t)(kind == 0 ? 0 : (kind == 2 ? 1 : 2)) * gridDim.x + bk) * 3;
      const float dt = P22[ao] * ga[0] + P22[ao + 1] * ga[1] + P22[ao + 2] * ga[2];
      const float sg = dt > 0.f ? 1.f : (dt < 0.f ? -1.f : 0.f);
      pl = 1.0f - fabsf(dt);
      d0 = -sg * ga[0]; d1 = -sg * ga[1]; d2 = -sg * ga[2];
    }
    out[bk * 2 + 1] = pl;
    dout[bk * 10 + 7] = d0; dout[bk * 10 + 8] = d1; dout[bk * 10 + 9] = d2;
  }
}

// gparams[b][m][slot] = sum over the GT instances k assigned to prediction m (ascending k) of
//   g_res . d residue / d slot + g_par . d axis-loss / d slot.
// One lane per (b, m, slot): every element of gparams is WRITTEN (zeros where nothing arrives), so the caller needs no
// zero fill, and the sum has a fixed order (was: one lane per (b, k) scattering with float atomics into a zeroed tensor).
__global__ void residue_bwd_kernel(const float *__restrict__ gout, const float *__restrict__ dout,
                                   const long long *__restrict__ match, const long long *__restrict__ Tgt, int K,
                                   int BK, int tid_plane, int tid_sphere, int tid_cyl, int tid_cone,
                                   float *__restrict__ gparams) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= BK * 22) return;
  const int slot = e % 22, bm = e / 22, b = bm / K, m = bm - b * K;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    const int bk = b * K + k;
    if (match[bk] != m) continue;
    const long long ty = Tgt[bk];
    const int kind = ty == tid_plane ? 0 : (ty == tid_sphere ? 1 : (ty == tid_cyl ? 2 : 3));
    const float gr = gout[bk * 2], ga = gout[bk * 2 + 1];
    const int off = kind == 0 ? 0 : (kind == 1 ? 4 : (kind == 2 ? 8 : 15));
    const int nq = kind == 0 ? 4 : (kind == 1 ? 4 : 7);
    if (gr != 0.f && slot >= off && slot < off + nq) acc += gr * dout[bk * 10 + slot - off];
    if (ga != 0.f && kind != 1) {
      const int ao = kind == 0 ? 0 : (kind == 2 ? 8 : 18);
      if (slot >= ao && slot < ao + 3) acc += ga * dout[bk * 10 + 7 + slot - ao];
    }
  }
  gparams[e] = acc;
}

inline int loss_chunks(int B, int N, int *ppb) {
  int want = (512 + B - 1) / (B > 0 ? B : 1);
  if (want < 1) want = 1;
  if (want > 64) want = 64;
  int p = (N + want - 1) / want;
  p = ((p + SS_TILE - 1) / SS_TILE) * SS_TILE;
  *ppb = p;
  return (N + p - 1) / p;
}

// ------------------------------------------------------------------------------------ P coverage (evaluation)
// compute_P_coverage (SPFN/metric_implementation.py:409-415): for every point of the cloud the residue against
// EVERY instance slot k (prediction matching[b,k], evaluated as the primitive type type_of_slot[b,k]), the minimum
// over k, and the fraction of points below each epsilon.  The reference expands P to [B,K,N,3] and evaluates all
// four residue formulas for every (k, n) ([B,K,N,4]: 77 MB at one 131072-point cloud with K = 49); here one lane
// per point walks the K slots with the slot parameters in LDS.
// residue value only (no tangents): the same formulas as residue_of above
// (sqrt_safe_f, residue_value: residue.h — shared with the evaluation tail, metrics.hip)
constexpr int PC_MAXEPS = 4;
struct PcEps { float e[PC_MAXEPS]; };
// partial[b][chunk][n_eps] = number of points of the chunk with min_k sqrt_safe(residue) < eps
__global__ __launch_bounds__(256) void p_coverage_kernel(const float *__restrict__ P, const float *__restrict__ params,
                                                         const long long *__restrict__ match,
                                                         const long long *__restrict__ slot_type, int N, int K,
                                                         int tid_plane, int tid_sphere, int tid_cyl, PcEps eps, int n_eps,
                                                         float *__restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float s_q[64][8];
  __shared__ int s_kind[64];
  __shared__ float s_cnt[4][PC_MAXEPS];
  const int b = blockIdx.y, t = threadIdx.x;
  const int n = blockIdx.x * 256 + t;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (n < N) {
    const float *p = P + ((size_t)b * N + n) * 3;
    px = p[0]; py = p[1]; pz = p[2];
  }
  float best = INFINITY;
  for (int k0 = 0; k0 < K; k0 += 64) {          // slot parameters through LDS, 64 slots at a time (any K)
    const int kw = min(64, K - k0);
    __syncthreads();
    if (t < kw) {
      const int k = k0 + t;
      const long long m = match[(size_t)b * K + k], ty = slot_type[(size_t)b * K + k];
      const int kind = ty == tid_plane ? 0 : (ty == tid_sphere ? 1 : (ty == tid_cyl ? 2 : 3));
      const float *P22 = params + ((size_t)b * K + m) * 22;
      const int off = kind == 0 ? 0 : (kind == 1 ? 4 : (kind == 2 ? 8 : 15));
      const int nq = kind == 0 ? 4 : (kind == 1 ? 4 : 7);
      for (int i = 0; i < 8; ++i) s_q[t][i] = i < nq ? P22[off + i] : 0.f;
      s_kind[t] = kind;
    }
    __syncthreads();
    if (n < N)
      for (int k = 0; k < kw; ++k) {
        const cpfn_f32x4 q0 = cpfn_lds_read4(&s_q[k][0]), q1 = cpfn_lds_read4(&s_q[k][4]);   // never a 96-bit LDS read
        const float q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        best = fminf(best, sqrt_safe_f(residue_value(s_kind[k], q, px, py, pz)));
      }
  }
  float cnt[PC_MAXEPS] = {0, 0, 0, 0};
  if (n < N)
    for (int i = 0; i < PC_MAXEPS; ++i) cnt[i] = (i < n_eps && best < eps.e[i]) ? 1.f : 0.f;
  for (int i = 0; i < PC_MAXEPS; ++i)
    for (int msk = 32; msk >= 1; msk >>= 1) cnt[i] += __shfl_xor(cnt[i], msk, 64);
  if ((t & 63) == 0)
    for (int i = 0; i < PC_MAXEPS; ++i) s_cnt[t >> 6][i] = cnt[i];
  __syncthreads();
  if (t < n_eps) partial[((size_t)b * gridDim.x + blockIdx.x) * n_eps + t] = s_cnt[0][t] + s_cnt[1][t] + s_cnt[2][t] + s_cnt[3][t];
}

__global__ void p_coverage_reduce_kernel(const float *__restrict__ partial, int chunks, int n_eps, int N, int total,
                                         float *__restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;   // (b, eps)
  if (e >= total) return;
  const int b = e / n_eps, i = e - b * n_eps;
  double s = 0.0;
  for (int c = 0; c < chunks; ++c) s += partial[((size_t)b * chunks + c) * n_eps + i];
  out[e] = (float)(s / (double)N);
}

// ------------------------------------------------------------------------------------ assignment
// The reference solves one linear assignment per cloud on the HOST (losses_implementation.py:19-29:
// scipy.optimize.linear_sum_assignment(-cost) on the relaxed-IoU matrix of the n_gt GT instances against the K
// predictions): the step's only device->host->device round trip.  Here the same solver runs on the device, one
// wave per cloud, so the whole step is a single graph with no host synchronisation.
// Algorithm = SciPy's rectangular_lsap.cpp (Crouse's shortest augmenting path, IEEE TAES 2016) restated, fp64,
// INCLUDING its two tie-breaking rules, because they decide which of several optimal assignments comes out:
// the unvisited-column list starts in reverse order, and among equal reduced costs an unassigned column (the
// last one in list order) wins.  The column scan of every path step is done by the lanes in parallel; the
// arg-min reproduces the sequential scan's choice exactly (ballots over list positions).
// (solver: lsap.h)
__global__ __launch_bounds__(64) void lsap_kernel(const float *__restrict__ S, const long long *__restrict__ n_gt, int K,
                                                  long long *__restrict__ match) {
  if (K <= LSAP_MAXK) lsap_one_cloud<LSAP_MAXK>(S, n_gt, K, match, blockIdx.x, threadIdx.x);
  else lsap_one_cloud<LSAP_WIDE_MAXK>(S, n_gt, K, match, blockIdx.x, threadIdx.x);
}

// ------------------------------------------------------------------------------------ loss tail
// Everything after the fitters that is [B,K]-sized (losses_implementation.py:77-90, 603-606, 633-673):
// relaxed IoU of the matched pairs from the segmented sums S, the per-cloud masked means of the IoU /
// residue / axis losses over the n_gt existing instances, the batch means and the weighted total — and, in the
// same pass, d total / d (S, rp, nl, tl), which are closed-form.  One workgroup; one lane per cloud.
//   out[6] = total, normal, type, miou, residue, parameter      mult[6] = normal, type, miou, residue, parameter, total
constexpr int LT_MAXB = 1024;
struct LossMult { float m[6]; };
__global__ __launch_bounds__(256) void loss_tail_kernel(const float *__restrict__ S, const float *__restrict__ rp,
                                                        const float *__restrict__ nl, const float *__restrict__ tl,
                                                        int nl_stride, const long long *__restrict__ match,
                                                        const long long *__restrict__ n_gt, int B, int K, LossMult mu,
                                                        float *__restrict__ out, float *__restrict__ gS,
                                                        float *__restrict__ grp, float *__restrict__ gnl,
                                                        float *__restrict__ gtl) {
  __shared__ float s_v[5][LT_MAXB];
  __shared__ float s_k[3][256];            // per-(cloud, instance) terms of the clouds in flight
  const int t = threadIdx.x;
  const bool on_miou = mu.m[2] > 0.f, on_res = mu.m[3] > 0.f && rp, on_par = mu.m[4] > 0.f && rp;
  const float invB = 1.f / (float)B, mt = mu.m[5];
  for (long long e = t; e < (long long)B * (K + 2) * K; e += 256) gS[e] = 0.f;
  __syncthreads();
  // One lane per (cloud, instance) — the first version walked the K instances of a cloud serially in one lane, a chain
  // of K dependent (match -> S gather) round trips: 20 us for a few KB.  The assignment is a permutation, so every
  // gS entry is written by exactly one lane; the per-cloud sums are taken from LDS in instance order.
  const int cpb = K <= 256 ? 256 / K : 0;      // clouds per pass (K <= MAXK = 64)
  for (int b0 = 0; b0 < B; b0 += cpb) {
    const int bl = t / K, k = t - bl * K, b = b0 + bl;
    float v_miou = 0.f, v_res = 0.f, v_par = 0.f;
    if (bl < cpb && b < B) {
      long long nn = n_gt[b];
      const int n = (int)(nn < 0 ? 0 : (nn > K ? K : nn));
      const float *Sb = S + (size_t)b * (K + 2) * K;
      float *gSb = gS + (size_t)b * (K + 2) * K;
      const float inv_n = n > 0 ? 1.f / (float)n : 0.f;
      const float c_miou = on_miou ? mt * mu.m[2] * invB * inv_n : 0.f;
      const bool live = k < n;
      if (live && on_miou) {
        long long m = match[(size_t)b * K + k];
        m = m < 0 ? 0 : (m >= K ? K - 1 : m);
        const float dot = Sb[k * K + m], colv = Sb[K * K + m], cntv = Sb[(K + 1) * K + k];
        const float q = cntv + colv - dot + 1e-10f;
        v_miou = 1.f - dot / q;
        const float dq = dot / (q * q);
        gSb[k * K + m] = -c_miou * (1.f / q + dq);
        gSb[K * K + m] = c_miou * dq;
        gSb[(K + 1) * K + k] = c_miou * dq;
      }
      if (rp) {
        if (live) { v_res = rp[((size_t)b * K + k) * 2]; v_par = rp[((size_t)b * K + k) * 2 + 1]; }
        grp[((size_t)b * K + k) * 2] = (live && on_res) ? mt * mu.m[3] * invB * inv_n : 0.f;
        grp[((size_t)b * K + k) * 2 + 1] = (live && on_par) ? mt * mu.m[4] * invB * inv_n : 0.f;
      }
    }
    s_k[0][t] = v_miou; s_k[1][t] = v_res; s_k[2][t] = v_par;
    __syncthreads();
    if (t < cpb && b0 + t < B) {
      const int bb = b0 + t;
      long long nn = n_gt[bb];
      const int n = (int)(nn < 0 ? 0 : (nn > K ? K : nn));
      const float inv_n = n > 0 ? 1.f / (float)n : 0.f;
      float miou = 0.f, res = 0.f, par = 0.f;
      for (int kk = 0; kk < K; ++kk) { miou += s_k[0][t * K + kk]; res += s_k[1][t * K + kk]; par += s_k[2][t * K + kk]; }
      s_v[0][bb] = nl[(size_t)bb * nl_stride];
      s_v[1][bb] = tl[(size_t)bb * nl_stride];
      s_v[2][bb] = miou * inv_n;
      s_v[3][bb] = res * inv_n;
      s_v[4][bb] = par * inv_n;
      gnl[bb] = mu.m[0] > 0.f ? mt * mu.m[0] * invB : 0.f;
      gtl[bb] = mu.m[1] > 0.f ? mt * mu.m[1] * invB : 0.f;
    }
    __syncthreads();
  }
  __syncthreads();
  if (t < 5) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += s_v[t][b];
    s *= invB;
    const bool on = t == 3 ? on_res : (t == 4 ? on_par : mu.m[t] > 0.f);
    s_v[t][0] = on ? s : 0.f;
  }
  __syncthreads();
  if (t == 0) {
    float total = 0.f;
    for (int i = 0; i < 5; ++i) {
      out[1 + i] = s_v[i][0];
      if (mu.m[i] > 0.f) total += mu.m[i] * s_v[i][0];
    }
    out[0] = total * mt;
  }
}

// ------------------------------------------------------------------------------------ two-class cross-entropy (PatchSelection)
// loss = mean_p (logsumexp(y_p) - y_p[label_p]) over P rows of two logits — F.cross_entropy of
// patch_selection_train_val_epoch (Utils/training_utils.py:66-68) — and, in the same pass, d loss / d y = (softmax - onehot) / P.
// As stock ops this was ~8 launches around a [131072, 2] tensor, among them torch's one-workgroup mean reduction: ~200 us of a
// 1.6 ms step.  Like cpfn_head_post_bwd the pass also leaves what the heads' backward makes of that gradient first: the rows as
// zero-padded bf16 [P, 64] and the per-256-row column sums in colsum_f32_kernel's order (bit-identical to that launch).
__global__ __launch_bounds__(256) void ce2_kernel(const float *__restrict__ logits, const long long *__restrict__ labels,
                                                  long long P, float inv_P, float *__restrict__ partial,
                                                  float *__restrict__ dlogits, unsigned short *__restrict__ pad_bf16,
                                                  float *__restrict__ colsum_partial) {
  __shared__ float s_g[256][2];
  __shared__ float s_l[4], s_cs[4][2];
  const int t = threadIdx.x;
  const long long p0 = (long long)blockIdx.x * 256, row = p0 + t;
  const int rows = (int)min(256ll, P - p0);
  float li = 0.f, g0 = 0.f, g1 = 0.f;
  if (t < rows) {
    const float2 y = ((const float2 *)logits)[row];
    const long long lab = labels[row];
    const bool one = lab != 0;
    const float m = fmaxf(y.x, y.y), e0 = expf(y.x - m), e1 = expf(y.y - m), sum = e0 + e1;
    li = (m + logf(sum)) - (one ? y.y : y.x);
    g0 = (e0 / sum - (one ? 0.f : 1.f)) * inv_P;
    g1 = (e1 / sum - (one ? 1.f : 0.f)) * inv_P;
    // A label outside {0, 1}: F.cross_entropy raises on it (or, for ignore_index = -100, leaves the row out of the mean); a
    // captured launch cannot raise, so such a row POISONS the loss and its gradient with NaN — the trainer's finite check then
    // skips the step and the caller sees a NaN loss instead of a row silently counted as class 1.
    if (lab != 0 && lab != 1) li = g0 = g1 = __builtin_nanf("");
    ((float2 *)dlogits)[row] = (float2){g0, g1};
  }
  s_g[t][0] = g0; s_g[t][1] = g1;
  for (int msk = 32; msk >= 1; msk >>= 1) li += __shfl_xor(li, msk, 64);
  if ((t & 63) == 0) s_l[t >> 6] = li;
  __syncthreads();
  if (t == 0) partial[blockIdx.x] = ((s_l[0] + s_l[1]) + s_l[2]) + s_l[3];
  if (pad_bf16)
    for (int e = t; e < rows * 8; e += 256) {
      const int r = e >> 3;
      unsigned w0 = 0u;
      if ((e & 7) == 0)
        w0 = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)s_g[r][0]) |
             ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)s_g[r][1]) << 16);
      *(uint4 *)(pad_bf16 + (p0 + r) * 64 + (e & 7) * 8) = (uint4){w0, 0u, 0u, 0u};
    }
  if (colsum_partial) {
    const int c = t & 63, rs = t >> 6;
    if (c < 2) {
      float a = 0.f;
      for (int r = rs; r < rows; r += 4) a += s_g[r][c];
      s_cs[rs][c] = a;
    }
    __syncthreads();
    if (t < 2) colsum_partial[(size_t)blockIdx.x * 2 + t] = ((s_cs[0][t] + s_cs[1][t]) + s_cs[2][t]) + s_cs[3][t];
  }
}

__global__ void ce2_finish_kernel(const float *__restrict__ partial, int nblk, float inv_P, float *__restrict__ loss) {
  double s = 0.0;
  for (int i = 0; i < nblk; ++i) s += (double)partial[i];
  *loss = (float)(s * (double)inv_P);
}

}  // namespace

extern "C" int cpfn_head_post_chunks(int N) { return cpfn_cdiv(N, LP_THREADS); }

extern "C" int cpfn_head_post_fwd(const float *Y, const float *Xgt, const int64_t *Igt, const int64_t *Tgt, int B,
                                  int N, int K, float *Xn, float *Wsm, float *workspace, float *stats,
                                  float *seg_workspace, float *S, int *lab_workspace, int64_t *n_gt, void *stream) {
  if (B <= 0 || N <= 0 || K <= 0 || K > MAXK || !Y || !Xgt || !Igt || !Tgt || !Xn || !Wsm || !workspace || !stats)
    return CPFN_EINVAL;
  if ((seg_workspace != nullptr) != (S != nullptr) || (S && K >= MAXK)) return CPFN_EINVAL;   // the ones row needs K < 32
  if ((lab_workspace != nullptr) != (n_gt != nullptr)) return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = cpfn_cdiv(N, LP_THREADS);
  head_post_fwd_kernel<<<dim3(chunks, B), LP_THREADS, 0, st>>>(Y, Xgt, (const long long *)Igt, (const long long *)Tgt, N, K,
                                                               Xn, Wsm, workspace, seg_workspace, lab_workspace);
  const long long total = S ? (long long)B * (K + 2) * K : 0;
  head_post_finish_kernel<<<cpfn_cdiv(total + B, 256), 256, 0, st>>>(workspace, seg_workspace, lab_workspace, chunks, N, B,
                                                                    (K + 2) * K, total, stats, S, (long long *)n_gt);
  return cpfn_launch_status();
}

extern "C" int cpfn_head_post_bwd(const float *Y, const float *Xgt, const int64_t *Igt, const int64_t *Tgt,
                                  const float *Wsm, const float *stats, const float *gXn, const float *gW,
                                  const float *gloss, int gloss_planar, int B, int N, int K, float *gY, const float *gS,
                                  void *pad_bf16, float *colsum_partial, void *stream) {
  if (B <= 0 || N <= 0 || K <= 0 || K > MAXK || !Y || !Xgt || !Igt || !Tgt || !Wsm || !stats || !gloss || !gY)
    return CPFN_EINVAL;
  if ((pad_bf16 || colsum_partial) && (N % LP_THREADS || 7 + K > 64 || !pad_bf16 || !colsum_partial)) return CPFN_EINVAL;
  head_post_bwd_kernel<<<dim3(cpfn_cdiv(N, LP_THREADS), B), LP_THREADS, 0, (hipStream_t)stream>>>(
      Y, Xgt, (const long long *)Igt, (const long long *)Tgt, Wsm, stats, gXn, gW, gloss, gloss_planar, N, K, gY, gS,
      (unsigned short *)pad_bf16, colsum_partial);
  return cpfn_launch_status();
}

extern "C" int cpfn_seg_stats_chunks(int B, int N) {
  int ppb;
  return loss_chunks(B, N, &ppb);
}

extern "C" int cpfn_seg_stats_fwd(const float *W, const int64_t *Igt, int B, int N, int K, float *workspace, float *S,
                                  void *stream) {
  if (B <= 0 || N <= 0 || K <= 0 || K > 1024 || !W || !Igt || !workspace || !S) return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int ppb;
  const int chunks = loss_chunks(B, N, &ppb);
  if (K <= MAXK) {
    seg_stats_fwd_kernel<false><<<dim3(chunks, B), LP_THREADS, 0, st>>>(W, (const long long *)Igt, N, K, ppb, workspace);
  } else {        // evaluation-sized label sets: one 32 x 32 tile of S per workgroup
    const int kt = (K + MAXK - 1) / MAXK;
    seg_stats_fwd_kernel<true><<<dim3(chunks, B, kt * kt), LP_THREADS, 0, st>>>(W, (const long long *)Igt, N, K, ppb, workspace);
  }
  const long long total = (long long)B * (K + 2) * K;
  chunk_sum_f32_kernel<<<cpfn_cdiv(total, 256), 256, 0, st>>>(workspace, chunks, (K + 2) * K, total, S);
  return cpfn_launch_status();
}

extern "C" int cpfn_seg_stats_bwd(const float *gS, const int64_t *Igt, int B, int N, int K, float *dW, void *stream) {
  if (B <= 0 || N <= 0 || K <= 0 || K > MAXK || !gS || !Igt || !dW) return CPFN_EINVAL;
  seg_stats_bwd_kernel<<<dim3(cpfn_cdiv(N, LP_THREADS), B), LP_THREADS, 0, (hipStream_t)stream>>>(
      gS, (const long long *)Igt, N, K, dW);
  return cpfn_launch_status();
}

extern "C" int cpfn_residue_fwd(const float *params, const int64_t *match, const int64_t *Tgt, const float *pts,
                                const float *gt_axes, int B, int K, int NP, const int *type_ids, float *out, float *dout,
                                void *stream) {
  if (B <= 0 || K <= 0 || NP <= 0 || !params || !match || !Tgt || !pts || !gt_axes || !type_ids || !out || !dout)
    return CPFN_EINVAL;
  residue_fwd_kernel<<<B * K, 128, 0, (hipStream_t)stream>>>(params, (const long long *)match, (const long long *)Tgt, pts,
                                                             gt_axes, K, NP, type_ids[0], type_ids[1], type_ids[2],
                                                             type_ids[3], out, dout);
  return cpfn_launch_status();
}

extern "C" int cpfn_residue_bwd(const float *gout, const float *dout, const int64_t *match, const int64_t *Tgt, int B,
                                int K, const int *type_ids, float *gparams, void *stream) {
  if (B <= 0 || K <= 0 || !gout || !dout || !match || !Tgt || !type_ids || !gparams) return CPFN_EINVAL;
  residue_bwd_kernel<<<cpfn_cdiv(B * K * 22, 256), 256, 0, (hipStream_t)stream>>>(gout, dout, (const long long *)match,
                                                                          (const long long *)Tgt, K, B * K, type_ids[0],
                                                                          type_ids[1], type_ids[2], type_ids[3], gparams);
  return cpfn_launch_status();
}

extern "C" int cpfn_loss_tail(const float *S, const float *rp, const float *nl, const float *tl, int nl_stride,
                              const int64_t *match, const int64_t *n_gt, int B, int K, const float *mult6, float *out6,
                              float *gS, float *grp, float *gnl, float *gtl, void *stream) {
  if (B <= 0 || B > LT_MAXB || K <= 0 || nl_stride < 1 || !S || !nl || !tl || !match || !n_gt || !mult6 || !out6 || !gS ||
      !gnl || !gtl || (rp && !grp))
    return CPFN_EINVAL;
  LossMult mu;
  for (int i = 0; i < 6; ++i) mu.m[i] = mult6[i];   // host array, passed by value
  loss_tail_kernel<<<1, 256, 0, (hipStream_t)stream>>>(S, rp, nl, tl, nl_stride, (const long long *)match,
                                                       (const long long *)n_gt, B, K, mu, out6, gS, grp, gnl, gtl);
  return cpfn_launch_status();
}

extern "C" int cpfn_hungarian_match(const float *S, const int64_t *n_gt, int B, int K, int64_t *match, void *stream) {
  if (B < 0 || K <= 0 || K > LSAP_WIDE_MAXK || !S || !n_gt || !match) return CPFN_EINVAL;
  if (B == 0) return 0;
  lsap_kernel<<<B, 64, 0, (hipStream_t)stream>>>(S, (const long long *)n_gt, K, (long long *)match);
  return cpfn_launch_status();
}

extern "C" int cpfn_p_coverage(const float *P, const float *params22, const int64_t *match, const int64_t *slot_type, int B,
                               int N, int K, const int *type_ids, const float *eps, int n_eps, float *workspace, float *out,
                               void *stream) {
  if (B <= 0 || N <= 0 || K <= 0 || n_eps <= 0 || n_eps > PC_MAXEPS || !P || !params22 || !match || !slot_type ||
      !type_ids || !eps || !workspace || !out)
    return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  PcEps pe;
  for (int i = 0; i < PC_MAXEPS; ++i) pe.e[i] = i < n_eps ? eps[i] : 0.f;
  const int chunks = cpfn_cdiv(N, 256);
  p_coverage_kernel<<<dim3(chunks, B), 256, 0, st>>>(P, params22, (const long long *)match, (const long long *)slot_type, N, K,
                                                     type_ids[0], type_ids[1], type_ids[2], pe, n_eps, workspace);
  p_coverage_reduce_kernel<<<cpfn_cdiv(B * n_eps, 64), 64, 0, st>>>(workspace, chunks, n_eps, N, B * n_eps, out);
  return cpfn_launch_status();
}

extern "C" int cpfn_ce2_blocks(long long P) { return P > 0 ? (int)((P + 255) / 256) : 0; }

extern "C" int cpfn_ce2(const float *logits, const int64_t *labels, long long P, float *workspace, float *loss, float *dlogits,
                        void *pad_bf16, float *colsum_partial, void *stream) {
  if (P <= 0 || P > 2000000000LL || !logits || !labels || !workspace || !loss || !dlogits || (!pad_bf16 != !colsum_partial) ||
      (pad_bf16 && P % 256))
    return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = cpfn_ce2_blocks(P);
  const float inv_P = 1.0f / (float)P;
  ce2_kernel<<<nblk, 256, 0, st>>>(logits, (const long long *)labels, P, inv_P, workspace, dlogits, (unsigned short *)pad_bf16,
                                   colsum_partial);
  ce2_finish_kernel<<<1, 1, 0, st>>>(workspace, nblk, inv_P, loss);
  return cpfn_launch_status();
}
