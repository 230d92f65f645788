// BatchNorm passes of the per-point MLP stacks (finalize, normalise + ReLU (+ max-pool), backward passes), the fixed-order
// split reductions, the fp32 small-K first layer of sa1 and the column sums of the heads.  See mlp_fwd.hip for the data layout.
#include "mlp_common.h"
#include "seam.h"

namespace {

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
    // (measured and not kept, twice: a subset's <= 8 partial rows all requested before the first addition — with clamped row
    //  numbers +8 us per step, with masked loads (no traffic for rows that do not exist) +31 us, same-box A/B both; the large
    //  layers' 342-448 rows are 6-7 iterations of this loop, a batch of four and a remainder loop that takes a round trip per
    //  row, and that is the faster form)
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
    // same order of additions as a plain loop over q, but the LDS reads of eight subsets are issued before the first add (round 4:
    // as a plain loop this tail was 64 dependent read -> add steps on the critical path of all 34 finalize launches of a step)
    for (int q0 = 0; q0 < RSUB; q0 += 8) {
      double v1[8], v2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { v1[j] = s_acc[q0 + j][threadIdx.x & 15][0]; v2[j] = s_acc[q0 + j][threadIdx.x & 15][1]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1 += v1[j]; s2 += v2[j]; }
    }
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
  // (the per-channel parameters are requested BEFORE the reduction: their latency then overlaps the partials' instead of
  //  following the sums on this launch's chain)
  float g_ = 0.f, b_ = 0.f, cb_ = 0.f, rm_ = 0.f, rv_ = 0.f;
  if (r == 0 && c < N) {
    g_ = gamma[c]; b_ = beta[c];
    if (running_mean) { cb_ = conv_bias ? conv_bias[c] : 0.f; rm_ = running_mean[c]; rv_ = running_var[c]; }
  }
  double s1, s2;
  partial_sums_16x16(partial, nblk, N, c, r, s_acc, s1, s2);
  if (r != 0 || c >= N) return;
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = g_ * rstd;
  scale[c] = sc;
  shift[c] = b_ - (float)mean * sc;
  mean_out[c] = (float)mean;
  rstd_out[c] = rstd;
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * rm_ + momentum * ((float)mean + cb_);
    const double unbiased = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
    running_var[c] = (1.f - momentum) * rv_ + momentum * (float)unbiased;
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

// ---------------------------------------------------------------- max-pool, second half (round 6)
// The pooled last layer's GEMM left, per wave of 32 rows and channel, the raw y of the wave's winner of max(+-y) and its row
// inside the group (cpfn_mlp_gemm_pool; +- = the sign of gamma).  Here: the WPG = pool_k / 32 wave results of a group are
// combined (ascending rows: `>` keeps the first), the affine map goes on the winner only —
//   out = relu(scale * y + shift), arg = its row, yarg = y —
// and scale / shift come either from cpfn_bn_finalize's vectors or are folded from the layer's seam (seam.h) by the first lanes
// of every workgroup: [G, C]-sized work in large workgroups (PF_GROUPS groups x 256 channels each), which is what a seam's
// consumer has to be.  out has the bits of cpfn_bn_relu_maxpool; arg / yarg name another row only where two DIFFERENT y round
// to the same z (this takes the first row of the extreme y, that one the first row of the z-tie).  scale == 0 / NaN, or no
// winner (a group of NaNs): row 0, like that kernel.
constexpr int PF_GROUPS = 32;
__global__ __launch_bounds__(256) void bn_pool_finish_kernel(const unsigned short *__restrict__ pmax, const unsigned char *__restrict__ pidx,
                                                             const unsigned short *__restrict__ Yr, int G, int pool_k, int C,
                                                             const float *__restrict__ scale, const float *__restrict__ shift,
                                                             const SeamIn si, unsigned short *__restrict__ out,
                                                             unsigned char *__restrict__ arg, unsigned short *__restrict__ yarg) {
  __shared__ float s_sc[256], s_sh[256];
  const int t = threadIdx.x;
  const int cb = blockIdx.y * 256, nc = min(256, C - cb);        // this workgroup's channels
  if (t < nc) {
    float sc, sh;
    if (si.acc) seam_fold_fwd(si, cb + t, blockIdx.x == 0, sc, sh);
    else { sc = scale[cb + t]; sh = shift[cb + t]; }
    s_sc[t] = sc; s_sh[t] = sh;
  }
  __syncthreads();
  const int nch = nc / 8, gsub = 256 / nch;                      // chunk lanes x groups per pass
  const int ch = t % nch, gs = t / nch, c0 = cb + ch * 8;
  const int wpg = pool_k / 32;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = s_sc[ch * 8 + j]; sh[j] = s_sh[ch * 8 + j]; }
  const int g1 = min(G, ((int)blockIdx.x + 1) * PF_GROUPS);
  for (int g = blockIdx.x * PF_GROUPS + gs; g < g1; g += gsub) {
    uint4 pv[4];
    uint2 pi[4];
#pragma unroll
    for (int w = 0; w < 4; ++w)
      if (w < wpg) {
        pv[w] = *(const uint4 *)(pmax + ((size_t)g * wpg + w) * C + c0);
        pi[w] = *(const uint2 *)(pidx + ((size_t)g * wpg + w) * C + c0);
      }
    unsigned short o[8], ya[8];
    unsigned char ka[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool neg = !(sc[j] >= 0.f) && sc[j] < 0.f;           // (the producer used the sign of gamma = the sign of scale)
      float best = -INFINITY, ybest = 0.f;
      int kbest = 255;
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (w < wpg) {
          const unsigned short *hv = (const unsigned short *)&pv[w];
          const unsigned char *hk = (const unsigned char *)&pi[w];
          const float y = bf2f(hv[j]), v = neg ? -y : y;
          if (hk[j] != 255 && v > best) { best = v; ybest = y; kbest = hk[j]; }
        }
      if (kbest == 255 || !(sc[j] != 0.f)) {                      // no winner, or no ordering: row 0
        kbest = 0;
        ybest = bf2f(Yr[(size_t)g * pool_k * C + c0 + j]);
      }
      o[j] = f2bf(fmaxf(fmaf(sc[j], ybest, sh[j]), 0.f));
      ya[j] = f2bf(ybest);
      ka[j] = (unsigned char)kbest;
    }
    *(uint4 *)(out + (size_t)g * C + c0) = *(const uint4 *)o;
    *(uint4 *)(yarg + (size_t)g * C + c0) = *(const uint4 *)ya;
    *(uint2 *)(arg + (size_t)g * C + c0) = *(const uint2 *)ka;
  }
}

// ---------------------------------------------------------------- BatchNorm backward, pass 1
// JOIN (round 6): the gradient is the SUM of two row-strided bf16 tensors — the two consumers' gradients of a tensor that autograd's
// input buffer would have added with a framework kernel between their backward nodes (sa2's pooled features: sa3's input rows,
// columns 3.. of a wider gradient and therefore only 2-byte aligned, + sfp1's skip columns; autograd_ops.SkipJoin) — formed on
// load with that add's rounding, bf16(a + b), and left contiguous in Gsum for the apply pass.
template <bool DROP, bool JOIN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(JOIN ? 2 : (DROP ? 3 : 4)))) void bn_relu_bwd_kernel(const unsigned short *__restrict__ Ga,
                                                          const unsigned short *__restrict__ Yr,
                                                          const float *__restrict__ scale,
                                                          const float *__restrict__ shift, long long P, int C,
                                                          unsigned short *__restrict__ Gz,
                                                          float *__restrict__ partial, int rpb,
                                                          const unsigned long long *__restrict__ drop_seed,
                                                          unsigned thresh16, float inv_keep, int ldg = 0,
                                                          const unsigned short *__restrict__ Gb = nullptr, int ldb = 0,
                                                          unsigned short *__restrict__ Gsum = nullptr) {
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
          if (JOIN) {
            const unsigned short *pa = Ga + rr * ldg + c0;        // (2-byte aligned rows: element loads; Gb's are 16-byte aligned)
            const uint4 qb = *(const uint4 *)(Gb + rr * ldb + c0);
            const unsigned wb[4] = {qb.x, qb.y, qb.z, qb.w};
            unsigned w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
              w[j] = (unsigned)f2bf(bf2f(pa[2 * j]) + bf2f((unsigned short)(wb[j] & 0xffffu))) |
                     ((unsigned)f2bf(bf2f(pa[2 * j + 1]) + bf2f((unsigned short)(wb[j] >> 16))) << 16);
            rg[u] = make_uint4(w[0], w[1], w[2], w[3]);
          } else
          rg[u] = *(const uint4 *)(Ga + rr * C + c0);
          ry[u] = *(const uint4 *)(Yr + rr * C + c0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long rr = r + (long long)u * rsub;
          const bool live = rr < rend;
          if (JOIN && live) *(uint4 *)(Gsum + rr * C + c0) = rg[u];
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
__device__ __forceinline__ void bn_bwd_finalize_body(const float *__restrict__ partial, int nblk, int C, float count,
                                                     const float *__restrict__ gamma, const float *__restrict__ mean,
                                                     const float *__restrict__ rstd, int training, float *__restrict__ dgamma,
                                                     float *__restrict__ dbeta, float *__restrict__ coef /*[3][C]*/,
                                                     unsigned *__restrict__ flags = nullptr /* one word: see msr_body_checked */) {
  __shared__ double s_acc[RSUB][16][2];
  const int c = blockIdx.x * 16 + (threadIdx.x & 15), r = threadIdx.x >> 4;
  float m_ = 0.f, rs_ = 0.f, g_ = 0.f;                  // (requested before the reduction, like bn_finalize_kernel's)
  if (r == 0 && c < C) { m_ = mean[c]; rs_ = rstd[c]; g_ = gamma[c]; }
  double s1, s2;
  partial_sums_16x16(partial, nblk, C, c, r, s_acc, s1, s2);
  if (r != 0 || c >= C) return;
  const double m = m_, rs = rs_;
  const double dg = rs * (s2 - m * s1);
  dgamma[c] = (float)dg;
  dbeta[c] = (float)s1;
  if (flags) {      // (the lanes still here: r == 0 and c < C, all in the first wave; lane 0 always is one of them)
    const unsigned long long m = __ballot((cpfn_nonfinite((float)dg) | cpfn_nonfinite((float)s1)) != 0);
    if (threadIdx.x == 0 && m != 0ull) atomicOr(flags, 1u);
  }
  const double s = (double)g_ * rs;
  const double c2 = training ? -s * dg * rs / count : 0.0;
  const double c3 = training ? -s * s1 / count - c2 * m : 0.0;
  coef[c] = (float)s;
  coef[C + c] = (float)c2;
  coef[2 * C + c] = (float)c3;
}
__global__ __launch_bounds__(RTPB) void bn_bwd_finalize_kernel(const float *__restrict__ partial, int nblk, int C, float count,
                                       const float *__restrict__ gamma, const float *__restrict__ mean,
                                       const float *__restrict__ rstd, int training, float *__restrict__ dgamma,
                                       float *__restrict__ dbeta, float *__restrict__ coef /*[3][C]*/,
                                       unsigned *__restrict__ flags) {
  bn_bwd_finalize_body(partial, nblk, C, count, gamma, mean, rstd, training, dgamma, dbeta, coef, flags);
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
constexpr int MSR_RIDE_MAX = 6;      // buffers that may ride on one bn_bwd_finalize launch (bn_bwd_finalize_ride_kernel)
template <int MAXB>
struct MsrArgsT {
  const float *partial[MAXB];
  float *out[MAXB];
  long long n[MAXB];
  int splits[MAXB];
  int row_in[MAXB], row_out[MAXB];   // 0, 0: flat; else only the first row_out of every row_in elements are kept
  int out_ld[MAXB];                  // ... at row stride out_ld of the output (>= row_out; a slice of a wider matrix)
  int deep[MAXB];              // 1: few outputs, many splits: 16 elements x 16 split-subsets per workgroup instead of 64 x 4; 2: wide
  const float *coef[MAXB];     // non-NULL: the "xyz weight gradient" form (see msr_body)
  int block0[MAXB + 1];        // first workgroup of buffer i
  int count;
};
typedef MsrArgsT<MSR_MAX> MsrArgs;
// (the body of multi_split_reduce_kernel for workgroup `bid` of the launch described by `a`; 256 lanes)
template <class ARGS>
__device__ __forceinline__ unsigned msr_body(const ARGS &a, const int bid) {
  // -> non-zero in the lanes that stored a NaN / inf (all of them lanes of the workgroup's FIRST wave): msr_body_checked below
  unsigned bad = 0;
  // 64 consecutive elements x 4 split-subsets per workgroup: 256-byte coalesced rows of the partial buffers.  "deep"
  // buffers (the 192 outputs x 1024 partials of the fp32-xyz layer, the 35 x 512 of the heads' bias: three / one workgroup
  // walking 256 / 128 rows each was a 20 us serial tail of this launch, which is why they had their own launches): 16 x 16.
  // "wide" buffers (mode 2: n % 4 == 0, 16-byte aligned — every large weight matrix): 256 elements x 4 subsets, one
  // float4 per lane and row and eight rows in flight — the launch reads ~250 MB and was latency-bound with 4-byte loads
  // (3.7 TB/s).  The order of the additions per element is the same in all three modes' common case (subset r adds rows
  // r, r + 4, ...; then (s0 + s1) + (s2 + s3)), so mode 2 is bit-identical to mode 0.  Mode 4 (round 4, below) has its own fixed order:
  // 16 subsets.  Which mode a buffer takes depends on its size and row count only, so a gradient has the same bits on every path.
  __shared__ __attribute__((aligned(16))) float s_acc[16][64];
  int d = 0;
  while (d + 1 < a.count && bid >= a.block0[d + 1]) ++d;
  const float *__restrict__ partial = a.partial[d];
  const long long n = a.n[d];
  const int splits = a.splits[d];
  if (a.coef[d]) {
    // The weight gradient of an fp32-xyz first layer (sa1) from sums that rode on the one-pass kernel of the layer above
    // (mlp_bwd_fused_kernel, XW): dW[c][j] = sum_p g_y[p,c] x[p,j] with g_y = c0 g_z + c1 y + c2 is LINEAR in the BatchNorm
    // coefficients, so the kernel that produces g_z — before the coefficients exist — accumulates S1 = sum g_z x_j, S2 = sum y x_j,
    // S3 = sum x_j per split and this reduction forms c0[c] S1[c][j] + c1[c] S2[c][j] + c2[c] S3[j]: the layer's stand-alone
    // weight-gradient launch and the [P, C] gradient tensor it read never exist.  partial [splits][7][C] (rows 0-2 S1, 3-5 S2,
    // row 6: S3 in its first three entries), C = row_in, out [C][3].  16 outputs x 16 split subsets per workgroup, fixed order.
    float (*s3)[16][3] = (float (*)[16][3])s_acc;        // [16 subsets][16 lanes][3]
    const int C = a.row_in[d];
    const int lane = threadIdx.x & 15, r = threadIdx.x >> 4;
    const long long e = (long long)(bid - a.block0[d]) * 16 + lane;
    const int c = (int)(e / 3), j = (int)(e - (long long)c * 3);
    float a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < n) {
#pragma unroll 4
      for (int i = r; i < splits; i += 16) {
        const float *pp = partial + (size_t)i * 7 * C;
        a1 += pp[j * C + c]; a2 += pp[(3 + j) * C + c]; a3 += pp[6 * C + j];
      }
    }
    s3[r][lane][0] = a1; s3[r][lane][1] = a2; s3[r][lane][2] = a3;
    __syncthreads();
    if (r == 0 && e < n) {
      float t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) { t1 += s3[q][lane][0]; t2 += s3[q][lane][1]; t3 += s3[q][lane][2]; }
      const float *cf = a.coef[d];
      const float ov = fmaf(cf[c], t1, fmaf(cf[C + c], t2, cf[2 * C + c] * t3));
      a.out[d][e] = ov;
      bad |= cpfn_nonfinite(ov);
    }
    return bad;
  }
  if (a.deep[d] == 4) {
    // "wide-deep" (round 4, large buffers with >= 64 partial rows): 64 elements (16 lanes x float4: 256-byte runs) x 16 subsets
    // per workgroup — a lane adds 16 of 256 rows (two batches of eight loads) instead of 64 (eight batches): four times the
    // workgroups and a quarter of the dependent round trips per lane.  What mattered when the reductions began to ride on the
    // finalize launches of the backward pass: a rider should be done when the finalize workgroups are.
    float4 (*s4)[16] = (float4 (*)[16])s_acc;          // [16 subsets][16 lanes] float4 = 4 KB
    const int lane = threadIdx.x & 15, r = threadIdx.x >> 4;
    const long long e = ((long long)(bid - a.block0[d]) * 16 + lane) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < n) {
      const float *src = partial + e;
      int i = r;
      for (; i + 112 < splits; i += 128) {             // eight rows of this subset in flight
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const float4 *)(src + (size_t)(i + 16 * u) * n);
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      }
      for (; i < splits; i += 16) {
        const float4 v = *(const float4 *)(src + (size_t)i * n);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    s4[r][lane] = acc;
    __syncthreads();
    if (r == 0 && e < n) {
      float4 t4 = s4[0][lane];
#pragma unroll
      for (int q = 1; q < 16; ++q) { const float4 u4 = s4[q][lane]; t4.x += u4.x; t4.y += u4.y; t4.z += u4.z; t4.w += u4.w; }
      const float v[4] = {t4.x, t4.y, t4.z, t4.w};
      const int ri = a.row_in[d];
      if (ri == 0) {
        bad |= cpfn_nonfinite(v[0]) | cpfn_nonfinite(v[1]) | cpfn_nonfinite(v[2]) | cpfn_nonfinite(v[3]);
        *(float4 *)(a.out[d] + e) = t4;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long long row = (e + j) / ri;
          const int col = (int)((e + j) - row * ri);
          if (col < a.row_out[d]) { a.out[d][row * a.out_ld[d] + col] = v[j]; bad |= cpfn_nonfinite(v[j]); }
        }
      }
    }
    return bad;
  }
  if (a.deep[d] == 2) {
    float4 (*s4)[64] = (float4 (*)[64])s_acc;          // [4 subsets][64 lanes] float4 = 4 KB
    const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
    const long long e = ((long long)(bid - a.block0[d]) * 64 + lane) * 4;
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
        bad |= cpfn_nonfinite(v[0]) | cpfn_nonfinite(v[1]) | cpfn_nonfinite(v[2]) | cpfn_nonfinite(v[3]);
        *(float4 *)(a.out[d] + e) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long long row = (e + j) / ri;
          const int col = (int)((e + j) - row * ri);
          if (col < a.row_out[d]) { a.out[d][row * a.out_ld[d] + col] = v[j]; bad |= cpfn_nonfinite(v[j]); }
        }
      }
    }
    return bad;
  }
  const bool deep = a.deep[d] != 0;
  const int epw = deep ? 16 : 64, nsub = deep ? 16 : 4;
  const int lane = deep ? (threadIdx.x & 15) : (threadIdx.x & 63), r = deep ? (threadIdx.x >> 4) : (threadIdx.x >> 6);
  const long long e = (long long)(bid - a.block0[d]) * epw + lane;
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
      bad |= cpfn_nonfinite(v);
    } else {           // zero-padded K: drop the padding columns (the caller gets a compact [N, row_out] matrix)
      const long long row = e / ri;
      const int col = (int)(e - row * ri);
      if (col < a.row_out[d]) { a.out[d][row * a.out_ld[d] + col] = v; bad |= cpfn_nonfinite(v); }
    }
  }
  return bad;
}

// flag (nullable): ONE word, OR-ed with 1 by every workgroup that stored a NaN / inf (a no-return atomic, only in that rare case) and
// cleared by its consumer (cpfn_adam_flat_sticky): the finite check of a training step's gradients rides on the launches that write
// them (round 6; the gradients land in the flat bucket directly and the packing copy that used to carry the scan is gone)
template <class ARGS>
__device__ __forceinline__ void msr_body_checked(const ARGS &a, const int bid, unsigned *__restrict__ flag) {
  const unsigned bad = msr_body(a, bid);
  if (flag && threadIdx.x < 64) {
    const unsigned long long m = __ballot(bad != 0);
    if (threadIdx.x == 0 && m != 0ull) atomicOr(flag, 1u);
  }
}
__global__ __launch_bounds__(256) void multi_split_reduce_kernel(MsrArgs a, unsigned *__restrict__ flags) {
  msr_body_checked(a, (int)blockIdx.x, flags);
}

// bn_bwd_finalize with split reductions RIDING on it (round 4).  The 17 finalize launches of a backward pass are eight-odd
// workgroups each on an otherwise idle chip, and the weight-gradient partials of the layer ABOVE were written by the launch before
// them: reduced here, as further workgroups of the same launch (the first 256 lanes of a 1024-lane workgroup run msr_body), they are
// read while they are still in the infinity cache instead of ~250 MB from HBM in one launch at the end of the pass.
typedef MsrArgsT<MSR_RIDE_MAX> MsrRideArgs;
__global__ __launch_bounds__(RTPB) void bn_bwd_finalize_ride_kernel(const float *__restrict__ partial, int nblk, int C, float count,
                                                                    const float *__restrict__ gamma, const float *__restrict__ mean,
                                                                    const float *__restrict__ rstd, int training,
                                                                    float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                                    float *__restrict__ coef, int nfin, MsrRideArgs a,
                                                                    unsigned *__restrict__ flags) {
  if ((int)blockIdx.x >= nfin) {
    if (threadIdx.x >= 256) return;           // (waves that have ended do not count at the body's barrier)
    msr_body_checked(a, (int)blockIdx.x - nfin, flags);
    return;
  }
  bn_bwd_finalize_body(partial, nblk, C, count, gamma, mean, rstd, training, dgamma, dbeta, coef, flags);
}

// ---------------------------------------------------------------- fp32 small-K first layer (sa1: K = 3)
// Y[p,c] = Σ_{j<KS} W[c,j]·X[p,j]  (fp32 inputs: relative coordinates are NOT rounded to bf16),
// bf16 output + Σy, Σy² partials.  One lane per (row-sub, 8-channel chunk).
constexpr int KS_MAX = 4;
template <int KS>
__device__ __forceinline__ void smallk_fwd_body(int bx, const float *__restrict__ X, const float *__restrict__ W, long long P, int C,
                                                unsigned short *__restrict__ Y, float *__restrict__ partial, int rpb,
                                                const SeamOut &so = SeamOut()) {
  __shared__ float s_red[2][256][8 + 1];
  const int t = threadIdx.x;
  if (so.acc && bx == 0 && t == 0) seam_counters(so);
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
    if (so.acc) seam_add(so, C, which, rem, s, (unsigned)bx);          // (no finalize launch behind this layer: seam.h)
    else partial[((size_t)bx * 2 + which) * C + rem] = s;
  }
}
template <int KS>
__global__ __launch_bounds__(256) void smallk_fwd_kernel(const float *__restrict__ X,
                                                         const float *__restrict__ W, long long P, int C,
                                                         unsigned short *__restrict__ Y,
                                                         float *__restrict__ partial, int rpb, const SeamOut so = SeamOut()) {
  smallk_fwd_body<KS>((int)blockIdx.x, X, W, P, C, Y, partial, rpb, so);
}
// The same launch with the step's bf16 weight-panel refresh (cpfn_multi_cast) as its first workgroups: sa1's first layer reads
// the fp32 weight itself, so the two are independent — and both sit at the very start of the step's chain, where the refresh
// alone was a ~7 us launch of pure latency (cpfn_smallk_fwd_cast).
#include "cast_body.h"
template <int KS>
__global__ __launch_bounds__(256) void smallk_fwd_cast_kernel(McvArgs cast, int cast_blocks, const float *__restrict__ X,
                                                              const float *__restrict__ W, long long P, int C,
                                                              unsigned short *__restrict__ Y, float *__restrict__ partial, int rpb,
                                                              const SeamOut so = SeamOut()) {
  if ((int)blockIdx.x < cast_blocks) multi_cast_body<256>(cast, (int)blockIdx.x);
  else smallk_fwd_body<KS>((int)blockIdx.x - cast_blocks, X, W, P, C, Y, partial, rpb, so);
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

extern "C" int cpfn_bn_pool_finish(const void *pmax, const unsigned char *pidx, const void *Y, int G, int pool_k, int C,
                                   const float *scale, const float *shift, const cpfn_seam_in *in, void *out, unsigned char *arg,
                                   void *yarg, void *stream) {
  if (G < 0 || !(pool_k == 32 || pool_k == 64 || pool_k == 128) || C < 64 || (C & 63) || !pmax || !pidx || !Y || !out || !arg || !yarg ||
      (!in && (!scale || !shift)) || (in && (scale || shift)) || !seam_in_valid(in, C))
    return CPFN_EINVAL;
  if (G == 0) return 0;
  const int nc = C < 256 ? C : 256;
  if ((nc / 8) & (nc / 8 - 1)) return CPFN_EINVAL;               // chunk lanes: a power of two
  bn_pool_finish_kernel<<<dim3(cpfn_cdiv(G, PF_GROUPS), cpfn_cdiv(C, 256)), 256, 0, (hipStream_t)stream>>>(
      (const unsigned short *)pmax, pidx, (const unsigned short *)Y, G, pool_k, C, scale, shift, seam_in_arg(in), (unsigned short *)out,
      arg, (unsigned short *)yarg);
  return cpfn_launch_status();
}

void cpfn_launch_split_reduce(const float *ws, int splits, long long n, float *out, hipStream_t st) {
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

extern "C" int cpfn_bn_relu_bwd_join(const void *Ga, int ldg, const void *Gb, int ldb, const void *Y, const float *scale,
                                     const float *shift, long long P, int C, void *Gsum, float *partial, void *stream) {
  if (P <= 0 || C <= 0 || (C & 7) || !pow2(C / 8) || !Ga || !Gb || !Y || !scale || !shift || !partial || !Gsum || ldg < C || ldb < C)
    return CPFN_EINVAL;
  if (((uintptr_t)Ga & 1) || ((uintptr_t)Gb & 15) || (ldb & 7)) return CPFN_EINVAL;
  bn_relu_bwd_kernel<false, true><<<cpfn_bn_bwd_blocks(P), 256, 0, (hipStream_t)stream>>>(
      (const unsigned short *)Ga, (const unsigned short *)Y, scale, shift, P, C, nullptr, partial, bn_rows_per_block(P), nullptr, 0u,
      1.f, ldg, (const unsigned short *)Gb, ldb, (unsigned short *)Gsum);
  return cpfn_launch_status();
}

extern "C" int cpfn_bn_bwd_finalize_checked(const float *partial, int nblk, int C, float count, const float *gamma,
                                            const float *mean, const float *rstd, int training, float *dgamma,
                                            float *dbeta, float *coef, unsigned *flags /* nullable: one word */, void *stream) {
  if (nblk <= 0 || C <= 0 || !partial || !gamma || !mean || !rstd || !dgamma || !dbeta || !coef) return CPFN_EINVAL;
  bn_bwd_finalize_kernel<<<cpfn_cdiv(C, 16), RTPB, 0, (hipStream_t)stream>>>(partial, nblk, C, count, gamma, mean, rstd,
                                                                            training, dgamma, dbeta, coef, flags);
  return cpfn_launch_status();
}
extern "C" int cpfn_bn_bwd_finalize(const float *partial, int nblk, int C, float count, const float *gamma,
                                    const float *mean, const float *rstd, int training, float *dgamma,
                                    float *dbeta, float *coef, void *stream) {
  return cpfn_bn_bwd_finalize_checked(partial, nblk, C, count, gamma, mean, rstd, training, dgamma, dbeta, coef, nullptr, stream);
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

static int smallk_fwd_cast_launch(const cpfn_cast_desc *casts, int n_casts, const float *X, int KS, const float *W, long long P,
                                  int C, void *Y, float *partial, const cpfn_seam_out *seam_out, void *stream) {
  if (P <= 0 || KS != 3 || C <= 0 || (C & 7) || !pow2(C / 8) || C / 8 > 256 || !X || !W || !Y || (!partial == !seam_out) || n_casts <= 0 ||
      n_casts > MCV_MAX || !casts || !seam_out_valid(seam_out))
    return CPFN_EINVAL;
  McvArgs a;
  int cast_blocks = 0;
  const int rc = mcv_fill(casts, n_casts, a, &cast_blocks);
  if (rc) return rc;
  const int nblk = cpfn_bn_bwd_blocks(P), rpb = bn_rows_per_block(P);
  smallk_fwd_cast_kernel<3><<<cast_blocks + nblk, 256, 0, (hipStream_t)stream>>>(a, cast_blocks, X, W, P, C, (unsigned short *)Y,
                                                                                 partial, rpb, seam_out_arg(seam_out));
  return cpfn_launch_status();
}
extern "C" int cpfn_smallk_fwd_cast(const cpfn_cast_desc *casts, int n_casts, const float *X, int KS, const float *W, long long P,
                                    int C, void *Y, float *partial, void *stream) {
  return smallk_fwd_cast_launch(casts, n_casts, X, KS, W, P, C, Y, partial, nullptr, stream);
}

static int smallk_fwd_launch(const float *X, int KS, const float *W, long long P, int C, void *Y, float *partial,
                             const cpfn_seam_out *seam_out, void *stream) {
  if (P <= 0 || KS <= 0 || KS > KS_MAX || C <= 0 || (C & 7) || !pow2(C / 8) || C / 8 > 256 || !X || !W || !Y || (!partial == !seam_out) ||
      !seam_out_valid(seam_out))
    return CPFN_EINVAL;
  const int nblk = cpfn_bn_bwd_blocks(P), rpb = bn_rows_per_block(P);
  hipStream_t st = (hipStream_t)stream;
  unsigned short *y = (unsigned short *)Y;
  const SeamOut so = seam_out_arg(seam_out);
  switch (KS) {
    case 1: smallk_fwd_kernel<1><<<nblk, 256, 0, st>>>(X, W, P, C, y, partial, rpb, so); break;
    case 2: smallk_fwd_kernel<2><<<nblk, 256, 0, st>>>(X, W, P, C, y, partial, rpb, so); break;
    case 3: smallk_fwd_kernel<3><<<nblk, 256, 0, st>>>(X, W, P, C, y, partial, rpb, so); break;
    default: smallk_fwd_kernel<4><<<nblk, 256, 0, st>>>(X, W, P, C, y, partial, rpb, so); break;
  }
  return cpfn_launch_status();
}
extern "C" int cpfn_smallk_fwd(const float *X, int KS, const float *W, long long P, int C, void *Y, float *partial,
                               void *stream) {
  return smallk_fwd_launch(X, KS, W, P, C, Y, partial, nullptr, stream);
}
// the same layer as the PRODUCER of a BatchNorm seam (seam.h): casts == NULL: plain launch; else the panel refresh rides (KS = 3)
extern "C" int cpfn_smallk_fwd_seam(const cpfn_cast_desc *casts, int n_casts, const float *X, int KS, const float *W, long long P,
                                    int C, void *Y, const cpfn_seam_out *out, void *stream) {
  if (!out) return CPFN_EINVAL;
  if (casts) return smallk_fwd_cast_launch(casts, n_casts, X, KS, W, P, C, Y, nullptr, out, stream);
  return smallk_fwd_launch(X, KS, W, P, C, Y, nullptr, out, stream);
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
  if (dW) cpfn_launch_split_reduce(workspace, nblk, n, dW, st);    // NULL: the caller batches it (cpfn_multi_split_reduce)
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
  if (out) cpfn_launch_split_reduce(workspace, nblk, C, out, st);  // NULL: the caller batches it (cpfn_multi_split_reduce)
  return cpfn_launch_status();
}

template <class ARGS>
static int msr_fill(ARGS &a, const cpfn_reduce_desc *descs, int count, int *blocks_out) {
  a.count = count;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const cpfn_reduce_desc &d = descs[i];
    if (!d.partial || !d.out || d.splits <= 0 || d.n <= 0 || d.row_in < 0 || d.row_out < 0 || d.row_out > d.row_in || (d.out_ld != 0 && (d.row_in == 0 || d.out_ld < d.row_out)) ||
        (!d.coef && d.row_in > 0 && (d.row_out == 0 || d.n % d.row_in))) return CPFN_EINVAL;
    a.partial[i] = d.partial; a.out[i] = d.out; a.n[i] = d.n; a.splits[i] = d.splits;
    a.row_in[i] = d.row_in; a.row_out[i] = d.row_out;
    a.out_ld[i] = d.out_ld > 0 ? d.out_ld : d.row_out;
    a.coef[i] = d.coef;
    if (d.coef) {          // the xyz weight-gradient form: n = 3 C outputs from [splits][7][C] partials (row_in = C)
      if (d.row_in <= 0 || d.n != 3LL * d.row_in) return CPFN_EINVAL;
      a.deep[i] = 1;
      a.block0[i] = blocks;
      blocks += cpfn_cdiv(d.n, 16);
      continue;
    }
    a.deep[i] = d.n <= 1024 && d.splits >= 128;
    if (!a.deep[i] && d.n >= 4096 && d.n % 4 == 0 && (((uintptr_t)d.partial | (d.row_in == 0 ? (uintptr_t)d.out : 0)) & 15) == 0)
      a.deep[i] = d.splits >= 64 ? 4 : 2;   // wide (float4 per lane; 2: the 64 x 4 layout's order of additions) / wide-deep (4: 16 subsets)
    a.block0[i] = blocks;
    blocks += cpfn_cdiv(d.n, a.deep[i] == 2 ? 256 : a.deep[i] == 4 ? 64 : a.deep[i] ? 16 : 64);
  }
  a.block0[count] = blocks;
  *blocks_out = blocks;
  return 0;
}

extern "C" int cpfn_multi_split_reduce_checked(const cpfn_reduce_desc *descs, int count, unsigned *flags, void *stream) {
  if (count < 0 || (count > 0 && !descs)) return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < count; base += MSR_MAX) {
    MsrArgs a;
    int blocks = 0;
    const int rc = msr_fill(a, descs + base, count - base < MSR_MAX ? count - base : MSR_MAX, &blocks);
    if (rc) return rc;
    if (blocks) multi_split_reduce_kernel<<<blocks, 256, 0, st>>>(a, flags);
  }
  return cpfn_launch_status();
}
extern "C" int cpfn_multi_split_reduce(const cpfn_reduce_desc *descs, int count, void *stream) {
  return cpfn_multi_split_reduce_checked(descs, count, nullptr, stream);
}
extern "C" int cpfn_bn_bwd_finalize_ride_checked(const float *partial, int nblk, int C, float count, const float *gamma,
                                                 const float *mean, const float *rstd, int training, float *dgamma, float *dbeta,
                                                 float *coef, const cpfn_reduce_desc *descs, int ndesc, unsigned *flags, void *stream) {
  if (nblk <= 0 || C <= 0 || !partial || !gamma || !mean || !rstd || !dgamma || !dbeta || !coef || ndesc < 0 || ndesc > MSR_RIDE_MAX ||
      (ndesc > 0 && !descs))
    return CPFN_EINVAL;
  MsrRideArgs a;
  int blocks = 0;
  const int rc = msr_fill(a, descs, ndesc, &blocks);
  if (rc) return rc;
  const int nfin = cpfn_cdiv(C, 16);
  bn_bwd_finalize_ride_kernel<<<nfin + blocks, RTPB, 0, (hipStream_t)stream>>>(partial, nblk, C, count, gamma, mean, rstd, training,
                                                                              dgamma, dbeta, coef, nfin, a, flags);
  return cpfn_launch_status();
}
extern "C" int cpfn_bn_bwd_finalize_ride(const float *partial, int nblk, int C, float count, const float *gamma, const float *mean,
                                         const float *rstd, int training, float *dgamma, float *dbeta, float *coef,
                                         const cpfn_reduce_desc *descs, int ndesc, void *stream) {
  return cpfn_bn_bwd_finalize_ride_checked(partial, nblk, C, count, gamma, mean, rstd, training, dgamma, dbeta, coef, descs, ndesc,
                                           nullptr, stream);
}

