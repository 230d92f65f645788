// Adam on flat buffers for gfx950: all parameters, gradients and both moments are single contiguous fp32
// buffers (training.FlatGradBucket / optim.FlatAdam), so the optimizer step is ONE streaming kernel instead of
// a multi-tensor launch over ~100 small tensors (62 us -> a few us for the 1.4 M parameters of GlobalSPFN).
// Same arithmetic as torch.optim.Adam (reference: the optimizer of Utils/training_utils.py's epoch loop;
// torch/optim/adam.py, non-amsgrad, maximize=False), capturable: learning rate, step count and the
// "skip this step" flag live in device memory.
#include "common.h"

namespace {

// coef[0] = lr / (1 - beta1^t), coef[1] = sqrt(1 - beta2^t), coef[2] = 1 when the step is skipped; t = step + 1.
// One wave.  beta^t is carried as a running fp64 product in pows[2] (beta1^step, beta2^step): calling pow() here
// costs 25 us of single-lane fp64 software — more than the streaming pass over the parameters.
// The step is skipped when *found_inf != 0 or any of the nf_count per-block flags of the gradient scan
// (nonfinite_partial_kernel) is set — the scan's final reduction and the trainer's skipped-step counter ride here
// instead of being two more launches.
// n_sticky: the first n_sticky words of nf_partial are not rewritten every step by a scan but OR-ed into by the launches that
// wrote the gradients (cpfn_multi_split_reduce_checked, cpfn_bn_bwd_finalize(_ride)_checked): read here, then cleared for the next step.
// xw (cpfn_adam_flat_xw): the LAST gradient of a backward pass — the weight gradient [C][3] of an fp32-xyz first layer from sums that
// rode on the layer above (bn.hip, msr_body's coefficient form): out[c][j] = c0[c] S1[j][c] + c1[c] S2[j][c] + c2[c] S3[j] with
// S [7][C] already reduced over the splits — is finished HERE, by this wave, with that form's arithmetic, and checked with the rest:
// it needs the BatchNorm coefficients of the last finalize launch of the pass, so as a reduction it was a launch of its own
// (12 workgroups, 5.7 us + a kernel boundary) between that finalize and this kernel.
struct AdamXw { const float *S, *coef; int C; float *out; };
__global__ __launch_bounds__(64) void adam_prepare_kernel(const float *__restrict__ lr, float beta1, float beta2,
                                                         float *__restrict__ step, double *__restrict__ pows,
                                                         const float *__restrict__ found_inf, float *__restrict__ coef,
                                                         unsigned *__restrict__ nf_partial, int nf_count,
                                                         float *__restrict__ skipped, int n_sticky, const AdamXw xw) {
  unsigned bad = 0;
  if (xw.out) {
    const int C = xw.C;
    for (int e = threadIdx.x; e < 3 * C; e += 64) {
      const int c = e / 3, j = e - 3 * c;
      const float v = fmaf(xw.coef[c], xw.S[j * C + c], fmaf(xw.coef[C + c], xw.S[(3 + j) * C + c], xw.coef[2 * C + c] * xw.S[6 * C + j]));
      xw.out[e] = v;
      bad |= (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u;
    }
  }
  for (int i = threadIdx.x; i < nf_count; i += 64) {
    bad |= nf_partial[i];
    if (i < n_sticky) nf_partial[i] = 0u;
  }
  const bool skip = __ballot(bad != 0) != 0ull || (found_inf && *found_inf != 0.f);
  if (threadIdx.x != 0) return;
  const double b1t = pows[0] * (double)beta1, b2t = pows[1] * (double)beta2;
  coef[0] = *lr / (float)(1.0 - b1t);
  coef[1] = sqrtf((float)(1.0 - b2t));
  coef[2] = skip ? 1.f : 0.f;
  if (!skip) { *step += 1.f; pows[0] = b1t; pows[1] = b2t; }
  else if (skipped) *skipped += 1.f;
}

__global__ __launch_bounds__(256) void adam_flat_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                        float *__restrict__ m, float *__restrict__ v, long long n,
                                                        float beta1, float beta2, float eps, float weight_decay,
                                                        const float *__restrict__ coef) {
  if (coef[2] != 0.f) return;                                       // non-finite gradients: the step is skipped
  const float step_size = coef[0], bc2_sqrt = coef[1];
  const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= n) return;
  // (no dynamically indexed local arrays: they would live in scratch memory and cost ~25 us per launch)
  const bool full = i0 + 4 <= n;
  float4 p4, g4, m4, v4;
  if (full) {
    p4 = *(const float4 *)(p + i0); g4 = *(const float4 *)(g + i0); m4 = *(const float4 *)(m + i0); v4 = *(const float4 *)(v + i0);
  } else {
    const long long i1 = i0 + 1, i2 = i0 + 2;
    p4 = make_float4(p[i0], i1 < n ? p[i1] : 0.f, i2 < n ? p[i2] : 0.f, 0.f);
    g4 = make_float4(g[i0], i1 < n ? g[i1] : 0.f, i2 < n ? g[i2] : 0.f, 0.f);
    m4 = make_float4(m[i0], i1 < n ? m[i1] : 0.f, i2 < n ? m[i2] : 0.f, 0.f);
    v4 = make_float4(v[i0], i1 < n ? v[i1] : 0.f, i2 < n ? v[i2] : 0.f, 0.f);
  }
  float pv[4] = {p4.x, p4.y, p4.z, p4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w},
        vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float gr = gv[j];
    if (weight_decay != 0.f) gr = fmaf(weight_decay, pv[j], gr);
    mv[j] = mv[j] + (gr - mv[j]) * (1.f - beta1);                   // exp_avg.lerp_(grad, 1 - beta1)
    vv[j] = beta2 * vv[j] + (1.f - beta2) * gr * gr;                // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
    pv[j] = pv[j] - step_size * (mv[j] / denom);
  }
  if (full) {
    *(float4 *)(p + i0) = make_float4(pv[0], pv[1], pv[2], pv[3]);
    *(float4 *)(m + i0) = make_float4(mv[0], mv[1], mv[2], mv[3]);
    *(float4 *)(v + i0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
  } else {
    p[i0] = pv[0]; m[i0] = mv[0]; v[i0] = vv[0];
    if (i0 + 1 < n) { p[i0 + 1] = pv[1]; m[i0 + 1] = mv[1]; v[i0 + 1] = vv[1]; }
    if (i0 + 2 < n) { p[i0 + 2] = pv[2]; m[i0 + 2] = mv[2]; v[i0 + 2] = vv[2]; }
  }
}

// flag[0] = 1.0 if any of x[0..n) is NaN or +-inf, else 0.0 (the reference scans every parameter gradient with
// isinf/isnan and two host syncs per tensor, Utils/training_utils.py:151-156; torch needs five kernels for
// `(~isfinite(x).all()).float()`).  Two tiny passes so that the flag is always written (no memset needed).
__global__ __launch_bounds__(256) void nonfinite_partial_kernel(const float *__restrict__ x, long long n,
                                                                unsigned *__restrict__ partial) {
  __shared__ unsigned s_any[4];
  unsigned bad = 0;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * 1024) {
    if (i + 4 <= n) {
      const uint4 v = *(const uint4 *)(x + i);
      bad |= ((v.x & 0x7f800000u) == 0x7f800000u) | ((v.y & 0x7f800000u) == 0x7f800000u) |
             ((v.z & 0x7f800000u) == 0x7f800000u) | ((v.w & 0x7f800000u) == 0x7f800000u);
    } else {
      for (long long j = i; j < n; ++j) bad |= (__float_as_uint(x[j]) & 0x7f800000u) == 0x7f800000u;
    }
  }
  const unsigned long long m = __ballot(bad != 0);
  if ((threadIdx.x & 63) == 0) s_any[threadIdx.x >> 6] = m != 0;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = s_any[0] | s_any[1] | s_any[2] | s_any[3];
}
__global__ void nonfinite_final_kernel(const unsigned *__restrict__ partial, int nblk, float *__restrict__ flag) {
  unsigned bad = 0;
  for (int i = threadIdx.x; i < nblk; i += 64) bad |= partial[i];
  const unsigned long long m = __ballot(bad != 0);
  if (threadIdx.x == 0) *flag = m != 0 ? 1.f : 0.f;
}

}  // namespace

static inline int nonfinite_blocks(long long n) { return (int)(n / 4096 + 1 < 256 ? n / 4096 + 1 : 256); }

extern "C" int cpfn_nonfinite_blocks(long long n) { return n < 0 ? 0 : nonfinite_blocks(n); }

extern "C" int cpfn_nonfinite_partial(const float *x, long long n, unsigned *workspace256, void *stream) {
  if (n < 0 || !x || !workspace256 || ((uintptr_t)x & 15)) return CPFN_EINVAL;
  nonfinite_partial_kernel<<<nonfinite_blocks(n), 256, 0, (hipStream_t)stream>>>(x, n, workspace256);
  return cpfn_launch_status();
}

extern "C" int cpfn_nonfinite_flag(const float *x, long long n, unsigned *workspace256, float *flag, void *stream) {
  if (n < 0 || !x || !workspace256 || !flag || ((uintptr_t)x & 15)) return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = nonfinite_blocks(n);
  nonfinite_partial_kernel<<<nblk, 256, 0, st>>>(x, n, workspace256);
  nonfinite_final_kernel<<<1, 64, 0, st>>>(workspace256, nblk, flag);
  return cpfn_launch_status();
}

extern "C" int cpfn_adam_flat_xw(float *p, const float *g, float *m, float *v, long long n, const float *lr, float beta1,
                                 float beta2, float eps, float weight_decay, float *step, double *pows,
                                 const float *found_inf, float *coef3, unsigned *nf_partial, int nf_count, int n_sticky,
                                 float *skipped, const float *xw_S, const float *xw_coef, int xw_C, float *xw_out, void *stream) {
  if (n < 0 || !p || !g || !m || !v || !lr || !step || !pows || !coef3 || nf_count < 0 || (nf_count > 0 && !nf_partial) ||
      n_sticky < 0 || n_sticky > nf_count)
    return CPFN_EINVAL;
  if (xw_out && (!xw_S || !xw_coef || xw_C <= 0)) return CPFN_EINVAL;
  AdamXw xw;
  xw.S = xw_S; xw.coef = xw_coef; xw.C = xw_C; xw.out = xw_out;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return CPFN_EINVAL;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  adam_prepare_kernel<<<1, 64, 0, st>>>(lr, beta1, beta2, step, pows, found_inf, coef3, nf_partial, nf_count, skipped, n_sticky, xw);
  adam_flat_kernel<<<cpfn_cdiv(cpfn_cdiv(n, 4), 256), 256, 0, st>>>(p, g, m, v, n, beta1, beta2, eps, weight_decay, coef3);
  return cpfn_launch_status();
}

extern "C" int cpfn_adam_flat_sticky(float *p, const float *g, float *m, float *v, long long n, const float *lr, float beta1,
                                     float beta2, float eps, float weight_decay, float *step, double *pows,
                                     const float *found_inf, float *coef3, unsigned *nf_partial, int nf_count, int n_sticky,
                                     float *skipped, void *stream) {
  return cpfn_adam_flat_xw(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, pows, found_inf, coef3, nf_partial, nf_count,
                           n_sticky, skipped, nullptr, nullptr, 0, nullptr, stream);
}

extern "C" int cpfn_adam_flat(float *p, const float *g, float *m, float *v, long long n, const float *lr, float beta1,
                              float beta2, float eps, float weight_decay, float *step, double *pows,
                              const float *found_inf, float *coef3, const unsigned *nf_partial, int nf_count,
                              float *skipped, void *stream) {
  return cpfn_adam_flat_sticky(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, pows, found_inf, coef3,
                               (unsigned *)nf_partial, nf_count, 0, skipped, stream);
}
