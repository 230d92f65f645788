// BatchNorm seams WITHOUT a finalize launch (round 6; priced in round 5, profiles/r05_atomic_seam.txt).
//
// Between a layer's GEMM and whatever consumes its output sits a grid-wide dependency: the batch statistics.  Rounds 1-5 paid
// for it with a launch: the producer leaves per-workgroup partial rows, a [C]-sized kernel sums them in a fixed order and writes
// scale / shift (cpfn_bn_finalize), the consumer reads those.  Here the sums leave the PRODUCER as no-return 64-bit FIXED-POINT
// atomics into `replicas` copies of a [2][C] accumulator (replica = workgroup % replicas: ~50 same-address adds per word instead
// of ~400), and every workgroup of the CONSUMER folds the replicas into scale / shift in its prologue — the kernel boundary the
// two share anyway is the only synchronisation.  Integer addition commutes: the statistics are bit-reproducible from run to run,
// like the ordered sums they replace (and equal to them to ~1e-7 relative).
//
// What still needs the old launch: consumers whose workgroups are too small to fold C channels each (the stand-alone BatchNorm
// apply / max-pool passes over a stack's LAST layer) and the backward pass, whose finalize launches carry the weight-gradient
// split reductions as riders.
#pragma once
#include "common.h"

struct SeamOut {               // producer side (kernel argument, by value); acc == nullptr: off
  long long *acc;              // [replicas][2][C] sums + one poison word behind them; ZERO before the launch
  int replicas;
  float fx;                    // 2^log2_scale
  long long *counter_a;        // step counters the launch advances (workgroup 0): the layer's num_batches_tracked,
  long long *counter_b;        //   the step counter of a fused dropout
};
struct SeamIn {                // consumer side; acc == nullptr: off
  const long long *acc;
  int replicas;
  float inv_fx;
  int C;
  float count, eps, momentum;
  const float *gamma, *beta, *conv_bias;
  float *running_mean, *running_var;     // (nullable) updated by ONE workgroup of the consumer
  float *stats;                          // [4][C] scale, shift, mean, rstd: written by that workgroup (the backward pass reads them)
};

// A partial sum that is NaN / inf, or too large for 2048 of its kind to stay inside 63 bits, poisons the seam: the consumer
// then sees NaN statistics, exactly what the fp32 path would have propagated (and the trainer's finite check skips the step).
__device__ __forceinline__ void seam_add(const SeamOut &o, int C, int which, int c, float v, unsigned wg) {
  const float s = v * o.fx;
  if (!(fabsf(s) < 1.125899906842624e15f /* 2^50 */)) {
    __hip_atomic_fetch_add((unsigned long long *)&o.acc[(size_t)o.replicas * 2 * C], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const long long q = __float2ll_rn(s);
  __hip_atomic_fetch_add((unsigned long long *)&o.acc[((size_t)(wg % (unsigned)o.replicas) * 2 + which) * C + c], (unsigned long long)q,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void seam_counters(const SeamOut &o) {      // (one lane of one workgroup calls this)
  if (o.counter_a) ++*o.counter_a;
  if (o.counter_b) ++*o.counter_b;
}

// scale / shift of channel c from the producer's sums: the arithmetic of bn_finalize_kernel (bn.hip) on the folded totals.
// Two phases, so that a consumer can put its own first requests between them (a wave's loads return in order: issued FIRST, the
// accumulator words are there by the time the consumer's first operand rows are):
//   seam_fold_issue: every load of the channel in flight;   seam_fold_finish: the arithmetic.
// writer: this lane also leaves [scale, shift, mean, rstd] in s.stats and updates the running statistics.
// MAXR: the most replicas the caller's producers use (the held words cost 4 MAXR + 4 registers per lane between the phases)
template <int MAXR = 8>
struct SeamFoldRegsT { long long v1[MAXR], v2[MAXR], poison; float g, b; };
typedef SeamFoldRegsT<8> SeamFoldRegs;
template <int MAXR>
__device__ __forceinline__ void seam_fold_issue(const SeamIn &s, int c, SeamFoldRegsT<MAXR> &q) {
  const int C = s.C;
#pragma unroll
  for (int r = 0; r < MAXR; ++r) {        // (replicas <= MAXR; surplus slots re-read the last one: a cache hit)
    const int rr = r < s.replicas ? r : s.replicas - 1;
    q.v1[r] = s.acc[((size_t)rr * 2 + 0) * C + c];
    q.v2[r] = s.acc[((size_t)rr * 2 + 1) * C + c];
  }
  q.poison = s.acc[(size_t)s.replicas * 2 * C];
  q.g = s.gamma[c]; q.b = s.beta[c];
}
template <int MAXR>
__device__ __forceinline__ void seam_fold_finish(const SeamIn &s, int c, bool writer, const SeamFoldRegsT<MAXR> &q, float &scale, float &shift) {
  const int C = s.C;
  const long long poison = q.poison;
  const float g_ = q.g, b_ = q.b;
  long long a1 = 0, a2 = 0;
#pragma unroll
  for (int r = 0; r < MAXR; ++r)
    if (r < s.replicas) { a1 += q.v1[r]; a2 += q.v2[r]; }
  const double s1 = (double)a1 * (double)s.inv_fx, s2 = (double)a2 * (double)s.inv_fx;
  double mean = s1 / s.count;
  double var = s2 / s.count - mean * mean;
  var = var > 0.0 ? var : 0.0;
  if (poison != 0) { mean = __builtin_nan(""); var = __builtin_nan(""); }
  const float rstd = (float)(1.0 / sqrt(var + (double)s.eps));
  const float sc = g_ * rstd;
  scale = sc;
  shift = b_ - (float)mean * sc;
  if (writer) {
    s.stats[c] = sc;
    s.stats[C + c] = shift;
    s.stats[2 * C + c] = (float)mean;
    s.stats[3 * C + c] = rstd;
    if (s.running_mean) {
      const float cb = s.conv_bias ? s.conv_bias[c] : 0.f;
      s.running_mean[c] = (1.f - s.momentum) * s.running_mean[c] + s.momentum * ((float)mean + cb);
      const double unbiased = s.count > 1.f ? var * (double)s.count / ((double)s.count - 1.0) : var;
      s.running_var[c] = (1.f - s.momentum) * s.running_var[c] + s.momentum * (float)unbiased;
    }
  }
}
__device__ __forceinline__ void seam_fold_fwd(const SeamIn &s, int c, bool writer, float &scale, float &shift) {
  SeamFoldRegs q;
  seam_fold_issue(s, c, q);
  seam_fold_finish(s, c, writer, q, scale, shift);
}

// host side: the C-ABI descriptors (include/cpfn_hip.h) -> kernel arguments
#include "../../include/cpfn_hip.h"
static inline SeamOut seam_out_arg(const cpfn_seam_out *o) {
  SeamOut s;
  s.acc = o ? o->acc : nullptr;
  s.replicas = o ? o->replicas : 1;
  s.fx = o ? ldexpf(1.f, o->log2_scale) : 1.f;
  s.counter_a = o ? o->counter_a : nullptr;
  s.counter_b = o ? o->counter_b : nullptr;
  return s;
}
static inline SeamIn seam_in_arg(const cpfn_seam_in *i) {
  SeamIn s;
  s.acc = i ? i->acc : nullptr;
  s.replicas = i ? i->replicas : 1;
  s.inv_fx = i ? ldexpf(1.f, -i->log2_scale) : 1.f;
  s.C = i ? i->C : 0;
  s.count = i ? i->count : 1.f;
  s.eps = i ? i->eps : 0.f;
  s.momentum = i ? i->momentum : 0.f;
  s.gamma = i ? i->gamma : nullptr;
  s.beta = i ? i->beta : nullptr;
  s.conv_bias = i ? i->conv_bias : nullptr;
  s.running_mean = i ? i->running_mean : nullptr;
  s.running_var = i ? i->running_var : nullptr;
  s.stats = i ? i->stats : nullptr;
  return s;
}
static inline bool seam_out_valid(const cpfn_seam_out *o) {
  return !o || (o->acc && o->replicas >= 1 && o->replicas <= 8 && o->log2_scale >= 0 && o->log2_scale <= 44);
}
static inline bool seam_in_valid(const cpfn_seam_in *i, int C) {
  return !i || (i->acc && i->replicas >= 1 && i->replicas <= 8 && i->log2_scale >= 0 && i->log2_scale <= 44 && i->C == C &&
                i->gamma && i->beta && i->stats && i->count > 0.f && (!i->running_mean == !i->running_var));
}
