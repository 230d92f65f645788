// The device-side linear assignment of the loss section (one wave per cloud), shared by losses.hip (its own launch,
// cpfn_hungarian_match) and fitters.hip (riding as extra workgroups on the moments launch, cpfn_fit_moments_fwd_match).
#pragma once
#include "common.h"

namespace {
constexpr int LSAP_MAXK = 32;      // instances per cloud (28 global / 21 local)

// Cost as the reference builds it in fp32: D / clamp(cnt + col - D, 1e-10), negated for maximisation.
__device__ __forceinline__ double wave_min_f64(double x) {
  for (int m = 32; m >= 1; m >>= 1) {
    const unsigned long long o = cpfn_shfl_xor_u64((unsigned long long)__double_as_longlong(x), m);
    const double y = __longlong_as_double((long long)o);
    x = y < x ? y : x;
  }
  return x;
}

// One cloud, ONE wave (lanes 0..63 of the calling workgroup; every other wave of it must have left before the call: the
// workgroup barriers inside then count this wave alone).
__device__ __forceinline__ void lsap_one_cloud(const float *__restrict__ S, const long long *__restrict__ n_gt, int K,
                                               long long *__restrict__ match, int b, int lane) {
  constexpr int MAXK = LSAP_MAXK;
  __shared__ double s_cost[MAXK][MAXK + 1];
  __shared__ double s_u[MAXK], s_v[MAXK], s_spc[MAXK];
  __shared__ int s_path[MAXK], s_col4row[MAXK], s_row4col[MAXK], s_rem[MAXK], s_SR[MAXK], s_SC[MAXK];
  const int nc = K;
  long long nn = n_gt[b];
  const int nr = (int)(nn < 0 ? 0 : (nn > K ? K : nn));
  const float *Sb = S + (size_t)b * (K + 2) * K;
  for (int e = lane; e < nr * nc; e += 64) {
    const int i = e / nc, j = e - i * nc;
    const float D = Sb[i * K + j], col = Sb[K * K + j], cnt = Sb[(K + 1) * K + i];
    const float den = (cnt + col) - D;
    s_cost[i][j] = -(double)(D / fmaxf(den, 1e-10f));
  }
  if (lane < MAXK) { s_u[lane] = 0.0; s_v[lane] = 0.0; s_path[lane] = -1; s_col4row[lane] = -1; s_row4col[lane] = -1; }
  __syncthreads();
  const double INF = __longlong_as_double(0x7ff0000000000000LL);
  bool failed = false;
  for (int cur = 0; cur < nr && !failed; ++cur) {
    double min_val = 0.0;
    int num_remaining = nc;
    if (lane < nc) { s_rem[lane] = nc - lane - 1; s_SC[lane] = 0; s_spc[lane] = INF; }
    if (lane < nr) s_SR[lane] = 0;
    __syncthreads();
    int sink = -1, i = cur;
    while (sink == -1) {
      if (lane == 0) s_SR[i] = 1;
      double val = INF;
      int j = 0;
      const bool active = lane < num_remaining;       // lane = position `it` in the list of unvisited columns
      if (active) {
        j = s_rem[lane];
        const double r = ((min_val + s_cost[i][j]) - s_u[i]) - s_v[j];
        if (r < s_spc[j]) { s_path[j] = i; s_spc[j] = r; }
        val = s_spc[j];
      }
      const double lowest = wave_min_f64(val);
      const bool cand = active && val == lowest;
      const unsigned long long bc = __ballot(cand), bu = __ballot(cand && s_row4col[j] == -1);
      // sequential rule: first position with the lowest value, replaced by every later position of equal
      // value whose column is unassigned -> the LAST unassigned one if there is any, else the first
      if (bc == 0) { failed = true; break; }          // NaN / inf costs (SciPy raises): give up, never spin
      const int index = bu ? 63 - __clzll((long long)bu) : __ffsll((long long)bc) - 1;
      min_val = lowest;
      __syncthreads();
      const int jsel = s_rem[index];
      const int r4c = s_row4col[jsel];
      if (r4c == -1) sink = jsel; else i = r4c;
      __syncthreads();
      if (lane == 0) { s_SC[jsel] = 1; s_rem[index] = s_rem[num_remaining - 1]; }
      --num_remaining;
      __syncthreads();
    }
    if (failed) break;
    // dual variables
    if (lane == 0) s_u[cur] += min_val;
    if (lane < nr && lane != cur && s_SR[lane]) s_u[lane] += min_val - s_spc[s_col4row[lane]];
    if (lane < nc && s_SC[lane]) s_v[lane] -= min_val - s_spc[lane];
    __syncthreads();
    // augment along the path
    if (lane == 0) {
      int j = sink;
      while (true) {
        const int ii = s_path[j];
        s_row4col[j] = ii;
        const int t = s_col4row[ii];
        s_col4row[ii] = j;
        j = t;
        if (ii == cur) break;
      }
    }
    __syncthreads();
  }
  __syncthreads();
  if (lane < K) match[(size_t)b * K + lane] = lane < nr ? (failed ? (long long)lane : (long long)s_col4row[lane]) : 0LL;
}
}  // namespace
