// The device-side linear assignment of the loss section (one wave per cloud), shared by losses.hip (its own launch,
// cpfn_hungarian_match) and fitters.hip (riding as extra workgroups on the moments launch, cpfn_fit_moments_fwd_match).
#pragma once
#include "common.h"

namespace {
constexpr int LSAP_MAXK = 32;      // instances per cloud of the fused loss kernels and of the solver riding on the fits (28 global / 21 local)
constexpr int LSAP_WIDE_MAXK = 64; // ... of the solver as its own launch (one lane per column): merged label sets of the evaluation cascade

// Cost as the reference builds it in fp32: D / clamp(cnt + col - D, 1e-10), negated for maximisation.
//
// Round 3: the solver's state lives in REGISTERS — lane j owns column j (v_j, shortest-path cost, predecessor, assigned row,
// visited flag, position in the list of unvisited columns), lane i owns row i (u_i, visited flag, assigned column) — and
// what a step needs from another lane comes by v_readlane with a wave-uniform index.  Only the cost matrix is in LDS (one
// conflict-free 8-byte read per step).  The first version kept every array in LDS and paid three workgroup barriers, a
// dozen dependent LDS round trips and six ds_bpermute steps of the 64-bit minimum per step: ~1500 cycles per step, 40 us per
// cloud, the longest workgroup of the launch it rides on.  The minimum is now four DPP steps inside the rows of 16 lanes
// and four v_readlane pairs across them.
__device__ __forceinline__ double lsap_dpp_f64(double x, const int ctrl_tag) {
  // (ctrl must be an immediate: one instantiation per pattern below)
  int lo = __double2loint(x), hi = __double2hiint(x);
  switch (ctrl_tag) {
    case 0: lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false); break;   // quad_perm [1,0,3,2]
    case 1: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false); break;   // quad_perm [2,3,0,1]
    case 2: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, false); break; // row_half_mirror
    default: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xF, 0xF, false); break; // row_mirror
  }
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lsap_readlane_f64(double x, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}
// minimum over the wave (no NaNs among the operands: the cost matrix is checked when it is built), wave-uniform result.
// v_min_f64 by name: fmin() adds a canonicalisation of each operand, and compare + two selects is three instructions.
__device__ __forceinline__ double lsap_min_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double wave_min_f64(double x) {
  x = lsap_min_f64(x, lsap_dpp_f64(x, 0));
  x = lsap_min_f64(x, lsap_dpp_f64(x, 1));
  x = lsap_min_f64(x, lsap_dpp_f64(x, 2));
  x = lsap_min_f64(x, lsap_dpp_f64(x, 3));          // every lane: the minimum of its row of 16
  x = lsap_min_f64(x, lsap_readlane_f64(x, 16));
  x = lsap_min_f64(x, lsap_readlane_f64(x, 32));
  x = lsap_min_f64(x, lsap_readlane_f64(x, 48));    // lanes 0..15: the minimum of the wave
  return lsap_readlane_f64(x, 0);
}
__device__ __forceinline__ int lsap_wave_max_i32(int x) {
  for (int m = 32; m >= 1; m >>= 1) x = max(x, __shfl_xor(x, m, 64));
  return x;
}

// One cloud, ONE wave (lanes 0..63 of the calling workgroup; every other wave of it must have left before the call: the
// workgroup barrier inside then counts this wave alone).
template <int MAXK = LSAP_MAXK>
__device__ __forceinline__ void lsap_one_cloud(const float *__restrict__ S, const long long *__restrict__ n_gt, int K,
                                               long long *__restrict__ match, int b, int lane) {
  static_assert(MAXK <= 64, "one lane per column");
  __shared__ double s_cost[MAXK][MAXK + 1];
  const int nc = K;
  long long nn = n_gt[b];
  const int nr = (int)(nn < 0 ? 0 : (nn > K ? K : nn));
  const float *Sb = S + (size_t)b * (K + 2) * K;
  bool bad = false;
  for (int e = lane; e < nr * nc; e += 64) {
    const int i = e / nc, j = e - i * nc;
    const float D = Sb[i * K + j], col = Sb[K * K + j], cnt = Sb[(K + 1) * K + i];
    const float den = (cnt + col) - D;
    const double c = -(double)(D / fmaxf(den, 1e-10f));
    bad |= !(fabs(c) < 1.7e308);                       // NaN / inf costs (SciPy raises): give up, never spin
    s_cost[i][j] = c;
  }
  __syncthreads();
  const double INF = __longlong_as_double(0x7ff0000000000000LL);
  bool failed = __ballot(bad) != 0ull;
  // column state of lane j = lane, row state of lane i = lane
  double v = 0.0, spc = INF, u = 0.0;
  int path = -1, row4col = -1, col4row = -1;
  for (int cur = 0; cur < nr && !failed; ++cur) {
    double min_val = 0.0;
    int num_remaining = nc;
    int pos = lane < nc ? nc - lane - 1 : -1;          // the list of unvisited columns starts as nc-1, nc-2, ..., 0
    bool SC = false, SR = false;
    spc = INF;
    int sink = -1, i = cur;
    while (sink == -1) {
      if (lane == i) SR = true;
      const bool active = lane < nc && !SC;
      double val = INF;
      const double u_i = lsap_readlane_f64(u, i);
      if (active) {
        const double r = ((min_val + s_cost[i][lane]) - u_i) - v;
        if (r < spc) { path = i; spc = r; }
        val = spc;
      }
      const double lowest = wave_min_f64(val);
      const bool cand = active && val == lowest;
      const unsigned long long bc = __ballot(cand);
      if (bc == 0) { failed = true; break; }
      int jsel;
      if ((bc & (bc - 1)) == 0) {
        jsel = __ffsll((long long)bc) - 1;
      } else {
        // sequential rule over the LIST: first position with the lowest value, replaced by every later position of equal
        // value whose column is unassigned -> the LAST unassigned one in list order if there is any, else the first
        const unsigned long long bu = __ballot(cand && row4col == -1);
        int want;
        if (bu) want = lsap_wave_max_i32((cand && row4col == -1) ? pos : -1);
        else want = -lsap_wave_max_i32(cand ? -pos : -0x7fffffff);
        jsel = __ffsll((long long)__ballot(cand && pos == want)) - 1;
      }
      min_val = lowest;
      const int index = __builtin_amdgcn_readlane(pos, jsel);
      const int r4c = __builtin_amdgcn_readlane(row4col, jsel);
      if (r4c == -1) sink = jsel; else i = r4c;
      // remove position `index` from the list: the column in the last position takes it
      if (pos == num_remaining - 1) pos = index;
      if (lane == jsel) { SC = true; pos = -1; }
      --num_remaining;
    }
    if (failed) break;
    // dual variables
    const double spc_of_my_col = __hiloint2double(__shfl(__double2hiint(spc), col4row < 0 ? 0 : col4row, 64),
                                                  __shfl(__double2loint(spc), col4row < 0 ? 0 : col4row, 64));
    if (lane == cur) u += min_val;
    else if (lane < nr && SR) u += min_val - spc_of_my_col;
    if (lane < nc && SC) v -= min_val - spc;
    // augment along the path (wave-uniform walk)
    int j = sink;
    while (true) {
      const int ii = __builtin_amdgcn_readlane(path, j);
      if (lane == j) row4col = ii;
      const int t = __builtin_amdgcn_readlane(col4row, ii);
      if (lane == ii) col4row = j;
      j = t;
      if (ii == cur) break;
    }
  }
  if (lane < K) match[(size_t)b * K + lane] = lane < nr ? (failed ? (long long)lane : (long long)col4row) : 0LL;
}
}  // namespace
