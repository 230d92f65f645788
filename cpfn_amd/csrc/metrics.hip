// Evaluation metrics of SPFN (SPFN/metric_implementation.py:485-514, compute_all_metrics) as three launches around the
// assignment and the fits, instead of ~200 framework launches on [B,N,K] / [B,K,N',4] expansions:
//
//   metrics_points_kernel   one pass over the points: arg-max membership -> hard one-hot W (:33-37), the joint histogram
//                           (GT label x predicted label) that IS the assignment's cost input for one-hot memberships
//                           (:19-25), the per-instance type scores W^T T (:52-55), the normal difference (:170-172)
//   metrics_finish_kernel   [B,K]-sized: counters -> S[B,K+2,K] (the layout cpfn_hungarian_match reads), n_gt, instance types
//   metrics_tail_kernel     after the assignment and the fits: matched IoU, type accuracy, axis difference, and per GT
//                           instance the sqrt-safe residues of its N' points against the matched fit (mean, unbiased std,
//                           coverage per epsilon), reduced to the per-cloud figures — everything that was [B,K]-sized glue.
//
// Integer work (histogram, counts, arg-max indices) is exact; float sums are taken in a fixed order: the type scores of a
// block by the lane that owns a (label, type) entry walking the tile's rows in row order (round 5; LDS float atomics before:
// the order, and with it a near-tie's arg-max, changed from run to run — ADVICE r4), then in chunk order in fp64.
#include "common.h"
#include "residue.h"

namespace {

constexpr int MP_THREADS = 256, MP_TILE = 64, MP_TILES = 4;      // a workgroup = 256 rows as four 64-row tiles
constexpr int MP_MAXK = 128, MP_MAXT = 8, MT_MAXEPS = 4;

// dynamic LDS: tile [64 * K] floats | labels [64] | hist [(Kp + 2) * Kp] ints | type sums [Kp * NT] floats | 4 floats |
//              the tile's type rows [64 * NT] floats
__global__ __launch_bounds__(MP_THREADS) void metrics_points_kernel(
    const float *__restrict__ W, const float *__restrict__ T, const float *__restrict__ X, const float *__restrict__ Xgt,
    const long long *__restrict__ Igt, int N, int K, int Kp, int NT, float *__restrict__ hardW, int *__restrict__ counters,
    float *__restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *s_tile = (float *)smem;
  int *s_lab = (int *)(s_tile + MP_TILE * K);
  int *s_hist = s_lab + MP_TILE;
  float *s_ts = (float *)(s_hist + (Kp + 2) * Kp);
  float *s_nd = s_ts + Kp * NT;
  float *s_T = s_nd + 4;
  __shared__ int s_lmax;
  const int b = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
  const int nh = (Kp + 2) * Kp;
  for (int e = t; e < nh; e += MP_THREADS) s_hist[e] = 0;
  for (int e = t; e < Kp * NT; e += MP_THREADS) s_ts[e] = 0.f;
  if (t == 0) s_lmax = 0;
  const int r = t >> 2, sub = t & 3;
  float nd = 0.f;
  int lmax1 = 0;
  for (int tile = 0; tile < MP_TILES; ++tile) {
    const long long row0 = (long long)chunk * (MP_TILE * MP_TILES) + tile * MP_TILE;
    if (row0 >= N) break;
    const int rows = (int)min((long long)MP_TILE, N - row0);
    __syncthreads();
    const float *src = W + ((size_t)b * N + row0) * K;
    for (int e = t; e < rows * K; e += MP_THREADS) s_tile[e] = src[e];
    const float *srcT = T + ((size_t)b * N + row0) * NT;
    for (int e = t; e < rows * NT; e += MP_THREADS) s_T[e] = srcT[e];
    __syncthreads();
    // arg-max of a row by four lanes (first index wins a tie, like torch.argmax)
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    if (r < rows)
      for (int k = sub; k < K; k += 4) {
        const float v = s_tile[r * K + k];
        if (v > bv || bi == 0x7fffffff) { bv = v; bi = k; }
      }
    for (int m = 1; m <= 2; m <<= 1) {
      const float ov = __shfl_xor(bv, m, 64);
      const int oi = __shfl_xor(bi, m, 64);
      if (ov > bv || (ov == bv && oi < bi) || bi == 0x7fffffff) { bv = ov; bi = oi; }
    }
    if (sub == 0 && r < rows) {
      const int lab = bi == 0x7fffffff ? 0 : bi;
      s_lab[r] = lab;
      const size_t n = (size_t)b * N + row0 + r;
      const long long g = Igt[n];
      atomicAdd(&s_hist[Kp * Kp + lab], 1);                                 // row Kp: points per predicted label
      if (g >= 0 && g < Kp) {
        atomicAdd(&s_hist[(int)g * Kp + lab], 1);                          // rows g < Kp: joint histogram
        atomicAdd(&s_hist[(Kp + 1) * Kp + (int)g], 1);                     // row Kp + 1: points per GT label
        lmax1 = max(lmax1, (int)g + 1);
      }
      const float d = fabsf(X[n * 3] * Xgt[n * 3] + X[n * 3 + 1] * Xgt[n * 3 + 1] + X[n * 3 + 2] * Xgt[n * 3 + 2]);
      nd += acosf(fminf(fmaxf(d, -1.0f + 1e-6f), 1.0f - 1e-6f));
    }
    __syncthreads();
    // type scores W^T T of the tile: entry (label k, type c) belongs to ONE lane for the whole launch, which adds the rows that
    // chose k in row order — a fixed order of additions (no float atomics)
    for (int e = t; e < Kp * NT; e += MP_THREADS) {
      const int k = e / NT, c = e - k * NT;
      float acc = s_ts[e];
      for (int rr = 0; rr < rows; ++rr) acc += s_lab[rr] == k ? s_T[rr * NT + c] : 0.f;
      s_ts[e] = acc;
    }
    float *dst = hardW + ((size_t)b * N + row0) * Kp;
    for (int e = t; e < rows * Kp; e += MP_THREADS) {
      const int rr = e / Kp;
      dst[e] = (e - rr * Kp) == s_lab[rr] ? 1.f : 0.f;
    }
  }
  for (int m = 32; m >= 1; m >>= 1) {
    nd += __shfl_xor(nd, m, 64);
    lmax1 = max(lmax1, __shfl_xor(lmax1, m, 64));
  }
  if ((t & 63) == 0) { s_nd[t >> 6] = nd; atomicMax(&s_lmax, lmax1); }
  __syncthreads();
  int *cnt = counters + (size_t)b * (nh + 1);
  for (int e = t; e < nh; e += MP_THREADS) {
    const int v = s_hist[e];
    if (v) atomicAdd(&cnt[e], v);                                          // integers: exact, order-free
  }
  if (t == 0 && s_lmax > 0) atomicMax(&cnt[nh], s_lmax);
  float *po = partial + ((size_t)b * gridDim.x + chunk) * (Kp * NT + 1);
  for (int e = t; e < Kp * NT; e += MP_THREADS) po[e] = s_ts[e];
  if (t == 0) po[Kp * NT] = (s_nd[0] + s_nd[1]) + (s_nd[2] + s_nd[3]);
}

__global__ __launch_bounds__(256) void metrics_finish_kernel(const int *__restrict__ counters, const float *__restrict__ partial,
                                                             int chunks, int N, int Kp, int NT, float *__restrict__ S,
                                                             long long *__restrict__ n_gt, long long *__restrict__ T_inst,
                                                             float *__restrict__ normal_diff) {
  __shared__ double s_sum[MP_MAXK * MP_MAXT + 1];
  const int b = blockIdx.x, t = threadIdx.x;
  const int nh = (Kp + 2) * Kp, ne = Kp * NT + 1;
  const int *cnt = counters + (size_t)b * (nh + 1);
  for (int e = t; e < nh; e += 256) S[(size_t)b * nh + e] = (float)cnt[e];
  if (t == 0) n_gt[b] = cnt[nh];
  for (int e = t; e < ne; e += 256) {
    const float *p = partial + (size_t)b * chunks * ne + e;
    double s = 0.0;
#pragma unroll 8
    for (int c = 0; c < chunks; ++c) s += (double)p[(size_t)c * ne];
    s_sum[e] = s;
  }
  __syncthreads();
  for (int k = t; k < Kp; k += 256) {
    int best = 0;
    double bv = s_sum[k * NT];
    for (int c = 1; c < NT; ++c)
      if (s_sum[k * NT + c] > bv) { bv = s_sum[k * NT + c]; best = c; }
    T_inst[(size_t)b * Kp + k] = best;
  }
  if (t == 0) normal_diff[b] = (float)(s_sum[Kp * NT] / (double)N);
}

struct MtEps { float e[MT_MAXEPS]; };

// One workgroup per cloud; one WAVE per GT slot at a time (its N' points over the lanes, two passes: mean, then variance and
// coverage counts), then the [K]-sized sums in instance order by one lane.
// out[b] = (mIoU, type accuracy, axis difference, mean residual, std residual, Sk coverage[n_eps])
// dynamic LDS: 5 + MT_MAXEPS floats per slot
__global__ __launch_bounds__(256) void metrics_tail_kernel(
    const float *__restrict__ S, const long long *__restrict__ match, const long long *__restrict__ n_gt,
    const long long *__restrict__ T_inst, const long long *__restrict__ T_gt, const float *__restrict__ params,
    const float *__restrict__ ppi, const float *__restrict__ ax_plane, const float *__restrict__ ax_cyl,
    const float *__restrict__ ax_cone, int Kp, int Kgt, int Np, int tid_plane, int tid_sphere, int tid_cyl, int tid_cone,
    MtEps eps, int n_eps, float *__restrict__ out, long long *__restrict__ slot_type) {
  extern __shared__ float s_k[];                   // [Kp][5 + MT_MAXEPS]: iou, type hit, axis weight, axis loss, mean, std, cov...
  constexpr int F = 6 + MT_MAXEPS;
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  long long nn = n_gt[b];
  const int n = (int)(nn < 0 ? 0 : (nn > Kp ? Kp : nn));
  const float *Sb = S + (size_t)b * (Kp + 2) * Kp;
  for (int k = wave; k < Kp; k += 4) {
    long long m = match[(size_t)b * Kp + k];
    m = m < 0 ? 0 : (m >= Kp ? Kp - 1 : m);
    const long long tgt = k < Kgt ? T_gt[(size_t)b * Kgt + k] : 0;
    const long long t_m = T_inst[(size_t)b * Kp + m];
    const float *P22 = params + ((size_t)b * Kp + m) * 22;
    // axis agreement of the matched fit with the GT axis of slot k, by GT type (losses_implementation.py:480-497 with
    // is_eval: the angle); slots beyond the GT table carry zero axes, like the reference's padding (:505-508)
    float pl = 0.f;
    if (tgt != tid_sphere) {
      const float *ax = tgt == tid_plane ? ax_plane : (tgt == tid_cyl ? ax_cyl : (tgt == tid_cone ? ax_cone : nullptr));
      const int off = tgt == tid_plane ? 0 : (tgt == tid_cyl ? 8 : 18);
      float d = 0.f;
      if (ax && k < Kgt) {
        const float *a = ax + ((size_t)b * Kgt + k) * 3;
        d = fabsf(P22[off] * a[0] + P22[off + 1] * a[1] + P22[off + 2] * a[2]);
      }
      pl = ax ? acosf(fminf(fmaxf(d, -1.0f + 1e-6f), 1.0f - 1e-6f)) : 0.f;
    }
    float mean = 0.f, sd = 0.f, cov[MT_MAXEPS] = {0, 0, 0, 0};
    if (k < n) {
      const int kind = tgt == tid_plane ? 0 : (tgt == tid_sphere ? 1 : (tgt == tid_cyl ? 2 : 3));
      const int off = kind == 0 ? 0 : (kind == 1 ? 4 : (kind == 2 ? 8 : 15));
      const int nq = kind == 0 ? 4 : (kind == 1 ? 4 : 7);
      float q[8];
      for (int i = 0; i < 8; ++i) q[i] = i < nq ? P22[off + i] : 0.f;
      const float *pts = k < Kgt ? ppi + ((size_t)b * Kgt + k) * Np * 3 : nullptr;
      float s = 0.f;
      for (int p = lane; p < Np; p += 64) {
        const float x = pts ? pts[p * 3] : 0.f, y = pts ? pts[p * 3 + 1] : 0.f, z = pts ? pts[p * 3 + 2] : 0.f;
        s += sqrt_safe_f(residue_value(kind, q, x, y, z));
      }
      for (int msk = 32; msk >= 1; msk >>= 1) s += __shfl_xor(s, msk, 64);
      mean = s / (float)Np;
      float v = 0.f;
      for (int p = lane; p < Np; p += 64) {
        const float x = pts ? pts[p * 3] : 0.f, y = pts ? pts[p * 3 + 1] : 0.f, z = pts ? pts[p * 3 + 2] : 0.f;
        const float rr = sqrt_safe_f(residue_value(kind, q, x, y, z));
        v += (rr - mean) * (rr - mean);
        for (int i = 0; i < MT_MAXEPS; ++i) cov[i] += (i < n_eps && rr < eps.e[i]) ? 1.f : 0.f;
      }
      for (int msk = 32; msk >= 1; msk >>= 1) {
        v += __shfl_xor(v, msk, 64);
        for (int i = 0; i < MT_MAXEPS; ++i) cov[i] += __shfl_xor(cov[i], msk, 64);
      }
      sd = sqrtf(v / (float)(Np - 1));                                    // torch.std: unbiased
    }
    if (lane == 0) {
      float *o = s_k + k * F;
      // matched relaxed IoU from the joint histogram: 1 - (1 - dot / (den + 1e-10))   (losses_implementation.py:77-90,
      // metric_implementation.py:119-121)
      const float dot = Sb[k * Kp + m], colv = Sb[Kp * Kp + m], cntv = Sb[(Kp + 1) * Kp + k];
      const float loss = 1.0f - dot / ((cntv + colv - dot) + 1e-10f);
      o[0] = 1.0f - loss;
      o[1] = t_m == tgt ? 1.f : 0.f;
      o[2] = T_inst[(size_t)b * Kp + k] == tgt ? 1.f : 0.f;               // (the reference compares the UNmatched type here)
      o[3] = pl;
      o[4] = mean;
      o[5] = sd;
      for (int i = 0; i < MT_MAXEPS; ++i) o[6 + i] = cov[i] / (float)Np;
      slot_type[(size_t)b * Kp + k] = t_m;
    }
  }
  __syncthreads();
  if (t < 5 + n_eps) {
    // t: 0 mIoU, 1 type accuracy, 2 axis difference, 3 mean residual, 4 std residual, 5.. Sk coverage
    float num = 0.f, den = 0.f;
    if (t == 2) {
      for (int k = 0; k < Kp; ++k) {
        const float *o = s_k + k * F;
        if (k < n) num += o[2] * o[3];
        den += o[3];                                                       // (over ALL slots, unmasked: :189-193)
      }
      den = fmaxf(den, 1e-10f);
    } else {
      const int f = t == 0 ? 0 : (t == 1 ? 1 : (t == 3 ? 4 : (t == 4 ? 5 : 6 + (t - 5))));
      for (int k = 0; k < n; ++k) num += s_k[k * F + f];
      den = (float)n;
    }
    out[(size_t)b * (5 + n_eps) + t] = num / den;
  }
}

}  // namespace

extern "C" long long cpfn_metrics_workspace(int B, int N, int Kp, int n_types) {
  if (B <= 0 || N <= 0 || Kp <= 0 || Kp > MP_MAXK || n_types <= 0 || n_types > MP_MAXT) return -1;
  const long long chunks = cpfn_cdiv(N, MP_TILE * MP_TILES);
  const long long ints = (long long)B * ((Kp + 2) * Kp + 1);
  return ((ints * 4 + 15) / 16) * 16 + 4ll * B * chunks * (Kp * n_types + 1);
}

extern "C" int cpfn_metrics_points(const float *W, const float *T, const float *X, const float *Xgt, const int64_t *Igt, int B,
                                   int N, int K, int Kp, int n_types, float *hardW, void *workspace, float *S, int64_t *n_gt,
                                   int64_t *T_inst, float *normal_diff, void *stream) {
  if (B <= 0 || N <= 0 || K <= 0 || Kp < K || Kp > MP_MAXK || n_types <= 0 || n_types > MP_MAXT || !W || !T || !X || !Xgt ||
      !Igt || !hardW || !workspace || !S || !n_gt || !T_inst || !normal_diff)
    return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = cpfn_cdiv(N, MP_TILE * MP_TILES);
  const long long ints = (long long)B * ((Kp + 2) * Kp + 1);
  int *counters = (int *)workspace;
  float *partial = (float *)((char *)workspace + ((ints * 4 + 15) / 16) * 16);
  if (hipMemsetAsync(counters, 0, ints * 4, st) != hipSuccess) return (int)hipGetLastError();
  const size_t lds = sizeof(float) * (MP_TILE * K + Kp * n_types + 4 + MP_TILE * n_types) + sizeof(int) * (MP_TILE + (Kp + 2) * Kp);
  // Kp = 128 needs 103 KB: past 64 KB a launch needs the kernel's dynamic-LDS limit raised first (as csr_build, moments_bwd and
  // the sampling kernel's LDS claim do) — label sets of 98..128 columns, which evaluation_localSPFN's merged sets can reach,
  // would otherwise fail at launch (tests/test_gpu_metrics.py runs K = 100 and K = 128)
  static size_t lds_allowed = 64 * 1024;
  if (lds > lds_allowed) {
    if (hipFuncSetAttribute((const void *)metrics_points_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) !=
        hipSuccess)
      return (int)hipGetLastError();
    lds_allowed = 160 * 1024 - 64;
  }
  metrics_points_kernel<<<dim3(chunks, B), MP_THREADS, lds, st>>>(W, T, X, Xgt, (const long long *)Igt, N, K, Kp, n_types, hardW,
                                                                   counters, partial);
  metrics_finish_kernel<<<B, 256, 0, st>>>(counters, partial, chunks, N, Kp, n_types, S, (long long *)n_gt, (long long *)T_inst,
                                           normal_diff);
  return cpfn_launch_status();
}

extern "C" int cpfn_metrics_tail(const float *S, const int64_t *match, const int64_t *n_gt, const int64_t *T_inst,
                                 const int64_t *T_gt, const float *params22, const float *ppi, const float *axis_plane,
                                 const float *axis_cylinder, const float *axis_cone, int B, int Kp, int Kgt, int Np,
                                 const int *type_ids, const float *eps, int n_eps, float *out, int64_t *slot_type, void *stream) {
  if (B <= 0 || Kp <= 0 || Kp > 1024 || Kgt <= 0 || Np <= 1 || n_eps < 0 || n_eps > MT_MAXEPS || !S || !match || !n_gt || !T_inst ||
      !T_gt || !params22 || !ppi || !axis_plane || !axis_cylinder || !axis_cone || !type_ids || (n_eps > 0 && !eps) || !out ||
      !slot_type)
    return CPFN_EINVAL;
  MtEps pe;
  for (int i = 0; i < MT_MAXEPS; ++i) pe.e[i] = i < n_eps ? eps[i] : 0.f;
  const size_t lds = sizeof(float) * Kp * (6 + MT_MAXEPS);
  metrics_tail_kernel<<<B, 256, lds, (hipStream_t)stream>>>(
      S, (const long long *)match, (const long long *)n_gt, (const long long *)T_inst, (const long long *)T_gt, params22, ppi,
      axis_plane, axis_cylinder, axis_cone, Kp, Kgt, Np, type_ids[0], type_ids[1], type_ids[2], type_ids[3], pe, n_eps, out,
      (long long *)slot_type);
  return cpfn_launch_status();
}
