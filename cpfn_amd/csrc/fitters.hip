// Weighted-moment kernels behind the SPFN primitive fitters (plane / sphere / cylinder /
// cone) for gfx950.
//
// The reference tiles P and X to [B*K, N, 3] copies and builds [B*K, N, 3, 3] outer
// products per fitter (SPFN/plane_fitter.py:12-13, SPFN/differentiable_tls.py:205-207);
// at B=16, K=28, N=8192 that is ~130 MB per temporary.  Every quantity the four fitters
// need is a weighted sum over the N points of a low-order polynomial of (p, x), so one
// pass over P, X, W produces ALL of them for all K instances:
//
//   M[b,k,m] = Σ_n  ω_m(W[b,n,k]) · φ_m(P[b,n], X[b,n])          (a [K x N]·[N x 52] product)
//
// with ω = w for the "A" slots and ω = max(w, 1e-10) for the "B" slots (the reference's
// sqrt(clamp(W)) row scaling, SPFN/geometry_utils.py:127).  Sums are accumulated in fp64
// — raw (uncentred) moments would otherwise lose the thin direction of a plane — per
// point-chunk, then reduced over chunks in a fixed order (bitwise reproducible, no atomics).
// The tiny per-instance algebra (3x3 eigenvectors, guarded solves) runs on these
// [B,K,52] moments (cpfn_amd/SPFN/geometry_utils.py); its adjoint comes back as G = dL/dM
// and `moments_bwd` expands it to dL/dW [B,N,K] and dL/dX [B,N,3] in one pass.
//
// Slot map (FM_SLOTS = 52):
//   A (ω = w):            0: 1 | 1-3: p | 4-9: p⊗p (xx xy xz yy yz zz) | 10-12: x | 13-18: x⊗x | 19: pad
//   B (ω = max(w,1e-10)): 20: 1 | 21-23: p | 24-29: p⊗p | 30-39: p⊗p⊗p (xxx xxy xxz xyy xyz xzz yyy yyz yzz zzz)
//                         40-45: x⊗x | 46-48: x·(p·x) | 49-51: pad
//
// The cone's half-angle needs the fitted apex/axis first, so it is a second pass
// (`cone_pass_*`) over P and W only (SPFN/cone_fitter.py:25-35).
#include "common.h"
#include "lsap.h"
#include "fit_pack.h"
#include "fit_internal.h"

namespace {

constexpr int FM_SLOTS = 52;
constexpr int FM_A_GROUPS = 5;   // 4-slot groups weighted by w
constexpr int FM_GROUPS = 13;    // 52 / 4
constexpr int FM_TILE = 64;      // points staged per iteration
constexpr int FM_KB = 32;        // instances per k-block
constexpr int FM_THREADS = 256;
constexpr float FM_WEPS = 1e-10f;

template <typename T>
__device__ __forceinline__ void point_features(const float *p, const float *x, T *f) {
  const T px = p[0], py = p[1], pz = p[2], nx = x[0], ny = x[1], nz = x[2];
  const T xx = px * px, xy = px * py, xz = px * pz, yy = py * py, yz = py * pz, zz = pz * pz;
  const T pn = px * nx + py * ny + pz * nz;
  f[0] = 1; f[1] = px; f[2] = py; f[3] = pz;
  f[4] = xx; f[5] = xy; f[6] = xz; f[7] = yy; f[8] = yz; f[9] = zz;
  f[10] = nx; f[11] = ny; f[12] = nz;
  f[13] = nx * nx; f[14] = nx * ny; f[15] = nx * nz; f[16] = ny * ny; f[17] = ny * nz; f[18] = nz * nz;
  f[19] = 0;
  f[20] = 1; f[21] = px; f[22] = py; f[23] = pz;
  f[24] = xx; f[25] = xy; f[26] = xz; f[27] = yy; f[28] = yz; f[29] = zz;
  f[30] = xx * px; f[31] = xx * py; f[32] = xx * pz; f[33] = xy * py; f[34] = xy * pz; f[35] = xz * pz;
  f[36] = yy * py; f[37] = yy * pz; f[38] = yz * pz; f[39] = zz * pz;
  f[40] = f[13]; f[41] = f[14]; f[42] = f[15]; f[43] = f[16]; f[44] = f[17]; f[45] = f[18];
  f[46] = nx * pn; f[47] = ny * pn; f[48] = nz * pn;
  f[49] = 0; f[50] = 0; f[51] = 0;
}

// grid (chunks, B); partial[b][chunk][K][52]
// Staging is software-pipelined and spread over the whole workgroup: the next tile's coordinates, normals and
// memberships are requested into registers before the FMA loop of the current tile; four lanes share a point and
// each stores a quarter of its 52 features (the first version let 64 lanes compute and store all 52 doubles of a
// point each — 52 stores with a 416-byte lane stride = 16-way bank conflicts — while three waves waited, and loaded
// every tile synchronously: 48 us for 18 MB).
constexpr int FM_LDP = FM_SLOTS + 2;   // row stride of the feature tile in doubles: 432 B keeps the rows 16-byte aligned
                                       // (ds_read_b128 in the FMA loop) and spreads the staging stores (2-way, was 16-way)
constexpr int FM_SUB = 4;              // point subsets of a tile (each handled by 52 lanes)
// The FMA loop is bound by LDS bandwidth, not by fp64 FMAs: with a 2-instance x 4-slot register tile a lane read 40 B
// of LDS per 8 FMAs (1.09 GB of LDS reads per launch = 14 us at the chip's LDS peak, 45 us measured).  Now a lane owns
// 8 instances x 4 slots (64 B per 32 FMAs) and the 64 points of a tile are split over four lane groups whose partial
// sums are added in a fixed order at the end.
// MATCH: the loss section's assignment (lsap.h: one wave per cloud, a ~40 us latency chain on 16 CUs that does not depend
// on the fits, nor they on it) rides as workgroup x = 0 of every cloud — one launch and ~25 us of chain less than two
// launches in a row.
template <bool MATCH>
__global__ __launch_bounds__(FM_THREADS) void moments_fwd_kernel(const float *__restrict__ P,
                                                                 const float *__restrict__ X,
                                                                 const float *__restrict__ W, int N, int K,
                                                                 int pts_per_block, double *__restrict__ partial,
                                                                 const float *__restrict__ seg_S,
                                                                 const long long *__restrict__ n_gt,
                                                                 long long *__restrict__ match) {
  if (MATCH && blockIdx.x == 0) {
    if (threadIdx.x >= 64) return;            // (the solver's barriers then count one wave)
    lsap_one_cloud(seg_S, n_gt, K, match, blockIdx.y, threadIdx.x);
    return;
  }
  __shared__ __attribute__((aligned(16))) double s_phi[FM_TILE][FM_LDP];     // 27.6 KB; reused for the final combine
  __shared__ __attribute__((aligned(16))) float s_w[FM_TILE][FM_KB];
  __shared__ __attribute__((aligned(16))) float s_wc[FM_TILE][FM_KB];
  static_assert(FM_THREADS == 4 * FM_TILE && FM_TILE * FM_KB == 8 * FM_THREADS, "staging maps below");
  static_assert(FM_SUB * 16 * FM_SLOTS <= FM_TILE * FM_LDP, "combine buffer (half of the instances at a time)");
  const int b = blockIdx.y, chunk = blockIdx.x - (MATCH ? 1 : 0), nchunks = gridDim.x - (MATCH ? 1 : 0), t = threadIdx.x;
  const int n0 = chunk * pts_per_block;
  const int n1 = min(N, n0 + pts_per_block);
  const int sub = t / 52, r52 = t % 52;       // lane group (points sub, sub+4, ... of a tile) and position in it
  const int mg = r52 % FM_GROUPS;             // slot group: slots 4mg .. 4mg+3
  const int kq = r52 / FM_GROUPS;             // instance octet: kb + 8kq .. kb + 8kq + 7
  const bool active = sub < FM_SUB;
  const float(*wsel)[FM_KB] = mg < FM_A_GROUPS ? s_w : s_wc;
  const int pi = t & 63, pq = t >> 6;         // staging: lane = point of the tile, wave = quarter of the feature rows
                                              // (full-wave stores of ONE feature of 64 points: 13 per wave and tile)

  for (int kb = 0; kb < K; kb += FM_KB) {
    double acc[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc[j][0] = 0; acc[j][1] = 0; acc[j][2] = 0; acc[j][3] = 0; }
    float rp[3], rx[3], rw[8];
    // memberships of a tile: rows of K floats, contiguous in memory.  K % 4 == 0 (and a whole K block): float4 pieces
    // (2 per lane for K = 28 instead of 8 scalar loads and 16 scalar LDS stores); otherwise element by element.
    const bool wvec = (K & 3) == 0 && kb == 0 && K <= FM_KB && ((uintptr_t)W & 15) == 0;
    const int k4 = K >> 2;                      // float4 pieces per row
    auto fetch = [&](int base) __attribute__((always_inline)) {
      const int n = min(base + pi, n1 - 1);
      const float *pp = P + ((size_t)b * N + n) * 3, *xx = X + ((size_t)b * N + n) * 3;
#pragma unroll
      for (int j = 0; j < 3; ++j) { rp[j] = pp[j]; rx[j] = xx[j]; }
      if (wvec) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = t + u * FM_THREADS, i = min(e / k4, FM_TILE - 1), q = e % k4;
          const int nn = min(base + i, n1 - 1);
          *(cpfn_f32x4 *)&rw[4 * u] = *(const cpfn_f32x4 *)(W + ((size_t)b * N + nn) * K + 4 * q);
        }
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = t + u * FM_THREADS, i = e / FM_KB, kk = e % FM_KB;
          const int nn = min(base + i, n1 - 1), k = min(kb + kk, K - 1);
          rw[u] = W[((size_t)b * N + nn) * K + k];
        }
      }
    };
    if (wvec) {   // the columns K..31 of the membership tiles are never written by the float4 path
      for (int e = t; e < FM_TILE * FM_KB; e += FM_THREADS) { (&s_w[0][0])[e] = 0.f; (&s_wc[0][0])[e] = 0.f; }
    }
    fetch(n0);
    for (int base = n0; base < n1; base += FM_TILE) {
      __syncthreads();                       // the previous tile's FMA loop is done with the LDS tiles
      {
        double f[FM_SLOTS];
        point_features<double>(rp, rx, f);
        const double live = base + pi < n1 ? 1.0 : 0.0;
        double *row = s_phi[pi];
        if (pq == 0) {
#pragma unroll
          for (int j = 0; j < 13; ++j) row[j] = f[j] * live;
        } else if (pq == 1) {
#pragma unroll
          for (int j = 13; j < 26; ++j) row[j] = f[j] * live;
        } else if (pq == 2) {
#pragma unroll
          for (int j = 26; j < 39; ++j) row[j] = f[j] * live;
        } else {
#pragma unroll
          for (int j = 39; j < 52; ++j) row[j] = f[j] * live;
        }
      }
      if (wvec) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = t + u * FM_THREADS, i = e / k4, q = e % k4;
          if (i < FM_TILE) {
            const bool ok = base + i < n1;
            cpfn_f32x4 w = *(const cpfn_f32x4 *)&rw[4 * u], wc;
#pragma unroll
            for (int j = 0; j < 4; ++j) { w[j] = ok ? w[j] : 0.f; wc[j] = ok ? fmaxf(w[j], FM_WEPS) : 0.f; }
            *(cpfn_f32x4 *)&s_w[i][4 * q] = w;
            *(cpfn_f32x4 *)&s_wc[i][4 * q] = wc;
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = t + u * FM_THREADS, i = e / FM_KB, kk = e % FM_KB;
          const bool ok = base + i < n1 && kb + kk < K;
          const float w = ok ? rw[u] : 0.f;
          s_w[i][kk] = w;
          s_wc[i][kk] = ok ? fmaxf(w, FM_WEPS) : 0.f;
        }
      }
      __syncthreads();
      fetch(base + FM_TILE);                 // in flight during the FMA loop (clamped past the end)
      if (active) {
#pragma unroll 2
        for (int i = sub; i < FM_TILE; i += FM_SUB) {
          const cpfn_f32x4 wa = *(const cpfn_f32x4 *)&wsel[i][8 * kq], wb = *(const cpfn_f32x4 *)&wsel[i][8 * kq + 4];
          const double f0 = s_phi[i][4 * mg], f1 = s_phi[i][4 * mg + 1], f2 = s_phi[i][4 * mg + 2],
                       f3 = s_phi[i][4 * mg + 3];
          const float w8[8] = {wa[0], wa[1], wa[2], wa[3], wb[0], wb[1], wb[2], wb[3]};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const double w = (double)w8[j];
            acc[j][0] += w * f0; acc[j][1] += w * f1; acc[j][2] += w * f2; acc[j][3] += w * f3;
          }
        }
      }
    }
    // the four lane groups' partial sums, added in group order: half of the instances at a time through the
    // (now idle) feature tile
    double *s_cmb = &s_phi[0][0];              // [FM_SUB][16][52]
    for (int h = 0; h < 2; ++h) {
      __syncthreads();
      if (active && (kq >> 1) == h) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          double *o = s_cmb + ((size_t)sub * 16 + (kq & 1) * 8 + j) * FM_SLOTS + 4 * mg;
          o[0] = acc[j][0]; o[1] = acc[j][1]; o[2] = acc[j][2]; o[3] = acc[j][3];
        }
      }
      __syncthreads();
      for (int e = t; e < 16 * FM_SLOTS; e += FM_THREADS) {
        const int kl = e / FM_SLOTS, m = e - kl * FM_SLOTS;
        const int k = kb + 16 * h + kl;
        if (k < K) {
          const double v = ((s_cmb[(0 * 16 + kl) * FM_SLOTS + m] + s_cmb[(1 * 16 + kl) * FM_SLOTS + m]) +
                            s_cmb[(2 * 16 + kl) * FM_SLOTS + m]) + s_cmb[(3 * 16 + kl) * FM_SLOTS + m];
          partial[(((size_t)b * nchunks + chunk) * K + k) * FM_SLOTS + m] = v;
        }
      }
    }
  }
}

// out[r][c] = Σ_chunk partial[b][chunk][r-in-b][c], chunks in ascending order.
__global__ void chunk_reduce_kernel(const double *__restrict__ partial, int nchunks, int per_b,
                                    long long total, double *__restrict__ out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const long long b = e / per_b, r = e % per_b;
  double s = 0.0;
  // (eight loads in flight per trip, added in chunk order: left alone the loop is one load latency per chunk)
#pragma unroll 8
  for (int c = 0; c < nchunks; ++c) s += partial[((size_t)b * nchunks + c) * per_b + r];
  out[e] = s;
}

// same sum, written as rows of `width` values into a wider row-major matrix (row stride ld), optionally
// accumulating: lets the cone adjoint land directly in columns 15..20 of the [G,21] algebra adjoint.
__global__ void chunk_reduce_strided_kernel(const double *__restrict__ partial, int nchunks, int per_b,
                                            long long total, int width, int ld, int accumulate,
                                            double *__restrict__ out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const long long b = e / per_b, r = e % per_b;
  double s = 0.0;
  // (eight loads in flight per trip, added in chunk order: left alone the loop is one load latency per chunk)
#pragma unroll 8
  for (int c = 0; c < nchunks; ++c) s += partial[((size_t)b * nchunks + c) * per_b + r];
  double *dst = out + (e / width) * ld + (e % width);
  *dst = accumulate ? *dst + s : s;
}

// dW[b,n,k] = Σ_m φ_m G[b,k,m] (B slots only where w >= 1e-10: clamp's adjoint);
// dX[b,n,:] from the slots that involve the normal.  One lane per point, G[b] in LDS.
constexpr int FM_MAXK = 64;
__global__ __launch_bounds__(FM_THREADS) void moments_bwd_kernel(const float *__restrict__ P,
                                                                 const float *__restrict__ X,
                                                                 const float *__restrict__ W,
                                                                 const float *__restrict__ G, int N, int K,
                                                                 const float *__restrict__ dW_add,
                                                                 float *__restrict__ dW, float *__restrict__ dX) {
  __shared__ float s_p[FM_THREADS * 3], s_x[FM_THREADS * 3];
  extern __shared__ float s_dyn[];                   // W tile, then (in place) the dW tile; dW_add tile
  const int b = blockIdx.y, t = threadIdx.x;
  const int n0 = blockIdx.x * FM_THREADS, rows = min(FM_THREADS, N - n0);
  const size_t p0 = (size_t)b * N + n0;
  const int ld = K | 1;
  float *s_w = s_dyn, *s_add = s_dyn + FM_THREADS * ld;
  cpfn_rows_to_lds<FM_THREADS>(s_p, 3, P + p0 * 3, rows, 3, t);
  cpfn_rows_to_lds<FM_THREADS>(s_x, 3, X + p0 * 3, rows, 3, t);
  cpfn_rows_to_lds<FM_THREADS>(s_w, ld, W + p0 * K, rows, K, t);
  if (dW_add) cpfn_rows_to_lds<FM_THREADS>(s_add, ld, dW_add + p0 * K, rows, K, t);
  __syncthreads();
  if (t < rows) {
    const float p[3] = {s_p[t * 3], s_p[t * 3 + 1], s_p[t * 3 + 2]}, x[3] = {s_x[t * 3], s_x[t * 3 + 1], s_x[t * 3 + 2]};
    float f[FM_SLOTS];
    point_features<float>(p, x, f);
    float a[3] = {0, 0, 0}, sa[6] = {0, 0, 0, 0, 0, 0}, sb[6] = {0, 0, 0, 0, 0, 0}, gb[3] = {0, 0, 0};
    float *wrow = s_w + t * ld;
    const float *arow = s_add + t * ld;
    for (int k = 0; k < K; ++k) {
      // the 52-slot row of instance k is the same for every lane: read straight from memory, it arrives by SCALAR loads
      // (s_load_dwordx16: the FMAs below take it as scalar operands).  Round 3; before, G[b] was staged in LDS and every
      // lane read the row back with 13 16-byte LDS reads per instance — 364 LDS instructions per wave, the bulk of the
      // kernel's time (25.4 -> 20.2 us stand-alone, same bits).
      float g[FM_SLOTS];
      {
        const float *grow = G + ((size_t)b * K + k) * FM_SLOTS;
#pragma unroll
        for (int m = 0; m < FM_SLOTS; ++m) g[m] = grow[m];
      }
      float ga = 0.f, gB = 0.f;
#pragma unroll
      for (int m = 0; m < 19; ++m) ga = fmaf(f[m], g[m], ga);
#pragma unroll
      for (int m = 20; m < 49; ++m) gB = fmaf(f[m], g[m], gB);
      const float w = wrow[k];
      const float wc = fmaxf(w, FM_WEPS);
      wrow[k] = ga + (w >= FM_WEPS ? gB : 0.f) + (dW_add ? arow[k] : 0.f);     // the dW row replaces the W row
#pragma unroll
      for (int j = 0; j < 3; ++j) { a[j] = fmaf(w, g[10 + j], a[j]); gb[j] = fmaf(wc, g[46 + j], gb[j]); }
#pragma unroll
      for (int j = 0; j < 6; ++j) { sa[j] = fmaf(w, g[13 + j], sa[j]); sb[j] = fmaf(wc, g[40 + j], sb[j]); }
    }
    // d/dx of Σ_{i<=j} S_ij x_i x_j  = Ŝ x with Ŝ_ll = 2 S_ll, Ŝ_lj = S_(lj)
    float s[6];
    for (int j = 0; j < 6; ++j) s[j] = sa[j] + sb[j];
    const float nx = x[0], ny = x[1], nz = x[2], px = p[0], py = p[1], pz = p[2];
    const float pn = px * nx + py * ny + pz * nz;
    const float gdotx = gb[0] * nx + gb[1] * ny + gb[2] * nz;
    float *o = s_x + t * 3;                               // the dX row replaces the X row
    o[0] = a[0] + 2.f * s[0] * nx + s[1] * ny + s[2] * nz + gb[0] * pn + gdotx * px;
    o[1] = a[1] + s[1] * nx + 2.f * s[3] * ny + s[4] * nz + gb[1] * pn + gdotx * py;
    o[2] = a[2] + s[2] * nx + s[4] * ny + 2.f * s[5] * nz + gb[2] * pn + gdotx * pz;
  }
  __syncthreads();
  cpfn_rows_from_lds<FM_THREADS>(s_w, ld, dW + p0 * K, rows, K, t);
  cpfn_rows_from_lds<FM_THREADS>(s_x, 3, dX + p0 * 3, rows, 3, t);
}

// ---------------------------------------------------------------- cone second pass
// thread = (instance kk in k-block, point subset); partial[b][chunk][K][NV]
constexpr int CP_SUB = FM_THREADS / FM_KB;  // 8 point subsets
constexpr float CP_CLAMP = 1.0f - 1e-6f;

struct ConeTerm {
  float vx, vy, vz, inv, dot, c;  // v/|v|, 1/max(|v|,eps), axis·vn, clamp(|dot|)
};

__device__ __forceinline__ ConeTerm cone_term(const float *p, const float *apex, const float *axis) {
  ConeTerm r;
  const float vx = p[0] - apex[0], vy = p[1] - apex[1], vz = p[2] - apex[2];
  const float nv = sqrtf(vx * vx + vy * vy + vz * vz);
  r.inv = 1.0f / fmaxf(nv, 1e-12f);  // F.normalize(eps=1e-12), cone_fitter.py:26
  r.vx = vx * r.inv; r.vy = vy * r.inv; r.vz = vz * r.inv;
  r.dot = axis[0] * r.vx + axis[1] * r.vy + axis[2] * r.vz;
  r.c = fminf(fabsf(r.dot), CP_CLAMP);  // acos_safe(abs(.)) clamps to [-1+1e-6, 1-1e-6]
  return r;
}

// A 64-point tile of coordinates and memberships on its way to LDS: requested into registers one tile AHEAD (round 3; the
// first version loaded straight into LDS between two barriers, i.e. one exposed global latency per tile, four per workgroup).
struct ConeStage { float p, w[FM_TILE * FM_KB / FM_THREADS]; };
__device__ __forceinline__ void cone_fetch(ConeStage &st, const float *__restrict__ P, const float *__restrict__ W, int b, int N,
                                           int K, int kb, int base, int n1, int t) {
  st.p = 0.f;
  if (t < FM_TILE * 3) {
    const int i = t / 3, c = t % 3;
    if (base + i < n1) st.p = P[((size_t)b * N + base + i) * 3 + c];
  }
#pragma unroll
  for (int u = 0; u < FM_TILE * FM_KB / FM_THREADS; ++u) {
    const int e = t + u * FM_THREADS, i = e / FM_KB, k2 = kb + e % FM_KB;
    st.w[u] = (base + i < n1 && k2 < K) ? W[((size_t)b * N + base + i) * K + k2] : 0.f;
  }
}
__device__ __forceinline__ void cone_store(const ConeStage &st, float (*s_p)[3], float (*s_w)[FM_KB], int t) {
  if (t < FM_TILE * 3) s_p[t / 3][t % 3] = st.p;
#pragma unroll
  for (int u = 0; u < FM_TILE * FM_KB / FM_THREADS; ++u) {
    const int e = t + u * FM_THREADS;
    s_w[e / FM_KB][e % FM_KB] = st.w[u];
  }
}

template <int NV>
__device__ __forceinline__ void cone_block_reduce(double (*s_red)[FM_KB][NV], const double *v, int sub, int kk,
                                                  int k, int K, double *out_row) {
  for (int j = 0; j < NV; ++j) s_red[sub][kk][j] = v[j];
  __syncthreads();
  if (sub == 0 && k < K) {
    for (int j = 0; j < NV; ++j) {
      double s = 0.0;
      for (int q = 0; q < CP_SUB; ++q) s += s_red[q][kk][j];
      out_row[j] = s;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(FM_THREADS) void cone_fwd_kernel(const float *__restrict__ P,
                                                              const float *__restrict__ W,
                                                              const float *__restrict__ apex,
                                                              const float *__restrict__ axis, int N, int K,
                                                              int pts_per_block, double *__restrict__ partial) {
  __shared__ float s_p[FM_TILE][3];
  __shared__ float s_w[FM_TILE][FM_KB];
  __shared__ double s_red[CP_SUB][FM_KB][2];
  const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x, t = threadIdx.x;
  const int kk = t % FM_KB, sub = t / FM_KB;
  const int n0 = chunk * pts_per_block, n1 = min(N, n0 + pts_per_block);
  for (int kb = 0; kb < K; kb += FM_KB) {
    const int k = kb + kk;
    float ap[3] = {0, 0, 0}, ax[3] = {0, 0, 0};
    if (k < K) {
      for (int j = 0; j < 3; ++j) { ap[j] = apex[((size_t)b * K + k) * 3 + j]; ax[j] = axis[((size_t)b * K + k) * 3 + j]; }
    }
    double acc[2] = {0.0, 0.0};  // Σ w·dot, Σ w·acos(clamp|dot|)
    ConeStage stg;
    cone_fetch(stg, P, W, b, N, K, kb, n0, n1, t);
    for (int base = n0; base < n1; base += FM_TILE) {
      __syncthreads();
      cone_store(stg, s_p, s_w, t);
      __syncthreads();
      if (base + FM_TILE < n1) cone_fetch(stg, P, W, b, N, K, kb, base + FM_TILE, n1, t);   // in flight during this tile's loop
      float a0 = 0.f, a1 = 0.f;
      for (int i = sub; i < FM_TILE; i += CP_SUB) {
        const ConeTerm r = cone_term(s_p[i], ap, ax);
        const float w = s_w[i][kk];
        a0 = fmaf(w, r.dot, a0);
        a1 = fmaf(w, acosf(r.c), a1);
      }
      acc[0] += (double)a0;
      acc[1] += (double)a1;
    }
    cone_block_reduce<2>(s_red, acc, sub, kk, k, K,
                         partial + (((size_t)b * nchunks + chunk) * K + (k < K ? k : 0)) * 2);
  }
}

// g_acos[b,k] = dL/d(Σ w·acos).  dW[b,n,k] = g·acos(c);  partial adjoints of (apex, axis).
// (Σ w·dot only feeds a sign(), whose derivative is zero — cone_fitter.py:29.)
__global__ __launch_bounds__(FM_THREADS) void cone_bwd_kernel(const float *__restrict__ P,
                                                              const float *__restrict__ W,
                                                              const float *__restrict__ apex,
                                                              const float *__restrict__ axis,
                                                              const float *__restrict__ g_acos, int N, int K,
                                                              int pts_per_block, float *__restrict__ dW,
                                                              double *__restrict__ partial,
                                                              const float *__restrict__ gparams /* or NULL */,
                                                              const double *__restrict__ sums,
                                                              const double *__restrict__ M) {
  __shared__ float s_p[FM_TILE][3];
  __shared__ float s_w[FM_TILE][FM_KB];
  __shared__ double s_red[CP_SUB][FM_KB][6];
  const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x, t = threadIdx.x;
  const int kk = t % FM_KB, sub = t / FM_KB;
  const int n0 = chunk * pts_per_block, n1 = min(N, n0 + pts_per_block);
  for (int kb = 0; kb < K; kb += FM_KB) {
    const int k = kb + kk;
    float ap[3] = {0, 0, 0}, ax[3] = {0, 0, 0}, g = 0.f;
    if (k < K) {
      for (int j = 0; j < 3; ++j) { ap[j] = apex[((size_t)b * K + k) * 3 + j]; ax[j] = axis[((size_t)b * K + k) * 3 + j]; }
      const size_t gi = (size_t)b * K + k;
      // (gparams given: the packed parameters' adjoint comes in as it is and fit_pack_bwd's half-angle rule runs here)
      g = gparams ? (float)pack_half_angle_adjoint((double)gparams[gi * 22 + 21], sums[gi * 2 + 1], M[gi * FM_SLOTS]).g_acos
                  : g_acos[gi];
    }
    double acc[6] = {0, 0, 0, 0, 0, 0};  // d apex (3), d axis (3)
    ConeStage stg;
    cone_fetch(stg, P, W, b, N, K, kb, n0, n1, t);
    for (int base = n0; base < n1; base += FM_TILE) {
      __syncthreads();
      cone_store(stg, s_p, s_w, t);
      __syncthreads();
      if (base + FM_TILE < n1) cone_fetch(stg, P, W, b, N, K, kb, base + FM_TILE, n1, t);   // in flight during this tile's loop
      float f[6] = {0, 0, 0, 0, 0, 0};
      for (int i = sub; i < FM_TILE; i += CP_SUB) {
        const int n = base + i;
        if (n >= n1) break;
        const ConeTerm r = cone_term(s_p[i], ap, ax);
        const float w = s_w[i][kk];
        if (k < K) dW[((size_t)b * N + n) * K + k] = g * acosf(r.c);
        // d/d(dot) of w·acos(clamp(|dot|)): zero where the clamp is active
        float gd = 0.f;
        if (fabsf(r.dot) < CP_CLAMP) gd = -g * w * rsqrtf(1.f - r.c * r.c) * (r.dot >= 0.f ? 1.f : -1.f);
        // dot = axis·vn ; vn = v / max(|v|, eps)
        f[3] = fmaf(gd, r.vx, f[3]); f[4] = fmaf(gd, r.vy, f[4]); f[5] = fmaf(gd, r.vz, f[5]);
        const float tx = gd * ax[0], ty = gd * ax[1], tz = gd * ax[2];   // d vn
        const float proj = tx * r.vx + ty * r.vy + tz * r.vz;
        // d v = (d vn − vn (vn·d vn)) / |v|    (for |v| >= eps);   d apex = −d v
        f[0] -= (tx - r.vx * proj) * r.inv;
        f[1] -= (ty - r.vy * proj) * r.inv;
        f[2] -= (tz - r.vz * proj) * r.inv;
      }
      for (int j = 0; j < 6; ++j) acc[j] += (double)f[j];
    }
    cone_block_reduce<6>(s_red, acc, sub, kk, k, K,
                         partial + (((size_t)b * nchunks + chunk) * K + (k < K ? k : 0)) * 6);
  }
}

// Batched symmetric 3x3 eigen-decomposition (cyclic Jacobi, fp64), one lane per matrix.
// S6 = (xx xy xz yy yz zz); eigenvalues ascending; V[g][i][j] = i-th component of eigenvector j.
__global__ void eigh3_kernel(const double *__restrict__ S6, long long G, double *__restrict__ lam,
                             double *__restrict__ V) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const double *s = S6 + g * 6;
  double a[3][3] = {{s[0], s[1], s[2]}, {s[1], s[3], s[4]}, {s[2], s[4], s[5]}};
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 16; ++sweep) {
    const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
    if (off <= 1e-300 || off <= 1e-22 * diag) break;
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      const double apq = a[p][q];
      if (apq == 0.0) continue;
      const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
      const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
      for (int r = 0; r < 3; ++r) {  // A <- A J
        const double arp = a[r][p], arq = a[r][q];
        a[r][p] = c * arp - sn * arq;
        a[r][q] = sn * arp + c * arq;
      }
      for (int r = 0; r < 3; ++r) {  // A <- J^T A
        const double apr = a[p][r], aqr = a[q][r];
        a[p][r] = c * apr - sn * aqr;
        a[q][r] = sn * apr + c * aqr;
      }
      for (int r = 0; r < 3; ++r) {
        const double vrp = v[r][p], vrq = v[r][q];
        v[r][p] = c * vrp - sn * vrq;
        v[r][q] = sn * vrp + c * vrq;
      }
    }
  }
  int o0 = 0, o1 = 1, o2 = 2;
  double e0 = a[0][0], e1 = a[1][1], e2 = a[2][2];
  if (e0 > e1) { double te = e0; e0 = e1; e1 = te; int ti = o0; o0 = o1; o1 = ti; }
  if (e1 > e2) { double te = e1; e1 = e2; e2 = te; int ti = o1; o1 = o2; o2 = ti; }
  if (e0 > e1) { double te = e0; e0 = e1; e1 = te; int ti = o0; o0 = o1; o1 = ti; }
  lam[g * 3] = e0; lam[g * 3 + 1] = e1; lam[g * 3 + 2] = e2;
  for (int r = 0; r < 3; ++r) {
    V[g * 9 + r * 3 + 0] = v[r][o0];
    V[g * 9 + r * 3 + 1] = v[r][o1];
    V[g * 9 + r * 3 + 2] = v[r][o2];
  }
}

inline int pick_chunks(int B, int N, int *pts_per_block) {
  // ~2 workgroups per CU over the batch, at least one 64-point tile each, at most 64 chunks
  // (256 / 1024 / 2048 workgroups measured: the moments pass 77 / 45 / 41 us against 48 at 512, but the ordered chunk
  //  reduction behind each pass grows by 10 us per doubling)
  int want = (512 + B - 1) / (B > 0 ? B : 1);
  if (want < 1) want = 1;
  if (want > 64) want = 64;
  int ppb = (N + want - 1) / want;
  ppb = ((ppb + FM_TILE - 1) / FM_TILE) * FM_TILE;
  *pts_per_block = ppb;
  return (N + ppb - 1) / ppb;
}

// ---------------------------------------------------------------- parameter pack (the [B,K]-sized glue)
// params[g, 0:22] (fp32) = plane n(3) c | sphere centre(3) r² | cylinder axis(3) centre(3) r² | cone apex(3)
// axis(3) half-angle — the layout the residue / axis losses read.  Columns 0..17 are the algebra's;
// the cone axis is flipped to sgn = sign(Σ W·(axis·v̂)) with sign(0) -> +1 (cone_fitter.py:28-31) and the
// half angle is Σ W·acos / (Σ W + 1e-10) clamped to [1e-3, π/2 − 1e-3] (cone_fitter.py:33-35).
// (constants and the sign rule: fit_pack.h)
__global__ void fit_pack_fwd_kernel(const double *__restrict__ alg, const double *__restrict__ sums,
                                    const double *__restrict__ M, long long G, float *__restrict__ params) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const double *a = alg + g * 21;
  float *p = params + g * 22;
  for (int i = 0; i < 18; ++i) p[i] = (float)a[i];
  const double sgn = cone_sign(sums[g * 2]);
  for (int i = 18; i < 21; ++i) p[i] = (float)(a[i] * sgn);
  const double h = sums[g * 2 + 1] / (M[g * FM_SLOTS] + PK_EPS);
  p[21] = (float)(h < PK_LO ? PK_LO : (h > PK_HI ? PK_HI : h));   // NaN stays NaN, like torch.clamp
}

// The same with the chunk reduction of the cone pass folded in (cpfn_cone_pass_fwd with out = NULL leaves its per-chunk
// partials [B][chunks][K][2] behind): sums[g] = Σ_c partial — chunk_reduce_kernel's order, so the same bits — is formed
// here and written out for the backward pass; one launch instead of two.
__global__ void fit_pack_fwd_partials_kernel(const double *__restrict__ alg, const double *__restrict__ cone_ws, int chunks,
                                             int K, const double *__restrict__ M, long long G, double *__restrict__ sums,
                                             float *__restrict__ params) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const long long b = g / K;
  const int k = (int)(g - b * K);
  double s0 = 0.0, s1 = 0.0;
#pragma unroll 32          // (all of a 16 x 8192 step's 32 chunks in flight at once: the loop is load latency per trip)
  for (int c = 0; c < chunks; ++c) {
    const double *w = cone_ws + ((size_t)b * chunks + c) * K * 2 + k * 2;
    s0 += w[0];
    s1 += w[1];
  }
  sums[g * 2] = s0;
  sums[g * 2 + 1] = s1;
  const double *a = alg + g * 21;
  float *p = params + g * 22;
  for (int i = 0; i < 18; ++i) p[i] = (float)a[i];
  const double sgn = cone_sign(s0);
  for (int i = 18; i < 21; ++i) p[i] = (float)(a[i] * sgn);
  const double h = s1 / (M[g * FM_SLOTS] + PK_EPS);
  p[21] = (float)(h < PK_LO ? PK_LO : (h > PK_HI ? PK_HI : h));   // NaN stays NaN, like torch.clamp
}

// adjoint: g_alg[G,21] (all columns written), g_acos[G] (fp32, feeds the cone pass adjoint), gA0[G] (slot 0 of M)
__global__ void fit_pack_bwd_kernel(const float *__restrict__ gparams, const double *__restrict__ sums,
                                    const double *__restrict__ M, long long G, double *__restrict__ g_alg,
                                    float *__restrict__ g_acos, double *__restrict__ gA0) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const float *gp = gparams + g * 22;
  double *ga = g_alg + g * 21;
  for (int i = 0; i < 18; ++i) ga[i] = (double)gp[i];
  const double sgn = cone_sign(sums[g * 2]);
  for (int i = 18; i < 21; ++i) ga[i] = (double)gp[i] * sgn;
  const PackAdj a = pack_half_angle_adjoint((double)gp[21], sums[g * 2 + 1], M[g * FM_SLOTS]);
  g_acos[g] = (float)a.g_acos;
  gA0[g] = a.gA0;
}

}  // namespace

extern "C" int cpfn_fit_num_chunks(int B, int N) {
  int ppb;
  return pick_chunks(B, N, &ppb);
}

static int fit_moments_fwd(const float *P, const float *X, const float *W, int B, int N, int K, double *workspace, double *M,
                           const float *seg_S, const int64_t *n_gt, int64_t *match, void *stream) {
  if (B < 0 || N <= 0 || K <= 0 || !P || !X || !W || !workspace || !M) return CPFN_EINVAL;
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int ppb;
  const int chunks = pick_chunks(B, N, &ppb);
  if (seg_S)
    moments_fwd_kernel<true><<<dim3(chunks + 1, B), FM_THREADS, 0, st>>>(P, X, W, N, K, ppb, workspace, seg_S, (const long long *)n_gt,
                                                                         (long long *)match);
  else
    moments_fwd_kernel<false><<<dim3(chunks, B), FM_THREADS, 0, st>>>(P, X, W, N, K, ppb, workspace, nullptr, nullptr, nullptr);
  const long long total = (long long)B * K * FM_SLOTS;
  chunk_reduce_kernel<<<cpfn_cdiv(total, 256), 256, 0, st>>>(workspace, chunks, K * FM_SLOTS, total, M);
  return cpfn_launch_status();
}
extern "C" int cpfn_fit_moments_fwd(const float *P, const float *X, const float *W, int B, int N, int K,
                                    double *workspace, double *M, void *stream) {
  return fit_moments_fwd(P, X, W, B, N, K, workspace, M, nullptr, nullptr, nullptr, stream);
}
extern "C" int cpfn_fit_moments_fwd_match(const float *P, const float *X, const float *W, int B, int N, int K,
                                          double *workspace, double *M, const float *S, const int64_t *n_gt, int64_t *match,
                                          void *stream) {
  if (!S || !n_gt || !match || K > LSAP_MAXK) return CPFN_EINVAL;
  return fit_moments_fwd(P, X, W, B, N, K, workspace, M, S, n_gt, match, stream);
}

// The moments pass (with the assignment riding on it when S is given) followed by ONE launch that sums the per-chunk
// partials and runs the per-instance algebra (cpfn_fit_moments_fwd[_match] + cpfn_fit_algebra_fwd, a launch less; same
// bits).  alg[B*K,21]; apex_axis32 as in cpfn_fit_algebra_fwd.
extern "C" int cpfn_fit_moments_algebra_fwd(const float *P, const float *X, const float *W, int B, int N, int K,
                                            double *workspace, double *M, double *alg, float *apex_axis32, const float *S,
                                            const int64_t *n_gt, int64_t *match, void *stream) {
  if (B < 0 || N <= 0 || K <= 0 || !P || !X || !W || !workspace || !M || !alg) return CPFN_EINVAL;
  if (S && (!n_gt || !match || K > LSAP_MAXK)) return CPFN_EINVAL;
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int ppb;
  const int chunks = pick_chunks(B, N, &ppb);
  if (S)
    moments_fwd_kernel<true><<<dim3(chunks + 1, B), FM_THREADS, 0, st>>>(P, X, W, N, K, ppb, workspace, S, (const long long *)n_gt,
                                                                         (long long *)match);
  else
    moments_fwd_kernel<false><<<dim3(chunks, B), FM_THREADS, 0, st>>>(P, X, W, N, K, ppb, workspace, nullptr, nullptr, nullptr);
  return cpfn_launch_reduce_algebra_fwd(workspace, chunks, B, K, M, alg, apex_axis32, st);
}

extern "C" int cpfn_fit_moments_bwd(const float *P, const float *X, const float *W, const float *G, int B,
                                    int N, int K, const float *dW_add, float *dW, float *dX, void *stream) {
  if (B < 0 || N <= 0 || K <= 0 || K > FM_MAXK || !P || !X || !W || !G || !dW || !dX) return CPFN_EINVAL;
  if (B == 0) return 0;
  const size_t lds = (size_t)2 * FM_THREADS * (K | 1) * sizeof(float);            // <= 130 KB at K = 64
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)moments_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       2 * FM_THREADS * (FM_MAXK | 1) * (int)sizeof(float));
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  moments_bwd_kernel<<<dim3(cpfn_cdiv(N, FM_THREADS), B), FM_THREADS, lds, (hipStream_t)stream>>>(P, X, W, G, N, K,
                                                                                                 dW_add, dW, dX);
  return cpfn_launch_status();
}

extern "C" int cpfn_cone_pass_fwd(const float *P, const float *W, const float *apex, const float *axis, int B,
                                  int N, int K, double *workspace, double *out, void *stream) {
  if (B < 0 || N <= 0 || K <= 0 || !P || !W || !apex || !axis || !workspace) return CPFN_EINVAL;
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int ppb;
  const int chunks = pick_chunks(B, N, &ppb);
  cone_fwd_kernel<<<dim3(chunks, B), FM_THREADS, 0, st>>>(P, W, apex, axis, N, K, ppb, workspace);
  if (out) {       // NULL: the per-chunk partials stay in workspace for cpfn_fit_pack_fwd_partials
    const long long total = (long long)B * K * 2;
    chunk_reduce_kernel<<<cpfn_cdiv(total, 256), 256, 0, st>>>(workspace, chunks, K * 2, total, out);
  }
  return cpfn_launch_status();
}

extern "C" int cpfn_fit_pack_fwd_partials(const double *alg, const double *cone_workspace, int B, int N, int K,
                                          const double *M, double *sums, float *params, void *stream) {
  if (B < 0 || N <= 0 || K <= 0 || !alg || !cone_workspace || !M || !sums || !params) return CPFN_EINVAL;
  if (B == 0) return 0;
  int ppb;
  const int chunks = pick_chunks(B, N, &ppb);
  const long long G = (long long)B * K;
  fit_pack_fwd_partials_kernel<<<cpfn_cdiv(G, 64), 64, 0, (hipStream_t)stream>>>(alg, cone_workspace, chunks, K, M, G, sums, params);
  return cpfn_launch_status();
}

extern "C" int cpfn_cone_pass_bwd(const float *P, const float *W, const float *apex, const float *axis,
                                  const float *g_acos, int B, int N, int K, float *dW, double *workspace,
                                  double *d_apex_axis, int ld, int accumulate, void *stream) {
  if (B < 0 || N <= 0 || K <= 0 || ld < 6 || !P || !W || !apex || !axis || !g_acos || !dW || !workspace || !d_apex_axis)
    return CPFN_EINVAL;
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int ppb;
  const int chunks = pick_chunks(B, N, &ppb);
  cone_bwd_kernel<<<dim3(chunks, B), FM_THREADS, 0, st>>>(P, W, apex, axis, g_acos, N, K, ppb, dW, workspace, nullptr, nullptr,
                                                          nullptr);
  const long long total = (long long)B * K * 6;
  chunk_reduce_strided_kernel<<<cpfn_cdiv(total, 256), 256, 0, st>>>(workspace, chunks, K * 6, total, 6, ld, accumulate,
                                                                     d_apex_axis);
  return cpfn_launch_status();
}

// The cone pass adjoint inside the packed-parameter backward (cpfn_fit_params_bwd_*): g_acos is derived from the packed
// parameters' adjoint here (no cpfn_fit_pack_bwd launch) and the per-chunk partials of d(apex, axis) stay in `workspace`
// (chunks * B * K * 6 doubles) for cpfn_fit_params_bwd_algebra to sum (no chunk reduction launch).
extern "C" int cpfn_fit_params_bwd_cone(const float *P, const float *W, const float *apex, const float *axis,
                                        const float *gparams, const double *sums, const double *M, int B, int N, int K,
                                        float *dW, double *workspace, void *stream) {
  if (B < 0 || N <= 0 || K <= 0 || !P || !W || !apex || !axis || !gparams || !sums || !M || !dW || !workspace) return CPFN_EINVAL;
  if (B == 0) return 0;
  int ppb;
  const int chunks = pick_chunks(B, N, &ppb);
  cone_bwd_kernel<<<dim3(chunks, B), FM_THREADS, 0, (hipStream_t)stream>>>(P, W, apex, axis, nullptr, N, K, ppb, dW, workspace,
                                                                           gparams, sums, M);
  return cpfn_launch_status();
}

extern "C" int cpfn_eigh3(const double *S6, int64_t G, double *lam, double *V, void *stream) {
  if (G < 0 || !S6 || !lam || !V) return CPFN_EINVAL;
  if (G == 0) return 0;
  eigh3_kernel<<<cpfn_cdiv(G, 64), 64, 0, (hipStream_t)stream>>>(S6, G, lam, V);
  return cpfn_launch_status();
}

extern "C" int cpfn_fit_pack_fwd(const double *alg, const double *sums, const double *M, int64_t G, float *params,
                                 void *stream) {
  if (G < 0 || !alg || !sums || !M || !params) return CPFN_EINVAL;
  if (G == 0) return 0;
  fit_pack_fwd_kernel<<<cpfn_cdiv(G, 64), 64, 0, (hipStream_t)stream>>>(alg, sums, M, G, params);
  return cpfn_launch_status();
}

extern "C" int cpfn_fit_pack_bwd(const float *gparams, const double *sums, const double *M, int64_t G, double *g_alg,
                                 float *g_acos, double *gA0, void *stream) {
  if (G < 0 || !gparams || !sums || !M || !g_alg || !g_acos || !gA0) return CPFN_EINVAL;
  if (G == 0) return 0;
  fit_pack_bwd_kernel<<<cpfn_cdiv(G, 64), 64, 0, (hipStream_t)stream>>>(gparams, sums, M, G, g_alg, g_acos, gA0);
  return cpfn_launch_status();
}
