// Index-driven data movement for gfx950: grouping / gathering / 3-point interpolation
// and their adjoints, in the reference's channel-major fp32 layout (drop-in for the
// `cuda_ops` names) and in the points-major row layout the MI355X path uses natively.
//
// All of these are HBM/L2-bound byte movers: one lane per output element (channel-major)
// or one 16-byte vector per lane (row layout), outputs written fully coalesced, the
// small gathered-from tensor left to L2.  Adjoints use fp32 atomics exactly like the
// reference's kernels (group_points_gpu.cu:60, interpolate_gpu.cu:139-141).
#include "common.h"
#include <type_traits>

namespace {

constexpr int TPB = 256;

// out[b,c,j] = points[b,c,idx[b,j]]   (j over S*K);  grid: (ceil(J/TPB), C-chunks, B)
__global__ __launch_bounds__(TPB) void group_fwd_kernel(const float *__restrict__ points,
                                                        const int *__restrict__ idx, int C, int N, int J,
                                                        int c_per_block, float *__restrict__ out) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * TPB + threadIdx.x;
  if (j >= J) return;
  int ii = idx[(size_t)b * J + j];
  ii = ii < 0 ? 0 : (ii >= N ? N - 1 : ii);
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(C, c0 + c_per_block);
  for (int c = c0; c < c1; ++c) out[((size_t)b * C + c) * J + j] = points[((size_t)b * C + c) * N + ii];
}

__global__ __launch_bounds__(TPB) void group_bwd_kernel(const float *__restrict__ grad_out,
                                                        const int *__restrict__ idx, int C, int N, int J,
                                                        int c_per_block, float *__restrict__ grad_points) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * TPB + threadIdx.x;
  if (j >= J) return;
  int ii = idx[(size_t)b * J + j];
  ii = ii < 0 ? 0 : (ii >= N ? N - 1 : ii);
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(C, c0 + c_per_block);
  for (int c = c0; c < c1; ++c)
    atomicAdd(grad_points + ((size_t)b * C + c) * N + ii, grad_out[((size_t)b * C + c) * J + j]);
}

// out[b,c,n] = Σ_t feats[b,c,idx[b,n,t]] * w[b,n,t]   (products rounded, summed in t order)
__global__ __launch_bounds__(TPB) void interp_fwd_kernel(const float *__restrict__ feats,
                                                         const int *__restrict__ idx,
                                                         const float *__restrict__ w, int C, int M, int N,
                                                         int c_per_block, float *__restrict__ out) {
  const int b = blockIdx.z;
  const int n = blockIdx.x * TPB + threadIdx.x;
  if (n >= N) return;
  const int *ii = idx + ((size_t)b * N + n) * 3;
  const float *ww = w + ((size_t)b * N + n) * 3;
  const int i0 = ii[0], i1 = ii[1], i2 = ii[2];
  const float w0 = ww[0], w1 = ww[1], w2 = ww[2];
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(C, c0 + c_per_block);
  for (int c = c0; c < c1; ++c) {
    const float *f = feats + ((size_t)b * C + c) * M;
    const float acc = __fadd_rn(__fmul_rn(f[i0], w0), __fmul_rn(f[i1], w1));
    out[((size_t)b * C + c) * N + n] = __fadd_rn(acc, __fmul_rn(f[i2], w2));
  }
}

__global__ __launch_bounds__(TPB) void interp_bwd_kernel(const float *__restrict__ grad_out,
                                                         const int *__restrict__ idx,
                                                         const float *__restrict__ w, int C, int N, int M,
                                                         int c_per_block, float *__restrict__ grad_feats) {
  const int b = blockIdx.z;
  const int n = blockIdx.x * TPB + threadIdx.x;
  if (n >= N) return;
  const int *ii = idx + ((size_t)b * N + n) * 3;
  const float *ww = w + ((size_t)b * N + n) * 3;
  const int i0 = ii[0], i1 = ii[1], i2 = ii[2];
  const float w0 = ww[0], w1 = ww[1], w2 = ww[2];
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(C, c0 + c_per_block);
  for (int c = c0; c < c1; ++c) {
    const float g = grad_out[((size_t)b * C + c) * N + n];
    float *o = grad_feats + ((size_t)b * C + c) * M;
    atomicAdd(o + i0, g * w0);
    atomicAdd(o + i1, g * w1);
    atomicAdd(o + i2, g * w2);
  }
}

// ---------------------------------------------------------------- points-major rows

// out[b,r,:] = rows[b,idx[b,r],:]; one VEC-byte vector per lane, consecutive lanes walk a row.
template <typename V>
__global__ __launch_bounds__(TPB) void gather_rows_kernel(const V *__restrict__ rows,
                                                          const int *__restrict__ idx, int N, int R,
                                                          int vec_per_row, V *__restrict__ out) {
  const int b = blockIdx.y;
  const long long e = (long long)blockIdx.x * TPB + threadIdx.x;
  if (e >= (long long)R * vec_per_row) return;
  const int r = (int)(e / vec_per_row);
  const int v = (int)(e - (long long)r * vec_per_row);
  int ii = idx[(size_t)b * R + r];
  ii = ii < 0 ? 0 : (ii >= N ? N - 1 : ii);
  out[((size_t)b * R + r) * vec_per_row + v] = rows[((size_t)b * N + ii) * vec_per_row + v];
}

__global__ __launch_bounds__(TPB) void scatter_add_rows_kernel(const float *__restrict__ grad_out,
                                                               const int *__restrict__ idx, int N, int R,
                                                               int C, float *__restrict__ grad_rows) {
  const int b = blockIdx.y;
  const long long e = (long long)blockIdx.x * TPB + threadIdx.x;
  if (e >= (long long)R * C) return;
  const int r = (int)(e / C);
  const int c = (int)(e - (long long)r * C);
  int ii = idx[(size_t)b * R + r];
  ii = ii < 0 ? 0 : (ii >= N ? N - 1 : ii);
  atomicAdd(grad_rows + ((size_t)b * N + ii) * C + c, grad_out[((size_t)b * R + r) * C + c]);
}

__global__ __launch_bounds__(TPB) void group_xyz_centered_kernel(const float *__restrict__ xyz,
                                                                 const float *__restrict__ new_xyz,
                                                                 const int *__restrict__ idx, int N, int S,
                                                                 int K, float *__restrict__ out) {
  const int b = blockIdx.y;
  const long long e = (long long)blockIdx.x * TPB + threadIdx.x;  // over S*K*3
  if (e >= (long long)S * K * 3) return;
  const int j = (int)(e / 3);
  const int c = (int)(e - (long long)j * 3);
  const int s = j / K;
  int ii = idx[(size_t)b * S * K + j];
  ii = ii < 0 ? 0 : (ii >= N ? N - 1 : ii);
  out[(size_t)b * S * K * 3 + e] =
      __fsub_rn(xyz[((size_t)b * N + ii) * 3 + c], new_xyz[((size_t)b * S + s) * 3 + c]);
}

__global__ __launch_bounds__(TPB) void interp_rows_fwd_kernel(const float *__restrict__ feats,
                                                              const int *__restrict__ idx,
                                                              const float *__restrict__ w, int M, int N,
                                                              int C, float *__restrict__ out) {
  const int b = blockIdx.y;
  const long long e = (long long)blockIdx.x * TPB + threadIdx.x;
  if (e >= (long long)N * C) return;
  const int n = (int)(e / C);
  const int c = (int)(e - (long long)n * C);
  const int *ii = idx + ((size_t)b * N + n) * 3;
  const float *ww = w + ((size_t)b * N + n) * 3;
  const float *f = feats + (size_t)b * M * C;
  const float acc = __fadd_rn(__fmul_rn(f[(size_t)ii[0] * C + c], ww[0]), __fmul_rn(f[(size_t)ii[1] * C + c], ww[1]));
  out[(size_t)b * N * C + e] = __fadd_rn(acc, __fmul_rn(f[(size_t)ii[2] * C + c], ww[2]));
}

__global__ __launch_bounds__(TPB) void interp_rows_bwd_kernel(const float *__restrict__ grad_out,
                                                              const int *__restrict__ idx,
                                                              const float *__restrict__ w, int M, int N,
                                                              int C, float *__restrict__ grad_feats) {
  const int b = blockIdx.y;
  const long long e = (long long)blockIdx.x * TPB + threadIdx.x;
  if (e >= (long long)N * C) return;
  const int n = (int)(e / C);
  const int c = (int)(e - (long long)n * C);
  const int *ii = idx + ((size_t)b * N + n) * 3;
  const float *ww = w + ((size_t)b * N + n) * 3;
  const float g = grad_out[(size_t)b * N * C + e];
  float *o = grad_feats + (size_t)b * M * C;
  atomicAdd(o + (size_t)ii[0] * C + c, g * ww[0]);
  atomicAdd(o + (size_t)ii[1] * C + c, g * ww[1]);
  atomicAdd(o + (size_t)ii[2] * C + c, g * ww[2]);
}

// ---------------------------------------------------------------- bf16 row movers of the MLP path
// XCD-aware (workgroup, cloud) of a (nx, B) grid.  Consecutive workgroup ids go round-robin over the chip's 8 XCDs, each with
// its own L2: with the plain (blockIdx.x, blockIdx.y = cloud) mapping every cloud's source rows are fetched by all eight — the
// interpolation adjoint, which reads each gradient row three times (once per neighbour), measured 1.93 x its bytes from HBM,
// the interpolation itself 1.40 x (round 5 counters).  Here XCD k walks clouds k, k + 8, ... one after the other, so a row's
// re-reads hit the L2 that fetched it.  (B % 8 != 0: the plain mapping.)
__device__ __forceinline__ void xcd_cloud_map(int &bx, int &b) {
  const int nx = (int)gridDim.x, B = (int)gridDim.y;
  bx = (int)blockIdx.x; b = (int)blockIdx.y;
  if (B & 7) return;
  const int L = bx + nx * b, j = L >> 3;
  b = (L & 7) + 8 * (j / nx);
  bx = j % nx;
}
__device__ __forceinline__ float bf2f_(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
__device__ __forceinline__ unsigned short f2bf_(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

// out[b,n,:] = Σ_t w[b,n,t]·feats[b,idx[b,n,t],:]   bf16 in / bf16 out, fp32 math, 8 channels per lane
__global__ __launch_bounds__(TPB) void interp_rows_bf16_kernel(const unsigned short *__restrict__ feats,
                                                               const int *__restrict__ idx,
                                                               const float *__restrict__ w, int M, int N, int C,
                                                               unsigned short *__restrict__ out) {
  int bx, b;
  xcd_cloud_map(bx, b);
  const int cpr = C / 8;
  const long long e = (long long)bx * TPB + threadIdx.x;
  if (e >= (long long)N * cpr) return;
  const int n = (int)(e / cpr), c0 = (int)(e - (long long)n * cpr) * 8;
  const int *ii = idx + ((size_t)b * N + n) * 3;
  const float *ww = w + ((size_t)b * N + n) * 3;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const uint4 r = *(const uint4 *)(feats + ((size_t)b * M + ii[t]) * C + c0);
    const unsigned short *h = (const unsigned short *)&r;
    const float wt = ww[t];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = fmaf(wt, bf2f_(h[j]), acc[j]);
  }
  unsigned short o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f2bf_(acc[j]);
  *(uint4 *)(out + ((size_t)b * N + n) * C + c0) = *(const uint4 *)o;
}

// Input rows of a feature-propagation stack in one pass (modules/pointset_feature_propagation.py:33-46):
//   out[b,n,:] = [ skip[b,n,:C1] | Σ_t w[b,n,t]·feats[b,idx[b,n,t],:C2] ]        (idx == NULL: | feats[b,0,:C2], the
// broadcast of a global feature vector).  Was: the interpolation (or an expand) + torch.cat = two launches and a
// [B,N,C2] intermediate; the adjoint of the expand was a 12 us framework reduction (colsum_rows_bf16_kernel below).
__global__ __launch_bounds__(TPB) void concat_interp_bf16_kernel(const unsigned short *__restrict__ skip, int C1,
                                                                 const unsigned short *__restrict__ feats,
                                                                 const int *__restrict__ idx, const float *__restrict__ w,
                                                                 int M, int N, int C2, unsigned short *__restrict__ out) {
  int bx, b;
  xcd_cloud_map(bx, b);
  const int cpr = (C1 + C2) / 8, cp1 = C1 / 8;
  const long long e = (long long)bx * TPB + threadIdx.x;
  if (e >= (long long)N * cpr) return;
  const int n = (int)(e / cpr), ch = (int)(e - (long long)n * cpr);
  unsigned short *o = out + ((size_t)b * N + n) * (C1 + C2) + ch * 8;
  if (ch < cp1) {
    *(uint4 *)o = *(const uint4 *)(skip + ((size_t)b * N + n) * C1 + ch * 8);
    return;
  }
  const int c0 = (ch - cp1) * 8;
  if (!idx) {
    *(uint4 *)o = *(const uint4 *)(feats + (size_t)b * M * C2 + c0);
    return;
  }
  const int *ii = idx + ((size_t)b * N + n) * 3;
  const float *ww = w + ((size_t)b * N + n) * 3;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const uint4 r = *(const uint4 *)(feats + ((size_t)b * M + ii[t]) * C2 + c0);
    const unsigned short *h = (const unsigned short *)&r;
    const float wt = ww[t];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = fmaf(wt, bf2f_(h[j]), acc[j]);
  }
  unsigned short ov[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) ov[j] = f2bf_(acc[j]);
  *(uint4 *)o = *(const uint4 *)ov;
}

// out[b,c] = Σ_n g[b,n,c] for a column block of a row-major bf16 tensor (row stride ldg): adjoint of the broadcast
// above.  One lane per (cloud, 8-channel chunk, row subset), fp32 sums, subsets combined in a fixed order through LDS.
// PASS1 (round 6): `out` [B, C] is the gradient of a max-pooled stack's output (sa3's global feature vector, one row per cloud),
// so BatchNorm-backward pass 1 of that stack's last layer — sum g_z, sum g_z y over its B rows, g_z = g [scale y + shift > 0] with y
// the pre-BN value at the arg-max row (bn_relu_bwd_kernel's arithmetic) — is taken from the row while it is being stored: one
// partial row per cloud, part [B][2][C], for cpfn_bn_bwd_finalize (nblk = B).  The one-workgroup cpfn_bn_relu_bwd launch that read
// the [B, C] gradient back (6 us + a kernel boundary on the backward chain) is not made.
template <bool PASS1>
__global__ __launch_bounds__(TPB) void colsum_rows_bf16_kernel(const unsigned short *__restrict__ g, int ldg, int N, int C,
                                                               unsigned short *__restrict__ out,
                                                               const unsigned short *__restrict__ yarg = nullptr,
                                                               const float *__restrict__ scale = nullptr,
                                                               const float *__restrict__ shift = nullptr,
                                                               float *__restrict__ part = nullptr) {
  __shared__ float s_acc[TPB][9];
  const int b = blockIdx.y, t = threadIdx.x;
  const int cpr = C / 8;                                   // chunks of the column block
  const int lanes = min(cpr - (int)blockIdx.x * 32, 32);   // up to 32 chunks per workgroup, 8 row subsets
  const int ch = t & 31, rs = t >> 5;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (ch < lanes) {
    const unsigned short *src = g + (size_t)b * N * ldg + ((size_t)blockIdx.x * 32 + ch) * 8;
    for (int n = rs; n < N; n += 8) {
      const uint4 r = *(const uint4 *)(src + (size_t)n * ldg);
      const unsigned short *h = (const unsigned short *)&r;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += bf2f_(h[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) s_acc[t][j] = acc[j];
  __syncthreads();
  if (rs == 0 && ch < lanes) {
    unsigned short ov[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) v += s_acc[q * 32 + ch][j];
      ov[j] = f2bf_(v);
    }
    const size_t c0 = ((size_t)blockIdx.x * 32 + ch) * 8;
    *(uint4 *)(out + (size_t)b * C + c0) = *(const uint4 *)ov;
    if (PASS1) {
      const uint4 ry = *(const uint4 *)(yarg + (size_t)b * C + c0);
      const unsigned short *y = (const unsigned short *)&ry;
      float z1[8], z2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float yv = bf2f_(y[j]);
        const float gz = fmaf(scale[c0 + j], yv, shift[c0 + j]) > 0.f ? bf2f_(ov[j]) : 0.f;
        z1[j] = gz;
        z2[j] = gz * yv;
      }
      float *p1 = part + ((size_t)b * 2 + 0) * C + c0, *p2 = part + ((size_t)b * 2 + 1) * C + c0;
      *(float4 *)p1 = make_float4(z1[0], z1[1], z1[2], z1[3]); *(float4 *)(p1 + 4) = make_float4(z1[4], z1[5], z1[6], z1[7]);
      *(float4 *)p2 = make_float4(z2[0], z2[1], z2[2], z2[3]); *(float4 *)(p2 + 4) = make_float4(z2[4], z2[5], z2[6], z2[7]);
    }
  }
}

// Scatter-add of bf16 gradient rows into a SMALL fp32 target [B,M,C] (M <= 1024 rows per cloud):
//   target[b, idx[b,r,t], c] += w[b,r,t] · g[b,r,c]        (T = 1 with w = NULL: plain gather adjoint)
// The target slab of a 32-channel chunk lives in LDS (M x 32 fp32 <= 128 KB), every contribution is an
// LDS atomic (64 lanes of one instruction hit 64 different banks), and the slab is flushed with one
// global atomic per element per workgroup: ~50x fewer global atomics than scattering row by row.
constexpr int SC_CH = 32;
constexpr int SC_LD = SC_CH + 1;   // padded slab row: with a 32-float stride every lane of an atomic hits bank (cg*8+j)%32 -> 16-way conflicts
__global__ __launch_bounds__(TPB) void scatter_rows_lds_kernel(const unsigned short *__restrict__ g, int ldg,
                                                               const int *__restrict__ idx,
                                                               const float *__restrict__ w, int T, int R, int M,
                                                               int C, int rows_per_block, float *__restrict__ out) {
  extern __shared__ float s_acc[];  // [M][SC_LD]
  const int b = blockIdx.z, c0 = blockIdx.y * SC_CH;
  const int t = threadIdx.x, cg = t % (SC_CH / 8), rs = t / (SC_CH / 8);  // 4 channel groups x 64 row lanes
  for (int e = t; e < M * SC_LD; e += TPB) s_acc[e] = 0.f;
  __syncthreads();
  const int r0 = blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
  const int cbase = c0 + cg * 8;
  if (cbase < C) {   // C % 8 == 0: a group is either fully inside or fully outside
    for (int r = r0 + rs; r < r1; r += TPB / (SC_CH / 8)) {
      const uint4 raw = *(const uint4 *)(g + ((size_t)b * R + r) * ldg + cbase);   // 8 channels, one 16-byte load
      const unsigned short *h = (const unsigned short *)&raw;
      float gv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) gv[j] = bf2f_(h[j]);
      for (int tt = 0; tt < T; ++tt) {
        int m = idx[((size_t)b * R + r) * T + tt];
        m = m < 0 ? 0 : (m >= M ? M - 1 : m);
        const float wt = w ? w[((size_t)b * R + r) * T + tt] : 1.f;
        float *dst = &s_acc[m * SC_LD + cg * 8];
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(dst + j, wt * gv[j]);
      }
    }
  }
  __syncthreads();
  for (int e = t; e < M * SC_CH; e += TPB) {
    const int m = e / SC_CH, c = e % SC_CH;
    const float v = s_acc[m * SC_LD + c];
    if (v != 0.f && c0 + c < C) atomicAdd(out + ((size_t)b * M + m) * C + c0 + c, v);
  }
}

// ---------------------------------------------------------------- inverse index (CSR) adjoints
// A gather's adjoint is a scatter-add; with float atomics it is slow (LDS float atomics retire about one
// lane per cycle on gfx950: 265 us for the sfp3 interpolation adjoint) and its summation order changes
// from run to run.  The index tensors depend on coordinates only, so the geometry stage (which runs one
// step ahead on a side stream) also builds their INVERSE: for every target row m the ascending list of
// source entries e = r*T + t that reference it.  The adjoint is then a gather-and-sum per target row:
// no atomics, fixed order, bitwise reproducible.
constexpr int CSR_MAXM = 2048;
constexpr int CSR_LDS_E = 32768;   // entries per cloud that fit the LDS staging slab (64 KB of 16-bit entry numbers)
// The slab holds entry numbers e < E <= 32768 as 16-bit values: with 32-bit ones the kernel held 128 + 16 KB of a
// CU's 160 KB for its whole 150-330 us on the geometry branch, and no main-stream workgroup that needs more than
// 16 KB of LDS could be placed on those 16 CUs meanwhile (measured with the in-kernel probe: a 256-workgroup,
// 60 KB launch of the backward pass ran in two rounds whenever it met this kernel).
template <bool LDS_SLAB>
__global__ __launch_bounds__(TPB) void csr_build_kernel(const int *__restrict__ idx, int E, int M,
                                                        int *__restrict__ offsets, int *__restrict__ entries) {
  __shared__ int s_cnt[CSR_MAXM + 1];
  __shared__ int s_cur[CSR_MAXM];
  extern __shared__ unsigned short s_ent[];
  typedef typename std::conditional<LDS_SLAB, unsigned short, int>::type ent_t;
  const int b = blockIdx.x, t = threadIdx.x;
  const int *ii = idx + (size_t)b * E;
  int *off = offsets + (size_t)b * (M + 1), *gent = entries + (size_t)b * E;
  ent_t *ent = LDS_SLAB ? (ent_t *)s_ent : (ent_t *)gent;
  for (int m = t; m <= M; m += TPB) s_cnt[m] = 0;
  __syncthreads();
  for (int e = t; e < E; e += TPB) {
    int m = ii[e];
    m = m < 0 ? 0 : (m >= M ? M - 1 : m);
    atomicAdd(&s_cnt[m], 1);
  }
  __syncthreads();
  if (t < 64) {   // exclusive scan by one wave: 64 lanes x (M/64) consecutive counters
    const int per = (M + 63) / 64, m0 = t * per, m1 = min(M, m0 + per);
    int sum = 0;
    for (int m = m0; m < m1; ++m) sum += s_cnt[m];
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if (t >= d) incl += o;
    }
    int run = incl - sum;
    for (int m = m0; m < m1; ++m) { const int c = s_cnt[m]; s_cnt[m] = run; run += c; }
    if (t == 63) s_cnt[M] = incl;
  }
  __syncthreads();
  for (int m = t; m <= M; m += TPB) { off[m] = s_cnt[m]; if (m < M) s_cur[m] = s_cnt[m]; }
  __syncthreads();
  for (int e = t; e < E; e += TPB) {
    int m = ii[e];
    m = m < 0 ? 0 : (m >= M ? M - 1 : m);
    ent[atomicAdd(&s_cur[m], 1)] = (ent_t)e;
  }
  __syncthreads();
  // the fill order above depends on scheduling: sort every list ascending (insertion sort; the lists are
  // short and arrive nearly sorted)
  for (int m = t; m < M; m += TPB) {
    const int a = s_cnt[m], z = s_cnt[m + 1];
    for (int i = a + 1; i < z; ++i) {
      const ent_t v = ent[i];
      int j = i - 1;
      while (j >= a && ent[j] > v) { ent[j + 1] = ent[j]; --j; }
      ent[j + 1] = v;
    }
  }
  if (LDS_SLAB) {
    __syncthreads();
    for (int e = t; e < E; e += TPB) gent[e] = (int)s_ent[e];
  }
}

// Round 5, the form the step likes ("ordered"): the first version's count / scan / LDS-atomic scatter — lanes that wait on LDS atomics
// issue nothing, which is what makes it a cheap neighbour (NOTEBOOK R5.4) — made to deliver ASCENDING lists without a sort phase:
// every wave owns a contiguous range of the entries and its own row of counters (position = start of the list + what the waves
// before it hold of that list + arrival order inside the wave), and walks its range in order, 64 entries per instruction.  Inside
// one ds_add_rtn instruction the hardware serialises the lanes that hit the same counter; on gfx950 it does so in ascending lane
// order, i.e. in entry order — which no manual promises, so the result is VERIFIED by all lanes (the whole array must be sorted by
// (target, entry): one compare per adjacent pair) and the first version's insertion sort runs if a single pair is out of order.
// The per-list sort that cost ~100 us with 80 KB of LDS held is then a ~3 us check.
template <int NW, bool VERIFY>
__global__ __launch_bounds__(64 * NW) void csr_build_ordered_kernel(const int *__restrict__ idx, int E, int M,
                                                                    int *__restrict__ offsets, int *__restrict__ entries,
                                                                    int *__restrict__ fallbacks) {
  extern __shared__ int s_dyn2[];                  // [NW][M] counters / running positions | [M + 1] offsets | [E] target << 16 | entry
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int *s_w = s_dyn2, *s_off = s_dyn2 + NW * M;
  unsigned *s_ent = (unsigned *)(s_off + M + 1);
  __shared__ int s_bad;
  const int *ii = idx + (size_t)b * E;
  int *off = offsets + (size_t)b * (M + 1), *gent = entries + (size_t)b * E;
  const int per = ((E + NW * 64 - 1) / (NW * 64)) * 64;          // entries per wave, a multiple of 64
  const int e0 = wave * per, e1 = min(E, e0 + per);
  for (int i = t; i < NW * M; i += 64 * NW) s_w[i] = 0;
  if (t == 0) s_bad = 0;
  __syncthreads();
#pragma unroll 4
  for (int e = e0 + lane; e < e1; e += 64) {
    int m = ii[e];
    m = m < 0 ? 0 : (m >= M ? M - 1 : m);
    atomicAdd(&s_w[wave * M + m], 1);
  }
  __syncthreads();
  if (t < 64) {   // exclusive scan of the per-target totals by one wave; the per-wave rows become running positions
    const int pm = (M + 63) / 64, m0 = t * pm, m1 = min(M, m0 + pm);
    int sum = 0;
    for (int m = m0; m < m1; ++m)
      for (int w = 0; w < NW; ++w) sum += s_w[w * M + m];
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if (t >= d) incl += o;
    }
    int run = incl - sum;
    for (int m = m0; m < m1; ++m) {
      s_off[m] = run;
      for (int w = 0; w < NW; ++w) { const int c = s_w[w * M + m]; s_w[w * M + m] = run; run += c; }
    }
    if (t == 63) s_off[M] = incl;
  }
  __syncthreads();
  for (int m = t; m <= M; m += 64 * NW) off[m] = s_off[m];
#pragma unroll 4
  for (int e = e0 + lane; e < e1; e += 64) {          // in order: one instruction = 64 consecutive entries
    int m = ii[e];
    m = m < 0 ? 0 : (m >= M ? M - 1 : m);
    s_ent[atomicAdd(&s_w[wave * M + m], 1)] = ((unsigned)m << 16) | (unsigned)e;
  }
  __syncthreads();
  // verification: sorted by (target, entry)?  (one LDS compare per adjacent pair)
  if (VERIFY) {
    int bad = 0;
    for (int i = t; i + 1 < E; i += 64 * NW) bad |= s_ent[i] >= s_ent[i + 1];
    if (bad) s_bad = 1;
    __syncthreads();
  }
  if (s_bad) {                                         // (never taken on the hardware seen so far; counted for the tests)
    if (t == 0 && fallbacks) atomicAdd(fallbacks, 1);
    for (int m = t; m < M; m += 64 * NW) {
      const int a = s_off[m], z = s_off[m + 1];
      for (int i = a + 1; i < z; ++i) {
        const unsigned v = s_ent[i];
        int j = i - 1;
        while (j >= a && s_ent[j] > v) { s_ent[j + 1] = s_ent[j]; --j; }
        s_ent[j + 1] = v;
      }
    }
    __syncthreads();
  }
  for (int e = t; e < E; e += 64 * NW) gent[e] = (int)(s_ent[e] & 0xffffu);
}

// Round 5: the inverse index as a STABLE LSD RADIX SORT of the entries by target — ascending lists by construction, no
// per-list sort.  The first version above counts, scatters with LDS atomics (order by scheduling) and then sorts every list with one
// lane per list: a ball query's padded rows give a few targets hundreds of entries, and three launches took 531 us per step with
// 64-80 KB of LDS per compute unit.  Here: 3 bits of the target per pass (ceil(log2 M) / 3 passes); a wave takes 64 consecutive
// entries at a time, its rank inside the chunk is a ballot + mbcnt per bucket (lane order = entry order), the rank of the chunk a
// prefix sum over the [bucket][chunk] table of counts (bucket-major: stable); entries travel packed as (target << 21 | e) between two
// global buffers (`entries` and a caller-supplied workspace, L2-resident: 96 KB per cloud), the last pass writes plain e into
// `entries`.  LDS: the table (32 B per chunk) + M counters = 14 KB for the step's largest launch.
constexpr int CSRX_BITS = 3, CSRX_NB = 1 << CSRX_BITS, CSRX_EBITS = 21, CSRX_MAX_E = 32768, CSRX_G = 8;

template <int NT>
__global__ __launch_bounds__(NT) void csr_build_radix_kernel(const int *__restrict__ idx, int E, int M, int npass,
                                                             int *__restrict__ offsets, int *__restrict__ entries,
                                                             int *__restrict__ ws) {
  constexpr int NW = NT / 64;
  extern __shared__ int s_dyn[];                   // [CSRX_NB][nch] chunk table | [M + 1] counters
  __shared__ int s_wsum[NW];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nch = (E + 63) / 64;
  int *s_tab = s_dyn, *s_cnt = s_dyn + CSRX_NB * nch;
  const int *ii = idx + (size_t)b * E;
  int *off = offsets + (size_t)b * (M + 1);
  unsigned *bufE = (unsigned *)(entries + (size_t)b * E), *bufW = (unsigned *)(ws + (size_t)b * E);
  // ---- offsets: histogram of the targets (integer LDS atomics: the COUNTS do not depend on the order) + exclusive scan
  for (int m = t; m <= M; m += NT) s_cnt[m] = 0;
  __syncthreads();
  for (int e = t; e < E; e += NT) {
    int m = ii[e];
    m = m < 0 ? 0 : (m >= M ? M - 1 : m);
    atomicAdd(&s_cnt[m], 1);
  }
  __syncthreads();
  if (t < 64) {
    const int per = (M + 63) / 64, m0 = t * per, m1 = min(M, m0 + per);
    int sum = 0;
    for (int m = m0; m < m1; ++m) sum += s_cnt[m];
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if (t >= d) incl += o;
    }
    int run = incl - sum;
    for (int m = m0; m < m1; ++m) { const int c = s_cnt[m]; off[m] = run; run += c; }
    if (t == 63) off[M] = incl;
  }
  // ---- the passes
  const int L = CSRX_NB * nch, per = (L + NT - 1) / NT;
  for (int p = 0; p < npass; ++p) {
    const int shift = CSRX_EBITS + p * CSRX_BITS;
    const bool last = p == npass - 1;
    // pass p writes dst_p; the last one writes `entries`; they alternate backwards from there
    unsigned *dst = ((npass - 1 - p) & 1) ? bufW : bufE;
    const unsigned *src = ((npass - 1 - p) & 1) ? bufE : bufW;           // = dst_{p-1} (unused in pass 0: the keys come from idx)
    auto fetch = [&](int e) -> unsigned {
      if (p == 0) {
        int m = ii[e];
        m = m < 0 ? 0 : (m >= M ? M - 1 : m);
        return ((unsigned)m << CSRX_EBITS) | (unsigned)e;
      }
      return src[e];
    };
    __syncthreads();                                                     // (previous pass's stores / table reads are done)
    // A: per chunk, the count of every bucket.  A wave takes CSRX_G consecutive chunks per trip and requests all their words before
    // it looks at the first (one chunk per trip was a chain of dependent L2 round trips: 850 cycles per chunk and phase)
    for (int c0 = wave * CSRX_G; c0 < nch; c0 += NW * CSRX_G) {
      unsigned v[CSRX_G];
#pragma unroll
      for (int g = 0; g < CSRX_G; ++g) {
        const int e = (c0 + g) * 64 + lane;
        v[g] = e < E ? fetch(e) : 0u;
      }
#pragma unroll
      for (int g = 0; g < CSRX_G; ++g) {
        const int c = c0 + g, e = c * 64 + lane;
        const int d = e < E ? (int)((v[g] >> shift) & (CSRX_NB - 1)) : CSRX_NB;
        int mine = 0;
#pragma unroll
        for (int q = 0; q < CSRX_NB; ++q) {
          const unsigned long long mk = __ballot(d == q);
          if (lane == q) mine = __popcll(mk);
        }
        if (lane < CSRX_NB && c < nch) s_tab[lane * nch + c] = mine;
      }
    }
    __syncthreads();
    // exclusive prefix sum over the table in bucket-major order: every lane sums `per` consecutive entries, then a block scan
    {
      const int i0 = min(t * per, L), i1 = min(i0 + per, L);
      int sum = 0;
      for (int i = i0; i < i1; ++i) sum += s_tab[i];
      int incl = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
      }
      if (lane == 63) s_wsum[wave] = incl;
      __syncthreads();
      int base = 0;
      for (int w = 0; w < wave; ++w) base += s_wsum[w];
      int run = base + incl - sum;
      for (int i = i0; i < i1; ++i) { const int c = s_tab[i]; s_tab[i] = run; run += c; }
    }
    __syncthreads();
    // B: scatter — position = start of (bucket, chunk) + the number of lower lanes of the chunk in the same bucket
    for (int c0 = wave * CSRX_G; c0 < nch; c0 += NW * CSRX_G) {
      unsigned v[CSRX_G];
#pragma unroll
      for (int g = 0; g < CSRX_G; ++g) {
        const int e = (c0 + g) * 64 + lane;
        v[g] = e < E ? fetch(e) : 0u;
      }
#pragma unroll
      for (int g = 0; g < CSRX_G; ++g) {
        const int c = c0 + g, e = c * 64 + lane;
        const int d = e < E ? (int)((v[g] >> shift) & (CSRX_NB - 1)) : CSRX_NB;
        int rank = 0;
#pragma unroll
        for (int q = 0; q < CSRX_NB; ++q) {      // (the ranks of phase A kept in LDS as one byte per entry instead: SLOWER, 307 against 248 us)
          const unsigned long long mk = __ballot(d == q);
          const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
          if (d == q) rank = below;
        }
        if (e < E) {
          const int pos = s_tab[d * nch + c] + rank;
          dst[pos] = last ? (v[g] & ((1u << CSRX_EBITS) - 1u)) : v[g];
        }
      }
    }
  }
}

// out[b,m,:] = Σ_{e in list(m)} w[b,e] · g[b, e / T, :]      (w may be NULL), bf16 in / bf16 out.
// CS_LANES lanes share one (target row, 8-channel chunk): lane q takes entries q, q+CS_LANES, ... of the list,
// CS_UNROLL at a time, and the partial sums are combined by a fixed butterfly.  The kernel is a chain of dependent loads
// (offsets -> entry -> row), so its time is (trips per lane) x latency: 32 entries per trip instead of 8 took the sa2 /
// sfp3 adjoints from 45 us to ~20 us.  Round 3: those 32 as 8 lanes x 4 instead of 16 lanes x 2 — a wave is then 8 chunks
// (128 contiguous bytes = whole cache lines) of 8 rows per load instead of 4 chunks (half lines) of 16 rows: the measured HBM
// traffic of the three launches was 1.94 x their bytes.
constexpr int CS_LANES = 8;
constexpr int CS_UNROLL = 4;
// ADD (round 4): out = bf16(bf16(sum) + addend[b, m, :]) — the OTHER gradient of a tensor with two consumers (sa2's grouping and
// sfp2's skip concatenation both read sa1's features) arrives as `addend` (bf16, row stride ld_add) and is added here with the
// roundings of the framework's bf16 add that autograd's input buffer launched between the two backward nodes (5 us + a boundary).
template <bool ADD>
__global__ __launch_bounds__(TPB) void csr_gather_sum_kernel(const unsigned short *__restrict__ g, int ldg,
                                                             const int *__restrict__ offsets,
                                                             const int *__restrict__ entries,
                                                             const float *__restrict__ w, int T, int R, int M, int C,
                                                             const unsigned short *__restrict__ addend, int ld_add,
                                                             unsigned short *__restrict__ out) {
  int bx, b;
  xcd_cloud_map(bx, b);
  const int cpr = C / 8;
  const long long xl = (long long)bx * TPB + threadIdx.x;
  const long long x = xl / CS_LANES;
  const int q = (int)(xl % CS_LANES);
  const bool live = x < (long long)M * cpr;
  const int m = live ? (int)(x / cpr) : 0, c0 = live ? (int)(x - (long long)m * cpr) * 8 : 0;
  const int *off = offsets + (size_t)b * (M + 1);
  const int *ent = entries + (size_t)b * R * T;
  const float *wb = w ? w + (size_t)b * R * T : nullptr;
  const unsigned short *gb = g + (size_t)b * R * ldg + c0;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint4 add4 = make_uint4(0, 0, 0, 0);
  if (ADD && live && q == 0) add4 = *(const uint4 *)(addend + ((size_t)b * M + m) * ld_add + c0);   // (in flight during the loop)
  const int i1 = live ? off[m + 1] : 0;
  int i = (live ? off[m] : 0) + q;
  for (; i < i1; i += CS_UNROLL * CS_LANES) {
    int e[CS_UNROLL];
    float wv[CS_UNROLL];
    uint4 r[CS_UNROLL];
#pragma unroll
    for (int u = 0; u < CS_UNROLL; ++u) e[u] = ent[min(i + u * CS_LANES, i1 - 1)];     // (past the end: clamped, weight 0)
#pragma unroll
    for (int u = 0; u < CS_UNROLL; ++u) wv[u] = i + u * CS_LANES < i1 ? (wb ? wb[e[u]] : 1.f) : 0.f;
#pragma unroll
    for (int u = 0; u < CS_UNROLL; ++u) r[u] = *(const uint4 *)(gb + (size_t)(e[u] / T) * ldg);
#pragma unroll
    for (int u = 0; u < CS_UNROLL; ++u) {
      const unsigned short *h = (const unsigned short *)&r[u];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(wv[u], bf2f_(h[j]), acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int msk = 1; msk < CS_LANES; msk <<= 1) acc[j] += __shfl_xor(acc[j], msk);
  if (live && q == 0) {
    unsigned short o[8];
    const unsigned short *ah = (const unsigned short *)&add4;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = ADD ? f2bf_(bf2f_(f2bf_(acc[j])) + bf2f_(ah[j])) : f2bf_(acc[j]);
    *(uint4 *)(out + ((size_t)b * M + m) * C + c0) = *(const uint4 *)o;
  }
}

// sa2-style grouped input rows: out[p,:] = [ feats[b, idx[p], 0:C] | rel[p, 0:3] | 0 ... ]  (bf16, width Cpad)
// (modules/pointset_abstraction.py:62-66: gathered features FIRST, then the centred coordinates)
__global__ __launch_bounds__(TPB) void group_concat_bf16_kernel(const unsigned short *__restrict__ feats,
                                                                const float *__restrict__ rel,
                                                                const int *__restrict__ idx, int N, int R, int C,
                                                                int Cpad, unsigned short *__restrict__ out) {
  // 32 chunk lanes x 8 rows per workgroup pass, GC_ROWS passes per workgroup: no integer divisions, one index load
  // per (row, lane) served from L1, and the rows of a pass are independent loads in flight
  constexpr int GC_ROWS = 4;
  int bx, b;
  xcd_cloud_map(bx, b);
  const int cpr = Cpad / 8;
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int row0 = bx * (8 * GC_ROWS) + ry;
  int ii[GC_ROWS];
#pragma unroll
  for (int u = 0; u < GC_ROWS; ++u) {
    const int r = min(row0 + 8 * u, R - 1);
    const int v = idx[(size_t)b * R + r];
    ii[u] = v < 0 ? 0 : (v >= N ? N - 1 : v);
  }
  for (int c = cx; c < cpr; c += 32) {
    const int c0 = c * 8;
    uint4 v[GC_ROWS];
#pragma unroll
    for (int u = 0; u < GC_ROWS; ++u) {
      const int r = min(row0 + 8 * u, R - 1);
      v[u] = (uint4){0, 0, 0, 0};
      if (c0 + 8 <= C) {
        v[u] = *(const uint4 *)(feats + ((size_t)b * N + ii[u]) * C + c0);
      } else if (c0 == C) {   // C % 8 == 0: the three relative coordinates start a chunk
        unsigned short h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const float *q = rel + ((size_t)b * R + r) * 3;
        h[0] = f2bf_(q[0]); h[1] = f2bf_(q[1]); h[2] = f2bf_(q[2]);
        v[u] = *(const uint4 *)h;
      }
    }
#pragma unroll
    for (int u = 0; u < GC_ROWS; ++u) {
      const int r = row0 + 8 * u;
      if (r < R) *(uint4 *)(out + ((size_t)b * R + r) * Cpad + c0) = v[u];
    }
  }
}

// Up to MC_MAX contiguous buffers copied by ONE launch (16-byte pieces, grid-stride inside each buffer's block
// range): the trainer's static-buffer hand-overs (geometry set B -> A, a new batch into the input buffers) are
// ~20 tensors of a few MB each; torch's multi-tensor copy takes ~25 us for them, a memcpy node per tensor more.
constexpr int MC_MAX = 96;   // (2.7 KB of kernel arguments; the ~57 gradient tensors of a step fit one launch — 40 needed two)
struct McArgs {
  const void *src[MC_MAX];
  void *dst[MC_MAX];
  long long bytes[MC_MAX];
  int block0[MC_MAX + 1];
  int count;
  unsigned *flags;     // CHECK: flags[flag_base + blockIdx.x] = 1 if the block copied an fp32 NaN / inf
  int flag_base;
};
// CHECK: the buffers are fp32 and every copied word is also tested for NaN / inf (exponent all ones) — the finite
// scan of the gradients rides on the copy that packs them into the flat bucket instead of being one more pass.
template <bool CHECK>
__global__ __launch_bounds__(TPB) void multi_copy_kernel(McArgs a) {
  int d = 0;
  while (d + 1 < a.count && (int)blockIdx.x >= a.block0[d + 1]) ++d;
  const long long nb = a.block0[d + 1] - a.block0[d];
  const long long first = (long long)(blockIdx.x - a.block0[d]) * TPB + threadIdx.x;
  const unsigned long long both = (unsigned long long)a.src[d] | (unsigned long long)a.dst[d];
  unsigned bad = 0;
  if ((both & 15) == 0) {
    const long long n16 = a.bytes[d] / 16;
    const uint4 *__restrict__ s = (const uint4 *)a.src[d];
    uint4 *__restrict__ o = (uint4 *)a.dst[d];
    for (long long i = first; i < n16; i += nb * TPB) {
      const uint4 v = s[i];
      o[i] = v;
      if (CHECK)
        bad |= ((v.x & 0x7f800000u) == 0x7f800000u) | ((v.y & 0x7f800000u) == 0x7f800000u) |
               ((v.z & 0x7f800000u) == 0x7f800000u) | ((v.w & 0x7f800000u) == 0x7f800000u);
    }
    if (blockIdx.x == a.block0[d] && threadIdx.x < (a.bytes[d] & 15)) {   // ragged tail, byte by byte
      ((unsigned char *)a.dst[d])[n16 * 16 + threadIdx.x] = ((const unsigned char *)a.src[d])[n16 * 16 + threadIdx.x];
      if (CHECK && (threadIdx.x & 3) == 0 && threadIdx.x + 4 <= (a.bytes[d] & 15)) {
        const unsigned w = *(const unsigned *)((const unsigned char *)a.src[d] + n16 * 16 + threadIdx.x);
        bad |= (w & 0x7f800000u) == 0x7f800000u;
      }
    }
  } else {                                                                // 4-byte aligned views (slices of a flat buffer)
    const long long n4 = a.bytes[d] / 4;
    const unsigned *__restrict__ s = (const unsigned *)a.src[d];
    unsigned *__restrict__ o = (unsigned *)a.dst[d];
    for (long long i = first; i < n4; i += nb * TPB) {
      const unsigned v = s[i];
      o[i] = v;
      if (CHECK) bad |= (v & 0x7f800000u) == 0x7f800000u;
    }
  }
  if (CHECK) {
    __shared__ unsigned s_bad[TPB / 64];
    const unsigned long long m = __ballot(bad != 0);
    if ((threadIdx.x & 63) == 0) s_bad[threadIdx.x >> 6] = m != 0;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned any = 0;
      for (int w = 0; w < TPB / 64; ++w) any |= s_bad[w];
      a.flags[a.flag_base + blockIdx.x] = any;
    }
  }
}

// fp32 matrices -> bf16 (or fp32) panels with a row stride, up to MCV_MAX of them in ONE launch: the per-step refresh
// of every bf16 weight panel of the network (plain, zero-padded-K and the packed heads panel with its fp32 bias
// vector).  torch._foreach_copy_ took 16 us for the ~20 plain panels and the padded ones were one strided copy each.
#include "cast_body.h"
__global__ __launch_bounds__(TPB) void multi_cast_kernel(McvArgs a) { multi_cast_body<TPB>(a, (int)blockIdx.x); }

// group_all set abstraction (sa3): rows [xyz(3) as bf16 | feats(C) bf16 | zeros] with the row length padded to the
// GEMM's K (modules/pointset_abstraction.py:56: pos FIRST).  Was: a dtype cast, torch.cat, torch.zeros and a strided
// slice copy.
__global__ __launch_bounds__(TPB) void concat_pos_feats_kernel(const float *__restrict__ xyz,
                                                               const unsigned short *__restrict__ feats, long long R,
                                                               int C, int Cpad, unsigned short *__restrict__ out) {
  const long long e = (long long)blockIdx.x * TPB + threadIdx.x;
  if (e >= R * Cpad) return;
  const long long r = e / Cpad;
  const int c = (int)(e - r * Cpad);
  unsigned short v = 0;
  if (c < 3) v = __builtin_bit_cast(unsigned short, (__bf16)xyz[r * 3 + c]);
  else if (c < 3 + C) v = feats[r * C + c - 3];
  out[e] = v;
}

// n_gt[b] = max label of cloud b + 1 (SPFN/losses_implementation.py:603-606 via `.max()`): one workgroup per cloud.
__global__ __launch_bounds__(1024) void count_labels_kernel(const long long *__restrict__ labels, int N,
                                                            long long *__restrict__ n_gt) {
  __shared__ long long s_max[16];
  const long long *__restrict__ row = labels + (size_t)blockIdx.x * N;
  long long m = -1;
  for (int i = threadIdx.x; i < N; i += 1024) m = max(m, row[i]);
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) m = max(m, s_max[w]);
    n_gt[blockIdx.x] = m + 1;
  }
}

inline int channel_chunk(int C, int blocks_x, int B) {
  // enough blocks to fill 256 CUs a few times over, but at least 8 channels per block
  // so the index / weight loads are amortised
  int want = (256 * 8) / (blocks_x * (B > 0 ? B : 1));
  if (want < 1) want = 1;
  int per = (C + want - 1) / want;
  if (per < 8) per = 8;
  if (per > C) per = C;
  return per;
}

}  // namespace

extern "C" int cpfn_group_fwd(const float *points, const int *idx, int B, int C, int N, int S, int K,
                              float *out, void *stream) {
  if (B < 0 || C < 0 || N <= 0 || S < 0 || K < 0 || !points || !idx || !out) return CPFN_EINVAL;
  const long long J = (long long)S * K;
  if (B == 0 || C == 0 || J == 0) return 0;
  const int bx = cpfn_cdiv(J, TPB);
  const int per = channel_chunk(C, bx, B);
  dim3 grid(bx, cpfn_cdiv(C, per), B);
  group_fwd_kernel<<<grid, TPB, 0, (hipStream_t)stream>>>(points, idx, C, N, (int)J, per, out);
  return cpfn_launch_status();
}

extern "C" int cpfn_group_bwd(const float *grad_out, const int *idx, int B, int C, int N, int S, int K,
                              float *grad_points, void *stream) {
  if (B < 0 || C < 0 || N <= 0 || S < 0 || K < 0 || !grad_out || !idx || !grad_points) return CPFN_EINVAL;
  const long long J = (long long)S * K;
  if (B == 0 || C == 0 || J == 0) return 0;
  const int bx = cpfn_cdiv(J, TPB);
  const int per = channel_chunk(C, bx, B);
  dim3 grid(bx, cpfn_cdiv(C, per), B);
  group_bwd_kernel<<<grid, TPB, 0, (hipStream_t)stream>>>(grad_out, idx, C, N, (int)J, per, grad_points);
  return cpfn_launch_status();
}

extern "C" int cpfn_three_interp_fwd(const float *feats, const int *idx, const float *w, int B, int C, int M,
                                     int N, float *out, void *stream) {
  if (B < 0 || C < 0 || M <= 0 || N < 0 || !feats || !idx || !w || !out) return CPFN_EINVAL;
  if (B == 0 || C == 0 || N == 0) return 0;
  const int bx = cpfn_cdiv(N, TPB);
  const int per = channel_chunk(C, bx, B);
  dim3 grid(bx, cpfn_cdiv(C, per), B);
  interp_fwd_kernel<<<grid, TPB, 0, (hipStream_t)stream>>>(feats, idx, w, C, M, N, per, out);
  return cpfn_launch_status();
}

extern "C" int cpfn_three_interp_bwd(const float *grad_out, const int *idx, const float *w, int B, int C,
                                     int N, int M, float *grad_feats, void *stream) {
  if (B < 0 || C < 0 || M <= 0 || N < 0 || !grad_out || !idx || !w || !grad_feats) return CPFN_EINVAL;
  if (B == 0 || C == 0 || N == 0) return 0;
  const int bx = cpfn_cdiv(N, TPB);
  const int per = channel_chunk(C, bx, B);
  dim3 grid(bx, cpfn_cdiv(C, per), B);
  interp_bwd_kernel<<<grid, TPB, 0, (hipStream_t)stream>>>(grad_out, idx, w, C, N, M, per, grad_feats);
  return cpfn_launch_status();
}

extern "C" int cpfn_gather_rows(const void *rows, const int *idx, int B, int N, int R, int row_bytes,
                                void *out, void *stream) {
  if (B < 0 || N <= 0 || R < 0 || row_bytes <= 0 || (row_bytes & 3) || !rows || !idx || !out)
    return CPFN_EINVAL;
  if (B == 0 || R == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const bool al16 = (row_bytes % 16 == 0) && (((uintptr_t)rows | (uintptr_t)out) % 16 == 0);
  if (al16) {
    const int vpr = row_bytes / 16;
    dim3 grid(cpfn_cdiv((long long)R * vpr, TPB), B);
    gather_rows_kernel<float4><<<grid, TPB, 0, st>>>((const float4 *)rows, idx, N, R, vpr, (float4 *)out);
  } else {
    const int vpr = row_bytes / 4;
    dim3 grid(cpfn_cdiv((long long)R * vpr, TPB), B);
    gather_rows_kernel<float><<<grid, TPB, 0, st>>>((const float *)rows, idx, N, R, vpr, (float *)out);
  }
  return cpfn_launch_status();
}

extern "C" int cpfn_scatter_add_rows_f32(const float *grad_out, const int *idx, int B, int N, int R, int C,
                                         float *grad_rows, void *stream) {
  if (B < 0 || N <= 0 || R < 0 || C <= 0 || !grad_out || !idx || !grad_rows) return CPFN_EINVAL;
  if (B == 0 || R == 0) return 0;
  dim3 grid(cpfn_cdiv((long long)R * C, TPB), B);
  scatter_add_rows_kernel<<<grid, TPB, 0, (hipStream_t)stream>>>(grad_out, idx, N, R, C, grad_rows);
  return cpfn_launch_status();
}

extern "C" int cpfn_group_xyz_centered(const float *xyz, const float *new_xyz, const int *idx, int B, int N,
                                       int S, int K, float *out, void *stream) {
  if (B < 0 || N <= 0 || S < 0 || K < 0 || !xyz || !new_xyz || !idx || !out) return CPFN_EINVAL;
  if (B == 0 || S == 0 || K == 0) return 0;
  dim3 grid(cpfn_cdiv((long long)S * K * 3, TPB), B);
  group_xyz_centered_kernel<<<grid, TPB, 0, (hipStream_t)stream>>>(xyz, new_xyz, idx, N, S, K, out);
  return cpfn_launch_status();
}

extern "C" int cpfn_interp_rows_fwd(const float *feats, const int *idx, const float *w, int B, int M, int N,
                                    int C, float *out, void *stream) {
  if (B < 0 || M <= 0 || N < 0 || C <= 0 || !feats || !idx || !w || !out) return CPFN_EINVAL;
  if (B == 0 || N == 0) return 0;
  dim3 grid(cpfn_cdiv((long long)N * C, TPB), B);
  interp_rows_fwd_kernel<<<grid, TPB, 0, (hipStream_t)stream>>>(feats, idx, w, M, N, C, out);
  return cpfn_launch_status();
}

extern "C" int cpfn_interp_rows_bwd(const float *grad_out, const int *idx, const float *w, int B, int M,
                                    int N, int C, float *grad_feats, void *stream) {
  if (B < 0 || M <= 0 || N < 0 || C <= 0 || !grad_out || !idx || !w || !grad_feats) return CPFN_EINVAL;
  if (B == 0 || N == 0) return 0;
  dim3 grid(cpfn_cdiv((long long)N * C, TPB), B);
  interp_rows_bwd_kernel<<<grid, TPB, 0, (hipStream_t)stream>>>(grad_out, idx, w, M, N, C, grad_feats);
  return cpfn_launch_status();
}

extern "C" int cpfn_interp_rows_bf16(const void *feats, const int *idx, const float *w, int B, int M, int N, int C,
                                     void *out, void *stream) {
  if (B < 0 || M <= 0 || N < 0 || C <= 0 || (C & 7) || !feats || !idx || !w || !out) return CPFN_EINVAL;
  if (B == 0 || N == 0) return 0;
  dim3 grid(cpfn_cdiv((long long)N * (C / 8), TPB), B);
  interp_rows_bf16_kernel<<<grid, TPB, 0, (hipStream_t)stream>>>((const unsigned short *)feats, idx, w, M, N, C,
                                                                 (unsigned short *)out);
  return cpfn_launch_status();
}

extern "C" int cpfn_concat_interp_bf16(const void *skip, int C1, const void *feats, const int *idx, const float *w, int B,
                                       int M, int N, int C2, void *out, void *stream) {
  if (B < 0 || M <= 0 || N < 0 || C1 < 0 || C2 <= 0 || (C1 & 7) || (C2 & 7) || !feats || !out || (C1 > 0 && !skip) ||
      (!idx != !w) || (!idx && M != 1))
    return CPFN_EINVAL;
  if (B == 0 || N == 0) return 0;
  dim3 grid(cpfn_cdiv((long long)N * ((C1 + C2) / 8), TPB), B);
  concat_interp_bf16_kernel<<<grid, TPB, 0, (hipStream_t)stream>>>((const unsigned short *)skip, C1, (const unsigned short *)feats,
                                                                   idx, w, M, N, C2, (unsigned short *)out);
  return cpfn_launch_status();
}

extern "C" int cpfn_colsum_rows_pass1_bf16(const void *g, int ldg, int B, int N, int C, void *out, const void *yarg, const float *scale,
                                          const float *shift, float *part, void *stream) {
  if (B < 0 || N <= 0 || C <= 0 || (C & 7) || (ldg & 7) || ldg < C || !g || !out) return CPFN_EINVAL;
  if (yarg && (!scale || !shift || !part)) return CPFN_EINVAL;
  if (B == 0) return 0;
  static_assert(TPB == 256, "32 chunks x 8 row subsets");
  const dim3 grid(cpfn_cdiv(C / 8, 32), B);
  if (yarg)
    colsum_rows_bf16_kernel<true><<<grid, TPB, 0, (hipStream_t)stream>>>((const unsigned short *)g, ldg, N, C, (unsigned short *)out,
                                                                         (const unsigned short *)yarg, scale, shift, part);
  else
    colsum_rows_bf16_kernel<false><<<grid, TPB, 0, (hipStream_t)stream>>>((const unsigned short *)g, ldg, N, C, (unsigned short *)out);
  return cpfn_launch_status();
}

extern "C" int cpfn_colsum_rows_bf16(const void *g, int ldg, int B, int N, int C, void *out, void *stream) {
  return cpfn_colsum_rows_pass1_bf16(g, ldg, B, N, C, out, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int cpfn_scatter_rows_bf16(const void *g, int ldg, const int *idx, const float *w, int T, int B, int R,
                                      int M, int C, float *out, void *stream) {
  if (B < 0 || R < 0 || M <= 0 || M > 1024 || C <= 0 || T < 1 || T > 3 || ldg < C || !g || !idx || !out)
    return CPFN_EINVAL;
  if (B == 0 || R == 0) return 0;
  const size_t lds = (size_t)M * SC_LD * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)scatter_rows_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       1024 * SC_LD * (int)sizeof(float));
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  // ~4 row-blocks per (cloud, channel chunk) beyond the first 256 workgroups
  const int cch = cpfn_cdiv(C, SC_CH);
  int xb = cpfn_cdiv(512, (long long)B * cch);
  if (xb < 1) xb = 1;
  int rpb = cpfn_cdiv(R, xb);
  rpb = ((rpb + 7) / 8) * 8;
  xb = cpfn_cdiv(R, rpb);
  scatter_rows_lds_kernel<<<dim3(xb, cch, B), TPB, lds, (hipStream_t)stream>>>((const unsigned short *)g, ldg, idx, w, T,
                                                                               R, M, C, rpb, out);
  return cpfn_launch_status();
}

extern "C" int cpfn_group_concat_bf16(const void *feats, const float *rel, const int *idx, int B, int N, int R, int C,
                                      int Cpad, void *out, void *stream) {
  // rel == NULL: gather only (Cpad == C: the coordinates travel separately, cpfn_mlp_gemm_xyz)
  if (B < 0 || N <= 0 || R < 0 || C <= 0 || (C & 7) || (Cpad & 7) || (rel ? Cpad < C + 8 : Cpad != C) || !feats || !idx || !out)
    return CPFN_EINVAL;
  if (B == 0 || R == 0) return 0;
  dim3 grid(cpfn_cdiv(R, 32), B);      // 8 rows x 4 passes per workgroup
  group_concat_bf16_kernel<<<grid, TPB, 0, (hipStream_t)stream>>>((const unsigned short *)feats, rel, idx, N, R, C, Cpad,
                                                                  (unsigned short *)out);
  return cpfn_launch_status();
}

extern "C" int cpfn_csr_build(const int *idx, int B, int E, int M, int *offsets, int *entries, void *stream) {
  if (B < 0 || E < 0 || M <= 0 || M > CSR_MAXM || !idx || !offsets || !entries) return CPFN_EINVAL;
  if (B == 0) return 0;
  if (E <= CSR_LDS_E) {
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void *)csr_build_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         CSR_LDS_E * (int)sizeof(unsigned short));
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
    csr_build_kernel<true><<<B, TPB, (size_t)E * sizeof(unsigned short), (hipStream_t)stream>>>(idx, E, M, offsets, entries);
  } else {
    csr_build_kernel<false><<<B, TPB, 0, (hipStream_t)stream>>>(idx, E, M, offsets, entries);
  }
  return cpfn_launch_status();
}

// cpfn_csr_build with a caller-supplied workspace [B, E] int32: the radix-sort form (E <= 65536); without it, or beyond, the first
// version.  threads: 0 = the default (256 lanes per cloud), or 512 / 1024.  E <= 32768.
extern "C" int cpfn_csr_build_ws(const int *idx, int B, int E, int M, int *offsets, int *entries, int *workspace, int threads,
                                 void *stream) {
  if (B < 0 || E < 0 || M <= 0 || M > CSR_MAXM || !idx || !offsets || !entries) return CPFN_EINVAL;
  if (B == 0) return 0;
  if (!workspace || E > CSRX_MAX_E || E == 0) return cpfn_csr_build(idx, B, E, M, offsets, entries, stream);
  if (threads < 0) {          // "ordered": per-wave ranges + in-order LDS atomics + verification (the workspace's first word counts fall-backs)
    const int NWr = threads == -8 ? 8 : threads == -16 ? 16 : 4;          // (-8 / -16: timing experiments; 4 waves is what the step likes)
    const size_t lds2 = sizeof(int) * ((size_t)NWr * M + M + 1) + sizeof(unsigned) * (size_t)E + 64;
    if (lds2 <= 150 * 1024 && M <= 65536 && E <= 65536) {
      static bool attr_set3 = false;
      if (!attr_set3) {
        hipError_t e = hipFuncSetAttribute((const void *)csr_build_ordered_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)csr_build_ordered_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)csr_build_ordered_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)csr_build_ordered_kernel<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set3 = true;
      }
      hipStream_t st = (hipStream_t)stream;
      if (threads == -2) csr_build_ordered_kernel<4, false><<<B, 256, lds2, st>>>(idx, E, M, offsets, entries, workspace);   // (timing experiments only)
      else if (NWr == 8) csr_build_ordered_kernel<8, true><<<B, 512, lds2, st>>>(idx, E, M, offsets, entries, workspace);
      else if (NWr == 16) csr_build_ordered_kernel<16, true><<<B, 1024, lds2, st>>>(idx, E, M, offsets, entries, workspace);
      else csr_build_ordered_kernel<4, true><<<B, 256, lds2, st>>>(idx, E, M, offsets, entries, workspace);
      return cpfn_launch_status();
    }
    return cpfn_csr_build(idx, B, E, M, offsets, entries, stream);     // (the one-word workspace is no scratch for the radix passes)
  }
  int bits = 1;
  while ((1 << bits) < M) ++bits;
  const int npass = (bits + CSRX_BITS - 1) / CSRX_BITS;
  const int nch = (E + 63) / 64;
  const size_t lds = sizeof(int) * ((size_t)CSRX_NB * nch + M + 1);
  hipStream_t st = (hipStream_t)stream;
  if (threads == 1024) csr_build_radix_kernel<1024><<<B, 1024, lds, st>>>(idx, E, M, npass, offsets, entries, workspace);
  else if (threads == 512) csr_build_radix_kernel<512><<<B, 512, lds, st>>>(idx, E, M, npass, offsets, entries, workspace);
  else csr_build_radix_kernel<256><<<B, 256, lds, st>>>(idx, E, M, npass, offsets, entries, workspace);
  return cpfn_launch_status();
}

extern "C" int cpfn_csr_gather_sum_add_bf16(const void *g, int ldg, const int *offsets, const int *entries, const float *w,
                                            int T, int B, int R, int M, int C, const void *addend, int ld_add, void *out,
                                            void *stream) {
  if (B < 0 || R < 0 || M <= 0 || C <= 0 || (C & 7) || (ldg & 7) || ldg < C || T < 1 || !g || !offsets || !entries || !out ||
      (addend && ((ld_add & 7) || ld_add < C || ((uintptr_t)addend & 15))))
    return CPFN_EINVAL;
  if (B == 0) return 0;
  dim3 grid(cpfn_cdiv((long long)M * (C / 8) * CS_LANES, TPB), B);
  if (addend)
    csr_gather_sum_kernel<true><<<grid, TPB, 0, (hipStream_t)stream>>>((const unsigned short *)g, ldg, offsets, entries, w, T, R, M,
                                                                       C, (const unsigned short *)addend, ld_add,
                                                                       (unsigned short *)out);
  else
    csr_gather_sum_kernel<false><<<grid, TPB, 0, (hipStream_t)stream>>>((const unsigned short *)g, ldg, offsets, entries, w, T, R,
                                                                        M, C, nullptr, 0, (unsigned short *)out);
  return cpfn_launch_status();
}

extern "C" int cpfn_csr_gather_sum_bf16(const void *g, int ldg, const int *offsets, const int *entries, const float *w,
                                        int T, int B, int R, int M, int C, void *out, void *stream) {
  return cpfn_csr_gather_sum_add_bf16(g, ldg, offsets, entries, w, T, B, R, M, C, nullptr, 0, out, stream);
}

static int multi_copy_impl(const cpfn_copy_desc *descs, int count, unsigned *flags, int flags_capacity, bool launch,
                           hipStream_t st, int *blocks_total) {
  int total = 0;
  for (int base = 0; base < count; base += MC_MAX) {
    McArgs a;
    a.count = count - base < MC_MAX ? count - base : MC_MAX;
    int blocks = 0;
    for (int i = 0; i < a.count; ++i) {
      const cpfn_copy_desc &d = descs[base + i];
      const uintptr_t both = (uintptr_t)d.src | (uintptr_t)d.dst;
      if (!d.src || !d.dst || d.bytes < 0 || (both & 3) || ((both & 15) && (d.bytes & 3))) return CPFN_EINVAL;
      if (flags && (d.bytes & 3)) return CPFN_EINVAL;                       // checked copies are whole fp32 words
      a.src[i] = d.src; a.dst[i] = d.dst; a.bytes[i] = d.bytes;
      a.block0[i] = blocks;
      long long nb = (d.bytes / 16 + TPB * 4 - 1) / (TPB * 4);            // ~4 pieces per lane
      if (nb < 1) nb = 1;
      if (nb > 1024) nb = 1024;
      blocks += (int)nb;
    }
    a.block0[a.count] = blocks;
    a.flags = flags;
    a.flag_base = total;
    if (launch && blocks) {
      if (flags) {
        if (total + blocks > flags_capacity) return CPFN_EINVAL;
        multi_copy_kernel<true><<<blocks, TPB, 0, st>>>(a);
      } else {
        multi_copy_kernel<false><<<blocks, TPB, 0, st>>>(a);
      }
    }
    total += blocks;
  }
  if (blocks_total) *blocks_total = total;
  return 0;
}

extern "C" int cpfn_multi_copy(const cpfn_copy_desc *descs, int count, void *stream) {
  if (count < 0 || (count > 0 && !descs)) return CPFN_EINVAL;
  const int e = multi_copy_impl(descs, count, nullptr, 0, true, (hipStream_t)stream, nullptr);
  return e ? e : cpfn_launch_status();
}

extern "C" int cpfn_multi_copy_blocks(const cpfn_copy_desc *descs, int count) {
  if (count < 0 || (count > 0 && !descs)) return -1;
  int total = 0;
  return multi_copy_impl(descs, count, nullptr, 0, false, nullptr, &total) ? -1 : total;
}

extern "C" int cpfn_multi_copy_checked(const cpfn_copy_desc *descs, int count, unsigned *flags, int flags_capacity,
                                       void *stream) {
  if (count < 0 || (count > 0 && !descs) || !flags || flags_capacity < 0) return CPFN_EINVAL;
  const int e = multi_copy_impl(descs, count, flags, flags_capacity, true, (hipStream_t)stream, nullptr);
  return e ? e : cpfn_launch_status();
}

extern "C" int cpfn_multi_cast(const cpfn_cast_desc *descs, int count, void *stream) {
  if (count < 0 || (count > 0 && !descs)) return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < count; base += MCV_MAX) {
    McvArgs a;
    int blocks = 0;
    const int rc = mcv_fill(descs + base, count - base < MCV_MAX ? count - base : MCV_MAX, a, &blocks);
    if (rc) return rc;
    if (blocks) multi_cast_kernel<<<blocks, TPB, 0, st>>>(a);
  }
  return cpfn_launch_status();
}

extern "C" int cpfn_concat_pos_feats_bf16(const float *xyz, const void *feats, long long R, int C, int Cpad, void *out,
                                          void *stream) {
  if (R < 0 || C < 0 || Cpad < C + 3 || !xyz || (C > 0 && !feats) || !out) return CPFN_EINVAL;
  if (R == 0) return 0;
  concat_pos_feats_kernel<<<(unsigned)cpfn_cdiv(R * Cpad, TPB), TPB, 0, (hipStream_t)stream>>>(
      xyz, (const unsigned short *)feats, R, C, Cpad, (unsigned short *)out);
  return cpfn_launch_status();
}

extern "C" int cpfn_count_labels(const int64_t *labels, int B, int N, int64_t *n_gt, void *stream) {
  if (B < 0 || N <= 0 || !labels || !n_gt) return CPFN_EINVAL;
  if (B == 0) return 0;
  count_labels_kernel<<<B, 1024, 0, (hipStream_t)stream>>>((const long long *)labels, N, (long long *)n_gt);
  return cpfn_launch_status();
}
