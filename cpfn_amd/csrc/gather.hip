// Index-driven data movement for gfx950: grouping / gathering / 3-point interpolation
// and their adjoints, in the reference's channel-major fp32 layout (drop-in for the
// `cuda_ops` names) and in the points-major row layout the MI355X path uses natively.
//
// All of these are HBM/L2-bound byte movers: one lane per output element (channel-major)
// or one 16-byte vector per lane (row layout), outputs written fully coalesced, the
// small gathered-from tensor left to L2.  Adjoints use fp32 atomics exactly like the
// reference's kernels (group_points_gpu.cu:60, interpolate_gpu.cu:139-141).
#include "common.h"

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
__device__ __forceinline__ float bf2f_(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
__device__ __forceinline__ unsigned short f2bf_(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

// out[b,n,:] = Σ_t w[b,n,t]·feats[b,idx[b,n,t],:]   bf16 in / bf16 out, fp32 math, 8 channels per lane
__global__ __launch_bounds__(TPB) void interp_rows_bf16_kernel(const unsigned short *__restrict__ feats,
                                                               const int *__restrict__ idx,
                                                               const float *__restrict__ w, int M, int N, int C,
                                                               unsigned short *__restrict__ out) {
  const int b = blockIdx.y;
  const int cpr = C / 8;
  const long long e = (long long)blockIdx.x * TPB + threadIdx.x;
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

// Scatter-add of bf16 gradient rows into a SMALL fp32 target [B,M,C] (M <= 1024 rows per cloud):
//   target[b, idx[b,r,t], c] += w[b,r,t] · g[b,r,c]        (T = 1 with w = NULL: plain gather adjoint)
// The target slab of a 32-channel chunk lives in LDS (M x 32 fp32 <= 128 KB), every contribution is an
// LDS atomic (64 lanes of one instruction hit 64 different banks), and the slab is flushed with one
// global atomic per element per workgroup: ~50x fewer global atomics than scattering row by row.
constexpr int SC_CH = 32;
__global__ __launch_bounds__(TPB) void scatter_rows_lds_kernel(const unsigned short *__restrict__ g, int ldg,
                                                               const int *__restrict__ idx,
                                                               const float *__restrict__ w, int T, int R, int M,
                                                               int C, int rows_per_block, float *__restrict__ out) {
  extern __shared__ float s_acc[];  // [M][SC_CH]
  const int b = blockIdx.z, c0 = blockIdx.y * SC_CH;
  const int t = threadIdx.x, cl = t % SC_CH, rs = t / SC_CH;  // 8 row sub-lanes
  for (int e = t; e < M * SC_CH; e += TPB) s_acc[e] = 0.f;
  __syncthreads();
  const int r0 = blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
  if (c0 + cl < C) {
    for (int r = r0 + rs; r < r1; r += TPB / SC_CH) {
      const float gv = bf2f_(g[((size_t)b * R + r) * ldg + c0 + cl]);
      for (int tt = 0; tt < T; ++tt) {
        int m = idx[((size_t)b * R + r) * T + tt];
        m = m < 0 ? 0 : (m >= M ? M - 1 : m);
        const float wt = w ? w[((size_t)b * R + r) * T + tt] : 1.f;
        atomicAdd(&s_acc[m * SC_CH + cl], wt * gv);
      }
    }
  }
  __syncthreads();
  for (int e = t; e < M * SC_CH; e += TPB) {
    const int m = e / SC_CH, c = e % SC_CH;
    const float v = s_acc[e];
    if (v != 0.f && c0 + c < C) atomicAdd(out + ((size_t)b * M + m) * C + c0 + c, v);
  }
}

// sa2-style grouped input rows: out[p,:] = [ feats[b, idx[p], 0:C] | rel[p, 0:3] | 0 ... ]  (bf16, width Cpad)
// (modules/pointset_abstraction.py:62-66: gathered features FIRST, then the centred coordinates)
__global__ __launch_bounds__(TPB) void group_concat_bf16_kernel(const unsigned short *__restrict__ feats,
                                                                const float *__restrict__ rel,
                                                                const int *__restrict__ idx, int N, int R, int C,
                                                                int Cpad, unsigned short *__restrict__ out) {
  const int b = blockIdx.y;
  const int cpr = Cpad / 8;
  const long long e = (long long)blockIdx.x * TPB + threadIdx.x;
  if (e >= (long long)R * cpr) return;
  const int r = (int)(e / cpr), c0 = (int)(e - (long long)r * cpr) * 8;
  uint4 v = {0, 0, 0, 0};
  if (c0 + 8 <= C) {
    int ii = idx[(size_t)b * R + r];
    ii = ii < 0 ? 0 : (ii >= N ? N - 1 : ii);
    v = *(const uint4 *)(feats + ((size_t)b * N + ii) * C + c0);
  } else if (c0 == C) {   // C % 8 == 0: the three relative coordinates start a chunk
    unsigned short h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float *q = rel + ((size_t)b * R + r) * 3;
    h[0] = f2bf_(q[0]); h[1] = f2bf_(q[1]); h[2] = f2bf_(q[2]);
    v = *(const uint4 *)h;
  }
  *(uint4 *)(out + ((size_t)b * R + r) * Cpad + c0) = v;
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

extern "C" int cpfn_scatter_rows_bf16(const void *g, int ldg, const int *idx, const float *w, int T, int B, int R,
                                      int M, int C, float *out, void *stream) {
  if (B < 0 || R < 0 || M <= 0 || M > 1024 || C <= 0 || T < 1 || T > 3 || ldg < C || !g || !idx || !out)
    return CPFN_EINVAL;
  if (B == 0 || R == 0) return 0;
  const size_t lds = (size_t)M * SC_CH * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)scatter_rows_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       1024 * SC_CH * (int)sizeof(float));
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
  if (B < 0 || N <= 0 || R < 0 || C <= 0 || (C & 7) || (Cpad & 7) || Cpad < C + 8 || !feats || !rel || !idx || !out)
    return CPFN_EINVAL;
  if (B == 0 || R == 0) return 0;
  dim3 grid(cpfn_cdiv((long long)R * (Cpad / 8), TPB), B);
  group_concat_bf16_kernel<<<grid, TPB, 0, (hipStream_t)stream>>>((const unsigned short *)feats, rel, idx, N, R, C, Cpad,
                                                                  (unsigned short *)out);
  return cpfn_launch_status();
}
