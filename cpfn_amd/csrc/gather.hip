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
