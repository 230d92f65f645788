// Patch merging at evaluation time (config 5: 32 local patches of 8192 points over a 131072-point cloud) for gfx950.
//
//  * cpfn_similarity_soft — Utils/merging_utils.py:6-15.  The reference scatters the soft labels of every patch
//    into a dense point-to-primitive matrix M [N, nb*Lp + Lo] (367 MB at N = 131072, C = 700) and returns the Gram
//    matrix M^T M with a dense fp32 GEMM (128 GFLOP).  M is block-sparse — a point lies in ~2 of the 32 patches —
//    so the Gram matrix is assembled block by block instead, without ever materialising M:
//        block(si, sj) = sum over rows r of source si whose point also lies in source sj of
//                        vals_i[r, :]^T vals_j[row_of_j(point(r)), :]
//    where a "source" is a patch (Lp columns, npp rows) or the global labelling (Lo columns, N rows).  One wave per
//    (si, sj, 256-row split): 32 rows at a time, index lookup through a [nb, N] row table, a ballot skips tiles
//    with no common point, the others go through 16 fp32 MFMAs (v_mfma_f32_32x32x2_f32; both operands gathered
//    straight from global memory, Lp, Lo <= 32 zero-padded in registers).  Splits are summed in a fixed order by a
//    second kernel: no atomics, reproducible, fp32 products and sums like the reference's GEMM.
//  * cpfn_label_pool — Utils/merging_utils.py:56-60 (get_point_final): the reference multiplies M by a normalised
//    one-hot [C, G] matrix; here every output is the sum of its label's columns (ascending column order) times
//    1/(count + 1e-10): one streaming pass over M.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int SIM_TILE = 32;             // rows per MFMA group
// rows per wave = 32 * 8: measured at config-5 size (patch rows | global rows, us): 32 tiles 186 | 65, 16: 141 | 34,
// 8: 125 | 19, 4: 128 | 12 (+ a 4x larger partial buffer to reduce)
constexpr int SIM_TILES_PER_SPLIT = 8;
static inline int sim_tps() { return SIM_TILES_PER_SPLIT; }

__global__ __launch_bounds__(256) void row_of_scatter_kernel(const long long *__restrict__ pidx, int nb, int npp, int N,
                                                             int *__restrict__ row_of) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)nb * npp) return;
  const int b = (int)(e / npp), r = (int)(e - (long long)b * npp);
  const long long p = pidx[e];
  if (p >= 0 && p < N) row_of[(size_t)b * N + p] = r;
}

// row block of source si0 + blockIdx.x / ncols against column source sj0 + blockIdx.x % ncols; blockIdx.y = row
// split.  Only blocks with sj >= si are computed (the Gram matrix is symmetric: sim_mirror_kernel fills the rest).
__global__ __launch_bounds__(256) void sim_pair_kernel(const float *__restrict__ spfn, const float *__restrict__ pred,
                                                      const long long *__restrict__ pidx,
                                                      const int *__restrict__ row_of, int N, int nb, int npp, int Lp,
                                                      int Lo, int si0, int sj0, int ncols, int nsplits, int tps,
                                                      float *__restrict__ partial /*[splits][rows_blk][C]*/,
                                                      int rows_blk /* rows of the partial slab */) {
  const int C = nb * Lp + Lo;
  const int si = si0 + blockIdx.x / ncols, sj = sj0 + blockIdx.x % ncols;
  const int split = blockIdx.y * 4 + (threadIdx.x >> 6);   // one wave per row split, four splits per workgroup
  if (sj < si || split >= nsplits) return;
  const bool gi = si == nb, gj = sj == nb;
  const int rows_i = gi ? N : npp, L_i = gi ? Lo : Lp, L_j = gj ? Lo : Lp;
  const float *__restrict__ vals_i = gi ? spfn : pred + (size_t)si * npp * Lp;
  const float *__restrict__ vals_j = gj ? spfn : pred + (size_t)sj * npp * Lp;
  const long long *__restrict__ pid_i = gi ? nullptr : pidx + (size_t)si * npp;
  const int *__restrict__ rof_j = gj ? nullptr : row_of + (size_t)sj * N;
  const int lane = threadIdx.x & 63, x = lane & 31, k = lane >> 5;
  const int ntiles = (rows_i + SIM_TILE - 1) / SIM_TILE;
  const int t0 = split * tps, t1 = min(t0 + tps, ntiles);

  // point of row r (or -1), and its row in source sj (or -1); lanes x and x+32 hold the same row
  auto point_of = [&](int tile) -> int {
    const int r = tile * SIM_TILE + x;
    if (tile >= t1 || r >= rows_i) return -1;
    if (gi) return r;
    const long long p = pid_i[r];
    return (p >= 0 && p < N) ? (int)p : -1;
  };
  auto row_in_j = [&](int p) -> int { return p < 0 ? -1 : (gj ? p : rof_j[p]); };

  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  // two-stage index pipeline: the point ids of tile n+2 and the row lookups of tile n+1 are in flight while tile n
  // is multiplied
  int p1 = point_of(t0 + 1);
  int rj0 = row_in_j(point_of(t0));
  for (int tile = t0; tile < t1; ++tile) {
    const int p2 = point_of(tile + 2);
    const int rj1 = row_in_j(p1);
    const int rj = rj0;
    if (__ballot(rj >= 0) != 0ull) {
      const int r0 = tile * SIM_TILE;
      // all 32 operand loads are issued before the first MFMA: unconditional loads from clamped addresses, zeroed
      // by a select afterwards (conditional loads were serialised load -> wait -> MFMA: 16 round trips per tile)
      float av[16], bv[16];
      const int xa = min(x, L_i - 1), xb = min(x, L_j - 1);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int row = 2 * t + k;                 // MFMA t contracts rows r0 + 2t and r0 + 2t + 1
        const int rjr = __shfl(rj, row, 64);
        const float a = vals_i[(size_t)min(r0 + row, rows_i - 1) * L_i + xa];
        const float b = vals_j[(size_t)max(rjr, 0) * L_j + xb];
        av[t] = (x < L_i && r0 + row < rows_i) ? a : 0.f;
        bv[t] = (x < L_j && rjr >= 0) ? b : 0.f;
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
    }
    p1 = p2;
    rj0 = rj1;
  }
  // D[i][j]: j = lane & 31, i = 8 (v / 4) + 4 (lane >> 5) + (v % 4)
  const int row_base = gi ? 0 : (si - si0) * Lp;           // row inside the partial slab
  const int col0 = gj ? nb * Lp : sj * Lp;
  float *__restrict__ o = partial + (size_t)split * rows_blk * C;
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const int i = 8 * (v >> 2) + 4 * k + (v & 3);
    if (i < L_i && x < L_j) o[(size_t)(row_base + i) * C + col0 + x] = acc[v];
  }
}

// out[e] = sum over splits of partial[split][e], fixed order: 64 elements x 4 split-subsets per workgroup
__global__ __launch_bounds__(256) void sim_reduce_kernel(const float *__restrict__ partial, int splits, long long n,
                                                         float *__restrict__ out) {
  __shared__ float s_acc[4][64];
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const long long e = (long long)blockIdx.x * 64 + lane;
  float s = 0.f;
  if (e < n) {
#pragma unroll 4
    for (int i = sub; i < splits; i += 4) s += partial[(size_t)i * n + e];
  }
  s_acc[sub][lane] = s;
  __syncthreads();
  if (sub == 0 && e < n) out[e] = ((s_acc[0][lane] + s_acc[1][lane]) + s_acc[2][lane]) + s_acc[3][lane];
}

// lower block triangle: out[r][c] = out[c][r] where the source of column c precedes the source of row r
__global__ __launch_bounds__(256) void sim_mirror_kernel(float *__restrict__ out, int nb, int Lp, int C) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)C * C) return;
  const int r = (int)(e / C), c = (int)(e - (long long)r * C);
  const int sr = r < nb * Lp ? r / Lp : nb, sc = c < nb * Lp ? c / Lp : nb;
  if (sc < sr) out[e] = out[(size_t)c * C + r];
}

// Column lists per label (one workgroup; C, G are a few hundred): offsets[G+1], order[C] = the columns of label 0
// in ascending order, then those of label 1, ...; inv_count[g] = 1 / (float(count) + 1e-10f) like merging_utils.py:58
// (labels outside [0, G) are ignored).
__global__ __launch_bounds__(1024) void label_index_kernel(const long long *__restrict__ labels, int C, int G,
                                                           int *__restrict__ order, int *__restrict__ offsets,
                                                           float *__restrict__ inv_count) {
  extern __shared__ int s_idx[];   // labels [C] | counts -> offsets [G+1]
  int *s_lab = s_idx, *s_off = s_idx + C;
  for (int c = threadIdx.x; c < C; c += 1024) {
    const long long l = labels[c];
    s_lab[c] = (l >= 0 && l < G) ? (int)l : -1;
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += 1024) {
    int n = 0;
    for (int c = 0; c < C; ++c) n += s_lab[c] == g;
    s_off[g + 1] = n;
    inv_count[g] = 1.0f / ((float)n + 1e-10f);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s_off[0] = 0;
    for (int g = 0; g < G; ++g) s_off[g + 1] += s_off[g];
  }
  __syncthreads();
  for (int g = threadIdx.x; g <= G; g += 1024) offsets[g] = s_off[g];
  for (int g = threadIdx.x; g < G; g += 1024) {
    int q = s_off[g];
    for (int c = 0; c < C; ++c)
      if (s_lab[c] == g) order[q++] = c;
  }
}

// out[p][g] = inv_count[g] * sum over columns c of label g (ascending c) of M[p][c]
constexpr int LP_MAX_FLOATS = 12288;   // row staging per workgroup (48 KB)
__global__ __launch_bounds__(256) void label_pool_kernel(const float *__restrict__ M, const int *__restrict__ order,
                                                         const int *__restrict__ offsets,
                                                         const float *__restrict__ inv_count, long long N, int C,
                                                         int G, int rows_per_block, float *__restrict__ out) {
  extern __shared__ float s_dyn[];    // rows [rows_per_block][C] | order [C] | offsets [G+1]
  float *s_rows = s_dyn;
  int *s_order = (int *)(s_dyn + (size_t)rows_per_block * C);
  int *s_off = s_order + C;
  const long long p0 = (long long)blockIdx.x * rows_per_block;
  const int nrows = (int)min((long long)rows_per_block, N - p0);
  const float *__restrict__ src = M + p0 * C;
  const int tot = nrows * C;
  if ((((uintptr_t)src) & 15) == 0 && (tot & 3) == 0) {
    // the whole slab (<= 12 float4 per thread) is requested before the first LDS store
    cpfn_f32x4 v[LP_MAX_FLOATS / 1024];   // (plain vector type: arrays of HIP's float4 struct end up in scratch memory)
#pragma unroll
    for (int i = 0; i < LP_MAX_FLOATS / 1024; ++i) {
      const int e = (threadIdx.x + i * 256) * 4;
      v[i] = e < tot ? *(const cpfn_f32x4 *)&src[e] : (cpfn_f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < LP_MAX_FLOATS / 1024; ++i) {
      const int e = (threadIdx.x + i * 256) * 4;
      if (e < tot) *(cpfn_f32x4 *)&s_rows[e] = v[i];
    }
  } else {
    for (int e = threadIdx.x; e < tot; e += 256) s_rows[e] = src[e];
  }
  for (int e = threadIdx.x; e < C; e += 256) s_order[e] = order[e];
  for (int e = threadIdx.x; e <= G; e += 256) s_off[e] = offsets[e];
  __syncthreads();
  for (int o = threadIdx.x; o < nrows * G; o += 256) {
    const int row = o / G, g = o - row * G;
    const float *r = s_rows + (size_t)row * C;
    float s = 0.f;
    const int q1 = s_off[g + 1];
#pragma unroll 4
    for (int q = s_off[g]; q < q1; ++q) s += r[s_order[q]];
    out[(p0 + row) * G + g] = s * inv_count[g];
  }
}

inline int sim_splits(int rows) {
  const int tiles = (rows + SIM_TILE - 1) / SIM_TILE;
  const int tps = sim_tps();
  return tiles > 0 ? (tiles + tps - 1) / tps : 1;
}
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" long long cpfn_similarity_soft_workspace(int N, int nb, int npp, int Lp, int Lo) {
  if (N <= 0 || nb < 0 || npp < 0 || Lp <= 0 || Lo <= 0) return -1;
  const size_t C = (size_t)nb * Lp + Lo;
  return (long long)(align256((size_t)nb * N * 4) + align256((size_t)sim_splits(npp) * nb * Lp * C * 4) +
                     align256((size_t)sim_splits(N) * Lo * C * 4));
}

extern "C" int cpfn_similarity_soft(const float *spfn_labels, const float *predicted_labels,
                                    const int64_t *point_indices, int N, int nb, int npp, int Lp, int Lo,
                                    void *workspace, float *out, void *stream) {
  if (N <= 0 || nb < 0 || npp < 0 || Lp <= 0 || Lo <= 0 || Lp > 32 || Lo > 32 || !spfn_labels || !workspace || !out)
    return CPFN_EINVAL;
  if (nb > 0 && npp > 0 && (!predicted_labels || !point_indices)) return CPFN_EINVAL;
  if ((long long)nb * N > 2000000000LL) return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int C = nb * Lp + Lo, S1 = nb + 1;
  char *ws = (char *)workspace;
  int *row_of = (int *)ws;
  float *part_p = (float *)(ws + align256((size_t)nb * N * 4));
  float *part_g = (float *)((char *)part_p + align256((size_t)sim_splits(npp) * nb * Lp * C * 4));
  const long long ne = (long long)nb * npp;
  if (nb > 0) {
    hipError_t e = hipMemsetAsync(row_of, 0xFF, (size_t)nb * N * 4, st);   // every entry -1
    if (e != hipSuccess) return (int)e;
    if (ne > 0)
      row_of_scatter_kernel<<<(unsigned)cpfn_cdiv(ne, 256), 256, 0, st>>>((const long long *)point_indices, nb, npp, N, row_of);
    const int sp = sim_splits(npp);
    sim_pair_kernel<<<dim3(nb * S1, (sp + 3) / 4), 256, 0, st>>>(spfn_labels, predicted_labels, (const long long *)point_indices,
                                                                row_of, N, nb, npp, Lp, Lo, 0, 0, S1, sp, sim_tps(), part_p, nb * Lp);
    const long long n1 = (long long)nb * Lp * C;
    sim_reduce_kernel<<<(unsigned)cpfn_cdiv(n1, 64), 256, 0, st>>>(part_p, sp, n1, out);
  }
  const int sg = sim_splits(N);
  sim_pair_kernel<<<dim3(1, (sg + 3) / 4), 256, 0, st>>>(spfn_labels, predicted_labels, (const long long *)point_indices, row_of,
                                                        N, nb, npp, Lp, Lo, nb, nb, 1, sg, sim_tps(), part_g, Lo);
  const long long n2 = (long long)Lo * C;
  sim_reduce_kernel<<<(unsigned)cpfn_cdiv(n2, 64), 256, 0, st>>>(part_g, sg, n2, out + (size_t)nb * Lp * C);
  if (nb > 0) sim_mirror_kernel<<<(unsigned)cpfn_cdiv((long long)C * C, 256), 256, 0, st>>>(out, nb, Lp, C);
  return cpfn_launch_status();
}

extern "C" int cpfn_label_pool(const float *M, const int64_t *labels, long long N, int C, int G, void *workspace,
                               float *out, void *stream) {
  if (N < 0 || C <= 0 || G <= 0 || C > LP_MAX_FLOATS || G > 16384 || !M || !labels || !workspace || !out) return CPFN_EINVAL;
  if (N == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int *order = (int *)workspace, *offsets = order + C;
  float *inv_count = (float *)(offsets + G + 1);
  label_index_kernel<<<1, 1024, (size_t)(C + G + 1) * 4, st>>>((const long long *)labels, C, G, order, offsets, inv_count);
  int rpb = LP_MAX_FLOATS / C;
  rpb = rpb > 8 ? 8 : rpb;   // (16 / 8 / 4 / 2 rows per workgroup measured: 183 / 161 / 157 / 182 us per call at config-5 size)
  const size_t lds = ((size_t)rpb * C + C + G + 1) * 4;
  label_pool_kernel<<<(unsigned)cpfn_cdiv(N, rpb), 256, lds, st>>>(M, order, offsets, inv_count, N, C, G, rpb, out);
  return cpfn_launch_status();
}
