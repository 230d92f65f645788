// Ball query and 3-nearest-neighbour search for gfx950.
//
// Both evaluate the reference's CPU distance (modules/geometry_utils.py:4-23) bit for
// bit: D = ((-2*dot) + |q|²) + |p|² with dot an fma chain over x, y, z.
// The DIRECT instantiations follow the reference's CUDA route instead (`fast=True`): the
// distance is (q-p)² summed over x, y, z (ball_query_gpu.cu:29-30, interpolate_gpu.cu:34),
// the ball is `d2 < radius*radius` in fp32 (:21, :31).  nvcc contracts a*a + b*b + c*c into
// an fma chain by default (-fmad=true); that is what is spelled here.  No CUDA build of the
// reference can run in this image, so the DIRECT route is NOT pinned bit for bit.
//
// ball_query : one 64-lane wave per query.  The wave strides over the cloud 64 points
//              at a time in index order; `__ballot` + popcount-of-lower-lanes gives each
//              kept point its output slot, so the first K kept indices land in index
//              order with no sort, and the wave stops as soon as K are found.
// three_nn   : one lane per query, candidates broadcast from LDS as float4 (x,y,z,|p|²),
//              three-deep insertion with strict '<' (ties keep the lower index).
#include "common.h"

namespace {

constexpr int BQ_WAVES = 2;     // (beside a training step: 1 / 2 / 4 / 8 waves per workgroup -> 1.846 / 1.844 / 1.852 / 1.849 ms per step)

__device__ __forceinline__ float direct_sqdist(float qx, float qy, float qz, float x, float y, float z) {
  const float dx = __fsub_rn(qx, x), dy = __fsub_rn(qy, y), dz = __fsub_rn(qz, z);
  return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
}

// PACKED: xyz is [B, N, 4] = (x, y, z, |p|^2) (cpfn_pack_xyzn): one 16-byte load per point and no norm per (point, query)
// — 6 instead of 11 vector operations per point in the scan of the wave-per-query kernel.
__global__ void pack_xyzn_kernel(const float *__restrict__ xyz, long long R, float4 *__restrict__ out) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float x = xyz[3 * r], y = xyz[3 * r + 1], z = xyz[3 * r + 2];
  out[r] = make_float4(x, y, z, cpfn_sqnorm3(x, y, z));
}

template <bool DIRECT, int BQW = BQ_WAVES, bool PACKED = false>
__global__ __launch_bounds__(BQW *CPFN_WAVE) void ball_query_kernel(
    const float *__restrict__ xyz, const float *__restrict__ new_xyz, int B, int N, int S, float thr, int K,
    int *__restrict__ idx_out, float *__restrict__ rel_out = nullptr /* [B,S,K,3]: xyz[idx] - centre (group_xyz_centered) */) {
  const int lane = threadIdx.x & (CPFN_WAVE - 1);
  // (grid-stride over the queries: the host may launch fewer workgroups than queries / 4 — see cpfn_ball_query)
  for (long long q = (long long)blockIdx.x * BQW + (threadIdx.x / CPFN_WAVE); q < (long long)B * S;
       q += (long long)gridDim.x * BQW) {   // wave-uniform
  const int b = (int)(q / S);
  const float *p = xyz + (size_t)b * N * (PACKED ? 4 : 3);
  const float *c = new_xyz + (size_t)q * 3;
  int *out = idx_out + (size_t)q * K;
  const float qx = c[0], qy = c[1], qz = c[2];
  const float qn = cpfn_sqnorm3(qx, qy, qz);

  int cnt = 0;
  int first = DIRECT ? 0 : N;  // empty ball: the CPU route's sort-based code pads with N, the CUDA route's output stays 0
  for (int base = 0; base < N && cnt < K; base += CPFN_WAVE) {
    const int n = base + lane;
    bool keep = false;
    float x = 0.f, y = 0.f, z = 0.f;
    if (n < N) {
      float pn;
      if (PACKED) {
        const float4 v = ((const float4 *)p)[n];
        x = v.x; y = v.y; z = v.z; pn = v.w;
      } else {
        x = p[3 * n]; y = p[3 * n + 1]; z = p[3 * n + 2];
        pn = DIRECT ? 0.f : cpfn_sqnorm3(x, y, z);
      }
      if (DIRECT) {
        keep = direct_sqdist(qx, qy, qz, x, y, z) < thr;
      } else {
        const float d = cpfn_pair_sqdist(qx, qy, qz, qn, x, y, z, pn);
        keep = !(d > thr);
      }
    }
    const unsigned long long mask = __ballot(keep);
    if (mask) {
      if (cnt == 0) first = base + __builtin_ctzll(mask);
      const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
      if (keep && pos < K) {
        out[pos] = n;
        if (rel_out) {           // the kept neighbour's centred coordinates: the lane holds them (group_xyz_centered's subtraction)
          float *r = rel_out + ((size_t)q * K + pos) * 3;
          r[0] = __fsub_rn(x, qx); r[1] = __fsub_rn(y, qy); r[2] = __fsub_rn(z, qz);
        }
      }
      cnt += __popcll(mask);
    }
  }
  if (cnt > K) cnt = K;
  if (rel_out && cnt + lane < K) {           // padding slots: the first kept neighbour again (an empty ball: index N, read as N - 1)
    const int fi = first < 0 ? 0 : (first >= N ? N - 1 : first);
    float x, y, z;
    if (PACKED) { const float4 v = ((const float4 *)p)[fi]; x = v.x; y = v.y; z = v.z; }
    else { x = p[3 * fi]; y = p[3 * fi + 1]; z = p[3 * fi + 2]; }
    const float rx = __fsub_rn(x, qx), ry = __fsub_rn(y, qy), rz = __fsub_rn(z, qz);
    for (int k = cnt + lane; k < K; k += CPFN_WAVE) {
      float *r = rel_out + ((size_t)q * K + k) * 3;
      r[0] = rx; r[1] = ry; r[2] = rz;
    }
  }
  for (int k = cnt + lane; k < K; k += CPFN_WAVE) out[k] = first;
  }
}

// The same query, sixteen per workgroup (sixteen waves, all of one cloud), with the cloud walked through LDS in tiles of
// 2048 points as (x, y, z, |p|^2): a wave then pays one 16-byte LDS read and the six distance operations per point
// instead of three global loads, the norm (five operations) and the distance — the wave-per-query kernel above is
// VALU / issue bound (8192 waves x ~100 trips x ~25 instructions for sa1).  Same arithmetic (the norm is the same
// cpfn_sqnorm3, computed once per point and tile instead of once per point and query), same index order, same early
// exit (per wave; the workgroup leaves when all sixteen are done).
constexpr int BQT_WAVES = 16, BQT_TILE = 2048;
template <bool DIRECT>
__global__ __launch_bounds__(BQT_WAVES *CPFN_WAVE) void ball_query_tiled_kernel(
    const float *__restrict__ xyz, const float *__restrict__ new_xyz, int N, int S, float thr, int K,
    int *__restrict__ idx_out) {
  __shared__ float4 s_pts[BQT_TILE];
  const int t = threadIdx.x, lane = t & (CPFN_WAVE - 1), wave = t / CPFN_WAVE;
  const int b = blockIdx.y;
  const int sq = blockIdx.x * BQT_WAVES + wave;          // query of this wave inside cloud b (S % 16 == 0: always valid)
  const float *p = xyz + (size_t)b * N * 3;
  const float *c = new_xyz + ((size_t)b * S + sq) * 3;
  int *out = idx_out + ((size_t)b * S + sq) * K;
  const float qx = c[0], qy = c[1], qz = c[2];
  const float qn = cpfn_sqnorm3(qx, qy, qz);
  int cnt = 0;
  int first = DIRECT ? 0 : N;
  for (int tile0 = 0; tile0 < N; tile0 += BQT_TILE) {
    const int tn = min(BQT_TILE, N - tile0);
    for (int j = t; j < tn; j += BQT_WAVES * CPFN_WAVE) {
      const float x = p[3 * (tile0 + j)], y = p[3 * (tile0 + j) + 1], z = p[3 * (tile0 + j) + 2];
      s_pts[j] = make_float4(x, y, z, cpfn_sqnorm3(x, y, z));
    }
    __syncthreads();
    for (int base = 0; base < tn && cnt < K; base += CPFN_WAVE) {
      const int j = base + lane, n = tile0 + j;
      bool keep = false;
      if (j < tn) {
        const cpfn_f32x4 k4 = cpfn_lds_read4((const float *)&s_pts[j]);
        if (DIRECT) {
          keep = direct_sqdist(qx, qy, qz, k4.x, k4.y, k4.z) < thr;
        } else {
          const float d = cpfn_pair_sqdist(qx, qy, qz, qn, k4.x, k4.y, k4.z, k4.w);
          keep = !(d > thr);
        }
      }
      const unsigned long long mask = __ballot(keep);
      if (mask) {
        if (cnt == 0) first = tile0 + base + __builtin_ctzll(mask);
        const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
        if (keep && pos < K) out[pos] = n;
        cnt += __popcll(mask);
      }
    }
    if (__syncthreads_or(cnt < K) == 0) break;            // (also the barrier in front of the next tile's fill)
  }
  if (cnt > K) cnt = K;
  for (int k = cnt + lane; k < K; k += CPFN_WAVE) out[k] = first;
}

constexpr int NN_THREADS = 256;
constexpr int NN_TILE = 1024;

template <bool DIRECT, int NNT = NN_THREADS>
__global__ __launch_bounds__(NNT) void three_nn_kernel(const float *__restrict__ unknown,
                                                              const float *__restrict__ known, int N, int M,
                                                              float *__restrict__ dist2, int *__restrict__ idx,
                                                              int sqrt_out, float *__restrict__ w_out = nullptr) {
  __shared__ float4 s_known[NN_TILE];
  const int b = blockIdx.y;
  const int i = blockIdx.x * NNT + threadIdx.x;
  const float *kn = known + (size_t)b * M * 3;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (i < N) {
    const float *u = unknown + ((size_t)b * N + i) * 3;
    ux = u[0]; uy = u[1]; uz = u[2];
  }
  const float un = cpfn_sqnorm3(ux, uy, uz);
  float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
  int i0 = DIRECT ? 0 : M, i1 = i0, i2 = i0;      // (interpolate_gpu.cu:29: the CUDA route starts from index 0)
  for (int base = 0; base < M; base += NN_TILE) {
    const int cntk = min(NN_TILE, M - base);
    __syncthreads();
    for (int j = threadIdx.x; j < cntk; j += NNT) {
      const float x = kn[3 * (base + j)], y = kn[3 * (base + j) + 1], z = kn[3 * (base + j) + 2];
      s_known[j] = make_float4(x, y, z, cpfn_sqnorm3(x, y, z));
    }
    __syncthreads();
    // four candidates per trip: their LDS reads are issued together (one exposed LDS round trip per FOUR candidates —
    // the rolled loop waited for every single one: 94 cycles per candidate and wave) and the distances are independent;
    // the insertions stay sequential and in index order (strict '<': the lower index keeps a tie)
#define CPFN_NN_INSERT(dv, jv)                                   \
    do {                                                         \
      const float d_ = (dv);                                     \
      const int jj_ = (jv);                                      \
      if (d_ < d2) {                                             \
        if (d_ < d1) {                                           \
          d2 = d1; i2 = i1;                                      \
          if (d_ < d0) { d1 = d0; i1 = i0; d0 = d_; i0 = jj_; }  \
          else { d1 = d_; i1 = jj_; }                            \
        } else { d2 = d_; i2 = jj_; }                            \
      }                                                          \
    } while (0)
#define CPFN_NN_DIST(k4) (DIRECT ? direct_sqdist(ux, uy, uz, (k4).x, (k4).y, (k4).z) \
                                 : cpfn_pair_sqdist(ux, uy, uz, un, (k4).x, (k4).y, (k4).z, (k4).w))
    int j = 0;
    for (; j + 4 <= cntk; j += 4) {
      const cpfn_f32x4 ka = cpfn_lds_read4((const float *)&s_known[j]), kb = cpfn_lds_read4((const float *)&s_known[j + 1]);
      const cpfn_f32x4 kc = cpfn_lds_read4((const float *)&s_known[j + 2]), kd = cpfn_lds_read4((const float *)&s_known[j + 3]);
      const float da = CPFN_NN_DIST(ka), db = CPFN_NN_DIST(kb), dc = CPFN_NN_DIST(kc), dd = CPFN_NN_DIST(kd);
      if (fminf(fminf(da, db), fminf(dc, dd)) < d2) {       // (NaN distances never enter, as in the rolled loop)
        CPFN_NN_INSERT(da, base + j);
        CPFN_NN_INSERT(db, base + j + 1);
        CPFN_NN_INSERT(dc, base + j + 2);
        CPFN_NN_INSERT(dd, base + j + 3);
      }
    }
    for (; j < cntk; ++j) {
      const cpfn_f32x4 k4 = cpfn_lds_read4((const float *)&s_known[j]);   // (the DIRECT variant uses 3 of the 4 floats)
      CPFN_NN_INSERT(CPFN_NN_DIST(k4), base + j);
    }
  }
  if (i < N) {
    float *od = dist2 + ((size_t)b * N + i) * 3;
    int *oi = idx + ((size_t)b * N + i) * 3;
    if (sqrt_out) {   // geometry_utils.py:184.  Through fp64: __fsqrt_rn was measured 1 ulp off torch.sqrt / sqrtf on
                      // gfx950; the fp64 root rounded to fp32 is the correctly rounded fp32 root
      d0 = (float)sqrt((double)d0); d1 = (float)sqrt((double)d1); d2 = (float)sqrt((double)d2);
    }
    od[0] = d0; od[1] = d1; od[2] = d2;
    oi[0] = i0; oi[1] = i1; oi[2] = i2;
    if (w_out) {      // three_weights_kernel's arithmetic on the three distances just stored (one launch less per interpolation)
      const float a = __fdiv_rn(1.0f, __fadd_rn(d0, 1e-8f)), bq = __fdiv_rn(1.0f, __fadd_rn(d1, 1e-8f)), c = __fdiv_rn(1.0f, __fadd_rn(d2, 1e-8f));
      const float sm = __fadd_rn(__fadd_rn(a, bq), c);
      float *ow = w_out + ((size_t)b * N + i) * 3;
      ow[0] = __fdiv_rn(a, sm); ow[1] = __fdiv_rn(bq, sm); ow[2] = __fdiv_rn(c, sm);
    }
  }
}

#undef CPFN_NN_INSERT

// The same search with FOUR lanes per query (adjacent lanes; lane s takes candidates s, s + 4, ...): the lane-per-query
// kernel above runs two waves per SIMD at 16 x 8192 queries and exposes its LDS round trips and branches; this one runs
// eight.  Each lane keeps its three best by strict '<' (its candidates come in ascending index order), then the four
// triples of a query are merged in the order (distance, index) — the sequential scan's result exactly: lowest distances,
// the lower index on a tie.
template <bool DIRECT>
__global__ __launch_bounds__(NN_THREADS) void three_nn_quad_kernel(const float *__restrict__ unknown,
                                                                   const float *__restrict__ known, int N, int M,
                                                                   float *__restrict__ dist2, int *__restrict__ idx,
                                                                   int sqrt_out, float *__restrict__ w_out = nullptr) {
  __shared__ float4 s_known[NN_TILE];
  const int b = blockIdx.y;
  const int sub = threadIdx.x & 3;
  const int i = blockIdx.x * (NN_THREADS / 4) + (threadIdx.x >> 2);
  const float *kn = known + (size_t)b * M * 3;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (i < N) {
    const float *u = unknown + ((size_t)b * N + i) * 3;
    ux = u[0]; uy = u[1]; uz = u[2];
  }
  const float un = cpfn_sqnorm3(ux, uy, uz);
  float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
  int i0 = DIRECT ? 0 : M, i1 = i0, i2 = i0;
  for (int base = 0; base < M; base += NN_TILE) {
    const int cntk = min(NN_TILE, M - base);
    __syncthreads();
    for (int j = threadIdx.x; j < cntk; j += NN_THREADS) {
      const float x = kn[3 * (base + j)], y = kn[3 * (base + j) + 1], z = kn[3 * (base + j) + 2];
      s_known[j] = make_float4(x, y, z, cpfn_sqnorm3(x, y, z));
    }
    __syncthreads();
    for (int j = sub; j < cntk; j += 8) {          // two candidates of this lane per trip
      const cpfn_f32x4 ka = cpfn_lds_read4((const float *)&s_known[j]);
      const bool two = j + 4 < cntk;
      const cpfn_f32x4 kb = cpfn_lds_read4((const float *)&s_known[two ? j + 4 : j]);
      const float da = CPFN_NN_DIST(ka), db = two ? CPFN_NN_DIST(kb) : INFINITY;
      const int ja = base + j, jb = base + j + 4;
      if (da < d2) {
        if (da < d1) {
          d2 = d1; i2 = i1;
          if (da < d0) { d1 = d0; i1 = i0; d0 = da; i0 = ja; } else { d1 = da; i1 = ja; }
        } else { d2 = da; i2 = ja; }
      }
      if (db < d2) {
        if (db < d1) {
          d2 = d1; i2 = i1;
          if (db < d0) { d1 = d0; i1 = i0; d0 = db; i0 = jb; } else { d1 = db; i1 = jb; }
        } else { d2 = db; i2 = jb; }
      }
    }
  }
  // merge the four lanes of a query: butterfly over the quad (xor 1, xor 2); after it every lane holds the query's result
#pragma unroll
  for (int m = 1; m <= 2; m <<= 1) {
    const float e0 = __shfl_xor(d0, m, 64), e1 = __shfl_xor(d1, m, 64), e2 = __shfl_xor(d2, m, 64);
    const int f0 = __shfl_xor(i0, m, 64), f1 = __shfl_xor(i1, m, 64), f2 = __shfl_xor(i2, m, 64);
    const float ed[3] = {e0, e1, e2};
    const int ei[3] = {f0, f1, f2};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float d = ed[q];
      const int jj = ei[q];
      // (distance, index) order; an entry equal to one already held (the defaults: +inf with the same index) changes nothing
      if (d < d2 || (d == d2 && jj < i2)) {
        if (d < d1 || (d == d1 && jj < i1)) {
          d2 = d1; i2 = i1;
          if (d < d0 || (d == d0 && jj < i0)) { d1 = d0; i1 = i0; d0 = d; i0 = jj; } else { d1 = d; i1 = jj; }
        } else { d2 = d; i2 = jj; }
      }
    }
  }
  if (i < N && sub == 0) {
    float *od = dist2 + ((size_t)b * N + i) * 3;
    int *oi = idx + ((size_t)b * N + i) * 3;
    if (sqrt_out) {
      d0 = (float)sqrt((double)d0); d1 = (float)sqrt((double)d1); d2 = (float)sqrt((double)d2);
    }
    od[0] = d0; od[1] = d1; od[2] = d2;
    oi[0] = i0; oi[1] = i1; oi[2] = i2;
    if (w_out) {      // three_weights_kernel's arithmetic on the three distances just stored (one launch less per interpolation)
      const float a = __fdiv_rn(1.0f, __fadd_rn(d0, 1e-8f)), bq = __fdiv_rn(1.0f, __fadd_rn(d1, 1e-8f)), c = __fdiv_rn(1.0f, __fadd_rn(d2, 1e-8f));
      const float sm = __fadd_rn(__fadd_rn(a, bq), c);
      float *ow = w_out + ((size_t)b * N + i) * 3;
      ow[0] = __fdiv_rn(a, sm); ow[1] = __fdiv_rn(bq, sm); ow[2] = __fdiv_rn(c, sm);
    }
  }
}
#undef CPFN_NN_DIST

__global__ void three_weights_kernel(const float *__restrict__ dist, long long R, float *__restrict__ w) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float a = __fdiv_rn(1.0f, __fadd_rn(dist[3 * r], 1e-8f));
  const float b = __fdiv_rn(1.0f, __fadd_rn(dist[3 * r + 1], 1e-8f));
  const float c = __fdiv_rn(1.0f, __fadd_rn(dist[3 * r + 2], 1e-8f));
  const float s = __fadd_rn(__fadd_rn(a, b), c);
  w[3 * r] = __fdiv_rn(a, s);
  w[3 * r + 1] = __fdiv_rn(b, s);
  w[3 * r + 2] = __fdiv_rn(c, s);
}

// Materialised distance matrix (the reference's pairwise_squared_distance itself); the hot
// path never needs it — ball_query / three_nn evaluate the same expression on the fly.
__global__ __launch_bounds__(256) void pairwise_sqdist_kernel(const float *__restrict__ src,
                                                              const float *__restrict__ dst, int N, int M,
                                                              float *__restrict__ out) {
  const int b = blockIdx.z;
  const int i = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= M) return;
  const float *s = src + ((size_t)b * N + i) * 3;
  const float *d = dst + ((size_t)b * M + j) * 3;
  const float sx = s[0], sy = s[1], sz = s[2], dx = d[0], dy = d[1], dz = d[2];
  out[((size_t)b * N + i) * M + j] =
      cpfn_pair_sqdist(sx, sy, sz, cpfn_sqnorm3(sx, sy, sz), dx, dy, dz, cpfn_sqnorm3(dx, dy, dz));
}

}  // namespace

// "Background" geometry (cpfn_set_background_geometry): the call runs on a side stream BESIDE other work (the next batch's
// geometry beside a training step).  The two fast variants — ball query through LDS tiles (84 -> 50 us for the step's two
// launches), four lanes per 3-NN query (78 -> 50 us) — then make the step they run beside SLOWER (interleaved A/B on one
// box: +19 us and +5 us per step): full 1024-thread workgroups / eight waves per SIMD take more from their neighbours than
// the time they save, and the geometry has 0.4 ms of slack anyway.  Throttling the wave-per-query kernel's grid the other
// way (1024 / 512 / 256 workgroups) is worse too (+14 / +31 / +45 us).  So: background calls use the wave-per-query and
// lane-per-query kernels, everything else (evaluation, stand-alone calls: the geometry is on the critical path) the fast ones.
// (The flag itself lives in abi.hip; FPS uses it too: sampling.hip.)
static int bq_grid(long long Q) { return cpfn_cdiv(Q, BQ_WAVES); }

extern "C" int cpfn_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S, float thr, int K,
                               int *idx_out, void *stream) {
  if (B < 0 || N <= 0 || S < 0 || K <= 0 || !xyz || !new_xyz || !idx_out) return CPFN_EINVAL;
  const long long Q = (long long)B * S;
  if (Q == 0) return 0;
  if (S % BQT_WAVES == 0 && N >= 512 && !cpfn_background_geometry())
    ball_query_tiled_kernel<false><<<dim3(S / BQT_WAVES, B), BQT_WAVES * CPFN_WAVE, 0, (hipStream_t)stream>>>(
        xyz, new_xyz, N, S, thr, K, idx_out);
  else
    ball_query_kernel<false><<<bq_grid(Q), BQ_WAVES * CPFN_WAVE, 0, (hipStream_t)stream>>>(
        xyz, new_xyz, B, N, S, thr, K, idx_out);
  return cpfn_launch_status();
}

extern "C" int cpfn_pack_xyzn(const float *xyz, int B, int N, float *out, void *stream) {
  if (B < 0 || N < 0 || !xyz || !out || ((uintptr_t)out & 15)) return CPFN_EINVAL;
  const long long R = (long long)B * N;
  if (R == 0) return 0;
  pack_xyzn_kernel<<<cpfn_cdiv(R, 256), 256, 0, (hipStream_t)stream>>>(xyz, R, (float4 *)out);
  return cpfn_launch_status();
}

extern "C" int cpfn_ball_query_packed_rel(const float *xyzn, const float *new_xyz, int B, int N, int S, float thr, int K,
                                          int *idx_out, float *rel_out, void *stream) {
  if (B < 0 || N <= 0 || S < 0 || K <= 0 || !xyzn || !new_xyz || !idx_out || ((uintptr_t)xyzn & 15)) return CPFN_EINVAL;
  const long long Q = (long long)B * S;
  if (Q == 0) return 0;
  ball_query_kernel<false, BQ_WAVES, true><<<bq_grid(Q), BQ_WAVES * CPFN_WAVE, 0, (hipStream_t)stream>>>(
      xyzn, new_xyz, B, N, S, thr, K, idx_out, rel_out);
  return cpfn_launch_status();
}

extern "C" int cpfn_ball_query_packed(const float *xyzn, const float *new_xyz, int B, int N, int S, float thr, int K,
                                      int *idx_out, void *stream) {
  return cpfn_ball_query_packed_rel(xyzn, new_xyz, B, N, S, thr, K, idx_out, nullptr, stream);
}

extern "C" int cpfn_ball_query_direct(const float *xyz, const float *new_xyz, int B, int N, int S, float radius, int K,
                                      int *idx_out, void *stream) {
  if (B < 0 || N <= 0 || S < 0 || K <= 0 || !xyz || !new_xyz || !idx_out) return CPFN_EINVAL;
  const long long Q = (long long)B * S;
  if (Q == 0) return 0;
  if (S % BQT_WAVES == 0 && N >= 512 && !cpfn_background_geometry())
    ball_query_tiled_kernel<true><<<dim3(S / BQT_WAVES, B), BQT_WAVES * CPFN_WAVE, 0, (hipStream_t)stream>>>(
        xyz, new_xyz, N, S, radius * radius, K, idx_out);
  else
    ball_query_kernel<true><<<cpfn_cdiv(Q, BQ_WAVES), BQ_WAVES * CPFN_WAVE, 0, (hipStream_t)stream>>>(
        xyz, new_xyz, B, N, S, radius * radius, K, idx_out);
  return cpfn_launch_status();
}

extern "C" int cpfn_three_nn_weights(const float *unknown, const float *known, int B, int N, int M, int direct, int sqrt_out,
                                     float *dist2, int *idx, float *w, void *stream) {
  if (B < 0 || N < 0 || M < 0 || !unknown || !known || !dist2 || !idx) return CPFN_EINVAL;
  if (B == 0 || N == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const bool quad = M >= 64 && !cpfn_background_geometry();
  const dim3 grid(cpfn_cdiv(N, quad ? NN_THREADS / 4 : NN_THREADS), B);
  if (direct) {
    if (quad) three_nn_quad_kernel<true><<<grid, NN_THREADS, 0, st>>>(unknown, known, N, M, dist2, idx, sqrt_out, w);
    else three_nn_kernel<true><<<grid, NN_THREADS, 0, st>>>(unknown, known, N, M, dist2, idx, sqrt_out, w);
  } else {
    if (quad) three_nn_quad_kernel<false><<<grid, NN_THREADS, 0, st>>>(unknown, known, N, M, dist2, idx, 0, w);
    else three_nn_kernel<false><<<grid, NN_THREADS, 0, st>>>(unknown, known, N, M, dist2, idx, 0, w);
  }
  return cpfn_launch_status();
}

extern "C" int cpfn_three_nn(const float *unknown, const float *known, int B, int N, int M, float *dist2,
                             int *idx, void *stream) {
  return cpfn_three_nn_weights(unknown, known, B, N, M, 0, 0, dist2, idx, nullptr, stream);
}

extern "C" int cpfn_three_nn_direct(const float *unknown, const float *known, int B, int N, int M, int sqrt_out,
                                    float *dist, int *idx, void *stream) {
  if (B < 0 || N < 0 || M < 0 || !unknown || !known || !dist || !idx) return CPFN_EINVAL;
  if (B == 0 || N == 0) return 0;
  if (M >= 64 && !cpfn_background_geometry()) {
    three_nn_quad_kernel<true><<<dim3(cpfn_cdiv(N, NN_THREADS / 4), B), NN_THREADS, 0, (hipStream_t)stream>>>(unknown, known, N, M,
                                                                                                     dist, idx, sqrt_out);
    return cpfn_launch_status();
  }
  dim3 grid(cpfn_cdiv(N, NN_THREADS), B);
  three_nn_kernel<true><<<grid, NN_THREADS, 0, (hipStream_t)stream>>>(unknown, known, N, M, dist, idx, sqrt_out);
  return cpfn_launch_status();
}

extern "C" int cpfn_three_weights(const float *dist, int64_t R, float *w, void *stream) {
  if (R < 0 || !dist || !w) return CPFN_EINVAL;
  if (R == 0) return 0;
  three_weights_kernel<<<cpfn_cdiv(R, 256), 256, 0, (hipStream_t)stream>>>(dist, R, w);
  return cpfn_launch_status();
}

extern "C" int cpfn_pairwise_sqdist(const float *src, const float *dst, int B, int N, int M, float *out,
                                    void *stream) {
  if (B < 0 || N < 0 || M < 0 || !src || !dst || !out) return CPFN_EINVAL;
  if (B == 0 || N == 0 || M == 0) return 0;
  if (N > 65535 || B > 65535) return CPFN_EINVAL;
  dim3 grid(cpfn_cdiv(M, 256), N, B);
  pairwise_sqdist_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src, dst, N, M, out);
  return cpfn_launch_status();
}
