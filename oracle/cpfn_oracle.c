/*
 * cpfn_oracle.c — CPU restatement of the reference's `fast=False` geometry path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cpfn_amd/ may import, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg do, and only as the checker / the timed CPU baseline.
 *
 * Every function restates, in scalar fp32 C, the arithmetic the reference's
 * pure-PyTorch route performs on CPU tensors (file:line relative to the
 * reference checkout).  Parity is pinned by the .npz files under tests/golden, which were
 * produced by importing the reference's Python in the build container
 * (tests/golden/make_golden.py); tests/test_oracle_golden.py replays them.
 *
 * Build:  gcc -O2 -fPIC -shared -ffp-contract=off -fno-fast-math cpfn_oracle.c -lm
 * `-ffp-contract=off` matters: every product/sum below is individually rounded
 * unless written as fmaf(), exactly as the reference's elementwise torch ops
 * (and its K=3 SGEMM inner product) round them.
 *
 * Layouts: point coordinates are [B, N, 3] row-major (what the reference's
 * native ops take, PointNet2/pointnet2_ops/modules/geometry_utils.py:86);
 * features are channel-major [B, C, N] like the reference's tensors.
 * Indices are int64 (the reference's torch.long).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(_OPENMP)
#include <omp.h>
#endif

/* ‖p‖² the way `torch.sum(src ** 2, dim=1)` rounds it on a [B,3,N] tensor:
 * ((x² + y²) + z²), each op rounded (geometry_utils.py:21-22). */
static inline float sqnorm3(float x, float y, float z) {
  float xx = x * x, yy = y * y, zz = z * z;
  float s = xx + yy;
  return s + zz;
}

/* One entry of pairwise_squared_distance (geometry_utils.py:4-23):
 *   dist  = -2 * matmul(srcᵀ, dst)        -> K=3 SGEMM: fma chain x, then y, then z
 *   dist += ‖src‖²   (line 21)
 *   dist += ‖dst‖²   (line 22)
 * `sn`, `dn` are the precomputed norms of the src / dst point. */
static inline float pair_sqdist(float sx, float sy, float sz, float sn,
                                float dx, float dy, float dz, float dn) {
  float dot = sx * dx;
  dot = fmaf(sy, dy, dot);
  dot = fmaf(sz, dz, dot);
  float d = -2.0f * dot;
  d = d + sn;
  d = d + dn;
  return d;
}

/* geometry_utils.py:4-23.  src [B,N,3], dst [B,M,3] -> out [B,N,M]. */
void orc_pairwise_sqdist(const float *src, const float *dst, int B, int N, int M,
                         float *out) {
  for (int b = 0; b < B; ++b) {
    const float *s = src + (size_t)b * N * 3;
    const float *d = dst + (size_t)b * M * 3;
    float *o = out + (size_t)b * N * M;
    for (int i = 0; i < N; ++i) {
      float sx = s[3 * i], sy = s[3 * i + 1], sz = s[3 * i + 2];
      float sn = sqnorm3(sx, sy, sz);
      for (int j = 0; j < M; ++j) {
        float dx = d[3 * j], dy = d[3 * j + 1], dz = d[3 * j + 2];
        o[(size_t)i * M + j] = pair_sqdist(sx, sy, sz, sn, dx, dy, dz, sqnorm3(dx, dy, dz));
      }
    }
  }
}

/* farthest_point_sample, fast=False branch (geometry_utils.py:88-101):
 *   distance = 1e10; farthest = start[b]           (:91-92; start is the CPU randint draw)
 *   loop i<S: idx[i] = farthest                    (:95)
 *             dist = Σ_c (p_c − far_c)²            (:97)  ((dx²+dy²)+dz², unfused)
 *             distance = where(dist < distance)    (:98-99)
 *             farthest = argmax(distance)          (:100) first index on ties
 * No near-origin skip, no fixed start (those are CUDA-kernel-only behaviours). */
void orc_fps(const float *xyz, int B, int N, int S, const int64_t *start,
             int64_t *idx_out) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int b = 0; b < B; ++b) {
    const float *p = xyz + (size_t)b * N * 3;
    float *mind = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    for (int k = 0; k < N; ++k) mind[k] = 1e10f;
    int64_t far = start[b];
    for (int i = 0; i < S; ++i) {
      idx_out[(size_t)b * S + i] = far;
      float fx = p[3 * far], fy = p[3 * far + 1], fz = p[3 * far + 2];
      float best = -INFINITY;
      int64_t besti = 0;
      for (int k = 0; k < N; ++k) {
        float dx = p[3 * k] - fx, dy = p[3 * k + 1] - fy, dz = p[3 * k + 2] - fz;
        float xx = dx * dx, yy = dy * dy, zz = dz * dz;
        float d = xx + yy;
        d = d + zz;
        if (d < mind[k]) mind[k] = d;
        if (mind[k] > best) {
          best = mind[k];
          besti = k;
        }
      }
      far = besti;
    }
    free(mind);
  }
}

/* ball_query, fast=False branch (geometry_utils.py:151-161):
 *   group[s, n] = n; group[sqrdists > radius**2] = N; sort; take first K;
 *   pad the N's with the first entry.  The comparison is done in fp32 against
 *   f32(radius**2) (`thr`), so a point is kept iff !(D > thr).  Equivalent to
 *   "first K kept points in index order, padded with the first".  A query with
 *   no kept point yields K copies of N (what the sort-based code produces).
 *   xyz [B,N,3], new_xyz [B,S,3] -> idx [B,S,K]. */
void orc_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S,
                    float thr, int K, int64_t *idx) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b) {
    for (int s = 0; s < S; ++s) {
      const float *p = xyz + (size_t)b * N * 3;
      const float *q = new_xyz + ((size_t)b * S + s) * 3;
      int64_t *o = idx + ((size_t)b * S + s) * K;
      float qx = q[0], qy = q[1], qz = q[2];
      float qn = sqnorm3(qx, qy, qz);
      int cnt = 0;
      for (int n = 0; n < N && cnt < K; ++n) {
        float px = p[3 * n], py = p[3 * n + 1], pz = p[3 * n + 2];
        float d = pair_sqdist(qx, qy, qz, qn, px, py, pz, sqnorm3(px, py, pz));
        if (!(d > thr)) o[cnt++] = n;
      }
      int64_t first = cnt > 0 ? o[0] : (int64_t)N;
      for (int k = cnt; k < K; ++k) o[k] = first;
    }
  }
}

/* three_nn, fast=False branch (geometry_utils.py:212-215): ascending sort of
 * pairwise_squared_distance(query, points) rows, first three -> *squared*
 * distances (they can be slightly negative at coincident points) and indices.
 * Ties are broken towards the lower index (what a stable sort gives).
 * unknown(query) [B,N,3], known(points) [B,M,3] -> dist [B,N,3], idx [B,N,3].
 * If M < 3 the missing slots hold +inf / index M. */
void orc_three_nn(const float *unknown, const float *known, int B, int N, int M,
                  float *dist, int64_t *idx) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b) {
    for (int i = 0; i < N; ++i) {
      const float *u = unknown + ((size_t)b * N + i) * 3;
      const float *kn = known + (size_t)b * M * 3;
      float ux = u[0], uy = u[1], uz = u[2];
      float un = sqnorm3(ux, uy, uz);
      float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
      int64_t i0 = M, i1 = M, i2 = M;
      for (int j = 0; j < M; ++j) {
        float x = kn[3 * j], y = kn[3 * j + 1], z = kn[3 * j + 2];
        float d = pair_sqdist(ux, uy, uz, un, x, y, z, sqnorm3(x, y, z));
        if (d < d0) {
          d2 = d1; i2 = i1; d1 = d0; i1 = i0; d0 = d; i0 = j;
        } else if (d < d1) {
          d2 = d1; i2 = i1; d1 = d; i1 = j;
        } else if (d < d2) {
          d2 = d; i2 = j;
        }
      }
      float *od = dist + ((size_t)b * N + i) * 3;
      int64_t *oi = idx + ((size_t)b * N + i) * 3;
      od[0] = d0; od[1] = d1; od[2] = d2;
      oi[0] = i0; oi[1] = i1; oi[2] = i2;
    }
  }
}

/* Interpolation weights of PointsetFeaturePropagation.forward
 * (modules/pointset_feature_propagation.py:40-42):
 *   recip = 1/(d + 1e-8); w = recip / Σ recip.   dist [R,3] -> w [R,3]. */
void orc_three_weights(const float *dist, int64_t R, float *w) {
  for (int64_t r = 0; r < R; ++r) {
    float a = 1.0f / (dist[3 * r] + 1e-8f);
    float b = 1.0f / (dist[3 * r + 1] + 1e-8f);
    float c = 1.0f / (dist[3 * r + 2] + 1e-8f);
    float s = a + b;
    s = s + c;
    w[3 * r] = a / s; w[3 * r + 1] = b / s; w[3 * r + 2] = c / s;
  }
}

/* three_weighted_sum, fast=False branch (geometry_utils.py:281-283):
 *   out[b,c,s] = Σ_t feats[b,c,idx[b,s,t]] * w[b,s,t]   (sum over t in order)
 * feats [B,C,M], idx [B,N,3], w [B,N,3] -> out [B,C,N]. */
void orc_three_weighted_sum(const float *feats, const int64_t *idx, const float *w,
                            int B, int C, int M, int N, float *out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b) {
    for (int c = 0; c < C; ++c) {
      const float *f = feats + ((size_t)b * C + c) * M;
      float *o = out + ((size_t)b * C + c) * N;
      for (int s = 0; s < N; ++s) {
        const int64_t *ii = idx + ((size_t)b * N + s) * 3;
        const float *ww = w + ((size_t)b * N + s) * 3;
        float t0 = f[ii[0]] * ww[0], t1 = f[ii[1]] * ww[1], t2 = f[ii[2]] * ww[2];
        float acc = t0 + t1;
        o[s] = acc + t2;
      }
    }
  }
}

/* Adjoint of the above w.r.t. feats (what autograd's index_put(accumulate) of
 * select_point_subset, geometry_utils.py:26-44, produces): scatter-add in
 * ascending (s,t) order, fp32.  grad_out [B,C,N] -> grad_feats [B,C,M]. */
void orc_three_weighted_sum_grad(const float *grad_out, const int64_t *idx,
                                 const float *w, int B, int C, int N, int M,
                                 float *grad_feats) {
  memset(grad_feats, 0, sizeof(float) * (size_t)B * C * M);
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b) {
    for (int c = 0; c < C; ++c) {
      const float *g = grad_out + ((size_t)b * C + c) * N;
      float *o = grad_feats + ((size_t)b * C + c) * M;
      for (int s = 0; s < N; ++s) {
        const int64_t *ii = idx + ((size_t)b * N + s) * 3;
        const float *ww = w + ((size_t)b * N + s) * 3;
        for (int t = 0; t < 3; ++t) o[ii[t]] += g[s] * ww[t];
      }
    }
  }
}

/* select_point_subset (geometry_utils.py:26-44) for idx [B,S,K]:
 *   out[b,c,s,k] = points[b,c,idx[b,s,k]].   K == 1 is the [B,S] case. */
void orc_group_points(const float *points, const int64_t *idx, int B, int C, int N,
                      int S, int K, float *out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b) {
    for (int c = 0; c < C; ++c) {
      const float *p = points + ((size_t)b * C + c) * N;
      float *o = out + ((size_t)b * C + c) * S * K;
      const int64_t *ii = idx + (size_t)b * S * K;
      for (int64_t j = 0; j < (int64_t)S * K; ++j) o[j] = p[ii[j]];
    }
  }
}

/* Adjoint of select_point_subset: grad_out [B,C,S,K] -> grad_points [B,C,N]. */
void orc_group_points_grad(const float *grad_out, const int64_t *idx, int B, int C,
                           int N, int S, int K, float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)B * C * N);
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b) {
    for (int c = 0; c < C; ++c) {
      const float *g = grad_out + ((size_t)b * C + c) * S * K;
      float *o = grad_points + ((size_t)b * C + c) * N;
      const int64_t *ii = idx + (size_t)b * S * K;
      for (int64_t j = 0; j < (int64_t)S * K; ++j) o[ii[j]] += g[j];
    }
  }
}

int orc_num_threads(void) {
#if defined(_OPENMP)
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * CUDA-ROUTE restatements (the reference's `fast=True` compiled ops).  PARITY UNPINNED: the
 * reference's CUDA extension cannot be built or run in this image, so nothing below has been
 * checked against it.  They restate the published kernels scalar-wise and serve as the checker
 * of the product's opt-in CUDA-route mode (tests/test_gpu_cuda_route.py).  nvcc contracts
 * a*a + b*b + c*c to an fma chain by default (-fmad=true); that rounding is assumed here. */
static inline float direct_sqdist(float qx, float qy, float qz, float x, float y, float z) {
  float dx = qx - x, dy = qy - y, dz = qz - z;
  return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

/* ball_query_gpu.cu:9-44: d2 < radius*radius (fp32), first K in index order, the first hit
 * fills every slot; an empty ball leaves the zero-initialised output (ball_query.cpp). */
void orc_ball_query_direct(const float *xyz, const float *new_xyz, int B, int N, int S,
                           float radius, int K, int64_t *idx) {
  const float r2 = radius * radius;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b) {
    for (int s = 0; s < S; ++s) {
      const float *p = xyz + (size_t)b * N * 3;
      const float *q = new_xyz + ((size_t)b * S + s) * 3;
      int64_t *o = idx + ((size_t)b * S + s) * K;
      for (int k = 0; k < K; ++k) o[k] = 0;
      int cnt = 0;
      for (int n = 0; n < N && cnt < K; ++n) {
        float d2 = direct_sqdist(q[0], q[1], q[2], p[3 * n], p[3 * n + 1], p[3 * n + 2]);
        if (d2 < r2) {
          if (cnt == 0)
            for (int l = 0; l < K; ++l) o[l] = n;
          o[cnt++] = n;
        }
      }
    }
  }
}

/* interpolate_gpu.cu:9-59 + modules/geometry_utils.py:184: direct distance, strict '<'
 * three-deep insertion starting from (1e40, index 0); sqrt_out != 0 -> sqrt of the result. */
void orc_three_nn_direct(const float *unknown, const float *known, int B, int N, int M,
                         int sqrt_out, float *dist, int64_t *idx) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b) {
    for (int j = 0; j < N; ++j) {
      const float *u = unknown + ((size_t)b * N + j) * 3;
      const float *kn = known + (size_t)b * M * 3;
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int64_t i1 = 0, i2 = 0, i3 = 0;
      for (int k = 0; k < M; ++k) {
        float d = direct_sqdist(u[0], u[1], u[2], kn[3 * k], kn[3 * k + 1], kn[3 * k + 2]);
        if (d < best1) {
          best3 = best2; i3 = i2; best2 = best1; i2 = i1; best1 = d; i1 = k;
        } else if (d < best2) {
          best3 = best2; i3 = i2; best2 = d; i2 = k;
        } else if (d < best3) {
          best3 = d; i3 = k;
        }
      }
      float *od = dist + ((size_t)b * N + j) * 3;
      int64_t *oi = idx + ((size_t)b * N + j) * 3;
      float f1 = (float)best1, f2 = (float)best2, f3 = (float)best3;
      if (sqrt_out) { f1 = sqrtf(f1); f2 = sqrtf(f2); f3 = sqrtf(f3); }
      od[0] = f1; od[1] = f2; od[2] = f3;
      oi[0] = i1; oi[1] = i2; oi[2] = i3;
    }
  }
}

/* sampling_gpu.cu:63-159: start at index 0; points with |p|^2 <= 1e-3 never take part; the
 * running minimum and the arg-max as in the CPU route (ties: lowest index here — the CUDA
 * kernel's tie order depends on its block size and is not reproduced). */
void orc_fps_cuda(const float *xyz, int B, int N, int S, int64_t *idx_out) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int b = 0; b < B; ++b) {
    const float *p = xyz + (size_t)b * N * 3;
    float *mind = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    for (int k = 0; k < N; ++k) mind[k] = sqnorm3(p[3 * k], p[3 * k + 1], p[3 * k + 2]) <= 1e-3f ? -1.0f : 1e10f;
    int64_t far = 0;
    for (int i = 0; i < S; ++i) {
      idx_out[(size_t)b * S + i] = far;
      float fx = p[3 * far], fy = p[3 * far + 1], fz = p[3 * far + 2];
      float best = -1.0f;
      int64_t besti = 0;
      for (int k = 0; k < N; ++k) {
        if (mind[k] < 0.0f) continue;
        float dx = p[3 * k] - fx, dy = p[3 * k + 1] - fy, dz = p[3 * k + 2] - fz;
        float xx = dx * dx, yy = dy * dy, zz = dz * dz;
        float d = xx + yy;
        d = d + zz;
        if (d < mind[k]) mind[k] = d;
        if (mind[k] > best) { best = mind[k]; besti = k; }
      }
      far = besti;
    }
    free(mind);
  }
}
