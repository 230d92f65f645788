/*
 * cpfn_hip.h — C ABI of libcpfn_hip.so, the MI355X (gfx950) replacement for the
 * reference's `cuda_ops` extension (PointNet2/pointnet2_ops/cuda_ops) plus the
 * fused kernels behind SPFN.*_fitter.compute_parameters.
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes; no torch / ATen types; `stream` is a hipStream_t
 *     passed as void* (NULL = the null stream);
 *   - nothing is allocated, nothing synchronises, the call only enqueues kernels —
 *     so a caller may capture it into a hipGraph;
 *   - return value: 0 on success, otherwise a hipError_t (launch errors) or
 *     CPFN_EINVAL for bad arguments.  Never exits the process (the reference's
 *     CUDA_CHECK_ERRORS() does: cuda_ops/include/cuda_utils.h:30-39);
 *   - indices are int32 (as in the reference's native ops, include/utils.h:17-21);
 *   - point coordinates are [B, N, 3] fp32 row-major (cuda_ops/src/sampling_gpu.cu:61).
 *
 * Each declaration cites the reference interface it replaces.  The Python shim that
 * re-creates the nine `cuda_ops.*` names on top of these is cpfn_amd/cuda_ops.py;
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 */
#ifndef CPFN_HIP_H
#define CPFN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define CPFN_API __attribute__((visibility("default")))
#else
#define CPFN_API
#endif

#define CPFN_EINVAL (-22)
#define CPFN_ABI_VERSION 3

/* Library / build identification (no GPU needed). */
CPFN_API int cpfn_abi_version(void);
CPFN_API const char *cpfn_build_info(void);

/* ------------------------------------------------------------------ sampling */

/* Furthest-point sampling.  Replaces farthest_point_sampling()
 * (cuda_ops/src/sampling.cpp:64-86, kernel sampling_gpu.cu:63-159) with the
 * semantics of the reference's CPU route (modules/geometry_utils.py:88-101):
 * sample 0 is start[b] (NULL -> 0), distances are ((dx*dx+dy*dy)+dz*dz) in
 * unfused fp32, ties go to the lowest index.  idx_out[B,S].
 * flags bit0 = 1 additionally ignores points with |p|^2 <= 1e-3 like the CUDA
 * kernel does (sampling_gpu.cu:90-91).
 * scratch: B*N floats (8-byte aligned), only touched when N > CPFN_FPS_MAX_RESIDENT (may be NULL
 * otherwise); the reference allocates the same [B,N] `tmp` itself (sampling.cpp:73).
 * Three kernels: N <= 8192: one workgroup per cloud, distances in registers, cloud mirrored in LDS;
 * 8192 < N <= 524288 (while all B * ceil(N / 2048..8192) workgroups can be resident together — occupancy of the kernel x
 * compute units of the device, queried once — and S <= 4094): ceil(N / (256*PPT)) workgroups per
 * cloud, PPT = 8|16|32 points per lane in registers, one 8-byte key per workgroup and sample exchanged through
 * the first words of `scratch` (zeroed by a memset node in front of the launch; bounded spin: a sibling workgroup
 * that never arrives ends the cloud's sampling with index 0 in the remaining outputs and a count in cpfn_fps_faults(),
 * not a hang); otherwise one workgroup per cloud streaming the distances
 * through `scratch`.  All three select identical indices. */
#define CPFN_FPS_MAX_RESIDENT 8192
#define CPFN_FPS_SKIP_NEAR_ORIGIN 1
CPFN_API int cpfn_fps(const float *xyz, int B, int N, int S, const int *start, int flags,
             int *idx_out, float *scratch, void *stream);
/* cpfn_fps for N <= cpfn_fps_max_resident() that also writes the sampled centres, centres[B,S,3] = xyz[b, idx_out[b,s], :] — what
 * select_point_subset (modules/geometry_utils.py:26-44) gathers right after the sampling (pointset_abstraction.py:50): one launch less
 * per set-abstraction level. */
CPFN_API int cpfn_fps_centres(const float *xyz, int B, int N, int S, const int *start, int flags, int *idx_out,
                              float *centres, void *stream);
CPFN_API int cpfn_fps_max_resident(void);
/* Sampling faults since the library was loaded.  0 in a healthy process; < 0 on error.  Two kinds are counted:
 * (i) clouds for which the several-workgroups FPS (8192 < N) gave up waiting for a sibling workgroup (their remaining samples
 * are index 0); (ii) TRIPWIRE, every kernel: a sample whose own min-distance was not zeroed by its update — the arg-max
 * returned the point just sampled with a positive distance — i.e. a lost update on the lane that owns the sample (round 4's
 * packed-fp32 fault beside a weight-gradient workgroup).  Detection only: the indices of such a launch are NOT those of
 * modules/geometry_utils.py:88-101 any more (the point repeats); the count says that the hardware / a neighbour misbehaved.
 * A point with an inf / NaN coordinate is NOT counted (its min-distance is never lowered from the initial 1e10 and the point
 * repeats, exactly as in the reference's loop): bad input, not a fault.
 * Reads a pinned host word (no synchronisation).  The word is allocated by the first sampling launch — also inside a stream
 * capture (the capture mode is relaxed around the allocation); only on a stack where it cannot be allocated at all does this
 * call fall back to reading the device counter, which SYNCHRONISES the device and must not be used while a capture is open. */
CPFN_API int cpfn_fps_faults(void);
/* Test hook for the tripwire: in every sampling launch issued from now on the wave that owns sample `sample` (0-based) skips its
 * distance update once (-1: off).  Returns the previous setting. */
CPFN_API int cpfn_fps_debug_drop(int sample);
/* Diagnostic twin of cpfn_fps for N <= 8192 (one workgroup per cloud; variant 0: 256 threads x 8 points per lane, N <= 2048;
 * 1: 512 x 16; 2: 256 x 32): same indices, plus prof[B][6] = shader-clock cycles, summed over the S samples, that wave 0
 * spent in {sample broadcast read, distance update + lane maximum, wave maximum, index ballots, LDS slot + barrier,
 * slot read + maximum over waves}.  The latency model of profiles/r03_fps_latency.md is built from it. */
CPFN_API int cpfn_fps_profile(const float *xyz, int B, int N, int S, const int *start, int variant, int *idx_out,
                              unsigned long long *prof, void *stream);

/* Ball query.  Replaces ball_query() (cuda_ops/src/ball_query.cpp, kernel
 * ball_query_gpu.cu:9-44) with the CPU route's arithmetic
 * (modules/geometry_utils.py:151-161): D = -2 q.p + |q|^2 + |p|^2 with the K=3
 * inner product as an fma chain, a point is kept iff !(D > thr) where
 * thr = (float)(radius**2 computed in double); first K kept indices in index
 * order, padded with the first; a query with no kept point gets K copies of N.
 * xyz[B,N,3], new_xyz[B,S,3] -> idx_out[B,S,K]. */
CPFN_API int cpfn_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S,
                    float thr, int K, int *idx_out, void *stream);

/* 3 nearest neighbours.  Replaces three_nn() (cuda_ops/src/interpolate.cpp,
 * kernel interpolate_gpu.cu:9-59) with the CPU route's arithmetic
 * (modules/geometry_utils.py:212-215): the same expanded D as the ball query,
 * ascending, ties to the lower index; returns SQUARED distances (can be < 0).
 * unknown[B,N,3] (queries), known[B,M,3] -> dist2[B,N,3], idx[B,N,3].
 * M < 3 leaves +inf / M in the unused slots. */
CPFN_API int cpfn_three_nn(const float *unknown, const float *known, int B, int N, int M,
                  float *dist2, int *idx, void *stream);
/* cpfn_three_nn (direct == 0) / cpfn_three_nn_direct (direct != 0) and, when w is not NULL, cpfn_three_weights on the distances
 * they return, in the same launch (pointset_feature_propagation.py:38-42). */
CPFN_API int cpfn_three_nn_weights(const float *unknown, const float *known, int B, int N, int M, int direct, int sqrt_out,
                                   float *dist2, int *idx, float *w, void *stream);

/* CUDA-route twins (the reference's `fast=True` results; opt-in, see DESIGN.md "CUDA route").
 * cpfn_ball_query_direct: ball_query_gpu.cu:9-44 — d2 = (q-p)^2 summed over x,y,z, kept iff
 * d2 < radius*radius (fp32), first K in index order, padded with the first; an empty ball
 * leaves index 0 (the reference's zero-initialised output, ball_query.cpp).
 * cpfn_three_nn_direct: interpolate_gpu.cu:9-59 — the same direct distance, strict '<'
 * insertion from index 0; sqrt_out != 0 returns the square roots, which is what the
 * reference's Python wrapper hands on (modules/geometry_utils.py:184).
 * nvcc's default fma contraction of the three-term sum is assumed; not pinned bit for bit
 * (no CUDA build of the reference can run next to this library). */
/* cpfn_ball_query on a cloud packed as [B, N, 4] = (x, y, z, |p|^2) by cpfn_pack_xyzn (16-byte aligned): the wave-per-query
 * scan then costs one 16-byte load and six operations per point instead of three loads, the norm and the distance.  Same
 * results as cpfn_ball_query, bit for bit (the norm is the same ((x*x + y*y) + z*z), computed once per point). */
CPFN_API int cpfn_pack_xyzn(const float *xyz, int B, int N, float *out, void *stream);
CPFN_API int cpfn_ball_query_packed(const float *xyzn, const float *new_xyz, int B, int N, int S, float thr, int K,
                                    int *idx_out, void *stream);
/* ... and, when rel_out is not NULL, the centred coordinates of the neighbours it returns, rel_out[B,S,K,3] = xyz[idx] - new_xyz
 * (cpfn_group_xyz_centered; pointset_abstraction.py:62-63) from the same launch. */
CPFN_API int cpfn_ball_query_packed_rel(const float *xyzn, const float *new_xyz, int B, int N, int S, float thr, int K,
                                        int *idx_out, float *rel_out, void *stream);
/* on != 0: the following cpfn_fps / cpfn_ball_query* / cpfn_three_nn* calls run BESIDE other work (a side stream next to a
 * training step) and use the kernel shapes that disturb their neighbours least; 0 (default): the fastest kernels.  Same
 * results, bit for bit.  Returns the previous setting.  (Per calling thread since round 5; no reference counterpart.) */
CPFN_API int cpfn_set_background_geometry(int on);
CPFN_API int cpfn_ball_query_direct(const float *xyz, const float *new_xyz, int B, int N, int S,
                                    float radius, int K, int *idx_out, void *stream);
CPFN_API int cpfn_three_nn_direct(const float *unknown, const float *known, int B, int N, int M,
                                  int sqrt_out, float *dist, int *idx, void *stream);

/* pairwise_squared_distance (modules/geometry_utils.py:4-23), materialised:
 * src[B,N,3], dst[B,M,3] -> out[B,N,M].  API parity only; N, B <= 65535. */
CPFN_API int cpfn_pairwise_sqdist(const float *src, const float *dst, int B, int N, int M,
                                  float *out, void *stream);

/* Inverse-distance weights of PointsetFeaturePropagation.forward
 * (modules/pointset_feature_propagation.py:40-42): w = (1/(d+1e-8)) / sum.
 * dist[R,3] -> w[R,3]. */
CPFN_API int cpfn_three_weights(const float *dist, int64_t R, float *w, void *stream);

/* ------------------------------------------------- channel-major fp32 ops
 * Drop-in layouts of the reference's bound functions (bindings.cpp:6-19). */

/* three_weighted_sum (interpolate_gpu.cu:72-101): feats[B,C,M], idx/w[B,N,3] -> out[B,C,N] */
CPFN_API int cpfn_three_interp_fwd(const float *feats, const int *idx, const float *w, int B,
                          int C, int M, int N, float *out, void *stream);
/* three_weighted_sum_grad (interpolate_gpu.cu:116-143): grad_out[B,C,N] -> grad_feats[B,C,M]
 * (grad_feats must be zero-filled by the caller, as the reference's wrapper does). */
CPFN_API int cpfn_three_interp_bwd(const float *grad_out, const int *idx, const float *w, int B,
                          int C, int N, int M, float *grad_feats, void *stream);
/* group_points (group_points_gpu.cu:8-28): points[B,C,N], idx[B,S,K] -> out[B,C,S,K].
 * K = 1 is gather_points (sampling_gpu.cu:8-20).  Indices >= N read point N-1. */
CPFN_API int cpfn_group_fwd(const float *points, const int *idx, int B, int C, int N, int S,
                   int K, float *out, void *stream);
/* group_points_grad / gather_points_grad (group_points_gpu.cu:43-64, sampling_gpu.cu:32-45):
 * grad_out[B,C,S,K] -> grad_points[B,C,N] (zero-filled by the caller). */
CPFN_API int cpfn_group_bwd(const float *grad_out, const int *idx, int B, int C, int N, int S,
                   int K, float *grad_points, void *stream);

/* ------------------------------------------------- points-major ops (native layout)
 * The MI355X path keeps features as [B, N, C] rows so that a neighbour gather is a
 * contiguous row copy and an MFMA operand fragment is one 16-byte load. */

/* Row gather: rows[B,N,row_bytes], idx[B,R] -> out[B,R,row_bytes].  row_bytes % 4 == 0. */
CPFN_API int cpfn_gather_rows(const void *rows, const int *idx, int B, int N, int R,
                     int row_bytes, void *out, void *stream);
/* Adjoint for fp32 rows: grad_out[B,R,C] scattered-added into grad_rows[B,N,C]
 * (zero-filled by the caller). */
CPFN_API int cpfn_scatter_add_rows_f32(const float *grad_out, const int *idx, int B, int N, int R,
                              int C, float *grad_rows, void *stream);
/* Grouped, centred coordinates of PointsetAbstraction.forward
 * (modules/pointset_abstraction.py:62-63): out[b,s,k,:] = xyz[b,idx[b,s,k],:] - new_xyz[b,s,:]. */
CPFN_API int cpfn_group_xyz_centered(const float *xyz, const float *new_xyz, const int *idx, int B,
                            int N, int S, int K, float *out, void *stream);
/* Points-major interpolation: feats[B,M,C], idx/w[B,N,3] -> out[B,N,C] (fp32). */
CPFN_API int cpfn_interp_rows_fwd(const float *feats, const int *idx, const float *w, int B, int M,
                         int N, int C, float *out, void *stream);
/* Adjoint: grad_out[B,N,C] -> grad_feats[B,M,C] (zero-filled by the caller). */
CPFN_API int cpfn_interp_rows_bwd(const float *grad_out, const int *idx, const float *w, int B,
                         int M, int N, int C, float *grad_feats, void *stream);

/* bf16 row movers used by the fused MLP path (same semantics as the fp32 entry points above):
 * interpolation feats[B,M,C] -> out[B,N,C];  its adjoint and the gather adjoint as one LDS-privatised
 * scatter-add  out[b, idx[b,r,t], :] += w[b,r,t] * g[b,r,:]  (g bf16 with row stride ldg, T in 1..3,
 * w may be NULL for T = 1, out fp32 [B,M,C] zero-filled by the caller, M <= 1024);
 * and the grouped sa-level input rows  out[p] = [feats[idx[p]] (C) | rel xyz (3) | zeros] of width Cpad
 * (modules/pointset_abstraction.py:62-66). */
CPFN_API int cpfn_interp_rows_bf16(const void *feats, const int *idx, const float *w, int B, int M, int N,
                                   int C, void *out, void *stream);
/* Input rows of a feature-propagation stack in one pass (modules/pointset_feature_propagation.py:33-46):
 * out[B,N,C1+C2] = [ skip[B,N,C1] | three-NN interpolation of feats[B,M,C2] ] — or, with idx = w = NULL and M = 1,
 * [ skip | feats[b,0,:] broadcast ] (the global feature vector of sfp1).  C1, C2 multiples of 8 (C1 may be 0). */
CPFN_API int cpfn_concat_interp_bf16(const void *skip, int C1, const void *feats, const int *idx, const float *w, int B,
                                     int M, int N, int C2, void *out, void *stream);
/* out[B,C] (bf16) = column sums over the N rows of a column block of g (bf16, row stride ldg): adjoint of that broadcast. */
CPFN_API int cpfn_colsum_rows_bf16(const void *g, int ldg, int B, int N, int C, void *out, void *stream);
/* ... which IS the gradient of sa3's max-pooled output, one row per cloud (PointNet2/pn2_network.py:49,56): with yarg [B,C] (bf16, the
 * pre-BN values at the arg-max rows), scale / shift [C] of that stack's last layer, the same launch leaves BatchNorm-backward pass 1
 * of it as one partial row per cloud, part [B][2][C] (sum g_z, sum g_z y — cpfn_bn_relu_bwd's arithmetic on a single row), for
 * cpfn_bn_bwd_finalize with nblk = B.  yarg == NULL: the plain entry. */
CPFN_API int cpfn_colsum_rows_pass1_bf16(const void *g, int ldg, int B, int N, int C, void *out, const void *yarg, const float *scale,
                                         const float *shift, float *part, void *stream);
CPFN_API int cpfn_scatter_rows_bf16(const void *g, int ldg, const int *idx, const float *w, int T, int B,
                                    int R, int M, int C, float *out, void *stream);
/* (rel == NULL with Cpad == C: the gather alone) */
CPFN_API int cpfn_group_concat_bf16(const void *feats, const float *rel, const int *idx, int B, int N, int R,
                                    int C, int Cpad, void *out, void *stream);

/* count device-to-device copies in ONE launch: pointers 16-byte aligned (any byte length), or 4-byte aligned
 * with a length that is a multiple of 4 (slices of a flat fp32 buffer). */
typedef struct { const void *src; void *dst; long long bytes; } cpfn_copy_desc;
CPFN_API int cpfn_multi_copy(const cpfn_copy_desc *descs /* HOST array */, int count, void *stream);
/* The same copies of fp32 buffers (bytes % 4 == 0) with a finite scan riding along: flags[i] = 1 if workgroup i copied
 * a NaN / inf, for i < cpfn_multi_copy_blocks(descs, count) <= flags_capacity.  The trainer packs the gradients into
 * the flat bucket with it and hands the flags to cpfn_adam_flat (nf_partial): no separate scan of the gradients. */
CPFN_API int cpfn_multi_copy_blocks(const cpfn_copy_desc *descs /* HOST array */, int count);
CPFN_API int cpfn_multi_copy_checked(const cpfn_copy_desc *descs /* HOST array */, int count, unsigned *flags,
                                     int flags_capacity, void *stream);
/* count fp32 matrices src[rows, cols] (contiguous) converted to bf16 (dst_f32 = 0) or copied as fp32 (dst_f32 = 1)
 * into dst with row stride dst_ld >= cols (elements; padding columns are left untouched), in ONE launch: the
 * per-step refresh of all bf16 weight panels of the network. */
/* out[R, Cpad] bf16 = [ bf16(xyz[R,3]) | feats[R,C] (bf16) | zeros ]: the input rows of the group-all set
 * abstraction (modules/pointset_abstraction.py:56, positions first), padded to the GEMM's K. */
CPFN_API int cpfn_concat_pos_feats_bf16(const float *xyz, const void *feats, long long R, int C, int Cpad, void *out,
                                        void *stream);
/* n_gt[b] = max(labels[b, :]) + 1 (SPFN/losses_implementation.py:603-606). */
CPFN_API int cpfn_count_labels(const int64_t *labels, int B, int N, int64_t *n_gt, void *stream);
typedef struct { const float *src; void *dst; int rows, cols, dst_ld, dst_f32;
                 int src_ld; /* 0 = cols (contiguous source); > cols: a column slice of a wider matrix */ } cpfn_cast_desc;
CPFN_API int cpfn_multi_cast(const cpfn_cast_desc *descs /* HOST array */, int count, void *stream);

/* Inverse index of a gather (geometry stage): for idx[B,E] with values in [0,M) (M <= 2048) build, per
 * cloud, offsets[M+1] and the ASCENDING list entries[E] of source positions e that reference each target.
 * The adjoint of the gather is then cpfn_csr_gather_sum_bf16: out[b,m,:] = sum over list(m) of
 * w[b,e] * g[b, e/T, :] (w may be NULL; g bf16 with row stride ldg; out bf16 [B,M,C]) — no atomics,
 * fixed summation order. */
CPFN_API int cpfn_csr_build(const int *idx, int B, int E, int M, int *offsets, int *entries, void *stream);
/* The same result by a stable radix sort of the entries by target (round 5: ascending by construction, no per-list sort; 3 bits of
 * the target per pass, ballot ranks inside 64-entry chunks, two global buffers): workspace [B, E] int32 (scratch), E <= 32768;
 * threads per cloud 0 (= 256), 512 or 1024.  Without a workspace, or beyond that size, cpfn_csr_build.
 * threads < 0 (-1: 4 waves per cloud, -8: 8 waves, -16: 16 waves): the "ordered" build — count, scan, then a scatter through LDS atomics in which every wave walks its own contiguous
 * range of the entries in order (ascending lists as long as one ds_add_rtn serves its lanes in lane order, which gfx950 does);
 * the kernel verifies the result (one LDS compare per adjacent pair) and sorts the lists itself where the check fails, so the
 * result is the same on any hardware.  workspace is then ONE int32 the caller zeroes once: the number of clouds that needed
 * the sort (0 on every MI355X seen).  Needs 4 ((waves + 1) M + 1) + 4 E + 64 bytes of LDS <= 150 KB, otherwise cpfn_csr_build runs. */
CPFN_API int cpfn_csr_build_ws(const int *idx, int B, int E, int M, int *offsets, int *entries, int *workspace, int threads,
                               void *stream);
CPFN_API int cpfn_csr_gather_sum_bf16(const void *g, int ldg, const int *offsets, const int *entries,
                                      const float *w, int T, int B, int R, int M, int C, void *out,
                                      void *stream);
/* The same with the OTHER gradient of a two-consumer tensor added in: out = bf16(bf16(sum) + addend[b,m,:]) (addend bf16 with row
 * stride ld_add, 16-byte aligned; NULL: the plain adjoint) — the roundings of the framework's bf16 add that autograd would launch
 * between the two backward nodes (PointNet2/pn2_network.py:45-46,55: l1_feats feeds sa2 AND sfp2). */
CPFN_API int cpfn_csr_gather_sum_add_bf16(const void *g, int ldg, const int *offsets, const int *entries,
                                          const float *w, int T, int B, int R, int M, int C, const void *addend,
                                          int ld_add, void *out, void *stream);

/* ------------------------------------------------------------------ SPFN fitters
 * One pass over P[B,N,3], X[B,N,3] (unit normals), W[B,N,K] (soft memberships) yields every
 * weighted sum the four primitive fitters need.  Replaces the tiled [B*K,N,3] /
 * [B*K,N,3,3] temporaries of SPFN/{plane,sphere,cylinder,cone}_fitter.compute_parameters,
 * SPFN/differentiable_tls.py:200-209 and SPFN/geometry_utils.py:74-84,121-142,209-223.
 *
 * M[B,K,52] (fp64) slot map — "A" slots are weighted by w, "B" slots by max(w,1e-10)
 * (the reference's sqrt(clamp(W,1e-10)) row scaling, geometry_utils.py:127):
 *   A  0:1  1-3:p  4-9:p(x)p (xx xy xz yy yz zz)  10-12:x  13-18:x(x)x  19:pad
 *   B  20:1 21-23:p 24-29:p(x)p 30-39:p(x)p(x)p (xxx xxy xxz xyy xyz xzz yyy yyz yzz zzz)
 *      40-45:x(x)x  46-48:x*(p.x)  49-51:pad
 * workspace: cpfn_fit_num_chunks(B,N) * B * K * 52 doubles. */
#define CPFN_FIT_SLOTS 52
#define CPFN_FIT_MAX_K 64
CPFN_API int cpfn_fit_num_chunks(int B, int N);
CPFN_API int cpfn_fit_moments_fwd(const float *P, const float *X, const float *W, int B, int N,
                                  int K, double *workspace, double *M, void *stream);
/* The same launch with the loss section's assignment riding on it as one extra workgroup per cloud (the fits do not
 * depend on the assignment, nor it on them; on its own it is a ~40 us one-wave-per-cloud latency chain):
 * S / n_gt / match as cpfn_hungarian_match (SPFN/losses_implementation.py:10-30), K <= 32.  Same results as the two
 * separate calls, bit for bit. */
CPFN_API int cpfn_fit_moments_fwd_match(const float *P, const float *X, const float *W, int B, int N, int K,
                                        double *workspace, double *M, const float *S, const int64_t *n_gt,
                                        int64_t *match, void *stream);
/* Adjoint: G[B,K,52] (fp32) = dL/dM  ->  dW[B,N,K], dX[B,N,3] (both overwritten). K <= 64.
 * dW_add (may be NULL): a [B,N,K] term added into dW (the cone pass's dW), saving a separate pass. */
CPFN_API int cpfn_fit_moments_bwd(const float *P, const float *X, const float *W, const float *G,
                                  int B, int N, int K, const float *dW_add, float *dW, float *dX,
                                  void *stream);
/* Cone second pass (SPFN/cone_fitter.py:25-34) for fitted apex/axis [B,K,3] (fp32):
 *   out[b,k,0] = sum_n W * (axis . normalize(p - apex)),  out[b,k,1] = sum_n W * acos_safe(|.|)
 * workspace: chunks * B * K * 2 doubles. */
CPFN_API int cpfn_cone_pass_fwd(const float *P, const float *W, const float *apex, const float *axis,
                                int B, int N, int K, double *workspace, double *out, void *stream);
/* Adjoint w.r.t. out[...,1] (g_acos[B,K] fp32): dW[B,N,K] (overwritten) and the fp64 adjoint of
 * (apex, axis): 6 values per instance written at d_apex_axis + (b*K+k)*ld (ld >= 6; ld = 6 for a dense
 * [B,K,6]; ld = 21 with the pointer at column 15 lands in the algebra adjoint), added to what is
 * there when `accumulate`.  workspace: chunks * B * K * 6 doubles. */
CPFN_API int cpfn_cone_pass_bwd(const float *P, const float *W, const float *apex, const float *axis,
                                const float *g_acos, int B, int N, int K, float *dW,
                                double *workspace, double *d_apex_axis, int ld, int accumulate,
                                void *stream);

/* Per-instance algebra of all four fitters on the moments (the [B,K]-sized tail of
 * SPFN/{plane,sphere,cylinder,cone}_fitter.compute_parameters; SPFN/geometry_utils.py:8-27,
 * 74-84, 121-142, 209-223; SPFN/differentiable_tls.py:123-143):
 *   M[G,52] -> out[G,21] = plane n(3) c(1) | sphere centre(3) r2(1) | cylinder axis(3) centre(3)
 *   r2(1) | cone apex(3) | cone axis before the sign fix(3).   G = B*K instances, fp64.
 * The backward entry returns gM[G,52] = J^T gout by forward-mode AD through the same code. */
#define CPFN_FIT_OUTPUTS 21
/* apex_axis32 (may be NULL): fp32 copy of the cone columns as apex[G,3] followed by axis[G,3] — the
 * inputs of cpfn_cone_pass_*.  Backward: gA0[G] (may be NULL) is added to slot 0 of the result (a
 * direct dependence on sum(W)); the result goes to gM (fp64) and/or gM32 (fp32), either may be NULL. */
CPFN_API int cpfn_fit_algebra_fwd(const double *M, int64_t G, double *out, float *apex_axis32,
                                  void *stream);
CPFN_API int cpfn_fit_algebra_bwd(const double *M, const double *gout, const double *gA0, int64_t G,
                                  double *gM, float *gM32, void *stream);
/* The 22 fp32 parameters per instance the residue / axis losses read, from the algebra's out[G,21],
 * the cone pass's sums[G,2] and M (slot 0): columns 0..17 as they are, cone axis flipped to
 * sign(sums0) (sign(0) = +1, SPFN/cone_fitter.py:28-31), half angle = sums1 / (M0 + 1e-10) clamped to
 * [1e-3, pi/2 - 1e-3] (cone_fitter.py:33-35).  Adjoint: g_alg[G,21] (every column written),
 * g_acos[G] fp32 (for cpfn_cone_pass_bwd) and gA0[G] (for cpfn_fit_algebra_bwd). */
CPFN_API int cpfn_fit_pack_fwd(const double *alg, const double *sums, const double *M, int64_t G,
                               float *params, void *stream);
/* cpfn_fit_moments_fwd (or _fwd_match when S is not NULL) + cpfn_fit_algebra_fwd with the chunk reduction of the first
 * folded into the second: two launches instead of three, the same M[B*K,52], alg[B*K,21] and apex_axis32. */
CPFN_API int cpfn_fit_moments_algebra_fwd(const float *P, const float *X, const float *W, int B, int N, int K,
                                          double *workspace, double *M, double *alg, float *apex_axis32,
                                          const float *S, const int64_t *n_gt, int64_t *match, void *stream);
/* Backward of the packed parameters as three launches instead of five (cone pass adjoint, algebra adjoint, then
 * cpfn_fit_moments_bwd): _cone derives g_acos from gparams[G,22] itself (cpfn_fit_pack_bwd's rule) and leaves the per-chunk
 * partials of d(apex, axis) in workspace (cpfn_fit_num_chunks * B * K * 6 doubles); _algebra sums them in chunk order
 * into columns 15..20 of the algebra adjoint it forms from gparams (no g_alg / gA0 tensors) and returns gM32[G,52].
 * Same bits as cpfn_fit_pack_bwd + cpfn_cone_pass_bwd(ld = 21, accumulate) + cpfn_fit_algebra_bwd. */
CPFN_API int cpfn_fit_params_bwd_cone(const float *P, const float *W, const float *apex, const float *axis,
                                      const float *gparams, const double *sums, const double *M, int B, int N,
                                      int K, float *dW, double *workspace, void *stream);
CPFN_API int cpfn_fit_params_bwd_algebra(const double *M, const float *gparams, const double *sums,
                                         const double *cone_workspace, int chunks, int B, int K, float *gM32,
                                         void *stream);
/* The same with the cone pass's chunk reduction folded in: cpfn_cone_pass_fwd(..., out = NULL) leaves its per-chunk
 * partials in its workspace; this launch sums them (same order, same bits), writes sums[B,K,2] for the backward pass and
 * packs the parameters. */
CPFN_API int cpfn_fit_pack_fwd_partials(const double *alg, const double *cone_workspace, int B, int N, int K,
                                        const double *M, double *sums, float *params, void *stream);
CPFN_API int cpfn_fit_pack_bwd(const float *gparams, const double *sums, const double *M, int64_t G,
                               double *g_alg, float *g_acos, double *gA0, void *stream);

/* Batched symmetric 3x3 eigen-decomposition (fp64 Jacobi) replacing the torch.svd call of
 * Custom_svd_v_colum (SPFN/differentiable_tls.py:126) on the PSD moment matrices.
 * S6[G,6] = (xx xy xz yy yz zz) -> lam[G,3] ascending, V[G,3,3] with eigenvectors in columns. */
CPFN_API int cpfn_eigh3(const double *S6, int64_t G, double *lam, double *V, void *stream);

/* ------------------------------------------------------------------ per-point MLP stacks
 * Replaces the Conv2d/Conv1d(1x1) + BatchNorm + ReLU (+ max over neighbours) chains of
 * modules/pointset_abstraction.py:70-74, modules/pointset_feature_propagation.py:49-51 and
 * PointNet2/pn2_network.py:60-68.  Activations are points-major bf16 rows [P, C]; weights
 * [N, K] bf16 (row n = output channel); accumulation fp32 (MFMA 16x16x32 bf16). */

/* Y[P,N] = A[P,K] . W[N,K]^T (+bias).  K % 32 == 0, N % 64 == 0, lda % 8 == 0.
 * w_trans = 1: W is stored [K,N] instead (the forward layer's weight, used as is for the data gradient).
 * gidx (optional): row p of A is A[gidx[p]] (fused neighbour gather).
 * y_f32 = 0: Y is bf16 with row stride ldy; 1: fp32.  Only channels < n_store are stored.
 * stats_partial (optional): [cpfn_mlp_gemm_blocks(P,N)][2][N] fp32 per-block sum(y), sum(y^2).
 * a_scale, a_shift (optional, [K] fp32, both or neither): A holds the PREVIOUS layer's pre-BN output and
 * the operand is relu(a_scale*A + a_shift) rounded to bf16, applied on the fly (bit-identical to
 * cpfn_bn_relu_apply followed by a plain call; the activated tensor is never materialised).
 * Three kernels behind the one entry point: P <= 16384 rows -> split-K small-P kernel (32|64 rows x 64 channels per
 * workgroup); K in {64,128} (and {192,256} for P >= 32768) -> whole-K streaming kernel; everything else (bias, fp32
 * or ragged output, gather, other K) -> 128-wide K chunks through a double-buffered LDS panel. */
CPFN_API int cpfn_mlp_gemm_blocks(long long P, int N);
/* bwd_y (optional; data-gradient launches, w_trans = 1): Y is then the gradient g_a of the layer BELOW, bwd_y that
 * layer's pre-BN output [P,N] bf16 (row stride ldy) and a_scale / a_shift ITS BatchNorm scale / shift [N]; the launch
 * also leaves pass 1 of that layer's BatchNorm backward in stats_partial — per-block sum(g_z), sum(g_z*y) with
 * g_z = g_a*[a_scale*y + a_shift > 0], the layout cpfn_bn_relu_bwd writes — so cpfn_bn_relu_bwd is not needed for
 * it.  Only where cpfn_mlp_gemm_can_fuse_bwd_stats(P,K,N) returns 1 (the streaming kernel). */
/* sa2's first layer: [A (K = 128 bf16 channels, row stride lda = K) | xyz [P,3] fp32] . [W [N,128] bf16 | Wx [N,3] fp32]^T ->
 * Y [P,N] bf16 (ldy = N) + the statistics rows of cpfn_mlp_gemm (stats_partial NULL: none).  The coordinate term is one more
 * MFMA k-step built in registers from x = hi + lo (bf16), accurate to ~2^-16: the K = 128 streaming kernel instead of a K = 192
 * operand with three bf16 coordinate columns and 61 columns of padding.  cpfn_mlp_gemm_xyz_ok: K = N = 128, P >= 32768. */
CPFN_API int cpfn_mlp_gemm_xyz_ok(long long P, int K, int N);
CPFN_API int cpfn_mlp_gemm_xyz(const void *A, int lda, const void *W, const float *xyz, const float *Wx, long long P, int K,
                               int N, void *Y, int ldy, float *stats_partial, void *stream);
CPFN_API int cpfn_mlp_gemm_can_fuse_bwd_stats(long long P, int K, int N);
/* Timing probe of the GEMM family (measurement only; bench.py's roofline leg).  buf = slots * (2 + 2*max_wg) u64 of
 * zero-filled device memory, or NULL to switch the probe off (default).  While installed, launch i of cpfn_mlp_gemm
 * — including launches captured into a hipGraph, which host-side events cannot bracket one by one — writes into slot
 * i % slots: [0] = its number of workgroups, [2 + 2w], [3 + 2w] = start / end of workgroup w in ticks of the device's
 * constant-rate wall clock (hipDeviceAttributeWallClockRate, 100 MHz on gfx950); plain stores, no atomics.  The
 * launch's duration is max(end) - min(start).  A captured launch keeps its slot across replays. */
CPFN_API int cpfn_mlp_gemm_set_probe(void *buf, int slots, int max_wg);
/* Rate in kHz of the device wall clock the probe's ticks are counted in (hipDeviceAttributeWallClockRate); <= 0 on error. */
CPFN_API int cpfn_wall_clock_khz(int device);
/* One reading of that clock into *dst, issued as a (capturable) 1-thread kernel on `stream`: a time stamp inside a
 * replayed graph (debugging aid: CPFN_STEP_STAMPS=1). */
CPFN_API int cpfn_stamp(unsigned long long *dst, void *stream);
/* Cross-stream ordering on ONE GPU by device flags (no reference counterpart: the reference has one stream): a one-lane
 * kernel on `stream` that polls *flag until (int)(*flag - value) >= 0 — giving up after timeout_ticks of the 100 MHz wall
 * clock, then storing 1 to *err (may be NULL; may be pinned host memory) and 1.0f to *fault (may be NULL; device memory: a
 * sticky word the caller passes to cpfn_adam_flat as found_inf, so that no step computed after a broken hand-over updates
 * the weights) — and one that stores `value` to *flag.  Data written by kernels BEFORE the
 * setter on its stream is visible to kernels AFTER the waiter on its stream (kernel boundaries), as with an event. */
CPFN_API int cpfn_flag_wait(const unsigned *flag, unsigned value, unsigned long long timeout_ticks, unsigned *err,
                            float *fault, void *stream);
CPFN_API int cpfn_flag_set(unsigned *flag, unsigned value, void *stream);
/* cpfn_flag_set with a payload of count <= 64 ints (HOST pointer: they travel in the launch's arguments) stored to the
 * device array dst before the flag: small per-step inputs of the waiting stream without a host-to-device copy. */
CPFN_API int cpfn_flag_set_payload(unsigned *flag, unsigned value, int *dst, const int *payload, int count, void *stream);
CPFN_API int cpfn_mlp_gemm(const void *A, int lda, const int *gidx, const void *W, int w_trans, long long P,
                           int K, int N, void *Y, int ldy, int y_f32, int n_store, const float *bias,
                           float *stats_partial, const float *a_scale, const float *a_shift, const void *bwd_y,
                           void *stream);
/* ---- BatchNorm seams without a finalize launch (round 6; no reference counterpart: the reference's batch_norm is one
 * framework op, modules/pointset_abstraction.py:70-74, pointset_feature_propagation.py:49-51).  The layer's GEMM (the
 * PRODUCER) adds its per-workgroup sum(y), sum(y^2) as 64-bit fixed-point integers (value * 2^log2_scale, no-return atomics)
 * into `replicas` copies of a [2][C] accumulator — workgroup w into copy w % replicas — and the NEXT layer's GEMM (the
 * CONSUMER) folds the copies into scale / shift in every workgroup's prologue: cpfn_bn_finalize is not launched for that
 * layer.  Integer sums: bit-reproducible.  acc = cpfn_seam_words(replicas, C) int64 words, ZERO before the producer runs
 * (the last word is a poison flag: a NaN / inf / oversized partial sum makes the consumer see NaN statistics).
 * One workgroup of the consumer also writes stats[4][C] = scale | shift | mean | rstd (what cpfn_bn_finalize leaves, for
 * the backward pass) and updates the running statistics; the step counters are advanced by the PRODUCER. */
typedef struct cpfn_seam_out {
  long long *acc; int replicas; int log2_scale;
  long long *counter_a, *counter_b;          /* optional int64 step counters (see cpfn_bn_finalize) */
} cpfn_seam_out;
typedef struct cpfn_seam_in {
  const long long *acc; int replicas; int log2_scale; int C;
  float count, eps, momentum;
  const float *gamma, *beta, *conv_bias;     /* conv_bias optional */
  float *running_mean, *running_var;         /* optional (both or neither) */
  float *stats;                              /* [4][C], written */
} cpfn_seam_in;
CPFN_API int cpfn_seam_words(int replicas, int C);
/* bit 0: cpfn_mlp_gemm_seam can PRODUCE a seam for a forward layer of this shape (bf16 rows, no gather / bias), bit 1: it can
 * CONSUME one (operand transform from the previous layer's sums; K = that layer's channels). */
CPFN_API int cpfn_mlp_gemm_seam_ok(long long P, int K, int N);
/* Forward layer Y[P,N] = f(A)[P,K] . W[N,K]^T, lda = K, ldy = N, bf16 — cpfn_mlp_gemm with the seams spelled out:
 * statistics either as partial rows (stats_partial) or into `out`; operand transform f = relu(scale*a + shift) either from
 * a_scale / a_shift or folded from `in` (NULL / NULL / NULL: plain operand).  Same kernels, same Y bits. */
CPFN_API int cpfn_mlp_gemm_seam(const void *A, const void *W, long long P, int K, int N, void *Y, float *stats_partial,
                                const cpfn_seam_out *out, const cpfn_seam_in *in, const float *a_scale, const float *a_shift,
                                void *stream);
/* cpfn_mlp_gemm_xyz / cpfn_smallk_fwd(_cast) as producers of a seam (stats_partial / partial replaced by `out`). */
CPFN_API int cpfn_mlp_gemm_xyz_seam(const void *A, const void *W, const float *xyz, const float *Wx, long long P, int K, int N,
                                    void *Y, const cpfn_seam_out *out, void *stream);
/* The pooled last layer of a set-abstraction stack (modules/pointset_abstraction.py:70-77: conv, batch_norm, relu, max over the K
 * neighbours) WITHOUT a second pass over its [P, N] output: cpfn_mlp_gemm_pool is cpfn_mlp_gemm_seam's streaming kernel with the
 * operand transform whose epilogue also leaves, per wave of 32 rows and channel, the raw y of the wave's winner of max(sign(gamma)*y)
 * (pmax [P/32][N] bf16) and its row inside the group (pidx [P/32][N] u8; 255: none) — z = scale*y + shift is monotone in y with the
 * sign of gamma, so the maximum over neighbours can be taken before the batch statistics exist.  cpfn_bn_pool_finish combines the
 * pool_k/32 wave results of every group and applies the affine map to the winner: out / arg / yarg as cpfn_bn_relu_maxpool leaves
 * them (out bit-identical; arg / yarg may name another row of a tie in z between different y), scale / shift from
 * cpfn_bn_finalize's vectors or folded from the layer's seam `in` (then scale = shift = NULL).  pool_k in {32, 64, 128}. */
CPFN_API int cpfn_mlp_gemm_pool_ok(long long P, int K, int N, int pool_k);
CPFN_API int cpfn_mlp_gemm_pool(const void *A, const void *W, long long P, int K, int N, void *Y, float *stats_partial,
                                const cpfn_seam_out *out, const cpfn_seam_in *in, const float *a_scale, const float *a_shift,
                                int pool_k, const float *gamma, void *pmax, unsigned char *pidx, void *stream);
CPFN_API int cpfn_bn_pool_finish(const void *pmax, const unsigned char *pidx, const void *Y, int G, int pool_k, int C,
                                 const float *scale, const float *shift, const cpfn_seam_in *in, void *out, unsigned char *arg,
                                 void *yarg, void *stream);
/* ... and with the [P, K] operand GATHERED while loading (round 6): row p of the operand is table[(p / rows_per_cloud) * n_src +
 * gidx[p]] — sa2's grouped input rows (modules/pointset_abstraction.py:62-66: select_point_subset of the features) read out of
 * the [B, n_src, K] feature table, so cpfn_group_concat_bf16's [P, K] copy is neither written nor read.  rows_per_cloud % 128 == 0;
 * exactly one of stats_partial / out (a seam). */
CPFN_API int cpfn_mlp_gemm_xyz_gather(const void *table, const int *gidx, int rows_per_cloud, int n_src, const void *W,
                                      const float *xyz, const float *Wx, long long P, int K, int N, void *Y, float *stats_partial,
                                      const cpfn_seam_out *out, void *stream);
CPFN_API int cpfn_smallk_fwd_seam(const cpfn_cast_desc *casts /* HOST array or NULL */, int n_casts, const float *X, int KS,
                                  const float *W, long long P, int C, void *Y, const cpfn_seam_out *out, void *stream);

/* Batch statistics -> scale = gamma*rstd, shift = beta - mean*scale (+ running-stat update with
 * torch's momentum / unbiased-variance convention; conv_bias re-enters the running mean).   counter_a / counter_b (optional): int64 step counters this launch advances by one — the BatchNorm
 * module's num_batches_tracked, and the dropout step counter of a stack whose fused output dropout reads it next. */
CPFN_API int cpfn_bn_finalize(const float *partial, int nblk, int N, float count, const float *gamma,
                              const float *beta, const float *conv_bias, float eps, float momentum,
                              float *running_mean, float *running_var, float *scale, float *shift,
                              float *mean, float *rstd, int64_t *counter_a, int64_t *counter_b, void *stream);
/* Evaluation-mode BatchNorm (running statistics; torch.nn.functional.batch_norm with training=False) as the
 * scale / shift of the bias-free GEMM output: out4C = [scale | shift | running_mean - conv_bias | rstd], C floats each —
 * the layout cpfn_bn_finalize writes.  conv_bias may be NULL. */
CPFN_API int cpfn_bn_eval_affine(const float *gamma, const float *beta, const float *conv_bias, const float *running_mean,
                                 const float *running_var, float eps, int C, float *out4C, void *stream);
/* out = relu(scale*y + shift), bf16 [P,C].
 * Fused dropout (optional; the reference's always-on F.dropout on the fc1 features, PointNet2/pn2_network.py:63):
 * with drop_counter non-NULL (a device int64 the caller advances once per forward pass) the output is multiplied by
 * a Bernoulli(1-drop_p) mask / (1-drop_p) generated from splitmix64(drop_base, *drop_counter, element index), and
 * the 8-byte seed is written to drop_seed_out; cpfn_bn_relu_bwd / cpfn_bn_bwd_apply given that seed (and the same
 * drop_p) apply the same mask to the incoming gradient, so no mask tensor exists.  0 <= drop_p < 1 (quantised to
 * 1/65536). */
CPFN_API int cpfn_bn_relu_apply(const void *Y, const float *scale, const float *shift, long long P,
                                int C, void *out, const long long *drop_counter, unsigned long long drop_base,
                                float drop_p, unsigned long long *drop_seed_out, void *stream);
/* out[g,c] = max_k relu(scale*y[g,k,c] + shift) over Kn <= 256 consecutive rows; arg = first k
 * attaining it (u8), yarg = raw y there.  C >= 64, C/8 a power of two. */
CPFN_API int cpfn_bn_relu_maxpool(const void *Y, const float *scale, const float *shift, int G, int Kn,
                                  int C, void *out, unsigned char *arg, void *yarg, void *stream);
/* Backward pass 1: partial[cpfn_bn_bwd_blocks(P)][2][C] = sum(Gz), sum(Gz*y) with Gz = Ga*[z>0];
 * Gz is also stored when the pointer is non-NULL (may alias Ga).  The pooled layers call it on the
 * [G,C] pooled gradient and the pre-BN values at the arg-max rows (only those rows carry gradient). */
CPFN_API int cpfn_bn_bwd_blocks(long long P);
CPFN_API int cpfn_bn_relu_bwd(const void *Ga, const void *Y, const float *scale, const float *shift,
                              long long P, int C, void *Gz, float *partial,
                              const unsigned long long *drop_seed /* NULL: no dropout */, float drop_p, void *stream);
/* The same pass on the SUM of two row-strided bf16 gradients (rows of ldg / ldb elements; Ga 2-byte, Gb 16-byte aligned, ldb % 8 == 0): the two consumers'
 * gradients of one tensor, which autograd (the reference's backward pass: torch's input buffer) adds with a kernel of its own
 * XX.  Gsum [P][C] receives
 * bf16(Ga + Gb), the bits of that add, for the apply pass. */
CPFN_API int cpfn_bn_relu_bwd_join(const void *Ga, int ldg, const void *Gb, int ldb, const void *Y, const float *scale,
                                   const float *shift, long long P, int C, void *Gsum, float *partial, void *stream);
/* dgamma, dbeta and coef[3][C] with g_y = coef0*g_z + coef1*y + coef2. */
CPFN_API int cpfn_bn_bwd_finalize(const float *partial, int nblk, int C, float count, const float *gamma,
                                  const float *mean, const float *rstd, int training, float *dgamma,
                                  float *dbeta, float *coef, void *stream);
/* g_y = coef0*g_z + coef1*y + coef2.  With scale/shift non-NULL the first argument is g_a and the ReLU
 * mask [scale*y+shift > 0] is recomputed (pass 1 then need not store g_z). */
CPFN_API int cpfn_bn_bwd_apply(const void *Gz, const void *Y, const float *coef, const float *scale,
                               const float *shift, long long P, int C, void *Gy,
                               const unsigned long long *drop_seed /* NULL: no dropout */, float drop_p, void *stream);
CPFN_API int cpfn_bn_pool_bwd_apply(const void *Gp, const unsigned char *arg, const void *yarg,
                                    const void *Y, const float *scale, const float *shift,
                                    const float *coef, int G, int Kn, int C, void *Gy, void *stream);
/* dW[N,K] (fp32) = Gy[P,N]^T . A[P,K]; workspace: cpfn_mlp_wgrad_splits(P,N,K)*N*K floats.
 * a_scale, a_shift (optional): as in cpfn_mlp_gemm, A = relu(a_scale*A + a_shift) on the fly.
 * dW may be NULL: only the split partials [splits][N*K] are left in `workspace`, to be finished later,
 * together with those of other layers, by ONE cpfn_multi_split_reduce launch (same fixed summation order). */
typedef struct { const float *partial; float *out; long long n; int splits;
                 int row_in, row_out; /* 0,0: out[n] flat; else partial rows have row_in elements of which the first
                                         row_out are kept: out is [n/row_in, row_out] (zero-padded K) ... */
                 int out_ld;          /* ... at row stride out_ld (0 = row_out: compact; > row_out: a column slice of a wider
                                         matrix, e.g. the 128 feature columns and the 3 coordinate columns of one [N,131] weight) */
                 const float *coef;   /* NULL — or the BatchNorm-backward coefficients [3][C] of an fp32-xyz first layer whose weight
                                         gradient is formed HERE from sums that rode on the layer above's cpfn_mlp_bwd_fused launch (xw_*):
                                         partial [splits][7][C] (S1 = sum g_z x_j, S2 = sum y x_j, S3 = sum x_j), row_in = C, n = 3 C,
                                         out [C][3] = c0 S1 + c1 S2 + c2 S3 */
} cpfn_reduce_desc;
CPFN_API int cpfn_multi_split_reduce(const cpfn_reduce_desc *descs /* HOST array */, int count, void *stream);
/* cpfn_bn_bwd_finalize with up to 6 such reductions riding on the same launch as further workgroups (the weight-gradient partials
 * the launch before it left: read while still in the infinity cache instead of at the end of the backward pass). */
CPFN_API int cpfn_bn_bwd_finalize_ride(const float *partial, int nblk, int C, float count, const float *gamma, const float *mean,
                                       const float *rstd, int training, float *dgamma, float *dbeta, float *coef,
                                       const cpfn_reduce_desc *descs /* HOST array */, int ndesc, void *stream);
/* The finite check of a step's gradients riding on the launches that WRITE them (round 6; the reference scans every parameter's
 * gradient with isinf / isnan after the backward pass, Utils/training_utils.py:151-156): the `_checked` forms OR 1 into ONE device
 * word `flag` (a no-return atomic, only from a workgroup that stored a NaN / inf); its consumer clears it (cpfn_adam_flat_sticky).
 * flag == NULL: the plain entries. */
CPFN_API int cpfn_multi_split_reduce_checked(const cpfn_reduce_desc *descs /* HOST array */, int count, unsigned *flag, void *stream);
CPFN_API int cpfn_bn_bwd_finalize_checked(const float *partial, int nblk, int C, float count, const float *gamma,
                                          const float *mean, const float *rstd, int training, float *dgamma,
                                          float *dbeta, float *coef, unsigned *flag, void *stream);
CPFN_API int cpfn_bn_bwd_finalize_ride_checked(const float *partial, int nblk, int C, float count, const float *gamma,
                                               const float *mean, const float *rstd, int training, float *dgamma, float *dbeta,
                                               float *coef, const cpfn_reduce_desc *descs /* HOST array */, int ndesc,
                                               unsigned *flag, void *stream);
CPFN_API int cpfn_mlp_wgrad_splits(long long P, int N, int K);
CPFN_API int cpfn_mlp_wgrad(const void *Gy, int ldg, const void *A, int lda, const int *gidx, long long P,
                            int N, int K, const float *a_scale, const float *a_shift, float *workspace,
                            float *dW, void *stream);
/* Small layers (P <= 16384 rows): cpfn_mlp_dgrad_small is the data gradient Gout[P,K] (bf16, row stride ldo) = Gy . W with
 * W the FORWARD weight panel [N][K], with pass 1 of the BatchNorm backward of the layer BELOW riding on the tile being
 * stored (bwd_y = that layer's pre-BN output [P,K], b_scale / b_shift [K], stats_partial [cpfn_mlp_gemm_blocks(P,K)][2][K]):
 * replaces that layer's cpfn_bn_relu_bwd launch.  cpfn_mlp_wgrad_apply_ok: the layer's weight gradient runs on 64 x 64
 * tiles (the shapes this small-layer route is taken for). */
CPFN_API int cpfn_mlp_wgrad_apply_ok(long long P, int N, int K);
/* ... and the two as ONE launch (the grid's head is cpfn_mlp_wgrad's 64 x 64 tiles, its tail the data gradient on 32-row
 * tiles): same workspace / Gout / statistics, bit for bit; bwd_y optional; stats_partial has cpfn_mlp_bwd_small_blocks(P) rows. */
CPFN_API int cpfn_mlp_bwd_small_ok(long long P, int N, int K);
CPFN_API int cpfn_mlp_bwd_small_blocks(long long P);
CPFN_API int cpfn_mlp_bwd_small(const void *Gy, int ldg, const void *A, int lda, const void *W, long long P, int N, int K,
                                const float *a_scale, const float *a_shift, float *workspace, void *Gout, int ldo,
                                const void *bwd_y, const float *b_scale, const float *b_shift, float *stats_partial,
                                void *stream);
CPFN_API int cpfn_mlp_dgrad_small_ok(long long P, int N, int K);
CPFN_API int cpfn_mlp_dgrad_small(const void *Gy, const void *W, long long P, int N, int K, void *Gout, int ldo,
                                  const void *bwd_y, const float *b_scale, const float *b_shift, float *stats_partial,
                                  void *stream);
/* Weight gradient AND data gradient of a dense 128 -> 128 layer from ONE read of its BatchNorm-adjoint gradient
 * (cpfn_mlp_bwd_fused_ok(P,N,K): P >= 32768): workspace receives the split partials of
 * dW = Gy^T . A exactly as cpfn_mlp_wgrad leaves them (cpfn_mlp_wgrad_splits(P,128,128) slabs; finish them with
 * cpfn_multi_split_reduce), Gout[P,128] (bf16, row stride ldo) = Gy . W with W the FORWARD weight panel [128][128]
 * bf16.  a_scale / a_shift: as in cpfn_mlp_wgrad.  bwd_y (optional) + b_scale / b_shift + stats_partial
 * [splits][2][128]: pass 1 of the BatchNorm backward of the layer below, as cpfn_mlp_gemm's bwd_y (one partial row per
 * split).  apply_y (optional) + apply_coef [3][128] + y_scale / y_shift: Gy is then the gradient with respect to the
 * layer's ACTIVATED output and cpfn_bn_bwd_apply's arithmetic (ReLU mask from apply_y = the layer's pre-BN output,
 * g_y = c0 . g_z + c1 . y + c2, bf16-rounded) runs on the staged chunks: g_y never exists in memory.  drop_seed /
 * drop_p: cpfn_bn_bwd_apply's fused dropout on that gradient.  pool_k > 0 (max-pooled layer, P = groups x pool_k rows,
 * pool_k a multiple of 64 - 32 for N = K = 128 - and <= 255): Gy is the POOLED gradient [P / pool_k, N] and pool_arg /
 * pool_yarg are cpfn_bn_relu_maxpool's arg-max rows and values: cpfn_bn_pool_bwd_apply's arithmetic instead.
 * (N, K) in {(128,128), (256,128), (64,64), (128,64)}.  xt_xyz [P,3] fp32 + xt_partial [splits][128][3] (N = K = 128, apply_y,
 * no bwd_y / drop_seed / pool_k): the layer has three more input channels, the coordinates (cpfn_mlp_gemm_xyz), whose
 * weight-gradient columns g_y^T . xyz leave as split partials in xt_partial.  Replaces a [cpfn_bn_bwd_apply | cpfn_bn_pool_bwd_apply +] cpfn_mlp_wgrad +
 * cpfn_mlp_gemm(w_trans) sequence, bit for bit. */
CPFN_API int cpfn_mlp_bwd_fused_ok(long long P, int N, int K);
CPFN_API int cpfn_mlp_bwd_fused(const void *Gy, int ldg, const void *A, int lda, const void *W, long long P, int N, int K,
                                const float *a_scale, const float *a_shift, float *workspace, void *Gout, int ldo,
                                const void *bwd_y, const float *b_scale, const float *b_shift, float *stats_partial,
                                const void *apply_y, const float *apply_coef, const float *y_scale, const float *y_shift,
                                const unsigned long long *drop_seed, float drop_p, const unsigned char *pool_arg,
                                const void *pool_yarg, int pool_k, const float *xt_xyz, float *xt_partial, void *stream);
/* The 64 <- 64 shape of cpfn_mlp_bwd_fused (sa1's second layer: dense apply pass, riding reduction of the layer below) when the
 * layer BELOW is the fp32-xyz first layer (modules/pointset_abstraction.py:70: the first 1x1 convolution over the centred
 * coordinates): that layer's weight gradient dW0[c][j] = sum_p g_y[p,c] x[p,j], g_y = c0 g_z + c1 y + c2, is linear in its
 * BatchNorm-backward coefficients, so its sums S1 = sum g_z x_j, S2 = sum y x_j, S3 = sum x_j ride on this launch
 * (xw_xyz [P,3], xw_partial [splits][7][64]) and cpfn_multi_split_reduce finishes c0 S1 + c1 S2 + c2 S3 once the coefficients exist
 * (cpfn_reduce_desc.coef): cpfn_smallk_wgrad_apply_xyz is not launched, and with Gout = NULL the gradient w.r.t. the first layer's
 * output is never stored.  ldg = N, lda = ldo = K. */
/* cpfn_mlp_bwd_fused's xyz-tail form (N = K = 128, dense apply pass, no layer below, xt_xyz / xt_partial as there) with the
 * layer's input rows gathered from the same table while loading (the operand of cpfn_mlp_gemm_xyz_gather). */
CPFN_API int cpfn_mlp_bwd_fused_xt_gather(const void *Gy, const void *table, const int *gidx, int rows_per_cloud, int n_src,
                                          const void *W, long long P, int N, int K, float *workspace, void *Gout,
                                          const void *apply_y, const float *apply_coef, const float *y_scale, const float *y_shift,
                                          const float *xt_xyz, float *xt_partial, void *stream);
CPFN_API int cpfn_mlp_bwd_fused_xw(const void *Gy, const void *A, const void *W, long long P, int N, int K, const float *a_scale,
                                   const float *a_shift, float *workspace, void *Gout, const void *bwd_y, const float *b_scale,
                                   const float *b_shift, float *stats_partial, const void *apply_y, const float *apply_coef,
                                   const float *y_scale, const float *y_shift, const float *xw_xyz, float *xw_partial,
                                   void *stream);
/* Column sums of a row-major fp32 matrix X[P,C], C <= 64 (bias gradient of the fc2 heads).
 * workspace: ceil(P/256)*C floats.  pad_bf16 (optional): [P,64] bf16, receives the rows of X converted to bf16
 * and zero-padded to 64 columns in the same pass (the gradient operand of the heads' GEMMs).
 * out == NULL: only the per-block partials are left in workspace (ceil(P/256) rows of C floats) for the caller
 * to finish with cpfn_multi_split_reduce; the same holds for dW == NULL in cpfn_mlp_wgrad / cpfn_smallk_wgrad. */
CPFN_API int cpfn_colsum_f32(const float *X, long long P, int C, float *workspace, float *out, void *pad_bf16,
                             void *stream);
/* fp32 first layer with K = KS <= 4 inputs (sa1: relative xyz stay fp32):
 * Y[P,C] bf16 = X[P,KS] . W[C,KS]^T, partial[cpfn_bn_bwd_blocks(P)][2][C]; and its weight gradient
 * (workspace: cpfn_bn_bwd_blocks(P)*C*KS floats). */
CPFN_API int cpfn_smallk_fwd(const float *X, int KS, const float *W, long long P, int C, void *Y,
                             float *partial, void *stream);
/* cpfn_smallk_fwd (KS = 3) and cpfn_multi_cast (n_casts <= 64 descriptors) as ONE launch: the step's weight-panel refresh rides as
 * the first workgroups of sa1's first layer, which reads the fp32 weight itself and does not depend on it. */
CPFN_API int cpfn_smallk_fwd_cast(const cpfn_cast_desc *casts /* HOST array */, int n_casts, const float *X, int KS, const float *W,
                                  long long P, int C, void *Y, float *partial, void *stream);
CPFN_API int cpfn_smallk_wgrad(const void *Gy, const float *X, int KS, long long P, int C,
                               float *workspace, float *dW, void *stream);
/* The same with cpfn_bn_bwd_apply folded in: Gz is the gradient w.r.t. the layer's ACTIVATED output, Y its pre-BN output;
 * g_y = bf16(coef0 . [y_scale . y + y_shift > 0] . g_z + coef1 . y + coef2) is formed on the operand load and never stored. */
CPFN_API int cpfn_smallk_wgrad_apply(const void *Gz, const void *Y, const float *coef, const float *y_scale,
                                     const float *y_shift, const float *X, int KS, long long P, int C, float *workspace,
                                     float *dW, void *stream);
/* ... and with the layer's pre-BN output y = bf16(W0 [C][KS] . X) RECOMPUTED from the coordinates (cpfn_smallk_fwd's
 * arithmetic) instead of read: the backward pass of an fp32-xyz first layer then never touches its [P,C] output. */
CPFN_API int cpfn_smallk_wgrad_apply_xyz(const void *Gz, const float *W0, const float *coef, const float *y_scale,
                                         const float *y_shift, const float *X, int KS, long long P, int C,
                                         float *workspace, float *dW, void *stream);

/* ------------------------------------------------------------------ loss-side fusions
 * (the callers on the far side of the fitters.)  K <= 32 for the training-side
 * kernels (head_post, seg_stats_bwd, hungarian_match); cpfn_seg_stats_fwd and cpfn_p_coverage, which the evaluation
 * metrics also run on merged label sets (evaluation_localSPFN.py:129-131), take any K. */

/* Heads post-processing: Y[B,N,7+K] fp32 (3 normal, 4 type logits, K membership logits) ->
 * Xn[B,N,3] = normalize (Utils/training_utils.py:141), Wsm[B,N,K] = softmax (:142),
 * stats[B,3] = (normal loss, type loss, #labelled points) per cloud
 * (SPFN/losses_implementation.py:152-159, 195-210; training forms).  Igt[B,N], Tgt[B,K] int64.
 * workspace: B * cpfn_head_post_chunks(N) * 3 floats.
 * seg_workspace + S (optional, both or neither; K <= 31): the same launch also leaves the label-segmented sums
 * S[B,K+2,K] of cpfn_seg_stats_fwd, taken from the soft-max rows while they are on chip (fp32 MFMA contraction
 * one-hot(label) x memberships); seg_workspace: B * cpfn_head_post_chunks(N) * (K+2)*K floats.
 * lab_workspace (B * chunks ints) + n_gt [B] (optional, both or neither): the number of GT instances per cloud (largest
 * label + 1, what cpfn_count_labels computes) from the same pass. */
CPFN_API int cpfn_head_post_chunks(int N);
CPFN_API int cpfn_head_post_fwd(const float *Y, const float *Xgt, const int64_t *Igt, const int64_t *Tgt,
                                int B, int N, int K, float *Xn, float *Wsm, float *workspace, float *stats,
                                float *seg_workspace, float *S, int *lab_workspace, int64_t *n_gt, void *stream);
/* Adjoint: gXn[B,N,3], gW[B,N,K] (either may be NULL), gloss = dL/d(normal, type loss) -> gY.
 * gloss_planar = 0: gloss is [B,2]; 1: [2,B] (the two gradient vectors one after the other, as cpfn_loss_tail leaves
 * them: no interleaving copy).  gS (optional) [B,K+2,K] = gradient w.r.t. the segmented sums cpfn_head_post_fwd
 * left in S: its adjoint (cpfn_seg_stats_bwd's dW) is added to gW inside this launch. */
/* (pad_bf16 [B*N,64] bf16 + colsum_partial [B*N/256][7+K], optional, N % 256 == 0: what cpfn_colsum_f32 would make of gY for the
 * heads' backward, produced from the tile in the same pass) */
CPFN_API int cpfn_head_post_bwd(const float *Y, const float *Xgt, const int64_t *Igt, const int64_t *Tgt,
                                const float *Wsm, const float *stats, const float *gXn, const float *gW,
                                const float *gloss, int gloss_planar, int B, int N, int K, float *gY, const float *gS, void *pad_bf16, float *colsum_partial,
                                void *stream);
/* The PatchSelection objective (Utils/training_utils.py:66-68: F.cross_entropy of the [B*N, 2] heat-map logits against
 * per-point labels, mean reduction) and its gradient in one pass: logits[P,2] fp32, labels[P] int64 (0 / non-zero) ->
 * loss[1] = mean_p (logsumexp - logit of the label), dlogits[P,2] = (softmax - onehot) / P.  Optionally (both or neither;
 * P % 256 == 0) what the fc2 heads' backward makes of dlogits first, as cpfn_head_post_bwd leaves it: pad_bf16[P,64] zero-padded
 * bf16 rows and colsum_partial[P/256][2] column sums in cpfn_colsum_f32's order.  workspace: cpfn_ce2_blocks(P) floats. */
CPFN_API int cpfn_ce2_blocks(long long P);
CPFN_API int cpfn_ce2(const float *logits, const int64_t *labels, long long P, float *workspace, float *loss, float *dlogits,
                      void *pad_bf16, float *colsum_partial, void *stream);
/* Label-segmented membership sums, shared by the Hungarian cost matrix and the relaxed-IoU loss
 * (SPFN/losses_implementation.py:19-24, 77-90):  S[B,K+2,K]: rows l<K = sum of W rows with label l,
 * row K = column sums of W, row K+1 = number of points per label.  fwd: any K <= 1024 (K > 32: one 32 x 32 tile
 * of S per workgroup); bwd: K <= 32.
 * workspace: B * cpfn_seg_stats_chunks(B,N) * (K+2)*K floats. */
CPFN_API int cpfn_seg_stats_chunks(int B, int N);
CPFN_API int cpfn_seg_stats_fwd(const float *W, const int64_t *Igt, int B, int N, int K, float *workspace,
                                float *S, void *stream);
CPFN_API int cpfn_seg_stats_bwd(const float *gS, const int64_t *Igt, int B, int N, int K, float *dW,
                                void *stream);
/* Residue + axis losses of every GT instance against its matched prediction, for the instance's GT
 * type only (SPFN/losses_implementation.py:351-387, 480-497; SPFN/{plane,sphere,cylinder,cone}_fitter.compute_residue_single).
 * params[B,K,22] = plane n(3) c | sphere c(3) r2 | cylinder a(3) c(3) r2 | cone apex(3) axis(3) half;
 * pts[B,K,NP,3]; gt_axes[3,B,K,3] = GT plane normal / cylinder axis / cone axis;
 * type_ids = HOST array of the ids of (plane, sphere, cylinder, cone).
 * out[B,K,2] = (mean residue, 1-|axis.axis_gt|); dout[B,K,10] = their derivatives (saved for bwd). */
CPFN_API int cpfn_residue_fwd(const float *params, const int64_t *match, const int64_t *Tgt,
                              const float *pts, const float *gt_axes, int B, int K, int NP,
                              const int *type_ids, float *out, float *dout, void *stream);
/* gparams[B,K,22] = gout[B,K,2] . dout, routed through match (every element written: no zero fill needed; the GT
 * instances assigned to one prediction are summed in ascending order). */
CPFN_API int cpfn_residue_bwd(const float *gout, const float *dout, const int64_t *match,
                              const int64_t *Tgt, int B, int K, const int *type_ids, float *gparams,
                              void *stream);

/* The assignment of GT instances to predictions on the device (the reference does it on the host:
 * SPFN/losses_implementation.py:10-30, scipy.optimize.linear_sum_assignment(-cost) per cloud):
 * S[B,K+2,K] from cpfn_seg_stats_fwd, n_gt[B] = number of GT instances -> match[B,K] int64 (zeros
 * beyond n_gt).  Same solver as SciPy 1.15 (Crouse's shortest augmenting path, fp64), same tie
 * breaking, hence the same matching.  K <= 64 (one lane per column; merged label sets of the evaluation
 * cascade).  Non-finite costs (SciPy raises): the cloud's rows get the identity. */
CPFN_API int cpfn_hungarian_match(const float *S, const int64_t *n_gt, int B, int K, int64_t *match,
                                  void *stream);
/* Evaluation metric "P coverage" (SPFN/metric_implementation.py:409-415): out[b, i] = fraction of the N
 * points of cloud b whose smallest residue over the K instance slots is below eps[i]; slot k uses the
 * parameters of prediction match[b,k] (params22 in the cpfn_fit_pack_fwd layout) evaluated as primitive
 * type slot_type[b,k]; residue = sqrt(|r| + 1e-10) of the fitters' compute_residue_single.
 * type_ids (HOST, 4 ints) = ids of plane, sphere, cylinder, cone; eps (HOST) n_eps <= 4 thresholds; any K.
 * workspace: B * ceil(N/256) * n_eps floats. */
CPFN_API int cpfn_p_coverage(const float *P, const float *params22, const int64_t *match,
                             const int64_t *slot_type, int B, int N, int K, const int *type_ids,
                             const float *eps, int n_eps, float *workspace, float *out, void *stream);
/* Evaluation metrics (SPFN/metric_implementation.py:485-514, compute_all_metrics) around the assignment and the fits.
 * cpfn_metrics_points — one pass over the points (replaces hard_W_encoding :33-37, get_instance_type :52-55, the cost
 * inputs of hungarian_matching :19-25 and compute_normal_difference): W[B,N,K] soft memberships, T[B,N,n_types] per-point
 * type scores, X / Xgt [B,N,3] unit normals, Igt[B,N] (gap-free labels, -1 = background) ->
 *   hardW[B,N,Kp]  one-hot of the arg-max membership (first index on ties), zero columns K..Kp-1 (Kp >= K: the label sets
 *                  padded to a common width like :487-492),
 *   S[B,Kp+2,Kp]   the segmented sums of hardW in cpfn_seg_stats_fwd's layout — for one-hot rows a joint histogram,
 *                  counted in integers (exact),
 *   n_gt[B]        largest GT label + 1,  T_inst[B,Kp]  arg-max over types of (hardW^T T),  normal_diff[B] = mean acos|x.x_gt|.
 * Kp <= 128, n_types <= 8.  workspace: cpfn_metrics_workspace(B, N, Kp, n_types) bytes (zeroed by the call itself). */
CPFN_API long long cpfn_metrics_workspace(int B, int N, int Kp, int n_types);
CPFN_API int cpfn_metrics_points(const float *W, const float *T, const float *X, const float *Xgt, const int64_t *Igt,
                                 int B, int N, int K, int Kp, int n_types, float *hardW, void *workspace, float *S,
                                 int64_t *n_gt, int64_t *T_inst, float *normal_diff, void *stream);
/* cpfn_metrics_tail — everything of compute_all_metrics behind the assignment and the fits that is [B,K]- or
 * [B,K,N']-sized (compute_segmentation_iou, compute_type_accuracy, compute_axis_difference, get_residual_loss :76-81,
 * compute_meanstd_Sk_residual, compute_Sk_coverage) in ONE launch: match[B,Kp] from cpfn_hungarian_match, params22[B,Kp,22]
 * from cpfn_fit_pack_fwd, T_gt[B,Kgt], ppi[B,Kgt,Np,3] points per GT instance, axis_*[B,Kgt,3] GT axes (slots beyond Kgt
 * count as zero axes / zero points, the reference's padding :505-508) ->
 *   out[B, 5 + n_eps] = mIoU, type accuracy, axis difference, mean residual, std residual (unbiased over the Np points,
 *                        averaged over the n_gt instances), Sk coverage per eps;
 *   slot_type[B,Kp]   = T_inst[b, match[b,k]]  (what cpfn_p_coverage evaluates slot k as, :412).
 * type_ids (HOST, 4 ints) = ids of plane, sphere, cylinder, cone; eps (HOST) n_eps <= 4.  Kp <= 1024, Np >= 2. */
CPFN_API int cpfn_metrics_tail(const float *S, const int64_t *match, const int64_t *n_gt, const int64_t *T_inst,
                               const int64_t *T_gt, const float *params22, const float *ppi, const float *axis_plane,
                               const float *axis_cylinder, const float *axis_cone, int B, int Kp, int Kgt, int Np,
                               const int *type_ids, const float *eps, int n_eps, float *out, int64_t *slot_type,
                               void *stream);
/* The [B,K]-sized tail of compute_all_losses (SPFN/losses_implementation.py:77-90, 603-606, 633-673)
 * in one launch: relaxed IoU of the matched pairs from S[B,K+2,K] (cpfn_seg_stats_fwd), masked means
 * over the n_gt[b] existing instances of that and of rp[B,K,2] (cpfn_residue_fwd; may be NULL), batch
 * means of those and of nl/tl (element stride nl_stride), the weighted total.  mult6 (HOST array) =
 * normal, type, miou, residue, parameter, total multipliers; a part with multiplier <= 0 is reported
 * as 0 and gets no gradient.  out6 = total, normal, type, miou, residue, parameter.  Also written:
 * d total / d S, rp, nl, tl (gS[B,K+2,K], grp[B,K,2], gnl[B], gtl[B]).  B <= 1024. */
CPFN_API int cpfn_loss_tail(const float *S, const float *rp, const float *nl, const float *tl,
                            int nl_stride, const int64_t *match, const int64_t *n_gt, int B, int K,
                            const float *mult6, float *out6, float *gS, float *grp, float *gnl,
                            float *gtl, void *stream);

/* flag[0] = 1.0f if any of x[0..n) (fp32, 16-byte aligned) is NaN or +-inf, else 0.0f: the finite
 * check of the gradients (Utils/training_utils.py:151-156) as one streaming pass.  workspace256: 256 uints. */
CPFN_API int cpfn_nonfinite_flag(const float *x, long long n, unsigned *workspace256, float *flag,
                                 void *stream);

/* Adam step on flat fp32 buffers p, g, m, v [n] (16-byte aligned), torch.optim.Adam arithmetic
 * (non-amsgrad; the optimizer of the reference's epoch loop, Utils/training_utils.py).  Capturable:
 * lr, step (count of steps taken so far, incremented here), pows (fp64 {beta1^step, beta2^step},
 * start at {1, 1}, advanced here) and found_inf (may be NULL; non-zero = skip the whole step) are
 * device scalars; coef3 = 3 floats of device scratch (bias-correction terms). */
CPFN_API int cpfn_adam_flat(float *p, const float *g, float *m, float *v, long long n, const float *lr,
                            float beta1, float beta2, float eps, float weight_decay, float *step,
                            double *pows, const float *found_inf, float *coef3,
                            const unsigned *nf_partial /* optional: the nf_count per-block flags left by
                            cpfn_nonfinite_partial over g; any set flag skips the step */, int nf_count,
                            float *skipped /* optional device counter, +1 for a skipped step */, void *stream);
/* The same with the first n_sticky words of nf_partial OR-accumulated by the launches that WROTE the gradients (the `_checked`
 * reductions / finalizes above) instead of rewritten by a scan: they are read and then CLEARED here for the next step. */
CPFN_API int cpfn_adam_flat_sticky(float *p, const float *g, float *m, float *v, long long n, const float *lr,
                                   float beta1, float beta2, float eps, float weight_decay, float *step,
                                   double *pows, const float *found_inf, float *coef3, unsigned *nf_partial, int nf_count,
                                   int n_sticky, float *skipped, void *stream);
/* ... and with the LAST gradient of the backward pass finished by the prepare kernel: the weight gradient xw_out [C][3] of an fp32-xyz
 * first layer, c0[c] S1[j][c] + c1[c] S2[j][c] + c2[c] S3[j] from xw_S [7][C] (the sums that rode on the layer above, cpfn_mlp_bwd_fused_xw,
 * already reduced over their splits) and xw_coef [3][C] (that layer's cpfn_bn_bwd_finalize coefficients) — the arithmetic of
 * cpfn_multi_split_reduce's coefficient form, checked for NaN / inf with the rest.  xw_out (a slice of g) == NULL: cpfn_adam_flat_sticky. */
CPFN_API int cpfn_adam_flat_xw(float *p, const float *g, float *m, float *v, long long n, const float *lr,
                               float beta1, float beta2, float eps, float weight_decay, float *step,
                               double *pows, const float *found_inf, float *coef3, unsigned *nf_partial, int nf_count,
                               int n_sticky, float *skipped, const float *xw_S, const float *xw_coef, int xw_C, float *xw_out,
                               void *stream);
/* First half of cpfn_nonfinite_flag: workspace256[i] = 1 if block i of x holds a NaN / inf, for
 * i < cpfn_nonfinite_blocks(n) (<= 256); the reduction is then done by cpfn_adam_flat's prepare kernel. */
CPFN_API int cpfn_nonfinite_blocks(long long n);
CPFN_API int cpfn_nonfinite_partial(const float *x, long long n, unsigned *workspace256, void *stream);

/* ------------------------------------------------------------------ patch merging (evaluation, config 5)
 * Replaces the two tensor functions of Utils/merging_utils.py that evaluation_localSPFN.py:101-110 runs on
 * the device.
 *
 * cpfn_similarity_soft (merging_utils.py:6-15): Gram matrix out[C,C] (C = nb*Lp + Lo, fp32) of the
 * point-to-primitive matrix whose columns b*Lp..(b+1)*Lp hold predicted_labels[b] [npp, Lp] scattered to the rows
 * point_indices[b] [npp] (int64, values in [0,N), no repeats inside a patch) and whose last Lo columns hold
 * spfn_labels [N, Lo] (fp32).  The dense matrix is never built.  Lp, Lo <= 32.  workspace: device memory of
 * cpfn_similarity_soft_workspace(...) bytes (256-byte aligned).
 *
 * cpfn_label_pool (merging_utils.py:56-60, get_point_final): out[N,G] with out[p][g] = (sum of M[p][c] over the
 * columns c with labels[c] == g, ascending c) / (number of such columns + 1e-10).  labels: C int64 values (those
 * outside [0,G) are ignored).  workspace: (C + 2*G + 2) * 4 bytes of device scratch.  C <= 12288, G <= 16384. */
CPFN_API long long cpfn_similarity_soft_workspace(int N, int nb, int npp, int Lp, int Lo);
CPFN_API int cpfn_similarity_soft(const float *spfn_labels, const float *predicted_labels,
                                  const int64_t *point_indices, int N, int nb, int npp, int Lp, int Lo,
                                  void *workspace, float *out, void *stream);
CPFN_API int cpfn_label_pool(const float *M, const int64_t *labels, long long N, int C, int G, void *workspace,
                             float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CPFN_HIP_H */
