// Per-instance algebra of the four primitive fitters on the fused moments, gfx950.
//
//   M[B*K, 52] (fp64 moments, slot map in include/cpfn_hip.h)  ->  out[B*K, 21] (fp64):
//     0-2 plane normal | 3 plane offset | 4-6 sphere centre | 7 sphere r² | 8-10 cylinder axis |
//     11-13 cylinder centre | 14 cylinder r² | 15-17 cone apex | 18-20 cone axis (before the sign fix)
//
// This is the [B,K]-sized tail of SPFN/{plane,sphere,cylinder,cone}_fitter.compute_parameters
// (3x3 TLS eigenvectors, guarded 3x3 / 2x2 least-squares solves, the cylinder's plane frame):
// a few thousand flops per instance, which as ~400 tiny framework kernels per direction was the
// largest launch-latency item of the step.  One lane per instance here.
//
// The SAME templated code is the forward (T = double) and the backward pass: with
// T = Dual (value + one tangent) lane d of an instance pushes the unit tangent e_d of moment
// slot d through the algebra and contracts the result with the upstream gradient, i.e. one
// row of Jᵀg per lane (forward-mode AD, 52 lanes per instance).  The only hand-written
// derivative is the TLS eigenvector, which uses the reference's guarded formula
// (SPFN/differentiable_tls.py:131-143; cpfn_amd/SPFN/differentiable_tls.py documents the
// collapse to the last column).  Value-only decisions (condition-number masks, the frame's
// arg-max, clamps) carry zero tangent exactly like the reference's detach()/argmax/clamp.
#include "common.h"
#include "fit_pack.h"
#include "fit_internal.h"

namespace {

constexpr int NM = 52, NO = 21;

struct Dual {
  double v, d;
  __device__ Dual() : v(0), d(0) {}
  __device__ Dual(double a) : v(a), d(0) {}
  __device__ Dual(double a, double b) : v(a), d(b) {}
};
__device__ inline Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.d + b.d); }
__device__ inline Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.d - b.d); }
__device__ inline Dual operator-(Dual a) { return Dual(-a.v, -a.d); }
__device__ inline Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, a.d * b.v + a.v * b.d); }
__device__ inline Dual operator/(Dual a, Dual b) {
  const double q = a.v / b.v;
  return Dual(q, (a.d - q * b.d) / b.v);
}
__device__ inline double val(double a) { return a; }
__device__ inline double val(Dual a) { return a.v; }
__device__ inline double tsqrt(double a) { return sqrt(a); }
__device__ inline Dual tsqrt(Dual a) {
  const double s = sqrt(a.v);
  return Dual(s, a.d / (2.0 * s));   // a.v = 0 gives inf like torch's sqrt backward
}
__device__ inline double clamp_min(double a, double lo) { return a < lo ? lo : a; }
__device__ inline Dual clamp_min(Dual a, double lo) { return a.v < lo ? Dual(lo, 0.0) : a; }

// ---- symmetric 3x3 Jacobi on plain doubles: ascending eigenvalues, eigenvectors in columns
__device__ void jacobi3(const double s[6], double lam[3], double V[3][3]) {
  double a[3][3] = {{s[0], s[1], s[2]}, {s[1], s[3], s[4]}, {s[2], s[4], s[5]}};
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 16; ++sweep) {
    const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
    if (off <= 1e-300 || off <= 1e-22 * diag) break;
#pragma unroll          // (p, q must be compile-time: a[][] / v[][] indexed by a loop variable live in scratch memory)
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      const double apq = a[p][q];
      if (apq == 0.0) continue;
      const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
      const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
      for (int r = 0; r < 3; ++r) { const double x = a[r][p], y = a[r][q]; a[r][p] = c * x - sn * y; a[r][q] = sn * x + c * y; }
      for (int r = 0; r < 3; ++r) { const double x = a[p][r], y = a[q][r]; a[p][r] = c * x - sn * y; a[q][r] = sn * x + c * y; }
      for (int r = 0; r < 3; ++r) { const double x = v[r][p], y = v[r][q]; v[r][p] = c * x - sn * y; v[r][q] = sn * x + c * y; }
    }
  }
  // ascending eigenvalues: the three compare-exchanges of a bubble sort, the eigenvector columns swapped along (statically
  // indexed: a permutation table would put v[][] in scratch memory)
  double e[3] = {a[0][0], a[1][1], a[2][2]};
#define CPFN_JACOBI_CSWAP(i, j)                                                              \
  if (e[i] > e[j]) {                                                                         \
    const double te = e[i]; e[i] = e[j]; e[j] = te;                                          \
    for (int r = 0; r < 3; ++r) { const double tv = v[r][i]; v[r][i] = v[r][j]; v[r][j] = tv; } \
  }
  CPFN_JACOBI_CSWAP(0, 1)
  CPFN_JACOBI_CSWAP(1, 2)
  CPFN_JACOBI_CSWAP(0, 1)
#undef CPFN_JACOBI_CSWAP
  for (int i = 0; i < 3; ++i) {
    lam[i] = e[i];
    for (int r = 0; r < 3; ++r) V[r][i] = v[r][i];
  }
}

// eigenvector of the smallest eigenvalue of the symmetric matrix with unique entries s6
__device__ void smallest_eigvec(const double s6[6], double out[3]) {
  double lam[3], V[3][3];
  jacobi3(s6, lam, V);
  for (int r = 0; r < 3; ++r) out[r] = V[r][0];
}
__device__ void smallest_eigvec(const Dual s6[6], Dual out[3]) {
  double sv[6], lam[3], V[3][3];
  for (int i = 0; i < 6; ++i) sv[i] = s6[i].v;
  jacobi3(sv, lam, V);
  // dM (symmetric) times v0
  const double dM[3][3] = {{s6[0].d, s6[1].d, s6[2].d}, {s6[1].d, s6[3].d, s6[4].d}, {s6[2].d, s6[4].d, s6[5].d}};
  double dMv[3];
  for (int r = 0; r < 3; ++r) dMv[r] = dM[r][0] * V[0][0] + dM[r][1] * V[1][0] + dM[r][2] * V[2][0];
  const double s2 = fabs(lam[0]);
  double dv[3] = {0, 0, 0};
  for (int i = 1; i < 3; ++i) {   // the two larger singular values
    const double si = fabs(lam[i]);
    double gap = s2 * s2 - si * si;            // K[2,i] of the reference, guarded to <= -1e-10
    gap = gap < -1e-10 ? gap : -1e-10;
    const double proj = V[0][i] * dMv[0] + V[1][i] * dMv[1] + V[2][i] * dMv[2];
    const double coef = proj * (si + s2) / gap;
    for (int r = 0; r < 3; ++r) dv[r] += coef * V[r][i];
  }
  for (int r = 0; r < 3; ++r) out[r] = Dual(V[r][0], dv[r]);
}

template <typename T>
__device__ void sym_from6(const T s[6], T S[3][3]) {
  S[0][0] = s[0]; S[0][1] = s[1]; S[0][2] = s[2];
  S[1][0] = s[1]; S[1][1] = s[3]; S[1][2] = s[4];
  S[2][0] = s[2]; S[2][1] = s[4]; S[2][2] = s[5];
}

// weighted_plane_fitting from raw sums (SPFN/geometry_utils.py:74-84)
template <typename T>
__device__ void fit_plane(T S0, const T S1[3], const T S2[6], T n[3], T &c) {
  const T den = clamp_min(S0, 1e-10);
  T mean[3];
  for (int i = 0; i < 3; ++i) mean[i] = S1[i] / den;
  T S[3][3];
  sym_from6(S2, S);
  T C6[6];
  int q = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = i; j < 3; ++j) C6[q++] = S[i][j] - mean[i] * S1[j] - S1[i] * mean[j] + S0 * mean[i] * mean[j];
  smallest_eigvec(C6, n);
  c = n[0] * mean[0] + n[1] * mean[1] + n[2] * mean[2];
}

// (AtA·mask + 1e-8 I) x = Atb·mask, mask = cond(AtA) < 1e5 on values (geometry_utils.py:132-140)
template <typename T>
__device__ void guarded_solve3(const T A[3][3], const T b[3], T x[3]) {
  double a6[6] = {val(A[0][0]), val(A[0][1]), val(A[0][2]), val(A[1][1]), val(A[1][2]), val(A[2][2])};
  double lam[3], V[3][3];
  jacobi3(a6, lam, V);
  const double s0 = fabs(lam[0]), s1 = fabs(lam[1]), s2 = fabs(lam[2]);
  const double smax = fmax(s0, fmax(s1, s2)), smin = fmin(s0, fmin(s1, s2));
  const double mask = (smax / smin < 1e5) ? 1.0 : 0.0;
  T M[3][3], r[3];
  for (int i = 0; i < 3; ++i) {
    r[i] = b[i] * T(mask);
    for (int j = 0; j < 3; ++j) M[i][j] = A[i][j] * T(mask) + T(i == j ? 1e-8 : 0.0);
  }
  // Cramer via the cross products of the columns
  T c0[3] = {M[1][1] * M[2][2] - M[2][1] * M[1][2], M[2][1] * M[0][2] - M[0][1] * M[2][2], M[0][1] * M[1][2] - M[1][1] * M[0][2]};
  T c1[3] = {M[1][2] * M[2][0] - M[2][2] * M[1][0], M[2][2] * M[0][0] - M[0][2] * M[2][0], M[0][2] * M[1][0] - M[1][2] * M[0][0]};
  T c2[3] = {M[1][0] * M[2][1] - M[2][0] * M[1][1], M[2][0] * M[0][1] - M[0][0] * M[2][1], M[0][0] * M[1][1] - M[1][0] * M[0][1]};
  const T det = M[0][0] * c0[0] + M[1][0] * c0[1] + M[2][0] * c0[2];
  x[0] = (c0[0] * r[0] + c0[1] * r[1] + c0[2] * r[2]) / det;
  x[1] = (c1[0] * r[0] + c1[1] * r[1] + c1[2] * r[2]) / det;
  x[2] = (c2[0] * r[0] + c2[1] * r[1] + c2[2] * r[2]) / det;
}

template <typename T>
__device__ void guarded_solve2(const T A[2][2], const T b[2], T x[2]) {
  const double a = val(A[0][0]), bb = val(A[0][1]), c = val(A[1][1]);
  const double mid = 0.5 * (a + c), rad = sqrt(0.25 * (a - c) * (a - c) + bb * bb);
  const double e0 = fabs(mid + rad), e1 = fabs(mid - rad);
  const double mask = (fmax(e0, e1) / fmin(e0, e1) < 1e5) ? 1.0 : 0.0;
  const T m00 = A[0][0] * T(mask) + T(1e-8), m01 = A[0][1] * T(mask), m10 = A[1][0] * T(mask), m11 = A[1][1] * T(mask) + T(1e-8);
  const T r0 = b[0] * T(mask), r1 = b[1] * T(mask);
  const T det = m00 * m11 - m01 * m10;
  x[0] = (m11 * r0 - m01 * r1) / det;
  x[1] = (m00 * r1 - m10 * r0) / det;
}

// weighted_sphere_fitting (geometry_utils.py:209-223) in D dimensions from raw sums:
// S* weighted by w (1, p, p pᵀ), T* by clamp(w) (1, p, p pᵀ, |p|² p)
template <typename T, int D>
__device__ void fit_sphere(T S0, const T S1[D], const T S2[D][D], T T0, const T T1[D], const T T2[D][D], const T T3c[D],
                           T centre[D], T &r2) {
  const T den = clamp_min(S0, 1e-10);
  T mean[D], trS2 = T(0.0), trT2 = T(0.0);
  for (int i = 0; i < D; ++i) { mean[i] = S1[i] / den; trS2 = trS2 + S2[i][i]; trT2 = trT2 + T2[i][i]; }
  const T m2 = trS2 / den;
  T AtA[D][D], Atb[D];
  for (int i = 0; i < D; ++i) {
    for (int j = 0; j < D; ++j) AtA[i][j] = T(4.0) * (T0 * mean[i] * mean[j] - mean[i] * T1[j] - T1[i] * mean[j] + T2[i][j]);
    Atb[i] = T(2.0) * (mean[i] * (m2 * T0 - trT2) - m2 * T1[i] + T3c[i]);
  }
  if constexpr (D == 3) guarded_solve3(AtA, Atb, centre);
  else guarded_solve2(AtA, Atb, centre);
  T cs1 = T(0.0), cc = T(0.0);
  for (int i = 0; i < D; ++i) { cs1 = cs1 + centre[i] * S1[i]; cc = cc + centre[i] * centre[i]; }
  r2 = (trS2 - T(2.0) * cs1 + cc * S0) / den;
}

// index of (i,j,k) in the 10 unique third-order moments (xxx xxy xxz xyy xyz xzz yyy yyz yzz zzz)
__device__ __constant__ int T3IDX[3][3][3] = {{{0, 1, 2}, {1, 3, 4}, {2, 4, 5}}, {{1, 3, 4}, {3, 6, 7}, {4, 7, 8}}, {{2, 4, 5}, {4, 7, 8}, {5, 8, 9}}};

// PART = -1: all four fits; 0 plane, 1 sphere, 2 cylinder, 3 cone first pass (each writes only its own columns of
// out: 0..3 | 4..7 | 8..14 | 15..20).  The kernels give every part its own WAVE, so the four fits of an instance
// run side by side on four SIMDs instead of one after the other in one lane.
template <typename T, int PART = -1>
__device__ void fit_all(const T *M, T *out) {
  const T S0 = M[0], T0 = M[20];
  const T *S1 = M + 1, *S2 = M + 4, *Sx = M + 10, *Sxx = M + 13;
  const T *T1 = M + 21, *T2 = M + 24, *T3 = M + 30, *Tnn = M + 40, *Tnpn = M + 46;
  // ---- plane (plane_fitter.py:9-17)
  if (PART < 0 || PART == 0) fit_plane(S0, S1, S2, out + 0, out[3]);
  // ---- sphere (sphere_fitter.py:9-19)
  if (PART < 0 || PART == 1) {
    T S2m[3][3], T2m[3][3], T3c[3];
    sym_from6(S2, S2m);
    sym_from6(T2, T2m);
    for (int k = 0; k < 3; ++k) T3c[k] = T3[T3IDX[0][0][k]] + T3[T3IDX[1][1][k]] + T3[T3IDX[2][2][k]];
    fit_sphere<T, 3>(S0, S1, S2m, T0, T1, T2m, T3c, out + 4, out[7]);
  }
  // ---- cylinder (cylinder_fitter.py:10-28)
  if (PART < 0 || PART == 2) {
    T n[3];
    smallest_eigvec(Sxx, n);
    // compute_consistent_plane_frame (geometry_utils.py:8-27): y = normalise(n x e_i) of largest norm
    // candidates n x e_i = (0, n2, -n1), (-n2, 0, n0), (n1, -n0, 0); the first of largest norm wins.  Written without a
    // [3][3] table: indexed by the run-time `pick`, the table lives in scratch memory (160 B per lane).
    const double n0v = val(n[0]), n1v = val(n[1]), n2v = val(n[2]);
    const double nn0 = sqrt(0.0 * 0.0 + n2v * n2v + n1v * n1v), nn1 = sqrt(n2v * n2v + 0.0 * 0.0 + n0v * n0v),
                 nn2 = sqrt(n1v * n1v + n0v * n0v + 0.0 * 0.0);
    int pick = 0;
    double best = nn0;                 // (NaN norms: no candidate beats -1 in the table form either; pick stays 0)
    if (!(nn0 > -1.0)) best = -1.0;
    if (nn1 > best) { best = nn1; pick = 1; }
    if (nn2 > best) { best = nn2; pick = 2; }
    T cp[3];
    if (pick == 0) { cp[0] = T(0.0); cp[1] = n[2]; cp[2] = -n[1]; }
    else if (pick == 1) { cp[0] = -n[2]; cp[1] = T(0.0); cp[2] = n[0]; }
    else { cp[0] = n[1]; cp[1] = -n[0]; cp[2] = T(0.0); }
    T yn = tsqrt(cp[0] * cp[0] + cp[1] * cp[1] + cp[2] * cp[2]);
    yn = clamp_min(yn, 1e-12);
    T ey[3], ex[3];
    for (int i = 0; i < 3; ++i) ey[i] = cp[i] / yn;
    ex[0] = ey[1] * n[2] - ey[2] * n[1];
    ex[1] = ey[2] * n[0] - ey[0] * n[2];
    ex[2] = ey[0] * n[1] - ey[1] * n[0];
    const T *E[2] = {ex, ey};
    T S2m[3][3], T2m[3][3];
    sym_from6(S2, S2m);
    sym_from6(T2, T2m);
    T s1q[2], t1q[2], s2q[2][2], t2q[2][2], t3q[2];
    for (int a = 0; a < 2; ++a) {
      s1q[a] = E[a][0] * S1[0] + E[a][1] * S1[1] + E[a][2] * S1[2];
      t1q[a] = E[a][0] * T1[0] + E[a][1] * T1[1] + E[a][2] * T1[2];
      for (int b = 0; b < 2; ++b) {
        T s = T(0.0), t = T(0.0);
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) { s = s + E[a][i] * S2m[i][j] * E[b][j]; t = t + E[a][i] * T2m[i][j] * E[b][j]; }
        s2q[a][b] = s;
        t2q[a][b] = t;
      }
    }
    // Σ ω |q|² q_a = Σ_ijk (E Eᵀ)_ij E_a,k T3_ijk
    T EEt[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) EEt[i][j] = ex[i] * ex[j] + ey[i] * ey[j];
    for (int a = 0; a < 2; ++a) {
      T acc = T(0.0);
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
          for (int k = 0; k < 3; ++k) acc = acc + EEt[i][j] * T3[T3IDX[i][j][k]] * E[a][k];
      t3q[a] = acc;
    }
    T cc[2], r2;
    fit_sphere<T, 2>(S0, s1q, s2q, T0, t1q, t2q, t3q, cc, r2);
    for (int i = 0; i < 3; ++i) { out[8 + i] = n[i]; out[11 + i] = cc[0] * ex[i] + cc[1] * ey[i]; }
    out[14] = r2;
  }
  // ---- cone, first pass (cone_fitter.py:17-23): apex = guarded LS(Σω' x xᵀ, Σω' x (p·x)); axis = plane fit of X
  if (PART < 0 || PART == 3) {
    T A[3][3], b[3] = {Tnpn[0], Tnpn[1], Tnpn[2]};
    sym_from6(Tnn, A);
    guarded_solve3(A, b, out + 15);
    T c_unused;
    fit_plane(S0, Sx, Sxx, out + 18, c_unused);
  }
}

constexpr int PART_LO[4] = {0, 4, 8, 15}, PART_HI[4] = {4, 8, 15, 21};

// fit PART and hand its outputs to `sink(i, o_i)` with COMPILE-TIME i (an o[] indexed by a run-time part range lives in
// scratch memory: 496 B per lane, 44 MB of HBM writes per backward launch in round 2's counters)
template <typename T, int PART, typename Sink>
__device__ __forceinline__ void fit_one(const T *m, Sink &&sink) {
  T o[NO];
  fit_all<T, PART>(m, o);
#pragma unroll
  for (int i = PART_LO[PART]; i < PART_HI[PART]; ++i) sink(i, o[i]);
}
template <typename T, typename Sink>
__device__ __forceinline__ void fit_part(int part, const T *m, Sink &&sink) {   // wave-uniform `part`
  switch (part) {
    case 0: fit_one<T, 0>(m, sink); break;
    case 1: fit_one<T, 1>(m, sink); break;
    case 2: fit_one<T, 2>(m, sink); break;
    default: fit_one<T, 3>(m, sink); break;
  }
}

// 256 lanes = 4 waves: wave p fits primitive type p for 64 instances
__global__ __launch_bounds__(256) void fit_algebra_fwd_kernel(const double *__restrict__ M, long long G,
                                                              double *__restrict__ out, float *__restrict__ apex_axis32) {
  const int part = threadIdx.x >> 6;
  const long long g = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
  if (g >= G) return;
  double m[NM];
  for (int i = 0; i < NM; ++i) m[i] = M[g * NM + i];
  fit_part<double>(part, m, [&](int i, double oi) {
    out[g * NO + i] = oi;
    if (apex_axis32 && i >= 15)   // fp32 copy of the cone pass's inputs: apex[G,3] then axis[G,3]
      apex_axis32[(i >= 18 ? (G + g) * 3 + (i - 18) : g * 3 + (i - 15))] = (float)oi;
  });
}

// The chunk reduction of the moments pass and the algebra in ONE launch (the packed-parameter path): a workgroup owns
// FR_INST instances; its 256 lanes first sum the per-chunk partials of those instances' 52 moments in chunk order (the
// bits of chunk_reduce_kernel; 32 loads in flight per lane), leave M in global memory and in LDS, and then wave p, lane
// l < FR_INST fits primitive type p of instance l.  The separate reduction was a 5 us launch plus its boundary, and 64
// instances per wave made the Jacobi sweeps of all of them wait for the slowest.
constexpr int FR_INST = 4;
__global__ __launch_bounds__(256) void reduce_algebra_fwd_kernel(const double *__restrict__ partial, int nchunks, int K,
                                                                 long long G, double *__restrict__ Mout,
                                                                 double *__restrict__ out, float *__restrict__ apex_axis32) {
  __shared__ double s_M[FR_INST][NM];
  const int t = threadIdx.x;
  const long long g0 = (long long)blockIdx.x * FR_INST;
  if (t < FR_INST * NM) {
    const int l = t / NM, slot = t - l * NM;
    const long long g = g0 + l;
    if (g < G) {
      const long long b = g / K;
      const size_t per_b = (size_t)K * NM;
      const double *src = partial + (size_t)b * nchunks * per_b + (size_t)(g - b * K) * NM + slot;
      double s = 0.0;
#pragma unroll 32
      for (int c = 0; c < nchunks; ++c) s += src[(size_t)c * per_b];
      s_M[l][slot] = s;
      Mout[g * NM + slot] = s;
    }
  }
  __syncthreads();
  const int part = t >> 6, l = t & 63;
  const long long g = g0 + l;
  if (l >= FR_INST || g >= G) return;
  double m[NM];
  for (int i = 0; i < NM; ++i) m[i] = s_M[l][i];
  fit_part<double>(part, m, [&](int i, double oi) {
    out[g * NO + i] = oi;
    if (apex_axis32 && i >= 15)   // fp32 copy of the cone pass's inputs: apex[G,3] then axis[G,3]
      apex_axis32[(i >= 18 ? (G + g) * 3 + (i - 18) : g * 3 + (i - 15))] = (float)oi;
  });
}

// one workgroup per instance, 4 waves: wave p, lane d < 52 computes the part-p share of dL/dM[g,d] by forward-mode
// AD through fit p; the four shares are added in a fixed order
__global__ __launch_bounds__(256) void fit_algebra_bwd_kernel(const double *__restrict__ M,
                                                              const double *__restrict__ gout,
                                                              const double *__restrict__ gA0, long long G,
                                                              double *__restrict__ gM, float *__restrict__ gM32) {
  __shared__ double s_share[4][64];
  const long long g = blockIdx.x;
  const int part = threadIdx.x >> 6, d = threadIdx.x & 63;
  double acc = 0.0;
  if (d < NM) {
    Dual m[NM];
    for (int i = 0; i < NM; ++i) m[i] = Dual(M[g * NM + i], i == d ? 1.0 : 0.0);
    fit_part<Dual>(part, m, [&](int i, Dual oi) { acc += gout[g * NO + i] * oi.d; });
  }
  s_share[part][d] = acc;
  __syncthreads();
  if (part == 0 && d < NM) {
    double tot = ((s_share[0][d] + s_share[1][d]) + s_share[2][d]) + s_share[3][d];
    if (gA0 && d == 0) tot += gA0[g];   // direct dependence of the caller on slot 0 (Σ W), e.g. the cone half angle
    if (gM) gM[g * NM + d] = tot;
    if (gM32) gM32[g * NM + d] = (float)tot;
  }
}

// The same inside the packed-parameter backward: the workgroup of instance g first forms its own gout[21] and gA0 from the
// packed parameters' adjoint (fit_pack_bwd_kernel's rules) and the cone pass's per-chunk partials of d(apex, axis), summed
// in chunk order like chunk_reduce_strided_kernel did — two [B,K]-sized launches of the loss section's backward chain less,
// the same bits.
constexpr int FA_MAXCHUNKS = 64;
__global__ __launch_bounds__(256) void fit_params_bwd_algebra_kernel(const double *__restrict__ M,
                                                                     const float *__restrict__ gparams,
                                                                     const double *__restrict__ sums,
                                                                     const double *__restrict__ cone_ws, int nchunks, int K,
                                                                     float *__restrict__ gM32) {
  __shared__ double s_share[4][64];
  __shared__ double s_cone[FA_MAXCHUNKS][6];
  __shared__ double s_g[NO + 1];
  const long long g = blockIdx.x;
  const int b = (int)(g / K), k = (int)(g - (long long)b * K);
  const int t = threadIdx.x;
  for (int e = t; e < nchunks * 6; e += 256) {
    const int c = e / 6, j = e - c * 6;
    s_cone[c][j] = cone_ws[(((size_t)b * nchunks + c) * K + k) * 6 + j];
  }
  __syncthreads();
  if (t < NO) {
    const double gp = (double)gparams[g * 22 + t];
    double v = t < 18 ? gp : gp * cone_sign(sums[g * 2]);
    if (t >= 15) {
      double sc = 0.0;
      for (int c = 0; c < nchunks; ++c) sc += s_cone[c][t - 15];
      v = v + sc;
    }
    s_g[t] = v;
  } else if (t == NO) {
    s_g[NO] = pack_half_angle_adjoint((double)gparams[g * 22 + 21], sums[g * 2 + 1], M[g * NM]).gA0;
  }
  __syncthreads();
  const int part = t >> 6, d = t & 63;
  double acc = 0.0;
  if (d < NM) {
    Dual m[NM];
    for (int i = 0; i < NM; ++i) m[i] = Dual(M[g * NM + i], i == d ? 1.0 : 0.0);
    fit_part<Dual>(part, m, [&](int i, Dual oi) { acc += s_g[i] * oi.d; });
  }
  s_share[part][d] = acc;
  __syncthreads();
  if (part == 0 && d < NM) {
    double tot = ((s_share[0][d] + s_share[1][d]) + s_share[2][d]) + s_share[3][d];
    if (d == 0) tot += s_g[NO];
    gM32[g * NM + d] = (float)tot;
  }
}

}  // namespace

int cpfn_launch_reduce_algebra_fwd(const double *partial, int chunks, int B, int K, double *M, double *out, float *apex_axis32,
                                   hipStream_t stream) {
  const long long G = (long long)B * K;
  reduce_algebra_fwd_kernel<<<(unsigned)cpfn_cdiv(G, FR_INST), 256, 0, stream>>>(partial, chunks, K, G, M, out, apex_axis32);
  return cpfn_launch_status();
}

extern "C" int cpfn_fit_params_bwd_algebra(const double *M, const float *gparams, const double *sums, const double *cone_workspace,
                                           int chunks, int B, int K, float *gM32, void *stream) {
  if (B < 0 || K <= 0 || chunks <= 0 || chunks > FA_MAXCHUNKS || !M || !gparams || !sums || !cone_workspace || !gM32)
    return CPFN_EINVAL;
  if (B == 0) return 0;
  fit_params_bwd_algebra_kernel<<<(unsigned)B * K, 256, 0, (hipStream_t)stream>>>(M, gparams, sums, cone_workspace, chunks, K,
                                                                               gM32);
  return cpfn_launch_status();
}

extern "C" int cpfn_fit_algebra_fwd(const double *M, int64_t G, double *out, float *apex_axis32, void *stream) {
  if (G < 0 || !M || !out) return CPFN_EINVAL;
  if (G == 0) return 0;
  fit_algebra_fwd_kernel<<<cpfn_cdiv(G, 64), 256, 0, (hipStream_t)stream>>>(M, G, out, apex_axis32);
  return cpfn_launch_status();
}

extern "C" int cpfn_fit_algebra_bwd(const double *M, const double *gout, const double *gA0, int64_t G, double *gM,
                                    float *gM32, void *stream) {
  if (G < 0 || !M || !gout || (!gM && !gM32)) return CPFN_EINVAL;
  if (G == 0) return 0;
  fit_algebra_bwd_kernel<<<(unsigned)G, 256, 0, (hipStream_t)stream>>>(M, gout, gA0, G, gM, gM32);
  return cpfn_launch_status();
}
