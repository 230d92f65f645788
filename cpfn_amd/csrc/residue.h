// Residue of a point against one fitted primitive — value only, no tangents — shared by the P-coverage kernel
// (losses.hip) and the evaluation tail (metrics.hip).  Formulas: SPFN/{plane,sphere,cylinder,cone}_fitter.compute_residue_single
// (plane_fitter.py:54-55, sphere_fitter.py:58-62, cylinder_fitter.py:82-89, cone_fitter.py:98-103).
// q = the primitive's slice of the packed 22-column parameter row (cpfn_fit_pack_fwd): plane (n, c), sphere (centre, r^2),
// cylinder (axis, centre, r^2), cone (apex, axis, half angle).
#pragma once
#include <hip/hip_runtime.h>

static __device__ inline float sqrt_safe_f(float x) { return sqrtf(fabsf(x) + 1e-10f); }   // metric_implementation.py:65-66
static __device__ inline float residue_value(int kind, const float *q, float px, float py, float pz) {
  if (kind == 0) {
    const float e = px * q[0] + py * q[1] + pz * q[2] - q[3];
    return e * e;
  } else if (kind == 1) {
    const float dx = px - q[0], dy = py - q[1], dz = pz - q[2];
    const float e = sqrt_safe_f(dx * dx + dy * dy + dz * dz) - sqrt_safe_f(q[3]);
    return e * e;
  } else if (kind == 2) {
    const float dx = px - q[3], dy = py - q[4], dz = pz - q[5];
    const float al = dx * q[0] + dy * q[1] + dz * q[2];
    const float e = sqrt_safe_f(dx * dx + dy * dy + dz * dz - al * al) - sqrt_safe_f(q[6]);
    return e * e;
  } else {
    const float vx = px - q[0], vy = py - q[1], vz = pz - q[2];
    const float n2 = vx * vx + vy * vy + vz * vz;
    const float inv = 1.f / fmaxf(sqrtf(n2), 1e-12f);
    float c = (vx * q[3] + vy * q[4] + vz * q[5]) * inv;
    const float lim = 1.0f - 1e-6f;
    c = fminf(fmaxf(c, -lim), lim);
    const float ad = fabsf(acosf(c) - q[6]);
    const float sn = sinf(fminf(ad, 1.57079632679f));
    return sn * sn * n2;
  }
}

