// The [B,K]-sized glue between the fitters' algebra, the cone pass and the packed parameters (shared by fitters.hip and
// fit_algebra.hip): the cone axis is flipped to sgn = sign(Σ W·(axis·v̂)) with sign(0) -> +1 (cone_fitter.py:28-31) and the
// half angle is Σ W·acos / (Σ W + 1e-10) clamped to [1e-3, π/2 − 1e-3] (cone_fitter.py:33-35).
#pragma once
namespace {
constexpr double PK_LO = 1e-3, PK_HI = 1.5707963267948966 - 1e-3, PK_EPS = 1e-10;

__device__ __forceinline__ double cone_sign(double s0) { return s0 > 0.0 ? 1.0 : (s0 < 0.0 ? -1.0 : (s0 == 0.0 ? 1.0 : s0)); }

// adjoint of the half angle h = s1 / (M0 + eps) clamped: gh = dL/dh where the clamp is inactive
struct PackAdj { double g_acos, gA0; };
__device__ __forceinline__ PackAdj pack_half_angle_adjoint(double gp21, double s1, double M0) {
  const double den = M0 + PK_EPS;
  const double h = s1 / den;
  const double gh = (h >= PK_LO && h <= PK_HI) ? gp21 : 0.0;   // clamp's adjoint
  return PackAdj{gh / den, -gh * s1 / (den * den)};
}
}  // namespace
