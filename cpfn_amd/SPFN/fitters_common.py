"""The four primitive fits on one shared set of moments."""
import math
import os

import torch

from . import geometry_utils as G
from . import moments as _m
from .differentiable_tls import smallest_eigvec, sym3


# ---------------------------------------------------------------------------------------------
# The per-instance algebra ([B,K]-sized: eigenvectors, guarded solves, the cylinder frame) runs
# in ONE HIP kernel per direction (csrc/fit_algebra.hip).  `_algebra_torch` is the same algebra
# as ~400 framework ops; it is what the kernel was validated against and can be selected with
# CPFN_FIT_ALGEBRA=torch for debugging.
ALGEBRA_IMPL = os.environ.get("CPFN_FIT_ALGEBRA", "hip")


def _algebra_torch(M):
    plane_n, plane_c = plane_from_moments(M)
    sph_c, sph_r2 = sphere_from_moments(M)
    cyl_n, cyl_c, cyl_r2 = cylinder_from_moments(M)
    apex = G.guarded_solve_normal_equations(sym3(M[..., _m.BXX]), M[..., _m.BXPX])   # cone_fitter.py:17-20
    axis, _ = G.fit_plane(M[..., _m.A0], M[..., _m.AX], sym3(M[..., _m.AXX]))         # cone_fitter.py:23
    return plane_n, plane_c, sph_c, sph_r2, cyl_n, cyl_c, cyl_r2, apex, axis


def algebra(M):
    """M [B,K,52] -> (plane_n, plane_c, sphere_c, sphere_r2, cyl_axis, cyl_c, cyl_r2, cone_apex,
    cone_axis_before_sign_fix)."""
    if ALGEBRA_IMPL == "torch":
        return _algebra_torch(M)
    o = _m.FitAlgebra.apply(M)
    return (o[..., 0:3], o[..., 3], o[..., 4:7], o[..., 7], o[..., 8:11], o[..., 11:14], o[..., 14],
            o[..., 15:18], o[..., 18:21])


def fit_params(P, W, X):
    """All four fits in the packed 22-column layout (moments.FitParams): the training path."""
    return _m.FitParams.apply(P, X, W)


def moments(P, W, X=None):
    """One fused pass: M [B,K,52] float64 (X defaults to P for the fits that ignore normals)."""
    return _m.FitMoments.apply(P, P if X is None else X, W)


def plane_from_moments(M):
    """SPFN/plane_fitter.py:9-17 -> n [B,K,3], c [B,K]."""
    return G.fit_plane(M[..., _m.A0], M[..., _m.AP], sym3(M[..., _m.APP]))


def sphere_from_moments(M):
    """SPFN/sphere_fitter.py:9-19 -> centre [B,K,3], r² [B,K]."""
    T3 = G.sym3x3x3(M[..., _m.BPPP])
    T3c = T3[..., 0, 0, :] + T3[..., 1, 1, :] + T3[..., 2, 2, :]
    return G.fit_sphere(M[..., _m.A0], M[..., _m.AP], sym3(M[..., _m.APP]),
                        M[..., _m.B0], M[..., _m.BP], sym3(M[..., _m.BPP]), T3c)


def cylinder_from_moments(M):
    """SPFN/cylinder_fitter.py:10-28 -> axis, centre [B,K,3], r² [B,K].
    The 2-D circle fit of the projected points q = Eᵀp needs Σω q, Σω q qᵀ, Σω |q|² q, all of
    which are contractions of the 3-D moments with the frame E = [x_axis y_axis]."""
    n = smallest_eigvec(M[..., _m.AXX])                              # TLS on the normals (:16)
    ex, ey = G.compute_consistent_plane_frame(n)                     # (:17)
    E = torch.stack([ex, ey], dim=-1)                                # [B,K,3,2]
    Et = E.transpose(-1, -2)
    proj1 = lambda v: (Et @ v.unsqueeze(-1)).squeeze(-1)             # [.,3] -> [.,2]
    proj2 = lambda S: Et @ S @ E                                     # [.,3,3] -> [.,2,2]
    S0, S1, S2 = M[..., _m.A0], M[..., _m.AP], sym3(M[..., _m.APP])
    T0, T1, T2 = M[..., _m.B0], M[..., _m.BP], sym3(M[..., _m.BPP])
    T3 = G.sym3x3x3(M[..., _m.BPPP])
    # Σ ω |q|² q_a = Σ_ijk (E Eᵀ)_ij E_ka T3_ijk
    EEt = E @ Et
    T3c = torch.einsum("...ij,...ijk,...ka->...a", EEt, T3, E)
    cc, r2 = G.fit_sphere(S0, proj1(S1), proj2(S2), T0, proj1(T1), proj2(T2), T3c)
    centre = cc[..., 0:1] * ex + cc[..., 1:2] * ey                   # (:26)
    return n, centre, r2


def cone_from_moments(M, P, W, div_eps=1e-10, apex=None, axis=None):
    """SPFN/cone_fitter.py:12-36 -> apex, axis [B,K,3], half_angle [B,K]."""
    if apex is None:
        apex = G.guarded_solve_normal_equations(sym3(M[..., _m.BXX]), M[..., _m.BXPX])   # (:17-20)
        axis, _ = G.fit_plane(M[..., _m.A0], M[..., _m.AX], sym3(M[..., _m.AXX]))         # (:23)
    sums = _m.ConePass.apply(P, W, apex, axis)                                        # (:25-34)
    sgn = torch.sign(sums[..., 0])
    sgn = sgn + (sgn == 0).to(sgn.dtype)                                              # (:30)
    axis = axis * sgn.unsqueeze(-1)
    half = sums[..., 1] / (M[..., _m.A0] + div_eps)
    half = half.clamp(min=1e-3, max=math.pi / 2 - 1e-3)                               # (:35)
    return apex, axis, half
