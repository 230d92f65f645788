"""Weighted plane / sphere / circle fits on fused moments (drop-in names for
SPFN/geometry_utils.py plus the batched `fit_*` routines the four fitters share).

All per-instance algebra runs on `[B, K]` batches of fp64 moments M[B,K,52]
(cpfn_amd/SPFN/moments.py); the N-sized work is inside the HIP kernels.  Guards are the
reference's: clamp(ΣW, 1e-10) (lines 80, 212-216), condition-number cap 1e5 with the
singular values detached and ridge 1e-8 (lines 132-139).
"""
import torch

from . import moments as _m
from .differentiable_tls import cached_index, smallest_eigvec, sym3

DIV_EPS = 1e-10
COND_CAP = 1e5
RIDGE = 1e-8

# index of (i<=j<=k) in (xxx xxy xxz xyy xyz xzz yyy yyz yzz zzz)
_T3 = [[[0, 1, 2], [1, 3, 4], [2, 4, 5]], [[1, 3, 4], [3, 6, 7], [4, 7, 8]], [[2, 4, 5], [4, 7, 8], [5, 8, 9]]]


def sym3x3x3(v10):
    return v10[..., cached_index("sym3x3x3", _T3, v10.device)]


def to6(S):
    """symmetric [...,3,3] -> [...,6]."""
    return torch.stack([S[..., 0, 0], S[..., 0, 1], S[..., 0, 2], S[..., 1, 1], S[..., 1, 2], S[..., 2, 2]], -1)


def centred_scatter(S0, S1, S2, mean):
    """Σ w (p−μ)(p−μ)ᵀ from raw sums:  S2 − μ S1ᵀ − S1 μᵀ + S0 μ μᵀ  (μ need not equal S1/S0:
    the reference divides by clamp(S0, 1e-10), geometry_utils.py:80)."""
    o = lambda a, b: a.unsqueeze(-1) * b.unsqueeze(-2)
    return S2 - o(mean, S1) - o(S1, mean) + S0[..., None, None] * o(mean, mean)


def _cond_mask(AtA, cap=COND_CAP):
    """mask = s_max / s_min < 1e5 on the detached singular values (reference lines 132-134)."""
    D = AtA.shape[-1]
    A = AtA.detach()
    if D not in (2, 3):                      # (the fitters only use D = 2, 3; any other width through the library routine)
        s = torch.linalg.svdvals(A)
        return ((s[..., 0] / s[..., -1]) < cap).to(AtA.dtype)
    if D == 3:
        lam, _ = _m.eigh3(to6(A))
        s = lam.abs()
        smax, smin = s.max(-1)[0], s.min(-1)[0]
    else:  # 2x2 symmetric: closed form
        a, b, c = A[..., 0, 0], A[..., 0, 1], A[..., 1, 1]
        mid, rad = 0.5 * (a + c), torch.sqrt((0.5 * (a - c)) ** 2 + b * b)
        smax, smin = torch.maximum((mid + rad).abs(), (mid - rad).abs()), torch.minimum((mid + rad).abs(), (mid - rad).abs())
    return ((smax / smin) < cap).to(AtA.dtype)


def _solve_small(A, b):
    """Closed-form solve of [...,D,D] x = [...,D] for D in (2, 3) (adjugate / determinant)."""
    D = A.shape[-1]
    if D not in (2, 3):
        return torch.linalg.solve(A, b.unsqueeze(-1)).squeeze(-1)
    if D == 2:
        det = A[..., 0, 0] * A[..., 1, 1] - A[..., 0, 1] * A[..., 1, 0]
        x0 = (A[..., 1, 1] * b[..., 0] - A[..., 0, 1] * b[..., 1]) / det
        x1 = (A[..., 0, 0] * b[..., 1] - A[..., 1, 0] * b[..., 0]) / det
        return torch.stack([x0, x1], -1)
    c0 = torch.linalg.cross(A[..., :, 1], A[..., :, 2])
    c1 = torch.linalg.cross(A[..., :, 2], A[..., :, 0])
    c2 = torch.linalg.cross(A[..., :, 0], A[..., :, 1])
    det = (A[..., :, 0] * c0).sum(-1)
    return torch.stack([(c0 * b).sum(-1), (c1 * b).sum(-1), (c2 * b).sum(-1)], -1) / det.unsqueeze(-1)


def guarded_solve_normal_equations(AtA, Atb, condition_number_cap=COND_CAP, ls_l2_regularizer=RIDGE):
    """(AtA·mask + 1e-8 I) x = Atb·mask   (reference lines 134-140)."""
    mask = _cond_mask(AtA, condition_number_cap)
    eye = torch.eye(AtA.shape[-1], dtype=AtA.dtype, device=AtA.device)
    return _solve_small(AtA * mask[..., None, None] + ls_l2_regularizer * eye, Atb * mask[..., None])


def guarded_matrix_solve_ls(A, b, W, condition_number_cap=1e5, sqrt_eps=1e-10, ls_l2_regularizer=1e-8):
    """Weighted least squares ‖√W (A x − b)‖² with the reference's guards   (reference lines 121-142, same signature):
    A [G,N,D], b [G,N,1], W [G,N] -> x [G,D].  Row weights clamp(W, sqrt_eps) (the square of the reference's
    √clamp(W)); the D x D normal equations are accumulated in fp64 and go through `guarded_solve_normal_equations` —
    the condition-number cap on the detached singular values and the ridge — which is what the sphere / circle / apex
    fits of this package use on their fused moments.  The reference's body calls `torch.solve`, removed from PyTorch."""
    if not A.is_cuda:
        raise RuntimeError("guarded_matrix_solve_ls: CPU not supported (cpfn_amd runs on the device path only)")
    w = W.clamp(min=sqrt_eps).unsqueeze(2).double()
    Ad = A.double()
    Aw = (Ad * w).transpose(1, 2)                                  # [G,D,N]
    x = guarded_solve_normal_equations(Aw @ Ad, (Aw @ b.double()).squeeze(2), condition_number_cap, ls_l2_regularizer)
    return x.to(A.dtype)


def fit_plane(S0, S1, S2):
    """weighted_plane_fitting (reference lines 74-84) from Σw, Σw p, Σw p pᵀ."""
    mean = S1 / S0.clamp(min=DIV_EPS).unsqueeze(-1)
    n = smallest_eigvec(to6(centred_scatter(S0, S1, S2, mean)))
    return n, (n * mean).sum(-1)


def fit_sphere(S0, S1, S2, T0, T1, T2, T3c):
    """weighted_sphere_fitting (reference lines 209-223) in any dimension D.
    S* are w-weighted sums (1, p, p pᵀ); T* are clamp(w)-weighted sums (1, p, p pᵀ, |p|² p)."""
    den = S0.clamp(min=DIV_EPS)
    mean = S1 / den.unsqueeze(-1)
    m2 = torch.diagonal(S2, dim1=-2, dim2=-1).sum(-1) / den
    o = lambda a, b: a.unsqueeze(-1) * b.unsqueeze(-2)
    # A = 2(μ − p), b = m2 − |p|²   ->   AtA = Σ w' A Aᵀ,  Atb = Σ w' A b
    AtA = 4.0 * (T0[..., None, None] * o(mean, mean) - o(mean, T1) - o(T1, mean) + T2)
    trT2 = torch.diagonal(T2, dim1=-2, dim2=-1).sum(-1)
    Atb = 2.0 * (mean * (m2 * T0 - trT2).unsqueeze(-1) - m2.unsqueeze(-1) * T1 + T3c)
    centre = guarded_solve_normal_equations(AtA, Atb)
    trS2 = torch.diagonal(S2, dim1=-2, dim2=-1).sum(-1)
    r2 = (trS2 - 2.0 * (centre * S1).sum(-1) + (centre * centre).sum(-1) * S0) / den
    return centre, r2


def compute_consistent_plane_frame(normal):
    """normal [...,3] -> (x_axis, y_axis): y = normalised n×e_i with the largest norm
    (first on ties), x = y×n   (reference lines 8-27)."""
    eye = torch.eye(3, dtype=normal.dtype, device=normal.device)
    cands = torch.stack([torch.linalg.cross(normal, eye[i].expand_as(normal)) for i in range(3)], 0)
    pick = cands.detach().norm(dim=-1).argmax(dim=0)
    y = torch.gather(cands, 0, pick[None, ..., None].expand(1, *normal.shape)).squeeze(0)
    y = torch.nn.functional.normalize(y, p=2, dim=-1, eps=1e-12)
    return torch.linalg.cross(y, normal), y


# ------------------------------------------------------------------ reference-shaped API
def weighted_plane_fitting(P, W, division_eps=1e-10):
    """P [G,N,3], W [G,N] -> n [G,3], c [G]   (reference lines 74-84)."""
    M = _m.FitMoments.apply(P, P, W.unsqueeze(2))[:, 0]
    # use the x-slots so the fit is differentiable in P when P is itself a network output
    n, c = fit_plane(M[..., _m.A0], M[..., _m.AX], sym3(M[..., _m.AXX]))
    return n.to(P.dtype), c.to(P.dtype)


def weighted_sphere_fitting(P, W, division_eps=1e-10):
    """P [G,N,3], W [G,N] -> centre [G,3], r² [G]   (reference lines 209-223; 3-D only here —
    the cylinder's 2-D circle fit goes through cylinder_fitter on projected moments)."""
    M = _m.FitMoments.apply(P, P, W.unsqueeze(2))[:, 0]
    T3 = sym3x3x3(M[..., _m.BPPP])
    T3c = T3[..., 0, 0, :] + T3[..., 1, 1, :] + T3[..., 2, 2, :]
    c, r2 = fit_sphere(M[..., _m.A0], M[..., _m.AP], sym3(M[..., _m.APP]),
                       M[..., _m.B0], M[..., _m.BP], sym3(M[..., _m.BPP]), T3c)
    return c.to(P.dtype), r2.to(P.dtype)


# Names the device path does not define (host-side GT parsing / JSON export, the TensorFlow twins) come from the
# reference's own SPFN/geometry_utils.py, found on sys.path (_reference.py): nothing of it is restated here.
from . import _reference as _ref  # noqa: E402

__getattr__ = _ref.module_fallback("geometry_utils")
