"""Shared comparison helpers for the parity tests."""
import torch

PARAM_KEYS = ["plane_normal", "plane_center", "sphere_center", "sphere_radius_squared",
              "cylinder_axis", "cylinder_center", "cylinder_radius_squared",
              "cone_apex", "cone_axis", "cone_half_angle"]


def rel_err(a, b):
    """max|a-b| / max|b| — the "1e-4 rel" of BASELINE.json's north_star, per tensor."""
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def per_instance_rel(a, b, floor=1e-3):
    """Per-instance relative error ‖a_bk − b_bk‖ / max(‖b_bk‖, floor) -> [B,K] (vector parameters: Euclidean norm
    over the last axis; scalar parameters [B,K]: absolute value).  Unlike `rel_err`, one large-magnitude instance
    cannot hide a wrong small one.  `floor` keeps instances whose true value is ~0 (a plane through the origin) from
    dividing by nothing: below it the error is measured in absolute units of `floor`."""
    a, b = a.double(), b.double()
    if a.dim() == 2:
        a, b = a.unsqueeze(-1), b.unsqueeze(-1)
    return (a - b).norm(dim=-1) / b.norm(dim=-1).clamp_min(floor)


def plane_eigen_gap(P, W):
    """(λ1 − λ0) / λ2 of every instance's weighted covariance (fp64, CPU): how well the TLS normal is determined.
    Printed next to a failing instance: the reference's own fp32 result moves by ~1e-7 / gap."""
    P, W = P.double().cpu(), W.double().cpu()
    Ws = W.sum(1).clamp_min(1e-10)                                     # [B,K]
    mu = torch.einsum("bnk,bnc->bkc", W, P) / Ws.unsqueeze(-1)
    d = P.unsqueeze(1) - mu.unsqueeze(2)                               # [B,K,N,3]
    C = torch.einsum("bnk,bknc,bknd->bkcd", W, d, d)
    ev = torch.linalg.eigvalsh(C)
    return (ev[..., 1] - ev[..., 0]) / ev[..., 2].clamp_min(1e-300)


def align_signs(mine, ref):
    """Plane normal / cylinder axis come out of an SVD and are defined up to a sign per
    instance (the plane offset flips with its normal).  Flip `mine` onto `ref`."""
    out = dict(mine)
    s = torch.sign((mine["plane_normal"] * ref["plane_normal"]).sum(-1, keepdim=True))
    s = torch.where(s == 0, torch.ones_like(s), s)
    out["plane_normal"] = mine["plane_normal"] * s
    out["plane_center"] = mine["plane_center"] * s.squeeze(-1)
    s = torch.sign((mine["cylinder_axis"] * ref["cylinder_axis"]).sum(-1, keepdim=True))
    s = torch.where(s == 0, torch.ones_like(s), s)
    out["cylinder_axis"] = mine["cylinder_axis"] * s
    return out


def sign_invariant_loss(params, coef):
    """The scalar the fitter fixtures differentiate (tests/golden/make_golden_spfn.py):
    linear in every sign-determined output, and in n⊗n / c·n for the sign-ambiguous ones."""
    L = 0
    for key in PARAM_KEYS:
        v = params[key]
        if key == "plane_normal":
            L = L + (coef["plane_normal_outer"] * (v.unsqueeze(-1) * v.unsqueeze(-2))).sum()
            L = L + (coef["plane_cn"] * (params["plane_center"].unsqueeze(-1) * v)).sum()
        elif key == "plane_center":
            continue
        elif key == "cylinder_axis":
            L = L + (coef["cylinder_axis_outer"] * (v.unsqueeze(-1) * v.unsqueeze(-2))).sum()
        else:
            L = L + (coef[key] * v).sum()
    return L
