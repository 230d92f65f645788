"""Shared comparison helpers for the parity tests."""
import torch

PARAM_KEYS = ["plane_normal", "plane_center", "sphere_center", "sphere_radius_squared",
              "cylinder_axis", "cylinder_center", "cylinder_radius_squared",
              "cone_apex", "cone_axis", "cone_half_angle"]


def rel_err(a, b):
    """max|a-b| / max|b| — the "1e-4 rel" of BASELINE.json's north_star, per tensor."""
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


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
