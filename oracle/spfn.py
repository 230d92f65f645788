"""torch-CPU restatement of the SPFN primitive fitters and losses.

TEST INFRASTRUCTURE ONLY — see ``oracle/__init__.py``.  Pinned against the
reference by ``tests/golden/fitters_*.npz`` and ``tests/golden/step_*.npz``.

The reference tiles P/X to ``[B*K, N, 3]`` copies and calls batched helpers
(SPFN/plane_fitter.py:9-17 etc.); this restatement keeps an explicit instance
axis ``[B, K, ...]`` and broadcasts instead, but performs the same fp32
arithmetic family (centred second moments, sqrt-weighted normal equations,
cond-number guard, ridge), so results agree to fp32 rounding.  ``dtype`` may be
set to float64 to obtain a high-precision arbiter for ill-conditioned cases.
"""
import math

import numpy as np
import torch

# --------------------------------------------------------------------------- TLS


def _guarded_gap_inverse(s):
    """K of SPFN/differentiable_tls.py:45-53 with guard_one_over_matrix (:8-17):
    K[i,j] = 1/(s_i² − s_j²) off the diagonal, where the difference is pushed to
    >= +1e-10 above the diagonal and <= −1e-10 below it; 0 on the diagonal."""
    s2 = s * s
    diff = s2.unsqueeze(-1) - s2.unsqueeze(-2)
    n = s.shape[-1]
    iu = torch.triu(torch.ones(n, n, dtype=torch.bool), diagonal=1)
    il = torch.tril(torch.ones(n, n, dtype=torch.bool), diagonal=-1)
    out = torch.zeros_like(diff)
    out = torch.where(iu, 1.0 / diff.clamp(min=1e-10), out)
    out = torch.where(il, 1.0 / diff.clamp(max=-1e-10), out)
    return out


class _SmallestRightSingularVector(torch.autograd.Function):
    """Custom_svd_v_colum (SPFN/differentiable_tls.py:123-143), col_index = -1."""

    @staticmethod
    def forward(ctx, M):
        U, S, Vh = torch.linalg.svd(M)
        V = Vh.transpose(-1, -2)
        ctx.save_for_backward(U, S, V)
        return V[..., -1]

    @staticmethod
    def backward(ctx, g):
        U, S, V = ctx.saved_tensors
        gV = torch.zeros_like(V)
        gV[..., -1] = g
        K = _guarded_gap_inverse(S)
        inner = K.transpose(-1, -2) * (V.transpose(-1, -2) @ gV)        # :137
        inner = 0.5 * (inner + inner.transpose(-1, -2))                  # :138
        return U @ ((2.0 * S.unsqueeze(-1) * inner) @ V.transpose(-1, -2))  # :140


def solve_weighted_tls(A, W):
    """SPFN/differentiable_tls.py:200-209.  A [...,N,3], W [...,N] -> unit x [...,3]
    minimising Σ w (a·x)²; sign arbitrary."""
    M = torch.einsum("...n,...ni,...nj->...ij", W, A, A)
    return _SmallestRightSingularVector.apply(M)


# --------------------------------------------------------------------------- helpers


def _instance_weights(W):
    """W [B,N,K] -> [B,K,N] (the reference's W.transpose(1,2), plane_fitter.py:12)."""
    return W.transpose(1, 2)


def weighted_plane_fitting(P, Wk, division_eps=1e-10):
    """SPFN/geometry_utils.py:74-84.  P [B,N,3] (or [B,K,N,3]), Wk [B,K,N]."""
    if P.dim() == 3:
        P = P.unsqueeze(1)
    wsum = Wk.sum(-1, keepdim=True)                                        # [B,K,1]
    mean = (Wk.unsqueeze(-1) * P).sum(-2) / wsum.clamp(min=division_eps)   # [B,K,3]
    A = P - mean.unsqueeze(-2)                                             # [B,K,N,3]
    n = solve_weighted_tls(A, Wk)
    c = (n * mean).sum(-1)
    return n, c


def guarded_matrix_solve_ls(A, b, Wk, condition_number_cap=1e5, sqrt_eps=1e-10,
                            ls_l2_regularizer=1e-8):
    """SPFN/geometry_utils.py:121-142.  A [B,K,N,D], b [B,K,N], Wk [B,K,N] -> x [B,K,D]."""
    D = A.shape[-1]
    sw = torch.sqrt(Wk.clamp(min=sqrt_eps)).unsqueeze(-1)                  # :127
    As = A * sw
    bs = (b.unsqueeze(-1) * sw)
    AtA = As.transpose(-1, -2) @ As                                        # :131
    s = torch.linalg.svdvals(AtA.detach())                                 # :132-133
    mask = (s[..., 0] / s[..., -1] < condition_number_cap).to(A.dtype)     # :134
    mask = mask.unsqueeze(-1).unsqueeze(-1)
    eye = torch.eye(D, dtype=A.dtype)
    lhs = AtA * mask + ls_l2_regularizer * eye                             # :138
    rhs = (As.transpose(-1, -2) * mask) @ bs                               # :139
    return torch.linalg.solve(lhs, rhs).squeeze(-1)                        # :140


def weighted_sphere_fitting(P, Wk, division_eps=1e-10):
    """SPFN/geometry_utils.py:209-223.  P [B,N,D] or [B,K,N,D]; Wk [B,K,N]."""
    if P.dim() == 3:
        P = P.unsqueeze(1)
    wsum = Wk.sum(-1)                                                      # [B,K]
    den = wsum.clamp(min=division_eps)
    psq = (P * P).sum(-1)                                                  # [B,1|K,N]
    b = ((Wk * psq).sum(-1) / den).unsqueeze(-1) - psq                     # [B,K,N]
    mean = (Wk.unsqueeze(-1) * P).sum(-2) / den.unsqueeze(-1)              # [B,K,D]
    A = 2.0 * (mean.unsqueeze(-2) - P)                                     # [B,K,N,D]
    centre = guarded_matrix_solve_ls(A, b, Wk)
    diff = P - centre.unsqueeze(-2)
    r2 = (Wk * (diff * diff).sum(-1)).sum(-1) / den
    return centre, r2


def compute_consistent_plane_frame(n):
    """SPFN/geometry_utils.py:8-27.  n [...,3] -> (x_axis, y_axis)."""
    eye = torch.eye(3, dtype=n.dtype)
    cands = torch.stack([torch.linalg.cross(n, eye[i].expand_as(n)) for i in range(3)], 0)
    pick = cands.norm(dim=-1).argmax(dim=0)                                # first max on ties
    y = torch.gather(cands, 0, pick.unsqueeze(0).unsqueeze(-1).expand(1, *n.shape)).squeeze(0)
    y = torch.nn.functional.normalize(y, p=2, dim=-1, eps=1e-12)
    x = torch.linalg.cross(y, n)
    return x, y


def acos_safe(x):
    """SPFN/cone_fitter.py:9-10."""
    return torch.acos(x.clamp(min=-1.0 + 1e-6, max=1.0 - 1e-6))


def sqrt_safe(x):
    """SPFN/sphere_fitter.py:58-59."""
    return torch.sqrt(x.abs() + 1e-10)


# --------------------------------------------------------------------------- fitters


def plane_parameters(P, W):
    """SPFN/plane_fitter.py:9-17 -> n [B,K,3], c [B,K]."""
    return weighted_plane_fitting(P, _instance_weights(W))


def sphere_parameters(P, W):
    """SPFN/sphere_fitter.py:9-19 -> centre [B,K,3], r² [B,K]."""
    return weighted_sphere_fitting(P, _instance_weights(W))


def cylinder_parameters(P, W, X):
    """SPFN/cylinder_fitter.py:10-28 -> axis [B,K,3], centre [B,K,3], r² [B,K]."""
    Wk = _instance_weights(W)
    n = solve_weighted_tls(X.unsqueeze(1), Wk)                             # :16
    x_ax, y_ax = compute_consistent_plane_frame(n)                         # :17
    xc = (P.unsqueeze(1) * x_ax.unsqueeze(2)).sum(-1)                      # :20  [B,K,N]
    yc = (P.unsqueeze(1) * y_ax.unsqueeze(2)).sum(-1)                      # :21
    cc, r2 = weighted_sphere_fitting(torch.stack([xc, yc], -1), Wk)        # :22-24
    centre = cc[..., 0:1] * x_ax + cc[..., 1:2] * y_ax                     # :26
    return n, centre, r2


def cone_parameters(P, W, X, div_eps=1e-10):
    """SPFN/cone_fitter.py:12-36 -> apex [B,K,3], axis [B,K,3], half_angle [B,K]."""
    Wk = _instance_weights(W)
    B, K, N = Wk.shape
    A = X.unsqueeze(1).expand(B, K, N, 3)
    b = (P * X).sum(-1).unsqueeze(1).expand(B, K, N)                       # :19
    apex = guarded_matrix_solve_ls(A, b, Wk)                               # :20
    axis, _ = weighted_plane_fitting(X, Wk)                                # :23
    v = P.unsqueeze(2) - apex.unsqueeze(1)                                 # :25 [B,N,K,3]
    vn = torch.nn.functional.normalize(v, p=2, dim=3, eps=1e-12)
    cosang = (axis.unsqueeze(1) * vn).sum(-1)                              # :27 [B,N,K]
    sgn = torch.sign((W * cosang).sum(1))                                  # :29
    sgn = sgn + (sgn == 0.0).to(sgn.dtype)                                 # :30
    axis = axis * sgn.unsqueeze(-1)
    half = (W * acos_safe(cosang.abs())).sum(1) / (W.sum(1) + div_eps)     # :32-34
    half = half.clamp(min=1e-3, max=math.pi / 2 - 1e-3)                    # :35
    return apex, axis, half


def compute_parameters(P, W, X, classes=("plane", "sphere", "cylinder", "cone")):
    """SPFN/losses_implementation.py:255-278 (same dict keys)."""
    out = {}
    for c in classes:
        if c == "plane":
            out["plane_normal"], out["plane_center"] = plane_parameters(P, W)
        elif c == "sphere":
            out["sphere_center"], out["sphere_radius_squared"] = sphere_parameters(P, W)
        elif c == "cylinder":
            (out["cylinder_axis"], out["cylinder_center"],
             out["cylinder_radius_squared"]) = cylinder_parameters(P, W, X)
        elif c == "cone":
            out["cone_apex"], out["cone_axis"], out["cone_half_angle"] = cone_parameters(P, W, X)
        else:
            raise NotImplementedError(c)
    return out


# --------------------------------------------------------------------------- residues


def plane_residue(n, c, p):
    """SPFN/plane_fitter.py:54-55."""
    return ((p * n).sum(-1) - c) ** 2


def sphere_residue(centre, r2, p):
    """SPFN/sphere_fitter.py:61-62."""
    return (sqrt_safe(((p - centre) ** 2).sum(-1)) - sqrt_safe(r2)) ** 2


def cylinder_residue(axis, centre, r2, p):
    """SPFN/cylinder_fitter.py:85-89."""
    d = p - centre
    return (sqrt_safe((d * d).sum(-1) - ((d * axis).sum(-1)) ** 2) - sqrt_safe(r2)) ** 2


def cone_residue(apex, axis, half, p):
    """SPFN/cone_fitter.py:98-103."""
    v = p - apex
    vn = torch.nn.functional.normalize(v, p=2, dim=-1, eps=1e-12)
    alpha = acos_safe((vn * axis).sum(-1))
    return torch.sin((alpha - half).abs().clamp(max=math.pi / 2)) ** 2 * (v * v).sum(-1)


# --------------------------------------------------------------------------- losses


def hungarian_matching(W, I_gt):
    """SPFN/losses_implementation.py:11-30 (relaxed-IoU cost, SciPy assignment)."""
    from scipy.optimize import linear_sum_assignment
    B, N, K = W.shape
    match = torch.zeros(B, K, dtype=torch.long)
    for b in range(B):
        n_gt = int(I_gt[b].max()) + 1
        onehot = torch.eye(n_gt + 1, dtype=W.dtype)[I_gt[b]]                # [N, n_gt+1]
        dot = onehot.t() @ W[b]
        den = onehot.sum(0).unsqueeze(1) + W[b].sum(0).unsqueeze(0) - dot
        cost = (dot / den.clamp(min=1e-10))[:n_gt]
        _, col = linear_sum_assignment(-cost.detach().numpy())
        match[b, :n_gt] = torch.from_numpy(col).long()
    return match


def _gather_k(t, match):
    """t [B,K,...] -> t[b, match[b,k], ...]."""
    idx = match
    while idx.dim() < t.dim():
        idx = idx.unsqueeze(-1)
    return torch.gather(t, 1, idx.expand(*match.shape, *t.shape[2:]))


def miou_loss(W, I_gt, match, div_eps=1e-10):
    """SPFN/losses_implementation.py:77-90."""
    B, N, K = W.shape
    Wr = torch.gather(W, 2, match.unsqueeze(1).expand(B, N, K))
    Wgt = torch.eye(K + 2, dtype=W.dtype)[I_gt][:, :, :K]
    dot = (Wgt * Wr).sum(1)
    den = Wgt.sum(1) + Wr.sum(1) - dot
    return 1.0 - dot / (den + div_eps)


def normal_loss(X, X_gt):
    """SPFN/losses_implementation.py:152-159 (training form)."""
    return (1.0 - (X * X_gt).sum(-1).abs()).mean(1)


def type_loss(T, I_gt, T_gt):
    """SPFN/losses_implementation.py:195-210 (training form)."""
    B, N = I_gt.shape
    tgt = torch.gather(T_gt, 1, I_gt.clamp(min=0))
    ce = torch.nn.functional.cross_entropy(T.reshape(B * N, -1), tgt.reshape(B * N),
                                           reduction="none").view(B, N)
    ce = torch.where(I_gt == -1, torch.zeros_like(ce), ce)
    return ce.sum(1) / (I_gt != -1).to(ce.dtype).sum(1)


def residue_loss(params, match, ppi, T_gt, classes):
    """SPFN/losses_implementation.py:351-387.  ppi [B,K,N',3]."""
    per_class = []
    for c in classes:
        g = lambda key: _gather_k(params[key], match).unsqueeze(2)
        if c == "plane":
            r = plane_residue(g("plane_normal"), g("plane_center"), ppi)
        elif c == "sphere":
            r = sphere_residue(g("sphere_center"), g("sphere_radius_squared"), ppi)
        elif c == "cylinder":
            r = cylinder_residue(g("cylinder_axis"), g("cylinder_center"),
                                 g("cylinder_radius_squared"), ppi)
        elif c == "cone":
            r = cone_residue(g("cone_apex"), g("cone_axis"), g("cone_half_angle"), ppi)
        per_class.append(r.mean(2))
    stacked = torch.stack(per_class, 2)                                   # [B,K,T]
    return torch.gather(stacked, 2, T_gt.unsqueeze(2)).squeeze(2)


def parameter_loss(params, gt, match, T_gt, classes):
    """SPFN/losses_implementation.py:480-497 (training form: 1 − |cos|)."""
    B, K = match.shape
    per_class = []
    for c in classes:
        key = {"plane": "plane_normal", "cylinder": "cylinder_axis", "cone": "cone_axis"}.get(c)
        if key is None:
            per_class.append(torch.zeros(B, K, dtype=gt["plane_normal"].dtype))
        else:
            pred = _gather_k(params[key], match)
            per_class.append(1.0 - (pred * gt[key]).sum(-1).abs())
    stacked = torch.stack(per_class, 2)
    return torch.gather(stacked, 2, T_gt.unsqueeze(2)).squeeze(2)


def masked_instance_mean(loss, I_gt):
    """get_mask_gt + reduce_mean_masked_instance, losses_implementation.py:603-638."""
    K = loss.shape[1]
    n_inst = I_gt.max(dim=1)[0] + 1
    mask = torch.arange(K).unsqueeze(0) < n_inst.unsqueeze(1)
    s = torch.where(mask, loss, torch.zeros_like(loss)).sum(1)
    den = mask.to(loss.dtype).sum(1)
    return torch.where(den > 0, s / den, torch.zeros_like(s))


def compute_all_losses(P, W, I_gt, X, X_gt, T, T_gt, gt_parameters, points_per_instance,
                       classes=("sphere", "plane", "cylinder", "cone"), multipliers=None,
                       match=None):
    """SPFN/losses_implementation.py:675-720 with the GlobalSPFN config
    (all multipliers 1.0, Configs/config_globalSPFN.yml:7-12; is_eval=False,
    mode_seg='mIoU').  Returns (total, normal, type, miou, residue, parameter, params)."""
    m = dict(normal=1.0, type=1.0, miou=1.0, residue=1.0, parameter=1.0, total=1.0)
    if multipliers:
        m.update(multipliers)
    if match is None:
        match = hungarian_matching(W, I_gt)
    params = None
    if m["residue"] > 0 or m["parameter"] > 0:
        params = compute_parameters(P, W, X, classes=classes)
    zero_b = torch.zeros(P.shape[0], dtype=P.dtype)
    l_normal = normal_loss(X, X_gt).mean() if m["normal"] > 0 else zero_b.mean()
    l_type = type_loss(T, I_gt, T_gt).mean() if m["type"] > 0 else zero_b.mean()
    l_miou = masked_instance_mean(miou_loss(W, I_gt, match), I_gt).mean() if m["miou"] > 0 else zero_b.mean()
    l_res = (masked_instance_mean(residue_loss(params, match, points_per_instance, T_gt, classes), I_gt).mean()
             if m["residue"] > 0 else zero_b.mean())
    l_par = (masked_instance_mean(parameter_loss(params, gt_parameters, match, T_gt, classes), I_gt).mean()
             if m["parameter"] > 0 else zero_b.mean())
    total = 0
    for key, val in (("normal", l_normal), ("type", l_type), ("miou", l_miou),
                     ("residue", l_res), ("parameter", l_par)):
        if m[key] > 0:
            total = total + m[key] * val
    total = total * m["total"]
    return total, l_normal, l_type, l_miou, l_res, l_par, params
