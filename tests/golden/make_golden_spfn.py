"""Fitter / network / training-step fixtures (see make_golden.py for the harness)."""
import os
import sys

import numpy as np
import torch

from make_golden import save

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

PARAM_KEYS = ["plane_normal", "plane_center", "sphere_center", "sphere_radius_squared",
              "cylinder_axis", "cylinder_center", "cylinder_radius_squared",
              "cone_apex", "cone_axis", "cone_half_angle"]
# parameters defined up to a global sign per instance (SVD sign ambiguity); the
# fixture loss uses sign-invariant functions of them so gradients are comparable
SIGNED = {"plane_normal": "plane_center", "cylinder_axis": None}


def sign_invariant_loss(params, coef):
    """L = Σ_key <coef_key, f(param_key)> with f = identity for sign-determined
    outputs and f(n, c) = (n⊗n, c·n) for the sign-ambiguous ones."""
    L = 0
    for key in PARAM_KEYS:
        v = params[key]
        if key == "plane_normal":
            outer = v.unsqueeze(-1) * v.unsqueeze(-2)
            L = L + (coef["plane_normal_outer"] * outer).sum()
            L = L + (coef["plane_cn"] * (params["plane_center"].unsqueeze(-1) * v)).sum()
        elif key == "plane_center":
            continue
        elif key == "cylinder_axis":
            outer = v.unsqueeze(-1) * v.unsqueeze(-2)
            L = L + (coef["cylinder_axis_outer"] * outer).sum()
        else:
            L = L + (coef[key] * v).sum()
    return L


def make_coef(B, K, g):
    coef = {}
    for key in PARAM_KEYS:
        shape = (B, K) if key in ("plane_center", "sphere_radius_squared", "cylinder_radius_squared",
                                  "cone_half_angle") else (B, K, 3)
        coef[key] = torch.randn(shape, generator=g)
    coef["plane_normal_outer"] = torch.randn(B, K, 3, 3, generator=g)
    coef["cylinder_axis_outer"] = torch.randn(B, K, 3, 3, generator=g)
    coef["plane_cn"] = torch.randn(B, K, 3, generator=g)
    return coef


def run_reference_fitters(losses, P, W, X, coef):
    W = W.clone().requires_grad_(True)
    X = X.clone().requires_grad_(True)
    params = losses.compute_parameters(P, W, X)
    L = sign_invariant_loss(params, coef)
    L.backward()
    return {k: params[k].detach() for k in PARAM_KEYS}, W.grad, X.grad, L.detach()


def make_fitters(losses):
    from cpfn_amd import synthetic
    from oracle import spfn as ospfn

    # (1) the reference's own self-test recipe (SPFN/plane_fitter.py:30-39,
    #     cylinder_fitter.py:51-63, cone_fitter.py:67-79) with batch reduced 100 -> 4
    np.random.seed(0)
    B, N, K = 4, 1024, 12
    P = torch.from_numpy(np.random.randn(B, N, 3)).float()
    W = torch.from_numpy(np.random.rand(B, N, K)).float()
    X = torch.from_numpy(np.random.randn(B, N, 3)).float()
    X = torch.nn.functional.normalize(X, p=2, dim=2, eps=1e-12)
    g = torch.Generator().manual_seed(3)
    coef = make_coef(B, K, g)
    params, gW, gX, L = run_reference_fitters(losses, P, W, X, coef)
    arrays = dict(P=P.numpy(), W=W.numpy(), X=X.numpy(), gW=gW.numpy(), gX=gX.numpy(), L=L.numpy())
    arrays.update({"out_" + k: v.numpy() for k, v in params.items()})
    arrays.update({"coef_" + k: v.numpy() for k, v in coef.items()})
    save("fitters_selftest.npz", **arrays)
    _report(ospfn, P, W, X, coef, params, gW, gX, "selftest")

    # (2) points that really lie on primitives with a peaky soft membership
    #     (well-conditioned instances + near-empty instances that trigger the guards)
    d = synthetic.primitive_cloud(2, 2048, n_prims=6, noise=0.002, seed=21)
    P2, X2 = d["P"], torch.nn.functional.normalize(d["X_gt"] + 0.02 * torch.randn(2, 2048, 3, generator=g), dim=2)
    K2 = 10
    logits = torch.randn(2, 2048, K2, generator=g) * 0.5
    logits.scatter_add_(2, d["I_gt"].unsqueeze(2), torch.full((2, 2048, 1), 6.0))
    W2 = torch.softmax(logits, dim=2)
    coef2 = make_coef(2, K2, g)
    params2, gW2, gX2, L2 = run_reference_fitters(losses, P2, W2, X2, coef2)
    arrays = dict(P=P2.numpy(), W=W2.numpy(), X=X2.numpy(), gW=gW2.numpy(), gX=gX2.numpy(), L=L2.numpy(),
                  I_gt=d["I_gt"].numpy().astype(np.int16), T_gt=d["T_gt"].numpy().astype(np.int16))
    arrays.update({"out_" + k: v.numpy() for k, v in params2.items()})
    arrays.update({"coef_" + k: v.numpy() for k, v in coef2.items()})
    save("fitters_primitives.npz", **arrays)
    _report(ospfn, P2, W2, X2, coef2, params2, gW2, gX2, "primitives")


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def _report(ospfn, P, W, X, coef, params, gW, gX, tag):
    Wq = W.clone().requires_grad_(True)
    Xq = X.clone().requires_grad_(True)
    mine = ospfn.compute_parameters(P, Wq, Xq)
    for k in PARAM_KEYS:
        a, b = mine[k].detach(), params[k]
        if k in ("plane_normal", "cylinder_axis"):
            s = torch.sign((a * b).sum(-1, keepdim=True))
            a = a * s
        if k == "plane_center":
            s = torch.sign((mine["plane_normal"].detach() * params["plane_normal"]).sum(-1))
            a = a * s
        print("  [%s] %-26s max-rel %.2e" % (tag, k, _rel(a, b)))
    sign_invariant_loss(mine, coef).backward()
    print("  [%s] dL/dW rel %.2e  dL/dX rel %.2e" % (tag, _rel(Wq.grad, gW), _rel(Xq.grad, gX)))


# --------------------------------------------------------------------------- network
def _load_reference_model(pn2_network, state):
    m = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 28])
    m.load_state_dict(state, strict=True)
    m.train()
    return m


class _no_dropout:
    def __enter__(self):
        self._orig = torch.nn.functional.dropout
        torch.nn.functional.dropout = lambda x, p=0.5, training=True, inplace=False: x

    def __exit__(self, *a):
        torch.nn.functional.dropout = self._orig


def _ref_forward(model, P, seed):
    """Reference forward on CPU (fast=False) and the FPS starts it drew."""
    B, N, _ = P.shape
    torch.manual_seed(seed)
    s1 = torch.randint(0, N, (B,), dtype=torch.long)
    s2 = torch.randint(0, 512, (B,), dtype=torch.long)
    torch.manual_seed(seed)
    with _no_dropout():
        out = model(P, fast=False)
    return out, (s1, s2)


def make_network(pn2_network, pn2_geo):
    from cpfn_amd import synthetic
    from oracle import pn2 as opn2

    state = synthetic.synthetic_state_dict(synthetic.pointnet2_state_shapes(), seed=0)
    model = _load_reference_model(pn2_network, state)
    P = synthetic.primitive_cloud(2, 2048, n_prims=6, seed=31)["P"]
    (X, T, W, l3, feat), starts = _ref_forward(model, P, seed=41)
    sub = np.arange(0, 2048, 8)
    save("network_2x2048.npz", P=P.numpy(), fps_start1=starts[0].numpy(), fps_start2=starts[1].numpy(),
         X=X.detach().numpy(), T=T.detach().numpy(), W=W.detach().numpy(),
         l3=l3.detach().numpy()[:, :, 0], feat_sub=feat.detach().numpy()[:, :, sub], sub=sub)
    with torch.no_grad():
        heads, ol3, ofeat, _ = opn2.pointnet2_forward(state, P, starts, training=True)
    for name, a, b in (("X", heads[0], X), ("T", heads[1], T), ("W", heads[2], W), ("l3", ol3, l3), ("feat", ofeat, feat)):
        print("  [network] %-5s max-abs %.2e (ref max %.2e)" % (name, float((a - b).abs().max()), float(b.abs().max())))


# --------------------------------------------------------------------------- step
def make_step(pn2_network, pn2_geo, losses):
    from cpfn_amd import synthetic
    from oracle import pn2 as opn2

    state = synthetic.synthetic_state_dict(synthetic.pointnet2_state_shapes(), seed=0)
    model = _load_reference_model(pn2_network, state)
    batch = synthetic.training_batch(2, N=1024, n_prims=5, n_inst_points=64, seed=51)
    P = batch["P"]
    torch.manual_seed(61)
    s1 = torch.randint(0, 1024, (2,), dtype=torch.long)
    s2 = torch.randint(0, 512, (2,), dtype=torch.long)
    torch.manual_seed(61)
    with _no_dropout():
        X, T, W, _, _ = model(P, fast=False)
    X = torch.nn.functional.normalize(X, p=2, dim=2, eps=1e-12)
    W = torch.softmax(W, dim=2)
    gt = {"plane_normal": batch["plane_n_gt"], "cylinder_axis": batch["cylinder_axis_gt"],
          "cone_axis": batch["cone_axis_gt"]}
    match = losses.hungarian_matching(W, batch["I_gt"])
    out = losses.compute_all_losses(P, W, batch["I_gt"], X, batch["X_gt"], T, batch["T_gt"], gt,
                                    batch["points_per_instance"], 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, False,
                                    mode_seg="mIoU", classes=["sphere", "plane", "cylinder", "cone"])
    total = out[0]
    total.backward()
    names = [n for n, _ in model.named_parameters()]
    gnorm = np.array([float(p.grad.norm()) for _, p in model.named_parameters()], np.float64)
    # a few raw gradient slices as well (first 8 entries of every tensor)
    ghead = np.stack([np.resize(p.grad.flatten()[:8].numpy(), 8) for _, p in model.named_parameters()])
    save("step_2x1024.npz", fps_start1=s1.numpy(), fps_start2=s2.numpy(),
         losses=np.array([float(v) for v in out[:6]], np.float64), match=match.numpy().astype(np.int16),
         grad_norm=gnorm, grad_head=ghead, names=np.array(names))
    print("  [step] reference losses", [round(float(v), 6) for v in out[:6]])
    # oracle cross-check
    st = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v)
          for k, v in state.items()}
    o = opn2.training_step_losses(st, batch, (s1, s2))
    print("  [step] oracle    losses", [round(float(v), 6) for v in o[:6]])
    o[0].backward()
    og = np.array([float(st[n].grad.norm()) for n in names])
    rel = np.abs(og - gnorm) / np.maximum(gnorm, 1e-12)
    print("  [step] grad-norm rel err: max %.2e median %.2e" % (rel.max(), np.median(rel)))


# --------------------------------------------------------------------------- loss section alone (§8 f1)
def make_losses(pn2_network, pn2_geo, losses):
    """The loss section pinned on its own: the reference network's RAW heads on a seeded batch are the input; outputs
    are the reference's six losses, its matching and dL/d(heads) through normalise / soft-max / Hungarian / fitters /
    residue + parameter losses (Utils/training_utils.py:141-146, SPFN/losses_implementation.py:675-720).  The
    product's fused loss kernels are fed exactly these heads (tests/test_gpu_network.py)."""
    from cpfn_amd import synthetic

    state = synthetic.synthetic_state_dict(synthetic.pointnet2_state_shapes(), seed=0)
    model = _load_reference_model(pn2_network, state)
    batch = synthetic.training_batch(2, N=1024, n_prims=5, n_inst_points=64, seed=51)
    torch.manual_seed(61)
    with _no_dropout(), torch.no_grad():
        X, T, W, _, _ = model(batch["P"], fast=False)
    Y = torch.cat([X, T, W], dim=2).detach().clone().requires_grad_(True)           # [2,1024,3+4+28]
    Xn = torch.nn.functional.normalize(Y[..., :3], p=2, dim=2, eps=1e-12)
    Ws = torch.softmax(Y[..., 7:], dim=2)
    gt = {"plane_normal": batch["plane_n_gt"], "cylinder_axis": batch["cylinder_axis_gt"],
          "cone_axis": batch["cone_axis_gt"]}
    match = losses.hungarian_matching(Ws, batch["I_gt"])
    out = losses.compute_all_losses(batch["P"], Ws, batch["I_gt"], Xn, batch["X_gt"], Y[..., 3:7], batch["T_gt"], gt,
                                    batch["points_per_instance"], 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, False,
                                    mode_seg="mIoU", classes=["sphere", "plane", "cylinder", "cone"])
    out[0].backward()
    save("losses_2x1024.npz", Y=Y.detach().numpy(), gY=Y.grad.numpy(),
         losses=np.array([float(v) for v in out[:6]], np.float64), match=match.numpy().astype(np.int16))
    print("  [losses] reference losses", [round(float(v), 6) for v in out[:6]])


# --------------------------------------------------------------------------- LocalSPFN step (config 3)
LOCAL_MULT = dict(normal=1.0, type=1.0, miou=1.0, residue=0.0, parameter=0.0, total=1.0)   # Configs/config_localSPFN.yml:6-11


def make_step_local(pn2_network, pn2_geo, losses):
    """One LocalSPFN training step of the reference: K = 21 local instances (training_SPFN.py:69-71,
    Configs/config_localSPFN.yml:19), fitter losses switched off (:10-11) so compute_all_losses never calls the
    fitters (SPFN/losses_implementation.py:681-682)."""
    from cpfn_amd import synthetic
    from oracle import pn2 as opn2

    shapes = synthetic.pointnet2_state_shapes(output_sizes=(3, 4, 21))
    state = synthetic.synthetic_state_dict(shapes, seed=3)
    model = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 21])
    model.load_state_dict(state, strict=True)
    model.train()
    batch = synthetic.training_batch(2, N=1024, n_max_instances=21, n_prims=6, n_inst_points=64, seed=71)
    torch.manual_seed(81)
    s1 = torch.randint(0, 1024, (2,), dtype=torch.long)
    s2 = torch.randint(0, 512, (2,), dtype=torch.long)
    torch.manual_seed(81)
    with _no_dropout():
        X, T, W, _, _ = model(batch["P"], fast=False)
    X = torch.nn.functional.normalize(X, p=2, dim=2, eps=1e-12)
    W = torch.softmax(W, dim=2)
    gt = {"plane_normal": batch["plane_n_gt"], "cylinder_axis": batch["cylinder_axis_gt"],
          "cone_axis": batch["cone_axis_gt"]}
    m = LOCAL_MULT
    match = losses.hungarian_matching(W, batch["I_gt"])
    out = losses.compute_all_losses(batch["P"], W, batch["I_gt"], X, batch["X_gt"], T, batch["T_gt"], gt,
                                    batch["points_per_instance"], m["normal"], m["type"], m["miou"], m["residue"],
                                    m["parameter"], m["total"], False, mode_seg="mIoU",
                                    classes=["sphere", "plane", "cylinder", "cone"])
    assert out[6] is None                       # the fitters were not called
    out[0].backward()
    names = [n for n, p in model.named_parameters() if p.grad is not None]
    gnorm = np.array([float(p.grad.norm()) for _, p in model.named_parameters() if p.grad is not None], np.float64)
    save("step_local_2x1024.npz", fps_start1=s1.numpy(), fps_start2=s2.numpy(),
         losses=np.array([float(v) for v in out[:6]], np.float64), match=match.numpy().astype(np.int16),
         grad_norm=gnorm, names=np.array(names))
    print("  [step_local] reference losses", [round(float(v), 6) for v in out[:6]])
    st = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v)
          for k, v in state.items()}
    o = opn2.training_step_losses(st, batch, (s1, s2), multipliers=m)
    print("  [step_local] oracle    losses", [round(float(v), 6) for v in o[:6]])
    o[0].backward()
    og = np.array([float(st[n].grad.norm()) for n in names])
    rel = np.abs(og - gnorm) / np.maximum(gnorm, 1e-12)
    print("  [step_local] grad-norm rel err: max %.2e median %.2e" % (rel.max(), np.median(rel)))


# --------------------------------------------------------------------------- the other network variants (config 5)
def make_variants(pn2_network):
    """The three PointNet2 variants BASELINE.json configs[4] runs beside GlobalSPFN / LocalSPFN, on a 2 x 2048 cloud:
      ps_  PatchSelection (output_sizes=[2]; training_PatchSelection.py:55, evaluation_PatchSelection.py:45): heat-map logits
           in training and in evaluation mode, and one cross-entropy training step of Utils/training_utils.py:62-75 (loss,
           per-parameter gradient norms, the first entries of every gradient);
      fe_  features_extractor=True (pn2_network.py:31-36, 70-71): (l3_feats, output_feat);
      gl_  use_glob_features=True, use_loc_features=True (pn2_network.py:22-27, 51-54) with seeded feature inputs.
    Dropout neutralised; weights = cpfn_amd.synthetic.synthetic_state_dict of the variant's shapes."""
    from cpfn_amd import synthetic
    out = {}
    cloud = synthetic.primitive_cloud(2, 2048, n_prims=6, seed=32)
    P = cloud["P"]
    sub = np.arange(0, 2048, 8)
    out.update(P=P.numpy(), sub=sub)

    def fwd(model, seed, **kw):
        B, N, _ = P.shape
        torch.manual_seed(seed)
        s1 = torch.randint(0, N, (B,), dtype=torch.long)
        s2 = torch.randint(0, 512, (B,), dtype=torch.long)
        torch.manual_seed(seed)
        with _no_dropout():
            r = model(P, fast=False, **kw)
        return r, (s1, s2)

    # ---- PatchSelection
    state = synthetic.synthetic_state_dict(synthetic.pointnet2_state_shapes([2]), seed=1)
    m = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[2])
    m.load_state_dict(state, strict=True)
    m.train()
    labels = (cloud["I_gt"] % 2).long()                                      # [2,2048] in {0,1}
    (heat, l3, feat), st = fwd(m, 43)
    loss = torch.nn.functional.cross_entropy(heat.contiguous().view(-1, 2), labels.view(-1))     # training_utils.py:66-68
    loss.backward()
    names = [n for n, _ in m.named_parameters()]
    out.update(ps_fps_start1=st[0].numpy(), ps_fps_start2=st[1].numpy(), ps_heat=heat.detach().numpy(), ps_labels=labels.numpy(),
               ps_l3=l3.detach().numpy()[:, :, 0], ps_feat_sub=feat.detach().numpy()[:, :, sub], ps_loss=np.float64(loss.item()),
               ps_grad_norm=np.array([float(p.grad.norm()) for _, p in m.named_parameters()], np.float64),
               ps_grad_head=np.stack([np.resize(p.grad.flatten()[:8].numpy(), 8) for _, p in m.named_parameters()]),
               ps_names=np.array(names))
    m.zero_grad()
    m.load_state_dict(state, strict=True)                                    # (the training pass moved the running statistics)
    m.eval()
    with torch.no_grad():
        (heat_e, _, _), st_e = fwd(m, 44)
    out.update(ps_eval_fps_start1=st_e[0].numpy(), ps_eval_fps_start2=st_e[1].numpy(), ps_eval_heat=heat_e.numpy())
    print("  [variants] PatchSelection loss %.6f, heat range %.3f .. %.3f" % (loss.item(), float(heat.min()), float(heat.max())))
    # ---- features extractor
    state = synthetic.synthetic_state_dict(synthetic.pointnet2_state_shapes([2], features_extractor=True), seed=2)
    m = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[2], features_extractor=True)
    m.load_state_dict(state, strict=True)
    m.train()
    with torch.no_grad():
        (l3, feat), st = fwd(m, 45)
    out.update(fe_fps_start1=st[0].numpy(), fe_fps_start2=st[1].numpy(), fe_l3=l3.numpy()[:, :, 0], fe_feat_sub=feat.numpy()[:, :, sub])
    # ---- LocalSPFN with global + local feature inputs
    state = synthetic.synthetic_state_dict(synthetic.pointnet2_state_shapes([3, 4, 21], True, True), seed=3)
    m = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 21], use_glob_features=True, use_loc_features=True)
    m.load_state_dict(state, strict=True)
    m.train()
    g = torch.Generator().manual_seed(46)
    glob, loc = torch.randn(2, 1024, generator=g), torch.randn(2, 128, generator=g)
    with torch.no_grad():
        (X, T, W, l3, feat), st = fwd(m, 47, glob_features=glob, loc_features=loc)
    out.update(gl_fps_start1=st[0].numpy(), gl_fps_start2=st[1].numpy(), gl_glob=glob.numpy(), gl_loc=loc.numpy(), gl_X=X.numpy(),
               gl_T=T.numpy(), gl_W=W.numpy(), gl_l3=l3.numpy()[:, :, 0], gl_feat_sub=feat.numpy()[:, :, sub])
    save("network_variants_2x2048.npz", **out)
