"""GPU parity of the fused SPFN fitters (HIP moment kernels + per-instance algebra) against
the golden fixtures of the imported reference and against the oracle.
Tolerance: 1e-4 relative PER INSTANCE (‖a_bk − b_bk‖ / max(‖b_bk‖, 1e-3) for every [b,k] of every
parameter, helpers.per_instance_rel) and per tensor (max|a-b| / max|b|), sign-invariant where the
reference's SVD leaves the sign free — BASELINE.json north_star.  Gradients: per instance column
for dL/dW, per cloud for dL/dX."""
import numpy as np
import pytest
import torch

from helpers import PARAM_KEYS, align_signs, per_instance_rel, plane_eigen_gap, rel_err, sign_invariant_loss
from oracle import spfn as ospfn

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev():
    return torch.device("cuda:0")


def _run_product(P, W, X, coef=None):
    from cpfn_amd.SPFN import losses_implementation as li
    Pd = P.to(dev())
    Wd = W.to(dev()).requires_grad_(True)
    Xd = X.to(dev()).requires_grad_(True)
    params = li.compute_parameters(Pd, Wd, Xd)
    grads = None
    if coef is not None:
        L = sign_invariant_loss(params, {k: v.to(dev()) for k, v in coef.items()})
        L.backward()
        grads = (Wd.grad.cpu(), Xd.grad.cpu())
    return {k: v.detach().cpu() for k, v in params.items()}, grads


def _check_case(g):
    P, W, X = (torch.from_numpy(g[k]) for k in ("P", "W", "X"))
    coef = {k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("coef_")}
    mine, (gW, gX) = _run_product(P, W, X, coef)
    ref = {k: torch.from_numpy(g["out_" + k]) for k in PARAM_KEYS}
    aligned = align_signs(mine, ref)
    worst = {}
    for k in PARAM_KEYS:
        assert mine[k].dtype == torch.float32
        assert rel_err(aligned[k], ref[k]) < TOL, (k, rel_err(aligned[k], ref[k]))
        e = per_instance_rel(aligned[k], ref[k])                       # [B,K]: EVERY instance of the fixture
        worst[k] = float(e.max())
        if not bool((e < TOL).all()):
            gap = plane_eigen_gap(P, W)
            bad = [(int(b), int(i), float(e[b, i]), float(gap[b, i])) for b, i in (e >= TOL).nonzero()]
            raise AssertionError("%s: instances (b, k, rel err, plane eigen-gap) over %g: %s" % (k, TOL, bad))
    print("per-instance max rel err:", {k: "%.1e" % v for k, v in worst.items()})
    rW, rX = torch.from_numpy(g["gW"]), torch.from_numpy(g["gX"])
    assert rel_err(gW, rW) < TOL
    assert rel_err(gX, rX) < TOL
    eW = (gW.double() - rW.double()).norm(dim=1) / rW.double().norm(dim=1).clamp_min(1e-12)       # per (cloud, instance) column
    eX = (gX.double() - rX.double()).norm(dim=(1, 2)) / rX.double().norm(dim=(1, 2))                # per cloud
    assert float(eW.max()) < TOL and float(eX.max()) < TOL, (float(eW.max()), float(eX.max()))


def test_golden_selftest_recipe(golden):
    _check_case(golden("fitters_selftest.npz"))


def test_golden_points_on_primitives(golden):
    _check_case(golden("fitters_primitives.npz"))


def test_full_size_vs_oracle():
    """B=2 x 8192 points x 28 instances (10 real primitives + 18 near-empty columns that hit
    the guards) against the fp32 oracle."""
    from cpfn_amd import synthetic
    d = synthetic.primitive_cloud(2, 8192, n_prims=10, noise=0.003, seed=77)
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(2, 8192, 28, generator=g) * 0.3
    logits.scatter_add_(2, d["I_gt"].unsqueeze(2), torch.full((2, 8192, 1), 7.0))
    W = torch.softmax(logits, dim=2)
    X = torch.nn.functional.normalize(d["X_gt"] + 0.05 * torch.randn(2, 8192, 3, generator=g), dim=2)
    P = d["P"]
    mine, _ = _run_product(P, W, X)
    ref = ospfn.compute_parameters(P, W, X)
    ref64 = ospfn.compute_parameters(P.double(), W.double(), X.double())
    aligned = align_signs(mine, ref)
    aligned64 = align_signs({k: v.double() for k, v in mine.items()}, ref64)
    ref_noise = align_signs({k: v.double() for k, v in ref.items()}, ref64)
    gap = None
    for k in PARAM_KEYS:
        assert torch.isfinite(mine[k]).all(), k
        # EVERY instance (all 28 columns, both clouds) against the fp32 restatement of the reference, per instance.
        e32 = per_instance_rel(aligned[k], ref[k])
        bad = (e32 >= TOL).nonzero()
        # An instance may only miss the fp32 oracle if that is the reference's own fp32 rounding noise: the fp32 oracle
        # itself is then further than TOL/4 from the fp64 evaluation of the same formulas on that instance, while the
        # product (fp64 moment accumulation) sits within TOL of the fp64 evaluation.  Justified per instance, printed
        # with the eigen-gap; anything else fails.
        e64 = per_instance_rel(aligned64[k], ref64[k])
        noise = per_instance_rel(ref_noise[k], ref64[k])
        for b, i in bad:
            gap = plane_eigen_gap(P, W) if gap is None else gap
            info = (k, int(b), int(i), float(e32[b, i]), float(e64[b, i]), float(noise[b, i]), float(gap[b, i]))
            assert float(e64[b, i]) < TOL and float(noise[b, i]) > TOL / 4, \
                "(param, b, k, err vs fp32 oracle, err vs fp64 oracle, fp32-oracle noise, plane eigen-gap) = %s" % (info,)
            print("instance accepted on fp32-reference noise:", info)


def test_reference_shaped_helpers_vs_oracle():
    from cpfn_amd.SPFN import differentiable_tls, geometry_utils
    g = torch.Generator().manual_seed(9)
    A = torch.randn(6, 500, 3, generator=g)
    W = torch.rand(6, 500, generator=g)
    x = differentiable_tls.solve_weighted_tls(A.to(dev()), W.to(dev())).cpu()
    xr = ospfn.solve_weighted_tls(A, W)
    s = torch.sign((x * xr).sum(-1, keepdim=True))
    assert rel_err(x * s, xr) < TOL
    n, c = geometry_utils.weighted_plane_fitting(A.to(dev()), W.to(dev()))
    nr, cr = ospfn.weighted_plane_fitting(A, W.unsqueeze(1))
    s = torch.sign((n.cpu() * nr[:, 0]).sum(-1, keepdim=True))
    assert rel_err(n.cpu() * s, nr[:, 0]) < TOL and rel_err(c.cpu() * s[:, 0], cr[:, 0]) < TOL
    ctr, r2 = geometry_utils.weighted_sphere_fitting(A.to(dev()), W.to(dev()))
    ctr_r, r2_r = ospfn.weighted_sphere_fitting(A, W.unsqueeze(1))
    assert rel_err(ctr.cpu(), ctr_r[:, 0]) < TOL and rel_err(r2.cpu(), r2_r[:, 0]) < TOL


def test_guarded_matrix_solve_ls_on_the_selftest_sphere_problem(golden):
    """`SPFN.geometry_utils.guarded_matrix_solve_ls` exists on the device path with the reference's signature
    (SPFN/geometry_utils.py:121-142) — given the (A, b, W) that `weighted_sphere_fitting` builds (:209-220) from the reference's own
    self-test recipe it returns the fixture's sphere centres (the reference's output) and the oracle's, per instance; its guards
    (condition-number cap, ridge) answer a rank-deficient problem with 0 like the oracle; gradients flow to A, b and W."""
    from cpfn_amd.SPFN import geometry_utils
    g = golden("fitters_selftest.npz")
    P, W = torch.from_numpy(g["P"]), torch.from_numpy(g["W"])
    B, N, K = W.shape
    Wk = W.transpose(1, 2).reshape(B * K, N)                                       # [BK,N]   (sphere_fitter.py:12-13)
    Pk = P.unsqueeze(1).expand(B, K, N, 3).reshape(B * K, N, 3)
    den = Wk.sum(1).clamp(min=1e-10)
    psq = (Pk * Pk).sum(-1)
    b = ((Wk * psq).sum(1) / den).unsqueeze(1) - psq                               # :213-214
    A = 2.0 * (((Wk.unsqueeze(2) * Pk).sum(1) / den.unsqueeze(1)).unsqueeze(1) - Pk)   # :215-216
    x = geometry_utils.guarded_matrix_solve_ls(A.to(dev()), b.unsqueeze(2).to(dev()), Wk.to(dev())).cpu()
    assert x.dtype == torch.float32 and x.shape == (B * K, 3)
    ref = torch.from_numpy(g["out_sphere_center"]).reshape(B * K, 3)
    xo = ospfn.guarded_matrix_solve_ls(A.view(B, K, N, 3), b.view(B, K, N), Wk.view(B, K, N)).reshape(B * K, 3)
    for want in (ref, xo):
        e = (x - want).norm(dim=-1) / want.norm(dim=-1).clamp(min=1e-3)
        assert float(e.max()) < TOL, float(e.max())
    # the guards: two identical columns -> condition number over the cap -> mask 0 -> x = 0 (ridge only), as in the oracle
    A2 = A.clone()
    A2[:, :, 1] = A2[:, :, 0]
    x2 = geometry_utils.guarded_matrix_solve_ls(A2.to(dev()), b.unsqueeze(2).to(dev()), Wk.to(dev())).cpu()
    xo2 = ospfn.guarded_matrix_solve_ls(A2.view(B, K, N, 3), b.view(B, K, N), Wk.view(B, K, N)).reshape(B * K, 3)
    assert float(x2.abs().max()) == 0.0 and float(xo2.abs().max()) == 0.0
    # D = 2 (the cylinder's circle fit) against the oracle
    x3 = geometry_utils.guarded_matrix_solve_ls(A[:, :, :2].contiguous().to(dev()), b.unsqueeze(2).to(dev()), Wk.to(dev())).cpu()
    xo3 = ospfn.guarded_matrix_solve_ls(A.view(B, K, N, 3)[..., :2], b.view(B, K, N), Wk.view(B, K, N)).reshape(B * K, 2)
    assert rel_err(x3, xo3) < TOL
    # differentiable in all three arguments, gradients against the oracle's autograd
    Ad, bd, Wd = (t.to(dev()).requires_grad_(True) for t in (A, b.unsqueeze(2), Wk))
    geometry_utils.guarded_matrix_solve_ls(Ad, bd, Wd).square().sum().backward()
    Ao, bo, Wo = (t.clone().requires_grad_(True) for t in (A.view(B, K, N, 3), b.view(B, K, N), Wk.view(B, K, N)))
    ospfn.guarded_matrix_solve_ls(Ao, bo, Wo).square().sum().backward()
    assert rel_err(Ad.grad.cpu().view(B, K, N, 3), Ao.grad) < 1e-3
    assert rel_err(bd.grad.cpu().view(B, K, N), bo.grad) < 1e-3
    assert rel_err(Wd.grad.cpu().view(B, K, N), Wo.grad) < 1e-3
    with pytest.raises(RuntimeError, match="CPU not supported"):
        geometry_utils.guarded_matrix_solve_ls(A, b.unsqueeze(2), Wk)
    from cpfn_amd.SPFN import cone_fitter, cylinder_fitter, plane_fitter
    for m in (cylinder_fitter, cone_fitter):                                        # own `def`s now, not fall-throughs
        assert m.acos_safe.__module__ == m.__name__ and m.compute_parameter_loss.__module__ == m.__name__
        pn, gn = torch.randn(2, 5, 3, device=dev()), torch.randn(2, 4, 3, device=dev())
        mi = torch.randint(0, 5, (2, 4), device=dev())
        for ad in (False, True):
            assert torch.equal(m.compute_parameter_loss(pn, gn, mi, ad), plane_fitter.compute_parameter_loss(pn, gn, mi, ad))


def test_moment_linearity_and_determinism():
    """Size-independent properties at the full benchmark size (B=16, N=8192, K=28):
    moments are linear in W, and two runs are bitwise identical (no atomics)."""
    from cpfn_amd.SPFN import moments
    g = torch.Generator().manual_seed(1)
    P = (torch.rand(16, 8192, 3, generator=g) * 2 - 1).to(dev())
    X = torch.nn.functional.normalize(torch.randn(16, 8192, 3, generator=g), dim=2).to(dev())
    W1 = torch.rand(16, 8192, 28, generator=g).to(dev()) + 1e-3
    W2 = torch.rand(16, 8192, 28, generator=g).to(dev()) + 1e-3
    M1, M2, M12 = (moments.FitMoments.apply(P, X, w) for w in (W1, W2, W1 + W2))
    assert torch.equal(M1, moments.FitMoments.apply(P, X, W1))
    torch.testing.assert_close(M1 + M2, M12, rtol=1e-6, atol=1e-4)   # W1+W2 itself is rounded to fp32
    # slot 0 is ΣW
    torch.testing.assert_close(M1[..., 0], W1.double().sum(1), rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("B,N,K", [(16, 8192, 28), (3, 1000, 7), (2, 4097, 21)])
def test_packed_parameters_backward_in_three_launches(B, N, K, monkeypatch):
    """FitParams.backward through cpfn_fit_params_bwd_cone / _algebra (g_acos, the algebra's adjoint and the chunk sums of
    the cone pass's d(apex, axis) formed inside the two launches) against cpfn_fit_pack_bwd + cpfn_cone_pass_bwd +
    chunk reduction + cpfn_fit_algebra_bwd: the same bits."""
    from cpfn_amd import lib as _l
    from cpfn_amd.SPFN import moments, fitters_common as fc
    g = torch.Generator().manual_seed(B * N + K)
    P = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(dev())
    X = torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=2).to(dev())
    logits = (torch.randn(B, N, K, generator=g) * 2).to(dev())
    gp = torch.randn(B, K, 22, generator=g).to(dev())
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(moments, "PARAMS_BWD_FUSED", fused)
        monkeypatch.setattr(moments, "PARAMS_FWD_FUSED", fused)      # (+ the forward's chunk reduction inside the algebra launch)
        W = torch.softmax(logits, 2).requires_grad_(True)
        Xr = X.clone().requires_grad_(True)
        _l.byte_census(True)
        params = fc.fit_params(P, W, Xr)
        params.backward(gp)
        census = _l.byte_census(False)
        assert ("cpfn_fit_pack_bwd" in census) == (not fused), sorted(census)
        res[fused] = (params.detach(), W.grad, Xr.grad)
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)
    assert float(res[True][1].abs().max()) > 0 and bool(torch.isfinite(res[True][1]).all())


@pytest.mark.parametrize("B,N,n_prims,K", [(1, 300, 2, 2), (3, 1000, 5, 7), (2, 5000, 9, 12), (1, 8192, 10, 33), (1, 4097, 12, 64)])
def test_shape_sweep_vs_oracle(B, N, n_prims, K):
    """Ragged sizes and instance counts from 2 to 64 (the moment kernels' limit): points on real primitives with peaky
    memberships for the first n_prims columns, background-level memberships for the rest.  Every instance of every
    parameter within 1e-4 (per instance) of the fp32 oracle, or — where that is the fp32 oracle's own rounding noise
    against its fp64 run — of the fp64 oracle; gradients of the sign-invariant loss against the fp64 oracle."""
    from cpfn_amd import synthetic
    d = synthetic.primitive_cloud(B, N, n_prims=n_prims, noise=0.003, seed=N + K)
    g = torch.Generator().manual_seed(K)
    logits = torch.randn(B, N, K, generator=g) * 0.3
    logits.scatter_add_(2, d["I_gt"].unsqueeze(2), torch.full((B, N, 1), 7.0))
    W = torch.softmax(logits, dim=2)
    X = torch.nn.functional.normalize(d["X_gt"] + 0.05 * torch.randn(B, N, 3, generator=g), dim=2)
    P = d["P"]
    coef = {"plane_normal_outer": torch.randn(B, K, 3, 3, generator=g), "plane_cn": torch.randn(B, K, 3, generator=g),
            "cylinder_axis_outer": torch.randn(B, K, 3, 3, generator=g)}
    for k in PARAM_KEYS:
        if k not in ("plane_normal", "plane_center", "cylinder_axis"):
            coef[k] = torch.randn((B, K) if k.endswith("squared") or k == "cone_half_angle" else (B, K, 3), generator=g)
    mine, (gW, gX) = _run_product(P, W, X, coef)
    ref = ospfn.compute_parameters(P, W, X)
    W64, X64 = W.double().requires_grad_(True), X.double().requires_grad_(True)
    ref64 = ospfn.compute_parameters(P.double(), W64, X64)
    sign_invariant_loss(ref64, {k: v.double() for k, v in coef.items()}).backward()
    ref64 = {k: v.detach() for k, v in ref64.items()}
    a32 = align_signs(mine, ref)
    a64 = align_signs({k: v.double() for k, v in mine.items()}, ref64)
    noise = align_signs({k: v.double() for k, v in ref.items()}, ref64)
    for k in PARAM_KEYS:
        e32, e64, en = per_instance_rel(a32[k], ref[k]), per_instance_rel(a64[k], ref64[k]), per_instance_rel(noise[k], ref64[k])
        ok = (e32 < TOL) | ((e64 < TOL) & (en > TOL / 4))
        assert bool(ok.all()), (k, [(int(b), int(i), float(e32[b, i]), float(e64[b, i]), float(en[b, i])) for b, i in (~ok).nonzero()])
    eW = (gW.double() - W64.grad).norm(dim=1) / W64.grad.norm(dim=1).clamp_min(1e-12)
    eX = (gX.double() - X64.grad).norm(dim=(1, 2)) / X64.grad.norm(dim=(1, 2))
    assert float(eW.max()) < 5e-4 and float(eX.max()) < 5e-4, (float(eW.max()), float(eX.max()))
