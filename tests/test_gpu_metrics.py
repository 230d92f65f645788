"""Evaluation metrics: cpfn_amd.SPFN.metric_implementation on the GPU against the reference's
own outputs (fixture) and against the oracle at a larger size; the fused P-coverage kernel against the
reference-shaped `[B,K,N,*]` expansion."""
import numpy as np
import pytest
import torch

from cpfn_amd import synthetic
from oracle import metrics as om

pytestmark = pytest.mark.gpu
CLASSES = ["plane", "sphere", "cylinder", "cone"]


def dev():
    return torch.device("cuda:0")


def _check(out, ref, n_inst_pts, n_pts):
    mIoU, type_acc, normal_diff, axis_diff, mean_res, std_res, Sk, Pc, Wh, params, Tinst = out
    got = dict(mIoU=mIoU, type_accuracy=type_acc, normal_difference=normal_diff, axis_difference=axis_diff,
               mean_residual=mean_res, std_residual=std_res)
    assert np.array_equal(Tinst.cpu().numpy(), np.asarray(ref["T_instance"]))
    for k, v in got.items():
        np.testing.assert_allclose(v.cpu().numpy(), np.asarray(ref[k]), rtol=1e-3, atol=1e-5, err_msg=k)
    np.testing.assert_allclose(torch.stack(Sk).cpu().numpy(), np.asarray(ref["Sk_coverage"]), atol=3.0 / n_inst_pts)
    np.testing.assert_allclose(torch.stack(Pc).cpu().numpy(), np.asarray(ref["P_coverage"]), atol=3.0 / n_pts)


def test_all_metrics_match_reference_fixture(golden):
    from cpfn_amd.SPFN import metric_implementation as mi
    g = golden("metrics_2x2048.npz")
    t = lambda k: torch.from_numpy(g[k]).to(dev())
    gt = {k: t("gt_" + k) for k in ("plane_normal", "cylinder_axis", "cone_axis")}
    out = mi.compute_all_metrics(t("P"), t("X"), t("X_gt"), t("W"), t("I_gt"), t("T"), t("T_gt"), t("points_per_instance"),
                                 gt, list_epsilon=[float(e) for e in g["epsilons"]], classes=CLASSES)
    match, mask = mi.hungarian_matching(out[8], t("I_gt"))
    assert np.array_equal(match.cpu().numpy(), g["matching"]) and np.array_equal(mask.cpu().numpy(), g["mask"])
    _check(out, g, 512, 2048)


@pytest.mark.parametrize("tag", ["few_", "many_"])
def test_all_metrics_padding_branches_match_reference_fixture(golden, tag):
    """Fewer / more prediction columns than GT slots (reference :487-492, :505-508; handled inside the kernels here, nothing is
    concatenated) against the reference's own outputs, including the hard W, the fitted parameters and the instance types of
    the widened label set."""
    from cpfn_amd.SPFN import metric_implementation as mi
    from helpers import PARAM_KEYS, per_instance_rel
    g = {k[len(tag):]: v for k, v in golden("metrics_padded_2x1024.npz").items() if k.startswith(tag)}
    t = lambda k: torch.from_numpy(g[k]).to(dev())
    gt = {k: t("gt_" + k) for k in ("plane_normal", "cylinder_axis", "cone_axis")}
    out = mi.compute_all_metrics(t("P"), t("X"), t("X_gt"), t("W"), t("I_gt"), t("T"), t("T_gt"), t("points_per_instance"),
                                 gt, list_epsilon=[float(e) for e in g["epsilons"]], classes=CLASSES)
    _check(out, g, 512, 1024)
    assert np.array_equal(out[8].cpu().numpy(), g["W_hard"])
    # fitted parameters of the columns that own points (an empty column's fit is the guards' output: not compared)
    used = torch.from_numpy(g["W_hard"]).sum(1) > 8
    for k in PARAM_KEYS:
        a, b = out[9][k].cpu(), torch.from_numpy(g["param_" + k])
        if k in ("plane_normal", "cylinder_axis"):
            a = a * torch.sign((a * b).sum(-1, keepdim=True))
        if k == "plane_center":
            a = a * torch.sign((out[9]["plane_normal"].cpu() * torch.from_numpy(g["param_plane_normal"])).sum(-1))
        assert float(per_instance_rel(a, b)[used].max()) < 2e-3, k


def test_all_metrics_match_oracle_at_8192():
    from cpfn_amd.SPFN import metric_implementation as mi
    B, N, K = 3, 8192, 28
    batch = synthetic.training_batch(B, N=N, n_prims=9, n_inst_points=256, seed=31)
    g = torch.Generator().manual_seed(9)
    lab = batch["I_gt"].clamp(min=0)
    logits = torch.randn(B, N, K, generator=g)
    logits.scatter_add_(2, lab.unsqueeze(2), torch.full((B, N, 1), 2.5))
    W = torch.softmax(logits, dim=2)
    X = torch.nn.functional.normalize(batch["X_gt"] + 0.3 * torch.randn(B, N, 3, generator=g), dim=2)
    T = torch.randn(B, N, 4, generator=g)
    T.scatter_add_(2, torch.gather(batch["T_gt"], 1, lab).unsqueeze(2), torch.full((B, N, 1), 0.7))
    gt = {"plane_normal": batch["plane_n_gt"], "cylinder_axis": batch["cylinder_axis_gt"], "cone_axis": batch["cone_axis_gt"]}
    ref = om.compute_all_metrics(batch["P"], X, batch["X_gt"], W, batch["I_gt"], T, batch["T_gt"],
                                 batch["points_per_instance"], gt, list_epsilon=[0.01, 0.03, 0.1], classes=CLASSES)
    d = lambda v: v.to(dev())
    out = mi.compute_all_metrics(d(batch["P"]), d(X), d(batch["X_gt"]), d(W), d(batch["I_gt"]), d(T), d(batch["T_gt"]),
                                 d(batch["points_per_instance"]), {k: d(v) for k, v in gt.items()},
                                 list_epsilon=[0.01, 0.03, 0.1], classes=CLASSES)
    _check(out, {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in ref.items() if k != "params"}, 256, N)


@pytest.mark.parametrize("K,n_prims", [(49, 40), (70, 33), (100, 60), (128, 90)])      # 100, 128: > 64 KB of LDS
def test_all_metrics_more_than_32_instances_vs_oracle(K, n_prims):
    """Evaluation on a MERGED label set (evaluation_localSPFN.py:129-131 hands compute_all_metrics a W_fusion with
    >= 28 columns and no upper bound; 21 local + 28 global = 49): more instance columns AND more GT labels than the
    32-wide tile of the training-side kernels — tiled segmented sums, host assignment, untiled fits, K-looped P
    coverage — against the oracle."""
    from cpfn_amd.SPFN import metric_implementation as mi
    B, N = 2, 8192
    batch = synthetic.training_batch(B, N=N, n_max_instances=K, n_prims=n_prims, n_inst_points=128, seed=K)
    g = torch.Generator().manual_seed(K)
    lab = batch["I_gt"].clamp(min=0)
    logits = torch.randn(B, N, K, generator=g)
    logits.scatter_add_(2, lab.unsqueeze(2), torch.full((B, N, 1), 3.0))
    W = torch.softmax(logits, dim=2)
    X = torch.nn.functional.normalize(batch["X_gt"] + 0.3 * torch.randn(B, N, 3, generator=g), dim=2)
    T = torch.randn(B, N, 4, generator=g)
    T.scatter_add_(2, torch.gather(batch["T_gt"], 1, lab).unsqueeze(2), torch.full((B, N, 1), 0.7))
    gt = {"plane_normal": batch["plane_n_gt"], "cylinder_axis": batch["cylinder_axis_gt"], "cone_axis": batch["cone_axis_gt"]}
    ref = om.compute_all_metrics(batch["P"], X, batch["X_gt"], W, batch["I_gt"], T, batch["T_gt"],
                                 batch["points_per_instance"], gt, list_epsilon=[0.01, 0.03], classes=CLASSES)
    d = lambda v: v.to(dev())
    out = mi.compute_all_metrics(d(batch["P"]), d(X), d(batch["X_gt"]), d(W), d(batch["I_gt"]), d(T), d(batch["T_gt"]),
                                 d(batch["points_per_instance"]), {k: d(v) for k, v in gt.items()},
                                 list_epsilon=[0.01, 0.03], classes=CLASSES)
    match, mask = mi.hungarian_matching(out[8], d(batch["I_gt"]))
    assert np.array_equal(match.cpu().numpy(), ref["matching"].numpy())
    assert int(mask.sum()) == B * n_prims
    _check(out, {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in ref.items() if k != "params"}, 128, N)
    # the tiled segmented sums themselves against a dense one-hot contraction
    from cpfn_amd.SPFN import fused_losses as fl
    S = fl.SegStats.apply(d(W), d(batch["I_gt"])).cpu()
    oh = torch.nn.functional.one_hot(lab, K).to(W.dtype) * (batch["I_gt"] >= 0).unsqueeze(2)
    np.testing.assert_allclose(S[:, :K].numpy(), (oh.transpose(1, 2) @ W).numpy(), rtol=2e-5, atol=1e-4)
    np.testing.assert_allclose(S[:, K].numpy(), W.sum(1).numpy(), rtol=2e-5, atol=1e-4)
    assert np.array_equal(S[:, K + 1].numpy(), oh.sum(1).numpy())
    # the point pass sums in a fixed order (no float atomics since round 5): a second call gives the same bits
    out2 = mi.compute_all_metrics(d(batch["P"]), d(X), d(batch["X_gt"]), d(W), d(batch["I_gt"]), d(T), d(batch["T_gt"]),
                                  d(batch["points_per_instance"]), {k: d(v) for k, v in gt.items()},
                                  list_epsilon=[0.01, 0.03], classes=CLASSES)
    for a, b in zip(out[:6], out2[:6]):
        assert torch.equal(a, b)
    assert torch.equal(out[10], out2[10]) and torch.equal(out[8], out2[8])


def test_p_coverage_kernel_vs_expanded_formula():
    """cpfn_p_coverage against the reference-shaped expansion (get_residual_loss on P broadcast to [B,K,N,3])."""
    from cpfn_amd.SPFN import metric_implementation as mi
    B, N, K = 2, 5000, 28                                  # ragged N: not a multiple of the 256-point chunk
    batch = {k: v.to(dev()) for k, v in synthetic.training_batch(B, N=N, n_prims=7, n_inst_points=64, seed=4).items()}
    W = mi.hard_W_encoding(torch.rand(B, N, K, device=dev()) + 3 * torch.nn.functional.one_hot(batch["I_gt"].clamp(min=0), K))
    params = mi.losses_implementation.compute_parameters(batch["P"], W, batch["X_gt"])
    match, _ = mi.hungarian_matching(W, batch["I_gt"])
    T = torch.randint(0, 4, (B, K), device=dev())
    eps = [0.005, 0.02, 0.05, 0.2]
    got = mi.compute_P_coverages(batch["P"], T, match, params, eps, classes=CLASSES).cpu().numpy()
    res = mi.get_residual_loss(params, match, batch["P"].unsqueeze(1).expand(B, K, N, 3), torch.gather(T, 1, match), classes=CLASSES)
    rmin = res.min(dim=1)[0]
    want = np.stack([(rmin < e).float().mean(1).cpu().numpy() for e in eps])
    np.testing.assert_allclose(got, want, atol=3.0 / N)
    assert mi.compute_P_coverage(batch["P"], T, match, params, 0.02, classes=CLASSES).shape == (B,)


def _metrics_case(B, N, K, n_prims, seed, background=0.0):
    batch = synthetic.training_batch(B, N=N, n_max_instances=K, n_prims=n_prims, n_inst_points=128, seed=seed)
    g = torch.Generator().manual_seed(seed)
    I = batch["I_gt"].clone()
    if background > 0:
        I[torch.rand(B, N, generator=g) < background] = -1            # unlabelled points (the reference's "may contain -1's")
        I[:, :n_prims] = torch.arange(n_prims)                         # (every label still present: gap-free)
    lab = I.clamp(min=0)
    logits = torch.randn(B, N, K, generator=g)
    logits.scatter_add_(2, lab.unsqueeze(2), torch.full((B, N, 1), 3.0))
    W = torch.softmax(logits, dim=2)
    X = torch.nn.functional.normalize(batch["X_gt"] + 0.3 * torch.randn(B, N, 3, generator=g), dim=2)
    T = torch.randn(B, N, 4, generator=g)
    T.scatter_add_(2, torch.gather(batch["T_gt"], 1, lab).unsqueeze(2), torch.full((B, N, 1), 0.7))
    gt = {"plane_normal": batch["plane_n_gt"], "cylinder_axis": batch["cylinder_axis_gt"], "cone_axis": batch["cone_axis_gt"]}
    return batch, I, W, X, T, gt


@pytest.mark.parametrize("case", ["background", "wide"])
def test_all_metrics_background_labels_and_label_sets_wider_than_the_on_chip_histogram(case):
    """(background) I_gt with -1 entries (metric_implementation.py:12: "may contain -1's"): such points count in the column sums
    of the assignment's cost but in no GT row; (wide) K = 140 > 128: the point pass falls back to the generic kernels
    (arg-max / one-hot / SegStats / host assignment), the tail and P-coverage kernels are the same."""
    from cpfn_amd.SPFN import metric_implementation as mi
    B, N = 2, 4096
    K, n_prims, bg = (28, 8, 0.2) if case == "background" else (140, 30, 0.0)
    batch, I, W, X, T, gt = _metrics_case(B, N, K, n_prims, seed=41 + K, background=bg)
    ref = om.compute_all_metrics(batch["P"], X, batch["X_gt"], W, I, T, batch["T_gt"], batch["points_per_instance"], gt,
                                 list_epsilon=[0.01, 0.03], classes=CLASSES)
    d = lambda v: v.to(dev())
    out = mi.compute_all_metrics(d(batch["P"]), d(X), d(batch["X_gt"]), d(W), d(I), d(T), d(batch["T_gt"]),
                                 d(batch["points_per_instance"]), {k: d(v) for k, v in gt.items()}, list_epsilon=[0.01, 0.03],
                                 classes=CLASSES)
    assert out[8].shape == (B, N, K) and np.array_equal(out[8].cpu().numpy(), ref["W_hard"].numpy())
    match, mask = mi.hungarian_matching(out[8], d(I))
    assert np.array_equal(match.cpu().numpy(), ref["matching"].numpy()) and int(mask.sum()) == B * n_prims
    _check(out, {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in ref.items() if k != "params"}, 128, N)
