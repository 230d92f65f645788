"""GPU parity of the PointNet++ network and of one training step against the golden
fixtures of the imported reference (dropout neutralised on both sides, fp32 compute)."""
import numpy as np
import pytest
import torch

from cpfn_amd import synthetic

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _model(dtype="fp32"):
    from cpfn_amd.PointNet2 import pn2_network
    m = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 28])
    m.load_state_dict(synthetic.synthetic_state_dict(synthetic.pointnet2_state_shapes(), seed=0), strict=True)
    m.dropout_p = 0.0
    return m.to(dev()).train()


def test_forward_matches_reference(golden):
    g = golden("network_2x2048.npz")
    m = _model()
    P = torch.from_numpy(g["P"]).to(dev())
    starts = (torch.from_numpy(g["fps_start1"]), torch.from_numpy(g["fps_start2"]))
    with torch.no_grad():
        X, T, W, l3, feat = m(P, fps_start=starts)
    assert X.shape == (2, 2048, 3) and T.shape == (2, 2048, 4) and W.shape == (2, 2048, 28)
    assert l3.shape == (2, 1024, 1) and feat.shape == (2, 128, 2048)
    for name, t in (("X", X), ("T", T), ("W", W)):
        np.testing.assert_allclose(t.cpu().numpy(), g[name], rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(l3.cpu().numpy()[:, :, 0], g["l3"], rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(feat.cpu().numpy()[:, :, g["sub"]], g["feat_sub"], rtol=2e-3, atol=2e-3)


def test_seeded_fps_start_matches_reference_rng(golden):
    """Without explicit starts the model draws them from the CPU generator exactly like the
    reference's CPU route, so torch.manual_seed reproduces the reference's sampled points."""
    g = golden("network_2x2048.npz")
    m = _model()
    torch.manual_seed(41)
    with torch.no_grad():
        m(torch.from_numpy(g["P"]).to(dev()))
    assert np.array_equal(m.aux_sa1["fps_idx"][:, 0].cpu().numpy(), g["fps_start1"])
    assert np.array_equal(m.aux_sa2["fps_idx"][:, 0].cpu().numpy(), g["fps_start2"])


def test_training_step_matches_reference(golden):
    from cpfn_amd.SPFN import losses_implementation as li
    g = golden("step_2x1024.npz")
    m = _model()
    batch = {k: v.to(dev()) for k, v in synthetic.training_batch(2, N=1024, n_prims=5, n_inst_points=64, seed=51).items()}
    starts = (torch.from_numpy(g["fps_start1"]), torch.from_numpy(g["fps_start2"]))
    X, T, W, _, _ = m(batch["P"], fps_start=starts)
    X = torch.nn.functional.normalize(X, p=2, dim=2, eps=1e-12)
    W = torch.softmax(W, dim=2)
    gt = {"plane_normal": batch["plane_n_gt"], "cylinder_axis": batch["cylinder_axis_gt"],
          "cone_axis": batch["cone_axis_gt"]}
    out = li.compute_all_losses(batch["P"], W, batch["I_gt"], X, batch["X_gt"], T, batch["T_gt"], gt,
                                batch["points_per_instance"], 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, False,
                                mode_seg="mIoU", classes=["sphere", "plane", "cylinder", "cone"])
    np.testing.assert_allclose([float(v) for v in out[:6]], g["losses"], rtol=2e-3, atol=1e-4)
    assert np.array_equal(li.hungarian_matching(W, batch["I_gt"]).cpu().numpy(), g["match"].astype(np.int64))
    out[0].backward()
    names = [str(n) for n in g["names"]]
    params = dict(m.named_parameters())
    gn = np.array([float(params[n].grad.norm()) for n in names])
    scale = g["grad_norm"].max()
    bad = np.abs(gn - g["grad_norm"]) > 2e-2 * g["grad_norm"] + 1e-4 * scale
    assert not bad.any(), [(names[i], gn[i], g["grad_norm"][i]) for i in np.nonzero(bad)[0]]


def test_fused_losses_match_reference_fixture(golden):
    """§8 f1 pinned to the REFERENCE, not to the sibling path: the fused HIP loss section (csrc/losses.hip +
    the fitter chain) is fed the reference network's raw heads of losses_2x1024.npz and must reproduce the
    reference's six losses, its Hungarian matching and dL/d(heads) (SPFN/losses_implementation.py:675-720 behind
    Utils/training_utils.py:141-146); the op-by-op twin is held to the same fixture."""
    from cpfn_amd.SPFN import fused_losses, losses_implementation as li
    g = golden("losses_2x1024.npz")
    batch = {k: v.to(dev()) for k, v in synthetic.training_batch(2, N=1024, n_prims=5, n_inst_points=64, seed=51).items()}
    classes = ["sphere", "plane", "cylinder", "cone"]
    mult = dict(miou=1.0, normal=1.0, type=1.0, parameter=1.0, residue=1.0, total=1.0)
    gY = torch.from_numpy(g["gY"]).to(dev())
    Ya = torch.from_numpy(g["Y"]).to(dev()).requires_grad_(True)
    out = fused_losses.fused_losses(batch["P"], Ya, batch, mult, classes)
    np.testing.assert_allclose([float(v) for v in out[:6]], g["losses"], rtol=1e-4, atol=1e-6)
    out[0].backward()
    err = float((Ya.grad - gY).norm() / gY.norm())
    assert err < 1e-3, err
    # per-head blocks separately (normals | type logits | membership logits): none may hide behind the largest
    for lo, hi in ((0, 3), (3, 7), (7, 35)):
        e = float((Ya.grad[..., lo:hi] - gY[..., lo:hi]).norm() / gY[..., lo:hi].norm())
        assert e < 1e-3, (lo, hi, e)
    W = torch.softmax(Ya.detach()[..., 7:], dim=2)
    S = fused_losses.SegStats.apply(W, batch["I_gt"])
    assert np.array_equal(fused_losses.hungarian_device(S, fused_losses.count_gt(batch["I_gt"])).cpu().numpy(),
                          g["match"].astype(np.int64))
    Yb = torch.from_numpy(g["Y"]).to(dev()).requires_grad_(True)
    X = torch.nn.functional.normalize(Yb[..., :3], p=2, dim=2, eps=1e-12)
    Wb = torch.softmax(Yb[..., 7:], dim=2)
    gt = {"plane_normal": batch["plane_n_gt"], "cylinder_axis": batch["cylinder_axis_gt"], "cone_axis": batch["cone_axis_gt"]}
    out_r = li.compute_all_losses(batch["P"], Wb, batch["I_gt"], X, batch["X_gt"], Yb[..., 3:7], batch["T_gt"], gt,
                                  batch["points_per_instance"], 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, False, mode_seg="mIoU",
                                  classes=classes)
    np.testing.assert_allclose([float(v) for v in out_r[:6]], g["losses"], rtol=1e-4, atol=1e-6)
    out_r[0].backward()
    assert float((Yb.grad - gY).norm() / gY.norm()) < 1e-3


def test_fused_losses_match_reference_shaped_losses():
    """The fused HIP loss section (csrc/losses.hip) against the op-by-op
    `losses_implementation.compute_all_losses` on the same packed heads: six scalars and the
    gradient w.r.t. the heads."""
    from cpfn_amd.SPFN import fused_losses, losses_implementation as li
    B, N, K = 3, 2048, 28
    batch = {k: v.to(dev()) for k, v in synthetic.training_batch(B, N=N, n_prims=7, n_inst_points=128, seed=9).items()}
    g = torch.Generator().manual_seed(4)
    Y0 = torch.randn(B, N, 7 + K, generator=g)
    Y0[:, :, 7:].scatter_add_(2, batch["I_gt"].cpu().unsqueeze(2), torch.full((B, N, 1), 4.0))   # peaky memberships
    Y0[:, :, :3] += 2 * batch["X_gt"].cpu()
    classes = ["sphere", "plane", "cylinder", "cone"]
    mult = dict(miou=1.0, normal=1.0, type=1.0, parameter=1.0, residue=1.0, total=1.0)
    Ya = Y0.to(dev()).requires_grad_(True)
    out_f = fused_losses.fused_losses(batch["P"], Ya, batch, mult, classes)
    out_f[0].backward()
    Yb = Y0.to(dev()).requires_grad_(True)
    X = torch.nn.functional.normalize(Yb[..., :3], p=2, dim=2, eps=1e-12)
    W = torch.softmax(Yb[..., 7:], dim=2)
    gt = {"plane_normal": batch["plane_n_gt"], "cylinder_axis": batch["cylinder_axis_gt"], "cone_axis": batch["cone_axis_gt"]}
    out_r = li.compute_all_losses(batch["P"], W, batch["I_gt"], X, batch["X_gt"], Yb[..., 3:7], batch["T_gt"], gt,
                                  batch["points_per_instance"], 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, False, mode_seg="mIoU",
                                  classes=classes)
    out_r[0].backward()
    for a, b in zip(out_f[:6], out_r[:6]):
        assert abs(float(a) - float(b)) <= 2e-4 * abs(float(b)) + 1e-6, (float(a), float(b))
    err = float((Ya.grad - Yb.grad).norm() / Yb.grad.norm())
    assert err < 2e-3, err
    m1 = fused_losses.hungarian_from_stats(fused_losses.SegStats.apply(W.detach(), batch["I_gt"]), batch["I_gt"])
    assert torch.equal(m1, li.hungarian_matching(W.detach(), batch["I_gt"]))


@pytest.mark.parametrize("mult", [dict(miou=0.7, normal=1.3, type=0.5, parameter=2.0, residue=1.5, total=0.9),
                                  dict(miou=1.0, normal=0.0, type=1.0, parameter=0.0, residue=1.0, total=1.0)])
def test_loss_tail_multipliers(mult):
    """cpfn_loss_tail / FitParams with non-unit and zero multipliers against the op-by-op losses."""
    from cpfn_amd.SPFN import fused_losses, losses_implementation as li
    B, N, K = 2, 2048, 28
    batch = {k: v.to(dev()) for k, v in synthetic.training_batch(B, N=N, n_prims=6, n_inst_points=128, seed=21).items()}
    g = torch.Generator().manual_seed(5)
    Y0 = torch.randn(B, N, 7 + K, generator=g)
    Y0[:, :, 7:].scatter_add_(2, batch["I_gt"].cpu().unsqueeze(2), torch.full((B, N, 1), 4.0))
    Y0[:, :, :3] += 2 * batch["X_gt"].cpu()
    classes = ["sphere", "plane", "cylinder", "cone"]
    Ya = Y0.to(dev()).requires_grad_(True)
    out_f = fused_losses.fused_losses(batch["P"], Ya, batch, mult, classes)
    out_f[0].backward()
    Yb = Y0.to(dev()).requires_grad_(True)
    X = torch.nn.functional.normalize(Yb[..., :3], p=2, dim=2, eps=1e-12)
    W = torch.softmax(Yb[..., 7:], dim=2)
    gt = {"plane_normal": batch["plane_n_gt"], "cylinder_axis": batch["cylinder_axis_gt"], "cone_axis": batch["cone_axis_gt"]}
    out_r = li.compute_all_losses(batch["P"], W, batch["I_gt"], X, batch["X_gt"], Yb[..., 3:7], batch["T_gt"], gt,
                                  batch["points_per_instance"], mult["normal"], mult["type"], mult["miou"], mult["residue"],
                                  mult["parameter"], mult["total"], False, mode_seg="mIoU", classes=classes)
    out_r[0].backward()
    for a, b in zip(out_f[:6], out_r[:6]):
        assert abs(float(a) - float(b)) <= 2e-4 * abs(float(b)) + 1e-6, (float(a), float(b))
    err = float((Ya.grad - Yb.grad).norm() / Yb.grad.norm())
    assert err < 2e-3, err


def test_device_assignment_matches_scipy():
    """cpfn_hungarian_match against the reference's host-side SciPy call on the same segmented sums: random soft
    memberships, peaky ones, clouds with few / all instances, exact ties (identical prediction columns, empty
    GT instances)."""
    from scipy.optimize import linear_sum_assignment
    from cpfn_amd.SPFN import fused_losses as fl
    from oracle import lsap
    rng = np.random.default_rng(5)
    B, N, K = 24, 1024, 28
    I = np.zeros((B, N), dtype=np.int64)
    W = np.zeros((B, N, K), dtype=np.float32)
    for b in range(B):
        n = int(rng.integers(1, K + 1)) if b % 5 else K
        I[b] = rng.integers(-1, n, N)
        I[b, :n] = np.arange(n)                                   # every label < n present (no gaps)
        logits = rng.normal(size=(N, K)) * (0.2 if b % 3 == 0 else 3.0)
        if b % 4 == 1:
            logits[:, 5] = logits[:, 9]                            # two identical prediction columns: exact ties
        if b % 4 == 2:
            logits[:] = 0.0                                        # uniform memberships: everything ties
        e = np.exp(logits - logits.max(1, keepdims=True))
        W[b] = e / e.sum(1, keepdims=True)
    Wd, Id = torch.from_numpy(W).to(dev()), torch.from_numpy(I).to(dev())
    S = fl.SegStats.apply(Wd, Id)
    n_gt = fl.count_gt(Id)
    got = fl.hungarian_device(S, n_gt).cpu().numpy()
    want_host = fl.hungarian_from_pack(fl.hungarian_cost_pack(S, Id, n_gt), K).cpu().numpy()       # SciPy
    assert np.array_equal(got, want_host)
    assert np.array_equal(got, lsap.hungarian_from_stats(S.cpu().numpy(), n_gt.cpu().numpy()))     # the oracle restatement
    # the same assignment riding on the fits' first launch (cpfn_fit_moments_fwd_match): identical matching, identical fits
    P = torch.from_numpy(rng.normal(size=(B, N, 3)).astype(np.float32)).to(dev())
    Xn = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=(B, N, 3)).astype(np.float32)).to(dev()), dim=2)
    mult = dict(miou=1.0, normal=1.0, type=1.0, parameter=1.0, residue=1.0, total=1.0)
    params_sep = fl.fit_params(P, Wd, Xn, mult)
    params, match = fl.fit_params_and_match(P, Wd, Xn, mult, S, n_gt)
    assert np.array_equal(match.cpu().numpy(), got)
    assert torch.equal(params, params_sep)
    from cpfn_amd.SPFN import moments
    assert moments.pending_match_rider() is None
    # K beyond the fused kernels: nothing takes the rider, the assignment falls back to its own launch
    off = dict(mult, residue=0.0, parameter=0.0)
    params0, match0 = fl.fit_params_and_match(P, Wd, Xn, off, S, n_gt)
    assert params0 is None and np.array_equal(match0.cpu().numpy(), got)


@pytest.mark.parametrize("K", [33, 49, 64])
def test_device_assignment_on_merged_label_sets(K):
    """cpfn_hungarian_match beyond the fused loss kernels' 32 columns (evaluation_localSPFN.py:129-131 merges 21 local + 28
    global predictions): against SciPy on the same segmented sums, ties included."""
    from cpfn_amd.SPFN import fused_losses as fl
    from oracle import lsap
    rng = np.random.default_rng(K)
    B, N = 6, 2048
    I = np.zeros((B, N), dtype=np.int64)
    W = np.zeros((B, N, K), dtype=np.float32)
    for b in range(B):
        n = int(rng.integers(1, K + 1)) if b % 3 else K
        I[b] = rng.integers(-1, n, N)
        I[b, :n] = np.arange(n)
        logits = rng.normal(size=(N, K)) * (0.3 if b % 2 else 3.0)
        if b == 1:
            logits[:, 7] = logits[:, K - 1]
        if b == 2:
            logits[:] = 0.0
        e = np.exp(logits - logits.max(1, keepdims=True))
        W[b] = e / e.sum(1, keepdims=True)
    Wd, Id = torch.from_numpy(W).to(dev()), torch.from_numpy(I).to(dev())
    S = fl.SegStats.apply(Wd, Id)
    n_gt = fl.count_gt(Id)
    got = fl.hungarian_device(S, n_gt).cpu().numpy()
    assert np.array_equal(got, fl.hungarian_from_pack(fl.hungarian_cost_pack(S, Id, n_gt), K).cpu().numpy())     # SciPy
    assert np.array_equal(got, lsap.hungarian_from_stats(S.cpu().numpy(), n_gt.cpu().numpy()))


@pytest.mark.parametrize("K,levels", [(28, 3), (28, 0), (64, 2), (7, 2)])
def test_device_assignment_many_random_and_tie_heavy_problems(K, levels):
    """256 assignment problems per case straight on the segmented sums S (no point clouds): random counts, quantised to a
    few integer levels so that equal costs abound (levels > 0), random numbers of GT instances — against the oracle's
    restatement of SciPy's solver (pinned against SciPy itself in the CPU suite), which must agree on EVERY optimal
    assignment it picks, ties included."""
    from cpfn_amd.SPFN import fused_losses as fl
    from oracle import lsap
    rng = np.random.default_rng(100 * K + levels)
    B = 256
    S = np.zeros((B, K + 2, K), dtype=np.float32)
    n_gt = rng.integers(0, K + 1, B).astype(np.int64)
    n_gt[:8] = K
    for b in range(B):
        D = rng.random((K, K)) * 50 if levels == 0 else rng.integers(0, levels + 1, (K, K)) * 10.0
        S[b, :K] = D                                   # sum_n [I = l] W[n, k]
        S[b, K] = D.sum(0) + rng.integers(0, 3, K) * (10.0 if levels else 1.0)   # sum_n W[n, k]  (>= the labelled part)
        S[b, K + 1] = np.maximum(D.sum(1) / 10.0, 1.0).round()                    # points per label
    Sd = torch.from_numpy(S).to(dev())
    got = fl.hungarian_device(Sd, torch.from_numpy(n_gt).to(dev())).cpu().numpy()
    want = lsap.hungarian_from_stats(S, n_gt)
    assert np.array_equal(got, want), int((got != want).any(1).sum())


def test_device_assignment_gives_up_on_non_finite_costs():
    """NaN / inf in the segmented sums (SciPy raises "matrix contains invalid numeric entries"): the device solver must not
    spin or read outside its tables — it returns the identity for that cloud's GT rows, and solves the other clouds."""
    from cpfn_amd.SPFN import fused_losses as fl
    rng = np.random.default_rng(9)
    B, N, K = 4, 512, 28
    I = rng.integers(0, K, (B, N))
    I[:, :K] = np.arange(K)
    W = rng.random((B, N, K)).astype(np.float32)
    W /= W.sum(2, keepdims=True)
    Wd, Id = torch.from_numpy(W).to(dev()), torch.from_numpy(I).to(dev())
    S = fl.SegStats.apply(Wd, Id).clone()
    n_gt = fl.count_gt(Id)
    good = fl.hungarian_device(S, n_gt).cpu().numpy()
    S[1, 3, 5] = float("nan")
    S[2, 0, 0] = float("inf")
    got = fl.hungarian_device(S, n_gt).cpu().numpy()
    assert np.array_equal(got[0], good[0]) and np.array_equal(got[3], good[3])
    assert np.array_equal(got[1], np.arange(K)) and np.array_equal(got[2], np.arange(K))


def test_heads_gradient_hint_is_bit_identical_to_the_colsum_launch(monkeypatch):
    """The heads post-processing backward (cpfn_head_post_bwd) also leaves what the fc2 heads' backward makes of its result
    first — zero-padded bf16 rows and per-256-row column sums — so that cpfn_colsum_f32 is not launched: every parameter
    gradient of a training step must have the same bits with and without that hand-over."""
    from cpfn_amd import lib as _l, synthetic, training
    from cpfn_amd.PointNet2 import pn2_network
    from cpfn_amd.SPFN import fused_losses as fl
    dev = torch.device("cuda:0")
    batch = {k: v.to(dev) for k, v in synthetic.training_batch(2, N=2048, n_prims=6, n_inst_points=128, seed=3).items()}
    starts = (torch.tensor([5, 17]), torch.tensor([1, 300]))
    res = {}
    for hint in (True, False):
        monkeypatch.setattr(fl, "HEADS_HINT", hint)
        torch.manual_seed(0)
        model = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 28]).to(dev)
        model.set_compute_dtype(torch.bfloat16)
        model.dropout_p = 0.0
        tr = training.SPFNTrainer(model, batch_size=2)
        tr.bucket.zero()
        _l.byte_census(True)
        out = tr.losses(batch, fps_start=starts)
        out[0].backward()
        census = _l.byte_census(False)
        assert ("cpfn_colsum_f32" in census) == (not hint), sorted(census)
        res[hint] = [None if p.grad is None else p.grad.clone() for p in model.parameters()]
    for a, b in zip(res[True], res[False]):
        assert (a is None and b is None) or torch.equal(a, b)


def _step_grads(model, batch, starts, aux=None, mult=None):
    """One forward + losses + backward of the trainer's eager path; aux(model outputs) -> extra scalar added to the total."""
    from cpfn_amd import training
    tr = training.SPFNTrainer(model, batch_size=batch["P"].shape[0], multipliers=mult)
    model.return_point_features = True
    tr.bucket.zero()
    out = tr.losses(batch, fps_start=starts)
    total = out[0]
    if aux is not None:
        total = total + aux(model)
    total.backward()
    return [None if p.grad is None else p.grad.clone() for p in model.parameters()], [float(o) for o in out]


def _fresh_model(K, seed=0):
    from cpfn_amd.PointNet2 import pn2_network
    torch.manual_seed(seed)
    m = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, K]).to(torch.device("cuda:0"))
    m.set_compute_dtype(torch.bfloat16)
    return m


@pytest.mark.parametrize("which", ["heads", "features"])
def test_hand_overs_are_dropped_when_a_second_consumer_adds_to_the_gradient(which, monkeypatch):
    """the side results one backward node hands to the next (padded gradient rows + column sums of the
    packed heads; fc1's BatchNorm pass-1 partials) are computed from ONE incoming gradient.  With a second consumer of the
    packed heads (an auxiliary loss on their slices) or of fc1's output (the per-point features), autograd adds the second
    gradient into the first tensor in place — same address — and a hand-over keyed by the address alone would silently drop
    it.  The hand-over also carries the tensor's in-place version: the result must equal the run with the hand-overs off."""
    from cpfn_amd import fused_mlp, synthetic
    from cpfn_amd.SPFN import fused_losses as fl
    dev = torch.device("cuda:0")
    batch = {k: v.to(dev) for k, v in synthetic.training_batch(2, N=2048, n_prims=6, n_inst_points=128, seed=3).items()}
    starts = (torch.tensor([5, 17]), torch.tensor([1, 300]))
    if which == "heads":
        aux = lambda m: (m.heads_packed[..., :7] ** 2).sum() * 1e-3 + m.heads_packed[..., 7:].abs().sum() * 1e-4
    else:
        aux = None        # (set below: needs the model's outputs)
    res = {}
    for on in (True, False):
        monkeypatch.setattr(fl, "HEADS_HINT", on)
        monkeypatch.setattr(fused_mlp, "HEADS_RIDE", on)
        model = _fresh_model(28)
        torch.manual_seed(5)                       # dropout base seed: the same mask both times
        if which == "features":
            outs = {}
            orig = model._forward

            def keep(*a, _o=orig, _d=outs, **k):
                r = _o(*a, **k)
                _d["feat"] = r[4]
                return r
            model._forward = keep
            aux = lambda m, _d=outs: (_d["feat"].float() ** 2).sum() * 1e-3
        res[on] = _step_grads(model, batch, starts, aux)
    for a, b in zip(res[True][0], res[False][0]):
        assert (a is None and b is None) or torch.equal(a, b)
    # ... and the auxiliary gradient is really in there (the test would not see a dropped one otherwise)
    model = _fresh_model(28)
    torch.manual_seed(5)
    plain, _ = _step_grads(model, batch, starts, None)
    assert any(a is not None and not torch.equal(a, b) for a, b in zip(res[True][0], plain))


def test_two_models_with_interleaved_backward_passes_match_their_solo_runs():
    """VERDICT r3 #8a: the hand-over state between backward nodes is owned by the forward pass of ONE model (fused_mlp.HandOver),
    not by the module: GlobalSPFN (K = 28) and LocalSPFN (K = 21) run forward, forward, backward, backward in one process and
    both get the gradients of their solo runs, bit for bit (same launches: nothing falls back because of the other model)."""
    from cpfn_amd import lib as _l, synthetic
    dev = torch.device("cuda:0")
    starts = (torch.tensor([5, 17]), torch.tensor([1, 300]))
    local_mult = dict(miou=1.0, normal=1.0, type=1.0, parameter=0.0, residue=0.0, total=1.0)
    cfgs = {"g": (28, None, synthetic.training_batch(2, N=2048, n_max_instances=28, n_prims=6, n_inst_points=128, seed=3)),
            "l": (21, local_mult, synthetic.training_batch(2, N=2048, n_max_instances=21, n_prims=5, n_inst_points=128, seed=4))}
    cfgs = {k: (K, m, {kk: v.to(dev) for kk, v in b.items()}) for k, (K, m, b) in cfgs.items()}
    solo, census_solo = {}, {}
    for name, (K, mult, batch) in cfgs.items():
        model = _fresh_model(K, seed=1)
        torch.manual_seed(9)
        _l.byte_census(True)
        solo[name] = _step_grads(model, batch, starts, None, mult)
        census_solo[name] = {k: v[0] for k, v in _l.byte_census(False).items()}
    from cpfn_amd import training
    models = {n: _fresh_model(K, seed=1) for n, (K, _, _) in cfgs.items()}
    trs, totals = {}, {}
    _l.byte_census(True)
    for n, (K, mult, batch) in cfgs.items():                     # forward g, forward l
        torch.manual_seed(9)
        trs[n] = training.SPFNTrainer(models[n], batch_size=2, multipliers=mult)
        models[n].return_point_features = True
        trs[n].bucket.zero()
        totals[n] = trs[n].losses(batch, fps_start=starts)[0]
    for n in ("g", "l"):                                        # backward g, backward l (the other model's forward lies between)
        totals[n].backward()
    census = {k: v[0] for k, v in _l.byte_census(False).items()}
    for n in cfgs:
        got = [None if p.grad is None else p.grad for p in models[n].parameters()]
        for a, b in zip(got, solo[n][0]):
            assert (a is None and b is None) or torch.equal(a, b), n
    # the same launches as the two solo runs together: no hand-over was refused
    both = {k: census_solo["g"].get(k, 0) + census_solo["l"].get(k, 0) for k in set(census_solo["g"]) | set(census_solo["l"])}
    assert census == both, sorted((k, census.get(k), both.get(k)) for k in set(census) | set(both) if census.get(k) != both.get(k))


def test_skip_join_gives_the_gradients_of_the_framework_add(monkeypatch):
    """sa1's features feed sa2's grouping AND sfp2's skip concatenation (PointNet2/pn2_network.py:45-46,55): autograd adds
    their two gradients with a framework bf16 add between the two backward nodes.  autograd_ops.SkipJoin hands the skip's
    gradient to the grouping adjoint instead, which adds it inside its own launch with the same roundings: every parameter
    gradient of a training step must have the same bits both ways, and the framework add must be gone.
    Round 6: sa2's features (sa3's input rows + sfp1's skip, :48-49,56) the same way — that sum is formed by the first launch of
    sa2's own backward (cpfn_bn_relu_bwd_join) — so BOTH framework adds of a step are gone."""
    from cpfn_amd import autograd_ops, lib as _l, synthetic
    dev = torch.device("cuda:0")
    batch = {k: v.to(dev) for k, v in synthetic.training_batch(2, N=2048, n_prims=6, n_inst_points=128, seed=3).items()}
    starts = (torch.tensor([5, 17]), torch.tensor([1, 300]))
    res, adds = {}, {}
    for on in (True, False):
        monkeypatch.setattr(autograd_ops, "SKIP_JOIN", on)
        model = _fresh_model(28)
        torch.manual_seed(5)
        n_add = [0]

        class Count(torch.utils._python_dispatch.TorchDispatchMode):
            def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                if func in (torch.ops.aten.add.Tensor, torch.ops.aten.add_.Tensor) and args[0].dtype == torch.bfloat16:
                    n_add[0] += 1
                return func(*args, **(kwargs or {}))
        with Count():
            res[on] = _step_grads(model, batch, starts, None)
        adds[on] = n_add[0]
    for a, b in zip(res[True][0], res[False][0]):
        assert (a is None and b is None) or torch.equal(a, b)
    assert res[True][1] == res[False][1]
    assert adds[True] == adds[False] - 2, adds


def test_sa3_pass1_rides_on_the_broadcast_adjoint(monkeypatch):
    """fused_mlp.TopRide: the gradient of sa3's global feature vector is the column sum of sfp1's broadcast adjoint, one row per
    cloud; BatchNorm-backward pass 1 of sa3's last layer is taken from those rows as they are stored (cpfn_colsum_rows_pass1_bf16:
    one partial row per cloud) instead of by a one-workgroup launch that reads them back.  The same sums in another grouping:
    losses identical, every parameter gradient within the tolerance of the other riding reductions, one cpfn_bn_relu_bwd launch
    fewer; the kernel itself against cpfn_colsum_rows_bf16 + cpfn_bn_relu_bwd bit for bit."""
    from cpfn_amd import fused_mlp, lib as _l, synthetic
    dev = torch.device("cuda:0")
    h = _l.lib()
    # kernel level
    g = torch.Generator().manual_seed(8)
    B, N, C1, C2 = 5, 128, 256, 1024
    wide = torch.randn(B, N, C1 + C2, generator=g).to(dev).to(torch.bfloat16)
    yarg = torch.randn(B, C2, generator=g).to(dev).to(torch.bfloat16)
    scale, shift = (torch.rand(C2, generator=g) - 0.3).to(dev), (torch.randn(C2, generator=g) * 0.2).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    gcol = wide[:, :, C1:]
    ref = torch.empty(B, C2, dtype=torch.bfloat16, device=dev)
    got = torch.empty_like(ref)
    part = torch.empty(B, 2, C2, device=dev)
    _l.check(h.cpfn_colsum_rows_bf16(gcol.data_ptr(), C1 + C2, B, N, C2, ref.data_ptr(), st), "colsum")
    _l.check(h.cpfn_colsum_rows_pass1_bf16(gcol.data_ptr(), C1 + C2, B, N, C2, got.data_ptr(), yarg.data_ptr(), scale.data_ptr(),
                                           shift.data_ptr(), part.data_ptr(), st), "colsum_pass1")
    assert torch.equal(got, ref)
    for b in range(B):            # a single-row cpfn_bn_relu_bwd per cloud is the same arithmetic
        p1 = torch.empty(h.cpfn_bn_bwd_blocks(1), 2, C2, device=dev)
        _l.check(h.cpfn_bn_relu_bwd(ref[b:b + 1].data_ptr(), yarg[b:b + 1].data_ptr(), scale.data_ptr(), shift.data_ptr(), 1, C2, None,
                                    p1.data_ptr(), None, 0.0, st), "bn_relu_bwd")
        assert torch.equal(part[b], p1[0])
    # network level
    batch = {k: v.to(dev) for k, v in synthetic.training_batch(2, N=2048, n_prims=6, n_inst_points=128, seed=3).items()}
    starts = (torch.tensor([5, 17]), torch.tensor([1, 300]))
    res, n_pass1 = {}, {}
    orig = h.cpfn_bn_relu_bwd
    for on in (True, False):
        monkeypatch.setattr(fused_mlp, "TOP_RIDE", on)
        model = _fresh_model(28)
        torch.manual_seed(5)
        n = [0]

        def spy(*a):
            n[0] += 1
            return orig(*a)
        monkeypatch.setattr(h, "cpfn_bn_relu_bwd", spy)
        res[on] = _step_grads(model, batch, starts, None)
        monkeypatch.setattr(h, "cpfn_bn_relu_bwd", orig)
        n_pass1[on] = n[0]
    assert res[True][1] == res[False][1]
    for a, b in zip(res[True][0], res[False][0]):
        assert (a is None and b is None) or float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)) < 2e-3
    assert n_pass1[True] == n_pass1[False] - 1, n_pass1


def test_skip_join_raises_when_the_grouping_adjoint_never_runs():
    """A gradient handed over and never picked up (a backward pass that stops above sa2's grouping) must not be lost silently."""
    from cpfn_amd import autograd_ops
    dev = torch.device("cuda:0")
    B, N, C, S, k = 2, 512, 128, 128, 16
    feats = torch.randn(B, N, C, device=dev).to(torch.bfloat16).requires_grad_(True)
    idx = torch.randint(0, N, (B, S, k), device=dev, dtype=torch.int32)
    from cpfn_amd import ops
    inv = ops.csr_build(idx, N)
    coarse = torch.randn(B, 1, 64, device=dev).to(torch.bfloat16).requires_grad_(True)
    join = autograd_ops.SkipJoin()
    f = feats * 1.0
    grouped = autograd_ops.GroupConcat.apply(f, None, idx, C, inv[0], inv[1], join)
    cat = autograd_ops.concat_interp(f, coarse, None, None, None, join)
    assert join.armed
    # complete backward pass: both nodes run, the sum arrives
    (grouped.float().sum() + cat.float().sum()).backward(retain_graph=True)
    ref = torch.autograd.grad(grouped.float().sum(), feats, retain_graph=True)[0].float() + 1.0
    assert torch.allclose(feats.grad.float(), ref, atol=0.51, rtol=1e-2)
    # partial backward pass: only the concatenation's branch
    with pytest.raises(RuntimeError, match="SkipJoin"):
        torch.autograd.grad(cat.float().sum(), feats, retain_graph=True)
    assert join.addend is None


def test_split_reductions_riding_on_the_finalize_launches_are_bit_identical(monkeypatch):
    """fused_mlp.REDUCE_RIDE: the queued weight-gradient split reductions leave with the bn_bwd_finalize launches of the backward
    pass (cpfn_bn_bwd_finalize_ride) instead of all waiting for its end — the same arithmetic per buffer: every parameter gradient
    of a training step must have the same bits both ways, and most of the reduction must really have moved."""
    from cpfn_amd import fused_mlp, lib as _l, synthetic
    dev = torch.device("cuda:0")
    batch = {k: v.to(dev) for k, v in synthetic.training_batch(2, N=2048, n_prims=6, n_inst_points=128, seed=3).items()}
    starts = (torch.tensor([5, 17]), torch.tensor([1, 300]))
    h = _l.lib()
    seen = {"ride": 0, "riders": 0, "tail": 0}
    o_ride, o_tail = h.cpfn_bn_bwd_finalize_ride_checked, h.cpfn_multi_split_reduce_checked      # (the entries fused_mlp calls)

    def spy_ride(*a):
        seen["ride"] += 1
        seen["riders"] += a[12]
        return o_ride(*a)

    def spy_tail(*a):
        seen["tail"] += a[1]
        return o_tail(*a)
    monkeypatch.setattr(h, "cpfn_bn_bwd_finalize_ride_checked", spy_ride)
    monkeypatch.setattr(h, "cpfn_multi_split_reduce_checked", spy_tail)
    res, counts = {}, {}
    for on in (True, False):
        monkeypatch.setattr(fused_mlp, "REDUCE_RIDE", on)
        for k in seen:
            seen[k] = 0
        model = _fresh_model(28)
        torch.manual_seed(5)
        res[on] = _step_grads(model, batch, starts, None)
        counts[on] = dict(seen)
    for a, b in zip(res[True][0], res[False][0]):
        assert (a is None and b is None) or torch.equal(a, b)
    assert counts[False]["ride"] == 0 and counts[False]["tail"] >= 15, counts
    assert counts[True]["riders"] + counts[True]["tail"] == counts[False]["tail"], counts      # every buffer reduced exactly once
    assert counts[True]["riders"] >= 12 and counts[True]["tail"] <= 6, counts
