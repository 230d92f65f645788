"""GPU: BASELINE.json configs[4] at its real size on one GPU ("replicas only": clouds and patches are independent,
SURVEY §8e) — the cascaded evaluation of evaluation_globalSPFN.py:62-64 (GlobalSPFN on the FULL-resolution cloud,
batch 1, eval mode) followed by evaluation_localSPFN.py:95-110 (32 patches x 8192 points through LocalSPFN, then
similarity_soft / get_point_final on its memberships), every stage against the oracle:

  * PatchSelection (output_sizes=[2], evaluation_PatchSelection.py:45-65) on 16 x 8192 low-resolution clouds: geometry bit-exact,
    heat-map logits of both compute modes vs the oracle; its hottest points are the patch centres of the local stage;
  * 131072-point forward: FPS (several workgroups per cloud), ball query and 3-NN indices / weights BIT-EXACT vs
    oracle/geometry; the heads of both compute modes vs the oracle's evaluation forward (oracle/pn2.py, training=False);
  * 32 x 8192 LocalSPFN eval forward (K = 21): geometry bit-exact vs the oracle, heads of both modes vs the oracle;
  * similarity_soft on those memberships vs the float64 oracle (sparse form of the same matrix), get_point_final vs
    the oracle; the evaluation metrics on the merged (K = 49) membership matrix vs oracle/metrics.

The greedy label solver between similarity_soft and get_point_final is numba host code in the reference and out of
scope; a deterministic stand-in labelling keeps the data flowing."""
import numpy as np
import pytest
import torch

from cpfn_amd import synthetic
from oracle import geometry as og
from oracle import merging as omg
from oracle import metrics as om

pytestmark = pytest.mark.gpu
N_HI, NB, NPP, K_GLOBAL, K_LOCAL = 131072, 32, 8192, 28, 21


def dev():
    return torch.device("cuda:0")


def _net(K, seed, sizes=None):
    from cpfn_amd.PointNet2 import pn2_network
    sizes = [3, 4, K] if sizes is None else list(sizes)
    m = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=sizes)
    m.load_state_dict(synthetic.synthetic_state_dict(synthetic.pointnet2_state_shapes(output_sizes=tuple(sizes)), seed=seed),
                      strict=True)
    m.dropout_p = 0.0                 # (the reference's dropout stays on in eval mode; neutralised to compare modes)
    return m.to(dev()).eval()


def _geometry_vs_oracle(m, xyz, starts):
    f1 = og.farthest_point_sample(xyz, 512, starts[0].numpy())
    assert np.array_equal(m.aux_sa1["fps_idx"].cpu().numpy(), f1.astype(np.int32)), "sa1 FPS"
    l1 = np.take_along_axis(xyz, f1[:, :, None], axis=1)
    assert np.array_equal(m.aux_sa1["ball_idx"].cpu().numpy(), og.ball_query(0.2, 64, xyz, l1).astype(np.int32)), "sa1 ball"
    f2 = og.farthest_point_sample(l1, 128, starts[1].numpy())
    assert np.array_equal(m.aux_sa2["fps_idx"].cpu().numpy(), f2.astype(np.int32)), "sa2 FPS"
    d, i = og.three_nn(xyz, l1)
    assert np.array_equal(m.aux_sfp3["nn_idx"].cpu().numpy(), i.astype(np.int32)), "sfp3 3-NN"
    assert np.array_equal(m.aux_sfp3["nn_w"].cpu().numpy().view(np.uint32), og.three_weights(d).view(np.uint32))


def _both_modes(m, P, starts):
    """Evaluation-mode forward (running statistics, evaluation_globalSPFN.py:60,85 / evaluation_localSPFN.py:95) in both
    compute modes of the product against the ORACLE's evaluation forward on the same weights, cloud and FPS seeds:
    the fp32 mode within 1e-5 relative L2 per head (achieved 2e-7 ... 4e-7: summation order), the bf16 mode within 1e-2
    (achieved 1e-3 ... 5e-3: with running statistics nothing amplifies the operand rounding)."""
    from oracle import pn2 as opn2
    state = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    prev = torch.get_num_threads()
    torch.set_num_threads(16)
    try:
        with torch.no_grad():
            heads, _, _, _ = opn2.pointnet2_forward(state, P.detach().cpu(), (starts[0].numpy(), starts[1].numpy()), training=False)
    finally:
        torch.set_num_threads(prev)
    with torch.no_grad():
        m.set_compute_dtype(torch.float32)
        ref = [t.clone() for t in m(P, fps_start=starts)[:len(heads)]]
        m.set_compute_dtype(torch.bfloat16)
        out = m(P, fps_start=starts)
    for name, a, b, o in zip("XTW" if len(heads) == 3 else "H", out[:len(heads)], ref, heads):
        o = o.to(a.device)
        e32, e16, e = float((b - o).norm() / o.norm()), float((a - o).norm() / o.norm()), float((a - b).norm() / b.norm())
        print("head %s (%d x %d): fp32 mode vs oracle %.2e | bf16 vs oracle %.2e | bf16 vs fp32 mode %.2e" % (name, P.shape[0], P.shape[1], e32, e16, e))
        assert torch.isfinite(a).all() and e32 < 1e-5 and e16 < 1e-2, (name, e32, e16)
    return out


def test_cascaded_eval_131072_points():
    from cpfn_amd.SPFN import metric_implementation as mi
    from cpfn_amd.Utils import merging_utils as mu
    cloud = synthetic.primitive_cloud(1, N_HI, n_prims=12, noise=0.002, seed=9)
    P = cloud["P"].to(dev())
    # ---------------- stage 1: GlobalSPFN on the full-resolution cloud, batch 1 (evaluation_globalSPFN.py:62-64)
    g_net = _net(K_GLOBAL, seed=0)
    starts = (torch.tensor([12345]), torch.tensor([77]))
    out = _both_modes(g_net, P, starts)
    _geometry_vs_oracle(g_net, cloud["P"].numpy(), starts)
    assert out[3].shape == (1, 1024, 1) and out[4].shape == (1, 128, N_HI)
    Xg = torch.nn.functional.normalize(out[0], p=2, dim=2, eps=1e-12)
    Wg = torch.softmax(out[2], dim=2)                                              # [1,N,28]
    spfn_labels = torch.nn.functional.one_hot(Wg[0].argmax(1), K_GLOBAL)           # [N,28] long, as the data loader stores it
    # ---------------- stage 0: PatchSelection (evaluation_PatchSelection.py:45-65): the heat-map network on a batch of
    # 16 low-resolution clouds x 8192 points — cloud 0 is the 8192-point subsample of the cloud above — both compute modes
    # against the oracle; its 32 hottest points are the patch centres of stage 2
    P_lo = torch.cat([P[:, ::N_HI // NPP], synthetic.primitive_cloud(15, NPP, n_prims=9, seed=21)["P"].to(dev())], 0).contiguous()
    ps_net = _net(None, seed=1, sizes=[2])
    gps = torch.Generator().manual_seed(6)
    pstarts = (torch.randint(0, NPP, (16,), generator=gps), torch.randint(0, 512, (16,), generator=gps))
    heat = _both_modes(ps_net, P_lo, pstarts)[0]                                  # [16,8192,2] logits
    _geometry_vs_oracle(ps_net, P_lo.cpu().numpy(), pstarts)
    assert heat.shape == (16, NPP, 2)
    hot = (heat[0, :, 1] - heat[0, :, 0]).topk(NB)[1]                             # points the network rates "small primitive" most
    # ---------------- stage 2: 32 patches x 8192 points (nearest neighbours of the 32 selected centres), LocalSPFN eval
    centres = P_lo[0, hot]                                                        # [32,3]
    d2 = ((P[0].unsqueeze(0) - centres.unsqueeze(1)) ** 2).sum(-1)                # [32,N]
    patch_indices = d2.topk(NPP, dim=1, largest=False)[1]                         # [32,8192] (a set per patch)
    patches = P[0][patch_indices]                                                 # [32,8192,3]
    patches = patches - patches.mean(1, keepdim=True)
    patches = (patches / patches.norm(dim=2).max(dim=1)[0].view(NB, 1, 1)).contiguous()
    l_net = _net(K_LOCAL, seed=3)
    g = torch.Generator().manual_seed(4)
    lstarts = (torch.randint(0, NPP, (NB,), generator=g), torch.randint(0, 512, (NB,), generator=g))
    lout = _both_modes(l_net, patches, lstarts)
    _geometry_vs_oracle(l_net, patches.cpu().numpy(), lstarts)
    Wl = torch.softmax(lout[2], dim=2)                                             # [32,8192,21]
    # ---------------- stage 3: merging (evaluation_localSPFN.py:101-110)
    sim = mu.similarity_soft(spfn_labels, Wl, patch_indices)
    C = NB * K_LOCAL + K_GLOBAL
    assert sim.shape == (C, C)
    want = omg.similarity_soft_sparse(spfn_labels.cpu().numpy(), Wl.cpu().numpy(), patch_indices.cpu().numpy())
    scale = np.abs(want).max()
    err = np.abs(sim.cpu().numpy() - want).max() / scale
    print("similarity_soft vs float64 oracle: max err / max %.2e" % err)
    assert err < 2e-5
    # stand-in for the host solver: every local column joins the global label it overlaps most, global columns keep theirs
    labels = torch.cat([sim[:NB * K_LOCAL, NB * K_LOCAL:].argmax(1), torch.arange(K_GLOBAL, device=dev())])
    M = torch.zeros(N_HI, C, device=dev())
    for b in range(NB):
        M[patch_indices[b], b * K_LOCAL:(b + 1) * K_LOCAL] = Wl[b]
    M[:, NB * K_LOCAL:] = spfn_labels.float()
    flag = M[:, :NB * K_LOCAL].sum(1) > 0
    M[flag, NB * K_LOCAL:] = 0                                                    # evaluation_localSPFN.py:107-109
    W_fusion = mu.get_point_final(M, labels)
    wf = omg.get_point_final(M.cpu().numpy(), labels.cpu().numpy())
    np.testing.assert_allclose(W_fusion.cpu().numpy(), wf, rtol=1e-5, atol=1e-6)
    # ---------------- stage 4: metrics on a merged label set wider than 32 columns (local + global = 49)
    # (the networks are untrained: their merged memberships cut the cloud into arbitrary blobs whose cone / cylinder fits
    #  are ill-conditioned — the fp32 oracle itself moves by several per cent against its fp64 run there.  The GT labels
    #  are mixed in so that the hard instances are the cloud's real primitives and the comparison means something.)
    Kf = K_LOCAL + K_GLOBAL
    I_gt = cloud["I_gt"].to(dev())
    Wf = torch.zeros(1, N_HI, Kf, device=dev())
    Wf[0, :, :K_GLOBAL] = W_fusion + 2.0 * torch.nn.functional.one_hot(I_gt[0], K_GLOBAL)
    Wf[0, :, K_GLOBAL:] = 1e-3 * torch.rand(N_HI, K_LOCAL, device=dev())
    T_gt = torch.zeros(1, Kf, dtype=torch.long, device=dev())
    T_gt[:, :12] = cloud["T_gt"].to(dev())
    ppi = torch.zeros(1, Kf, 512, 3, device=dev())
    for k in range(12):
        idx = (I_gt[0] == k).nonzero().squeeze(1)[:512]
        ppi[0, k, :idx.numel()] = P[0, idx]
        ppi[0, k, idx.numel():] = P[0, idx[0]]
    gen = torch.Generator().manual_seed(8)
    gt = {k: torch.nn.functional.normalize(torch.randn(1, Kf, 3, generator=gen), dim=2).to(dev())
          for k in ("plane_normal", "cylinder_axis", "cone_axis")}
    T = out[1]
    CLASSES = ["sphere", "plane", "cylinder", "cone"]
    res = mi.compute_all_metrics(P, Xg, cloud["X_gt"].to(dev()), Wf, I_gt, T, T_gt, ppi, gt, list_epsilon=[0.01, 0.02],
                                 classes=CLASSES)
    c = lambda t: (t.detach().float() if t.dtype.is_floating_point else t.detach()).cpu()
    # The arbiter at this size is the float64 evaluation of the oracle's formulas: over 131072 points the fp32
    # restatement's own summation error shows in the residual statistics (its mean residual sits ~1 % from its fp64 run,
    # printed below), while the product accumulates the fit moments in fp64.
    d = lambda t: t.double() if t.dtype.is_floating_point else t
    ref32 = om.compute_all_metrics(c(P), c(Xg), cloud["X_gt"], c(Wf), cloud["I_gt"], c(T), c(T_gt), c(ppi),
                                   {k: c(v) for k, v in gt.items()}, list_epsilon=[0.01, 0.02], classes=CLASSES)
    ref = om.compute_all_metrics(d(c(P)), d(c(Xg)), d(cloud["X_gt"]), d(c(Wf)), cloud["I_gt"], d(c(T)), c(T_gt), d(c(ppi)),
                                 {k: d(c(v)) for k, v in gt.items()}, list_epsilon=[0.01, 0.02], classes=CLASSES)
    print("fp32 oracle vs fp64 oracle: mean_res %.6f / %.6f, std_res %.6f / %.6f" % (
        float(ref32["mean_residual"]), float(ref["mean_residual"]), float(ref32["std_residual"]), float(ref["std_residual"])))
    assert np.array_equal(res[10].cpu().numpy(), ref["T_instance"].numpy())
    # (axis difference is left out HERE: its denominator sums the axis loss over ALL K slots, metric_implementation.py:183,
    #  and 37 of the 49 hard instances of this cloud are empty — their "axes" are eigenvectors of a zero matrix, arbitrary
    #  in the reference too.  It is asserted where every slot is populated: tests/test_gpu_metrics.py.)
    print("metric axis     product %.6f oracle %.6f (not asserted: empty slots)" % (float(res[3]), float(ref["axis_difference"])))
    for name, a, b in (("mIoU", res[0], ref["mIoU"]), ("type", res[1], ref["type_accuracy"]), ("normal", res[2], ref["normal_difference"]),
                       ("mean_res", res[4], ref["mean_residual"]), ("std_res", res[5], ref["std_residual"])):
        print("metric %-8s product %.6f oracle %.6f" % (name, float(a), float(b)))
        np.testing.assert_allclose(a.cpu().numpy(), b.float().numpy(), rtol=2e-3, atol=1e-5, err_msg=name)
    np.testing.assert_allclose(torch.stack(res[6]).cpu().numpy(), ref["Sk_coverage"].float().numpy(), atol=3.0 / 512)
    np.testing.assert_allclose(torch.stack(res[7]).cpu().numpy(), ref["P_coverage"].float().numpy(), atol=20.0 / N_HI)


def test_graphed_evaluation_forward_matches_eager():
    """cpfn_amd.inference.GraphedForward: the evaluation forward of a 131072-point cloud (batch 1) and of 32 x 8192
    patches replayed as one hipGraph — bit-identical to the eager forward, same FPS points for the same seed."""
    from cpfn_amd.inference import GraphedForward
    for K, shape, seed in ((K_GLOBAL, (1, N_HI), 0), (K_LOCAL, (NB, NPP), 3)):
        m = _net(K, seed=seed).set_compute_dtype(torch.bfloat16)
        P = synthetic.primitive_cloud(shape[0], shape[1], n_prims=8, seed=11)["P"].to(dev())
        gf = GraphedForward(m)
        with torch.no_grad():
            m.auto_graph = False                     # the plain eager forward is the reference point
            torch.manual_seed(5)
            want = [t.clone() for t in m(P)]
            fps_eager = m.aux_sa1["fps_idx"].clone()
            m.auto_graph = True
            # ... the module itself replays a graph for an evaluation forward under no_grad (what the reference's unedited
            # evaluation scripts call, evaluation_globalSPFN.py:85): same bits, outputs that do not alias the graph's buffers
            # — from the SECOND sighting of a shape on (the first runs eager launches: a shape that never returns must not cost
            # a warm-up + capture + replay)
            for rep in range(3):
                torch.manual_seed(5)
                auto = m(P)
                if rep == 0:
                    assert "_auto_graph" not in m.__dict__
                else:
                    assert len(m.__dict__["_auto_graph"]._graphs) == 1
                for a, b in zip(auto, want):
                    assert torch.equal(a, b)
                assert torch.equal(m.aux_sa1["fps_idx"], fps_eager)
            keep = auto[2].clone()
            torch.manual_seed(6)
            m(P)                                     # a second call must not overwrite what the first returned
            assert torch.equal(auto[2], keep)
            # weights updated IN PLACE (an optimizer step between two evaluations) are seen by the replayed graph: it reads the
            # parameters where they lie and refreshes its bf16 panels inside the graph
            m.fc2[0].weight.mul_(1.5)
            m.sa1.bn_blocks[0][0].running_mean.add_(0.01)
            torch.manual_seed(7)
            after = m(P)
            m.auto_graph = False
            torch.manual_seed(7)
            after_eager = m(P)
            m.auto_graph = True
            assert not torch.equal(after[0], auto[0])
            for a, b in zip(after, after_eager):
                assert torch.equal(a, b)
            m.fc2[0].weight.div_(1.5)
            m.sa1.bn_blocks[0][0].running_mean.sub_(0.01)
            for rep in range(2):                     # capture, then a pure replay
                torch.manual_seed(5)
                got = gf(P)
                for a, b in zip(got, want):
                    assert torch.equal(a, b)
                assert torch.equal(m.aux_sa1["fps_idx"], fps_eager)
            got2 = gf(P, fps_start=(torch.zeros(shape[0], dtype=torch.long), torch.ones(shape[0], dtype=torch.long)))
            m.auto_graph = False
            want2 = m(P, fps_start=(torch.zeros(shape[0], dtype=torch.long), torch.ones(shape[0], dtype=torch.long)))
            for a, b in zip(got2, want2):
                assert torch.equal(a, b)
    m.train()
    with pytest.raises(RuntimeError, match="evaluation-mode"):
        gf(P)
