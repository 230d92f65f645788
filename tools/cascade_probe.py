#!/usr/bin/env python
"""Times the stages of BASELINE.json configs[4] on ONE GPU (replicas only: clouds and patches are independent):
PatchSelection on the 8192-point low-resolution cloud (evaluation_PatchSelection.py:65), GlobalSPFN eval forward on one
131072-point cloud (batch 1, evaluation_globalSPFN.py:62-64), its geometry kernels one by
one, the LocalSPFN eval forward on 32 patches x 8192 points (evaluation_localSPFN.py:95), similarity_soft /
get_point_final, and compute_all_metrics on the merged 49-column label set.  HIP events, median of 5."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cpfn_amd import ops, synthetic
from cpfn_amd.PointNet2 import pn2_network
from cpfn_amd.SPFN import metric_implementation as mi
from cpfn_amd.Utils import merging_utils as mu

dev = torch.device("cuda:0")
N, NB, NPP = 131072, 32, 8192


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r = fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2], r


cloud = synthetic.primitive_cloud(1, N, n_prims=12, noise=0.002, seed=9)
P = cloud["P"].to(dev)
torch.manual_seed(0)
g = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 28]).to(dev).eval()
g.set_compute_dtype(torch.bfloat16)
g.dropout_p = 0.0
l = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 21]).to(dev).eval()
l.set_compute_dtype(torch.bfloat16)
l.dropout_p = 0.0
ps = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[2]).to(dev).eval()
ps.set_compute_dtype(torch.bfloat16)
ps.dropout_p = 0.0
start = torch.zeros(1, dtype=torch.int32, device=dev)
with torch.no_grad():
    P_lo = P[:, ::N // NPP].contiguous()                       # the low-resolution cloud PatchSelection looks at (evaluation_PatchSelection.py:65)
    ps.auto_graph = False
    t, _ = timed(lambda: ps(P_lo, fps_start=(torch.tensor([0]), torch.tensor([0])))); print("PatchSelection eval forward, 1 x 8192, eager launches %6.3f ms" % t)
    ps.auto_graph = True
    t, heat = timed(lambda: ps(P_lo, fps_start=(torch.tensor([0]), torch.tensor([0])))); print("  ... the unedited call model(P): auto-replayed graph  %6.3f ms" % t)
    t, sel = timed(lambda: ops.fps(P, 512, start)); print("FPS 131072 -> 512 (16 workgroups per cloud)         %8.3f ms" % t)
    ctr = ops.gather_rows(P, sel)
    t, _ = timed(lambda: ops.ball_query(ctr, P, 0.2, 64)); print("ball query 512 x 131072                            %8.3f ms" % t)
    t, _ = timed(lambda: ops.three_nn(P, ctr)); print("3-NN 131072 x 512                                  %8.3f ms" % t)
    g.auto_graph = False
    t, out = timed(lambda: g(P, fps_start=(torch.tensor([0]), torch.tensor([0])))); print("GlobalSPFN eval forward, 1 x 131072, eager launches %7.3f ms" % t)
    g.auto_graph = True
    t, out = timed(lambda: g(P, fps_start=(torch.tensor([0]), torch.tensor([0])))); print("  ... the unedited call model(P): auto-replayed graph %6.3f ms" % t)
    from cpfn_amd.inference import GraphedForward
    gg = GraphedForward(g)
    t, _ = timed(lambda: gg(P, fps_start=(torch.tensor([0]), torch.tensor([0])))); print("  ... replayed as one hipGraph (GraphedForward)       %8.3f ms" % t)
    centres = P[0, g.aux_sa1["fps_idx"][0, :NB].long()]
    d2 = ((P[0].unsqueeze(0) - centres.unsqueeze(1)) ** 2).sum(-1)
    pidx = d2.topk(NPP, dim=1, largest=False)[1]
    patches = P[0][pidx]
    patches = patches - patches.mean(1, keepdim=True)
    patches = (patches / patches.norm(dim=2).max(dim=1)[0].view(NB, 1, 1)).contiguous()
    st = (torch.zeros(NB, dtype=torch.long), torch.zeros(NB, dtype=torch.long))
    l.auto_graph = False
    t, lout = timed(lambda: l(patches, fps_start=st)); print("LocalSPFN eval forward, 32 x 8192, eager launches  %8.3f ms" % t)
    l.auto_graph = True
    t, lout = timed(lambda: l(patches, fps_start=st)); print("  ... the unedited call model(P): auto-replayed graph %6.3f ms" % t)
    gl = GraphedForward(l)
    t, _ = timed(lambda: gl(patches, fps_start=st)); print("  ... replayed as one hipGraph (GraphedForward)       %8.3f ms" % t)
    Wg, Wl = torch.softmax(out[2], 2), torch.softmax(lout[2], 2)
    labels = torch.nn.functional.one_hot(Wg[0].argmax(1), 28)
    t, sim = timed(lambda: mu.similarity_soft(labels, Wl, pidx)); print("similarity_soft (700 x 700)                        %8.3f ms" % t)
    C = NB * 21 + 28
    M = torch.zeros(N, C, device=dev)
    for b in range(NB):
        M[pidx[b], b * 21:(b + 1) * 21] = Wl[b]
    M[:, NB * 21:] = labels.float()
    lab = torch.cat([sim[:NB * 21, NB * 21:].argmax(1), torch.arange(28, device=dev)])
    t, Wf = timed(lambda: mu.get_point_final(M, lab)); print("get_point_final                                    %8.3f ms" % t)
    K = 49
    W49 = torch.zeros(1, N, K, device=dev); W49[0, :, :28] = Wf + 2.0 * torch.nn.functional.one_hot(cloud["I_gt"][0].to(dev), 28)
    T_gt = torch.zeros(1, K, dtype=torch.long, device=dev)
    ppi = torch.rand(1, K, 512, 3, device=dev)
    gt = {k: torch.nn.functional.normalize(torch.randn(1, K, 3, device=dev), dim=2) for k in ("plane_normal", "cylinder_axis", "cone_axis")}
    X = torch.nn.functional.normalize(out[0], dim=2)
    t, _ = timed(lambda: mi.compute_all_metrics(P, X, cloud["X_gt"].to(dev), W49, cloud["I_gt"].to(dev), out[1], T_gt, ppi, gt,
                                                classes=["sphere", "plane", "cylinder", "cone"]))
    print("compute_all_metrics, 131072 pts, 49 columns        %8.3f ms" % t)
