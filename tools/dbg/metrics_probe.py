#!/usr/bin/env python
"""compute_all_metrics at the cascade's size (1 x 131072 points, 49 merged columns): wall time per call (HIP events,
median) — run under `rocprofv3 --kernel-trace --stats` for the launch list."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cpfn_amd import synthetic
from cpfn_amd.SPFN import metric_implementation as mi

dev = torch.device("cuda:0")
N, K = 131072, 49
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
cloud = synthetic.primitive_cloud(1, N, n_prims=12, noise=0.002, seed=9)
g = torch.Generator().manual_seed(1)
P, X_gt, I_gt = cloud["P"].to(dev), cloud["X_gt"].to(dev), cloud["I_gt"].to(dev)
W = torch.rand(1, N, K, generator=g).to(dev) + 2.0 * torch.nn.functional.one_hot(I_gt, K)
X = torch.nn.functional.normalize(X_gt + 0.2 * torch.randn(1, N, 3, generator=g).to(dev), dim=2)
T = torch.randn(1, N, 4, generator=g).to(dev)
T_gt = torch.zeros(1, K, dtype=torch.long, device=dev)
T_gt[0, :12] = cloud["T_gt"][0].to(dev)
ppi = torch.rand(1, K, 512, 3, generator=g).to(dev)
gt = {k: torch.nn.functional.normalize(torch.randn(1, K, 3, generator=g), dim=2).to(dev) for k in ("plane_normal", "cylinder_axis", "cone_axis")}
fn = lambda: mi.compute_all_metrics(P, X, X_gt, W, I_gt, T, T_gt, ppi, gt, classes=["sphere", "plane", "cylinder", "cone"])
fn(); torch.cuda.synchronize()
ts = []
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); out = fn(); b.record(); b.synchronize()
    ts.append(a.elapsed_time(b))
print("compute_all_metrics 1 x %d x %d: median %.3f ms (min %.3f)" % (N, K, sorted(ts)[len(ts) // 2], min(ts)))
print("mIoU %.4f type %.4f normal %.4f axis %.4f res %.5f/%.5f Sk %s P %s" % tuple(
    [float(o) for o in out[:6]] + [[round(float(s), 4) for s in out[6]], [round(float(s), 4) for s in out[7]]]))
