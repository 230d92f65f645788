"""Run-to-run determinism of the sa1-like stack's gradients with / without the first-layer recompute (debugging aid)."""
import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_fused_mlp as T
from cpfn_amd import fused_mlp
P, widths, pool_k = 40016, [64, 64], None
convs, bns = T._stack(3, widths, seed=19)
g = torch.Generator().manual_seed(P)
xyz = (torch.rand(P, 3, generator=g) * 0.4 - 0.2).to("cuda")
gout = torch.randn(P, widths[-1], generator=g).to("cuda")
names = [n for n, _ in list(convs.named_parameters()) + list(bns.named_parameters())]
for rec in (False, True):
    fused_mlp.XYZ_RECOMPUTE = rec
    runs = [T._run(None, convs, bns, torch.bfloat16, pool_k, xyz, gout)[2] for _ in range(6)]
    for i, n in enumerate(names):
        if runs[0][i] is None: continue
        d = max((runs[0][i] - r[i]).abs().max().item() for r in runs[1:])
        if d > 0: print("recompute=%s  %s: run-to-run max diff %.3g" % (rec, n, d))
    print("recompute=%s done" % rec)
