"""Stress: run-to-run determinism of an sa1-like stack (fp32-xyz first layer, 64 -> 64, pooled 64 -> 128) — debugging aid."""
import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_fused_mlp as T
from cpfn_amd import fused_mlp
rec = int(sys.argv[1]) if len(sys.argv) > 1 else 0
runs_n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
fused_mlp.XYZ_RECOMPUTE = bool(rec)
for P, widths, pool_k in ((643 * 64, [64, 64, 128], 64), (40016, [64, 64], None), (8192 * 64, [64, 64, 128], 64)):
    convs, bns = T._stack(3, widths, seed=19)
    g = torch.Generator().manual_seed(P)
    xyz = (torch.rand(P, 3, generator=g) * 0.4 - 0.2).to("cuda")
    gout = torch.randn(P // pool_k if pool_k else P, widths[-1], generator=g).to("cuda")
    names = [n for n, _ in list(convs.named_parameters()) + list(bns.named_parameters())]
    ref = T._run(None, convs, bns, torch.bfloat16, pool_k, xyz, gout)[2]
    bad = {}
    for _ in range(runs_n):
        r = T._run(None, convs, bns, torch.bfloat16, pool_k, xyz, gout)[2]
        for i, n in enumerate(names):
            if ref[i] is not None and not torch.equal(ref[i], r[i]):
                bad[n + "#%d" % i] = bad.get(n + "#%d" % i, 0) + 1
    print("recompute=%d P=%d widths=%s: differing runs per gradient: %s" % (rec, P, widths, bad or "none"))
