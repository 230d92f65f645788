import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from cpfn_amd import fused_mlp
from test_gpu_fused_mlp import _stack, _run, _rel, dev
P, widths, pool_k = 40 * 512 * 64, [64, 64, 128], 64
convs, bns = _stack(3, widths, seed=11)
g = torch.Generator().manual_seed(P + 1)
xyz = (torch.rand(P, 3, generator=g) * 0.4 - 0.2).to(dev())
gout = torch.randn(P // pool_k, widths[-1], generator=g).to(dev())
cap, ys = {}, {}
orig = fused_mlp._defer_reduction
def spy(ws, out, n, splits, row_in=0, row_out=0, params=(), out_ld=0, coef=None):
    if coef is not None:
        cap[mode] = (ws, coef, out)
    return orig(ws, out, n, splits, row_in, row_out, params, out_ld, coef)
fused_mlp._defer_reduction = spy
of = fused_mlp._FusedStack.forward
def fspy(ctx, *a):
    o = of(ctx, *a)
    ys[mode] = ctx.saved[0][2]
    return o
fused_mlp._FusedStack.forward = staticmethod(fspy)
fused_mlp.XYZ_WGRAD_RIDE = True
for mode in ("seamA", "seamB", "fin"):
    fused_mlp.ATOMIC_SEAMS = mode != "fin"
    with fused_mlp.seam_pass(dev(), True):
        _run(None, convs, bns, torch.bfloat16, pool_k, xyz, gout)
torch.cuda.synchronize()
print("Y0 identical across modes:", torch.equal(ys["seamA"], ys["fin"]))
S2ref = ys["fin"].double().t() @ xyz.double()          # [64, 3]
for mode in ("seamA", "seamB", "fin"):
    S = cap[mode][0].double().sum(0)
    e2 = (S[3:6].t() - S2ref).abs() / S2ref.abs().clamp_min(1.0)
    print(mode, "S2 vs fp64 reference: max rel %.3e at" % float(e2.max()), torch.nonzero(e2 > 1e-5)[:8].tolist(), "| S3 vs ref %.3e" % float((S[6, :3] - xyz.double().sum(0)).abs().max()))
print("seamA vs seamB partials identical:", torch.equal(cap["seamA"][0], cap["seamB"][0]))
