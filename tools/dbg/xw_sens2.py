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
cap = {}
orig = fused_mlp._defer_reduction
def spy(ws, out, n, splits, row_in=0, row_out=0, params=(), out_ld=0, coef=None):
    if coef is not None:
        cap[mode] = (ws, coef, out)
    return orig(ws, out, n, splits, row_in, row_out, params, out_ld, coef)
fused_mlp._defer_reduction = spy
res = {}
fused_mlp.XYZ_WGRAD_RIDE = True
for mode in (True, False):
    fused_mlp.ATOMIC_SEAMS = mode
    with fused_mlp.seam_pass(dev(), True):
        res[mode] = _run(None, convs, bns, torch.bfloat16, pool_k, xyz, gout)[2]
torch.cuda.synchronize()
gam = bns[0].weight.detach()
for mode in (True, False):
    ws, coef, out = cap[mode]
    S = ws.double().sum(0)          # [7, 64]
    d = coef[0].double()[:, None] * S[0:3].t() + coef[1].double()[:, None] * S[3:6].t() + coef[2].double()[:, None] * S[6, :3][None, :]
    print("seams", mode, "| kernel combine vs fp64 combine of the same partials: %.3e" % _rel(out.double(), d),
          "| terms: |c0 S1| %.3e |c1 S2| %.3e |c2 S3| %.3e |dW0| %.3e" % (float((coef[0][:, None] * S[0:3].t()).abs().max()), float((coef[1][:, None] * S[3:6].t()).abs().max()),
             float((coef[2][:, None] * S[6, :3][None, :]).abs().max()), float(out.abs().max())))
a, b = cap[True], cap[False]
print("partials seam vs finalize: S1 %.3e S2 %.3e S3 %.3e | coef %.3e %.3e %.3e" % (_rel(a[0][:, 0:3], b[0][:, 0:3]), _rel(a[0][:, 3:6], b[0][:, 3:6]), _rel(a[0][:, 6, :3], b[0][:, 6, :3]),
      _rel(a[1][0], b[1][0]), _rel(a[1][1], b[1][1]), _rel(a[1][2], b[1][2])))
dd = (a[2] - b[2]).abs().max(1)[0]
print("channels whose dW0 differs:", [int(i) for i in torch.nonzero(dd > 0).flatten()], "gamma < 0:", [int(i) for i in torch.nonzero(gam < 0).flatten()])
print("coef1 of those:", a[1][1][dd > 0].tolist()[:6], b[1][1][dd > 0].tolist()[:6])
d1 = (a[0] - b[0]).abs()          # [splits, 7, 64]
bad_splits = torch.nonzero(d1.flatten(1).max(1)[0] > 0).flatten()
print("splits with a difference:", len(bad_splits), "of", d1.shape[0], bad_splits[:20].tolist())
s0 = int(bad_splits[0])
print("split", s0, "rows that differ:", torch.nonzero(d1[s0].max(1)[0] > 0).flatten().tolist(), "channels:", torch.nonzero(d1[s0].max(0)[0] > 0).flatten().tolist()[:20])
print("S1 row0 seam", a[0][s0, 0, :8].tolist()); print("S1 row0 fin ", b[0][s0, 0, :8].tolist())
print("S2 row3 seam", a[0][s0, 3, :8].tolist()); print("S2 row3 fin ", b[0][s0, 3, :8].tolist())
