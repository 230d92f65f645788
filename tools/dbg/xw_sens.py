# How sensitive is sa1's first-layer weight gradient to the two statistics routes (seam / finalize), in the riding form and in the
# stand-alone launch?  (round 6: is the riding form's c0 S1 + c1 S2 + c2 S3 fragile, or are these ReLU-mask flips?)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from cpfn_amd import fused_mlp
from test_gpu_fused_mlp import _stack, _run, _rel, dev
P, widths, pool_k = 40 * 512 * 64, [64, 64, 128], 64
convs, bns = _stack(3, widths, seed=11)
g = torch.Generator().manual_seed(P + 1)
xyz = (torch.rand(P, 3, generator=g) * 0.4 - 0.2).to(dev())
gout = torch.randn(P // pool_k, widths[-1], generator=g).to(dev())
res = {}
for ride in (True, False):
    for seams in (True, False):
        fused_mlp.XYZ_WGRAD_RIDE, fused_mlp.ATOMIC_SEAMS = ride, seams
        with fused_mlp.seam_pass(dev(), True):
            res[(ride, seams)] = _run(None, convs, bns, torch.bfloat16, pool_k, xyz, gout)[2]
ref = _run(None, convs, bns, "emulated", pool_k, xyz, gout)[2]
r32 = _run(None, convs, bns, torch.float32, pool_k, xyz, gout)[2]
for k, v in res.items():
    print("ride %-5s seams %-5s: dW0 vs emulated %.3e vs fp32 %.3e | dW1 vs emulated %.3e" % (k[0], k[1], _rel(v[0], ref[0]), _rel(v[0], r32[0]), _rel(v[2], ref[2])))
print("seam vs finalize: ride %.3e  stand-alone %.3e   | ride vs stand-alone: seams %.3e finalize %.3e"
      % (_rel(res[(True, True)][0], res[(True, False)][0]), _rel(res[(False, True)][0], res[(False, False)][0]),
         _rel(res[(True, True)][0], res[(False, True)][0]), _rel(res[(True, False)][0], res[(False, False)][0])))
d = (res[(True, True)][0] - res[(True, False)][0]).abs().flatten()
print("ride seam-vs-finalize: %d of %d entries differ, max abs %.3f, |dW0| max %.1f" % (int((d > 0).sum()), d.numel(), float(d.max()), float(res[(True, True)][0].abs().max())))
