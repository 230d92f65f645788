import sys; sys.path.insert(0, '.')
import torch, torch.nn.functional as F
from cpfn_amd import fused_mlp as fm, lib as _l
from cpfn_amd.ops import _ptr, _stream
dev = torch.device('cuda:0'); h = _l.lib()
torch.manual_seed(0)
def rel(a, b): return float((a.float()-b.float()).norm()/b.float().norm())
P, C = 3000, 128
Y = (torch.randn(P, C, device=dev)*2+0.5).bfloat16()
gamma = torch.rand(C, device=dev)+0.5; gamma[::5] *= -1; beta = torch.randn(C, device=dev)*0.3
ga = torch.randn(P, C, device=dev).bfloat16()
# reference (fp32 math on the same bf16-valued inputs)
y = Y.float().requires_grad_(True); g_ = gamma.clone().requires_grad_(True); b_ = beta.clone().requires_grad_(True)
a = F.relu(F.batch_norm(y, None, None, g_, b_, True, 0.1, 1e-5))
(a*ga.float()).sum().backward()
# ours: stats from Y
mean = Y.float().mean(0); var = Y.float().var(0, unbiased=False); rstd = torch.rsqrt(var+1e-5)
scale = gamma*rstd; shift = beta - mean*scale
nblk = h.cpfn_bn_bwd_blocks(P)
part = torch.empty(nblk, 2, C, device=dev); Gy = torch.empty(P, C, device=dev, dtype=torch.bfloat16)
_l.check(h.cpfn_bn_relu_bwd(_ptr(ga), _ptr(Y), _ptr(scale), _ptr(shift), P, C, _ptr(Gy), _ptr(part), _stream()), 'a')
gz_ref = ga.float()*((Y.float()*scale+shift) > 0)
print('Gz', rel(Gy, gz_ref), 'S1', rel(part[:,0].sum(0), gz_ref.sum(0)), 'S2', rel(part[:,1].sum(0), (gz_ref*Y.float()).sum(0)))
dgb = torch.empty(2, C, device=dev); coef = torch.empty(3, C, device=dev)
_l.check(h.cpfn_bn_bwd_finalize(_ptr(part), nblk, C, float(P), _ptr(gamma), _ptr(mean), _ptr(rstd), 1, _ptr(dgb[0]), _ptr(dgb[1]), _ptr(coef), _stream()), 'b')
print('dgamma', rel(dgb[0], g_.grad), 'dbeta', rel(dgb[1], b_.grad))
_l.check(h.cpfn_bn_bwd_apply(_ptr(Gy), _ptr(Y), _ptr(coef), None, None, P, C, _ptr(Gy), _stream()), "c")
print('Gy', rel(Gy, y.grad))
