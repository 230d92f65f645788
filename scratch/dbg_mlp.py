import sys; sys.path.insert(0, '.')
import torch
from cpfn_amd import fused_mlp as fm, lib as _l
from cpfn_amd.ops import _ptr, _stream
dev = torch.device('cuda:0')
torch.manual_seed(0)
def rel(a, b): return float((a.float()-b.float()).norm()/b.float().norm())
for (P, K, N) in [(1000, 128, 64), (4173, 64, 128), (384, 320, 256), (3000, 256, 192), (2048, 1280, 256), (3000,128,64)]:
    A = torch.randn(P, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev)/K**0.5).bfloat16()
    ref = A.float() @ W.float().t()
    Y, part, nblk = fm.gemm(A, W, stats=True)
    print('gemm', P, K, N, 'rel', rel(Y, ref), 'stats sum', rel(part[:,0].sum(0), ref.sum(0)), 'sq', rel(part[:,1].sum(0), (ref**2).sum(0)))
    b = torch.randn(N, device=dev)
    Yf, _, _ = fm.gemm(A, W, bias=b, out_f32=True, n_store=N-29)
    print('   f32+bias rel', rel(Yf, (ref + b)[:, :N-29]))
    # wgrad
    G = torch.randn(P, N, device=dev).bfloat16()
    h = _l.lib()
    splits = h.cpfn_mlp_wgrad_splits(P, N, K)
    ws = torch.empty(splits*N*K, device=dev); dW = torch.empty(N, K, device=dev)
    _l.check(h.cpfn_mlp_wgrad(_ptr(G), N, _ptr(A), K, None, P, N, K, _ptr(ws), _ptr(dW), _stream()), 'wgrad')
    print('   wgrad rel', rel(dW, G.float().t() @ A.float()), 'splits', splits)
# bn pieces
P, C = 3000, 128
Y = torch.randn(P, C, device=dev).bfloat16(); sc = torch.rand(C, device=dev)+0.5; sc[::5] *= -1; sh = torch.randn(C, device=dev)*0.3
out = fm.bn_relu_apply(Y, sc, sh)
print('apply', rel(out, torch.relu(Y.float()*sc+sh)))
Kn = 60
o, arg, yarg = fm.bn_relu_maxpool(Y, sc, sh, Kn)
z = (Y.float()*sc+sh).reshape(P//Kn, Kn, C)
zm, am = z.max(1)
print('pool', rel(o, torch.relu(zm)), 'arg eq', float((arg.long()==am).float().mean()), 'yarg', rel(yarg, torch.gather(Y.float().reshape(P//Kn,Kn,C),1,am.unsqueeze(1)).squeeze(1)))
