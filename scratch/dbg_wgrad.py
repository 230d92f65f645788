import sys; sys.path.insert(0, '.')
import torch
from cpfn_amd import lib as _l
from cpfn_amd.ops import _ptr, _stream
dev = torch.device('cuda:0')
h = _l.lib()
def wgrad(G, A):
    P, N = G.shape; K = A.shape[1]
    splits = h.cpfn_mlp_wgrad_splits(P, N, K)
    ws = torch.empty(splits*N*K, device=dev); dW = torch.empty(N, K, device=dev)
    _l.check(h.cpfn_mlp_wgrad(_ptr(G), N, _ptr(A), K, None, P, N, K, _ptr(ws), _ptr(dW), _stream()), 'wgrad')
    return dW
P, N, K = 32, 64, 64
A = (torch.arange(K, device=dev).float()[None, :] + 0*torch.arange(P, device=dev).float()[:, None]).bfloat16()
for (ps, ns) in [(0, 0), (1, 0), (5, 3), (9, 17), (31, 63)]:
    G = torch.zeros(P, N, device=dev).bfloat16(); G[ps, ns] = 1
    A2 = A.clone(); A2[ps] = (torch.arange(K, device=dev).float() + 1).bfloat16()
    dW = wgrad(G, A2)
    nz = dW.abs().sum(1).nonzero().flatten().tolist()
    print('one-hot at p=%d n=%d -> nonzero rows %s; row vals[:8]=%s' % (ps, ns, nz, dW[nz[0], :8].tolist() if nz else None))
