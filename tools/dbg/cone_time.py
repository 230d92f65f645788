"""Time cpfn_cone_pass_fwd / _bwd as their own launches and print a checksum: python tools/dbg/cone_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cpfn_amd import lib as _l

dev = torch.device("cuda:0")
B, N, K = 16, 8192, 28
g = torch.Generator().manual_seed(0)
P = torch.randn(B, N, 3, generator=g).to(dev)
W = torch.softmax(torch.randn(B, N, K, generator=g) * 3.0, 2).to(dev)
apex = torch.randn(B, K, 3, generator=g).to(dev)
axis = torch.nn.functional.normalize(torch.randn(B, K, 3, generator=g), dim=2).to(dev)
g_acos = torch.randn(B, K, generator=g).to(dev)
h = _l.lib()
chunks = h.cpfn_fit_num_chunks(B, N)
ws = torch.empty(chunks * B * K * 6, dtype=torch.float64, device=dev)
out = torch.empty(B, K, 2, dtype=torch.float64, device=dev)
dW = torch.empty_like(W)
d6 = torch.empty(B, K, 6, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
fwd = lambda: _l.check(h.cpfn_cone_pass_fwd(p(P), p(W), p(apex), p(axis), B, N, K, p(ws), p(out), st), "fwd")
bwd = lambda: _l.check(h.cpfn_cone_pass_bwd(p(P), p(W), p(apex), p(axis), p(g_acos), B, N, K, p(dW), p(ws), p(d6), 6, 0, st), "bwd")
for name, fn in (("cpfn_cone_pass_fwd (+ chunk reduce)", fwd), ("cpfn_cone_pass_bwd (+ chunk reduce)", bwd)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%s: %.1f us per call" % (name, e0.elapsed_time(e1) * 1000 / 200))
print("checksum", float(out.sum()), float(dW.double().sum()), float(d6.sum()))
