"""Time cpfn_fit_moments_fwd (moments kernel + chunk reduction) and cpfn_fit_moments_bwd as their own launches:
python tools/dbg/moments_time.py [B N K]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cpfn_amd import lib as _l

dev = torch.device("cuda:0")
B, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (16, 8192, 28)
g = torch.Generator().manual_seed(0)
P = torch.randn(B, N, 3, generator=g).to(dev)
X = torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=2).to(dev)
W = torch.softmax(torch.randn(B, N, K, generator=g) * 3.0, 2).to(dev)
h = _l.lib()
chunks = h.cpfn_fit_num_chunks(B, N)
ws = torch.empty(chunks * B * K * 52, dtype=torch.float64, device=dev)
M = torch.empty(B, K, 52, dtype=torch.float64, device=dev)
G32 = torch.randn(B, K, 52, generator=g).to(dev)
dW, dX = torch.empty_like(W), torch.empty_like(X)
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()


def fwd():
    _l.check(h.cpfn_fit_moments_fwd(p(P), p(X), p(W), B, N, K, p(ws), p(M), st), "fwd")


def bwd():
    _l.check(h.cpfn_fit_moments_bwd(p(P), p(X), p(W), p(G32), B, N, K, None, p(dW), p(dX), st), "bwd")


for name, fn in (("cpfn_fit_moments_fwd (+ chunk reduce)", fwd), ("cpfn_fit_moments_bwd", bwd)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%s: %.1f us per call (B=%d N=%d K=%d, %d chunks)" % (name, e0.elapsed_time(e1) * 1000 / 200, B, N, K, chunks))
print("checksum", float(M.sum()), float(dW.sum()), float(dX.sum()))
