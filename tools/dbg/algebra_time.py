"""Are the fitters' algebra kernels (23-34 KB of code, executed once per step by a few lanes) bound by instruction fetch?
Run under `rocprofv3 --kernel-trace`: 60 back-to-back launches of cpfn_fit_algebra_fwd / _bwd (code warm in the instruction
caches), then 60 launches each with a 134 MB bf16 GEMM and a handful of other kernels in between (code cold, as in a
training step).  tools/dbg/algebra_time.py prints event timings of the warm loops; the trace gives the per-launch durations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cpfn_amd import lib as _l

dev = torch.device("cuda:0")
B, N, K = 16, 8192, 28
g = torch.Generator().manual_seed(0)
P = torch.randn(B, N, 3, generator=g).to(dev)
X = torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=2).to(dev)
W = torch.softmax(torch.randn(B, N, K, generator=g) * 3.0, 2).to(dev)
h = _l.lib()
chunks = h.cpfn_fit_num_chunks(B, N)
ws = torch.empty(chunks * B * K * 52, dtype=torch.float64, device=dev)
M = torch.empty(B, K, 52, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
_l.check(h.cpfn_fit_moments_fwd(p(P), p(X), p(W), B, N, K, p(ws), p(M), st), "moments")
out = torch.empty(B * K, 21, dtype=torch.float64, device=dev)
gout = torch.randn(B * K, 21, dtype=torch.float64, device=dev)
gM = torch.empty(B * K, 52, dtype=torch.float64, device=dev)
G = B * K


def fwd():
    _l.check(h.cpfn_fit_algebra_fwd(p(M), G, p(out), None, st), "fwd")


def bwd():
    _l.check(h.cpfn_fit_algebra_bwd(p(M), p(gout), None, G, p(gM), None, st), "bwd")


a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
v = torch.randn(1 << 24, device=dev)


def evict():
    c = a @ b
    w = torch.sin(v) + torch.cos(v) * torch.tanh(v)
    return c, w.sort()[0][:4]


for name, fn in (("cpfn_fit_algebra_fwd", fwd), ("cpfn_fit_algebra_bwd", bwd)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(60):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%s warm, back to back: %.1f us per call (an empty launch: 3.7)" % (name, e0.elapsed_time(e1) * 1000 / 60))
for _ in range(60):
    evict(); fwd(); evict(); bwd()
torch.cuda.synchronize()
print("checksum", float(out.sum()), float(gM.sum()))
