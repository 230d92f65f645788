"""cpfn_fit_moments_fwd / cone passes alone in a loop (debugging aid): event-timed per launch."""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from cpfn_amd import lib as _l
from cpfn_amd.ops import _ptr, _stream
dev = torch.device("cuda:0")
B, N, K = 16, 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 28
g = torch.Generator().manual_seed(0)
P = torch.randn(B, N, 3, generator=g).to(dev)
X = torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=2).to(dev)
W = torch.softmax(torch.randn(B, N, K, generator=g), 2).to(dev)
h = _l.lib()
chunks = h.cpfn_fit_num_chunks(B, N)
ws = torch.empty(B * chunks * K * 52, dtype=torch.float64, device=dev)
M = torch.empty(B, K, 52, dtype=torch.float64, device=dev)
def run():
    _l.check(h.cpfn_fit_moments_fwd(_ptr(P), _ptr(X), _ptr(W), B, N, K, _ptr(ws), _ptr(M), _stream()), "moments")
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print("moments_fwd + chunk_reduce: %.1f us per call (chunks=%d)" % (e0.elapsed_time(e1) / 50 * 1e3, chunks))
