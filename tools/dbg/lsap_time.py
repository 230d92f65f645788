"""Time cpfn_hungarian_match as its own launch (B clouds, K = 28): python tools/dbg/lsap_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cpfn_amd.SPFN import fused_losses as fl

dev = torch.device("cuda:0")
B, N, K = 16, 8192, 28
g = torch.Generator().manual_seed(0)
W = torch.softmax(torch.randn(B, N, K, generator=g) * 3.0, 2).to(dev)
I = torch.randint(0, K, (B, N), generator=g).to(dev)
S = fl.SegStats.apply(W, I)
n_gt = fl.count_gt(I)
for _ in range(5):
    m = fl.hungarian_device(S, n_gt)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    m = fl.hungarian_device(S, n_gt)
e1.record()
torch.cuda.synchronize()
print("cpfn_hungarian_match: %.1f us per launch (B=%d, K=%d)" % (e0.elapsed_time(e1) * 1000 / 200, B, K))
