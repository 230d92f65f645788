"""The small-P GEMM shapes of one GlobalSPFN step (sa3, sfp1, sfp2 forward + data gradients), cycled so that every
launch finds its operands cold in L2; run under `rocprofv3 --kernel-trace --output-format csv` and aggregate the
trace by grid size (debugging aid)."""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from cpfn_amd import fused_mlp
dev = torch.device("cuda:0")
SHAPES = [  # P, K, N, stats, w_trans, atr
    (2048, 320, 256, 1, 0, 0), (2048, 256, 512, 1, 0, 1), (2048, 512, 1024, 1, 0, 1), (2048, 1280, 256, 1, 0, 0),
    (2048, 256, 256, 1, 0, 1), (8192, 384, 256, 1, 0, 0), (8192, 256, 128, 1, 0, 1), (8192, 128, 256, 0, 1, 0),
    (8192, 256, 384, 0, 1, 0), (2048, 256, 256, 0, 1, 0), (2048, 256, 1280, 0, 1, 0), (2048, 1024, 512, 0, 1, 0),
    (2048, 512, 256, 0, 1, 0), (2048, 256, 320, 0, 1, 0)]
ops = []
for P, K, N, st, wt, atr in SHAPES:
    A = torch.randn(P, K, device=dev).bfloat16()
    W = (torch.randn(K, N, device=dev) if wt else torch.randn(N, K, device=dev)).bfloat16()
    kw = dict(a_scale=torch.rand(K, device=dev) + 0.5, a_shift=torch.rand(K, device=dev) - 0.5) if atr else {}
    ops.append((A, W, bool(st), bool(wt), kw))
filler = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    for A, W, st, wt, kw in ops:
        fused_mlp.gemm(A, W, stats=st, w_trans=wt, **kw)
    filler.zero_()      # sweep the caches between rounds
torch.cuda.synchronize()
