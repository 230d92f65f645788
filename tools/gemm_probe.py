"""One shape of cpfn_mlp_gemm in a loop, for rocprofv3 counter passes (debugging aid).
    python tools/gemm_probe.py [P K N stats reps]"""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from cpfn_amd import fused_mlp
P, K, N, stats, reps = (int(v) for v in (sys.argv[1:6] + ["131072", "128", "128", "1", "50"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
A = torch.randn(P, K, device=dev).to(torch.bfloat16)
W = torch.randn(N, K, device=dev).to(torch.bfloat16)
for _ in range(5):
    fused_mlp.gemm(A, W, stats=bool(stats))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fused_mlp.gemm(A, W, stats=bool(stats))
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
print("P=%d K=%d N=%d stats=%d: %.1f us/launch, %.2f TB/s algorithmic" % (P, K, N, stats, us, (P * K + P * N + N * K) * 2 / us / 1e6))
if len(sys.argv) > 6 and sys.argv[6] == "trans":
    Wt = torch.randn(K, N, device=dev).to(torch.bfloat16)
    for _ in range(5):
        fused_mlp.gemm(A, Wt, w_trans=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fused_mlp.gemm(A, Wt, w_trans=True)
    e1.record()
    torch.cuda.synchronize()
    print("   w_trans: %.1f us/launch" % (e0.elapsed_time(e1) * 1e3 / reps))
