"""Times cpfn_csr_build on the three shapes of a GlobalSPFN step (16 clouds): sa2's grouping (8192 entries -> 512 targets, ball-query
rows with padding), sfp2's 3-NN (1536 -> 128), sfp3's 3-NN (24576 -> 512); HIP events, median of 20, launches back to back.
    CPFN_CSR_RADIX=0|1 CPFN_CSR_THREADS=-16|-8|-1|0|512|1024 python tools/dbg/csr_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from cpfn_amd import ops, synthetic          # noqa: E402

dev = torch.device("cuda:0")
B = 16
P = synthetic.primitive_cloud(B, 8192, n_prims=10, seed=3)["P"].to(dev)
start = torch.zeros(B, dtype=torch.int32, device=dev)
s1 = ops.fps(P, 512, start)
c1 = ops.gather_rows(P, s1)
s2 = ops.fps(c1, 128, start)
c2 = ops.gather_rows(c1, s2)
cases = [("sa2 grouping 128 x 64 -> 512", ops.ball_query(c2, c1, 0.4, 64).int(), 512),
         ("sfp2 3-NN 512 x 3 -> 128", ops.three_nn(c1, c2)[1].int(), 128),
         ("sfp3 3-NN 8192 x 3 -> 512", ops.three_nn(P, c1)[1].int(), 512)]
tot = 0.0
for name, idx, M in cases:
    idx = idx.contiguous()
    ops.csr_build(idx, M)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            off, ent = ops.csr_build(idx, M)
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) / 5 * 1e3)
    ts.sort()
    tot += ts[len(ts) // 2]
    print("%-32s %8.1f us   (checksum %d)" % (name, ts[len(ts) // 2], int(ent.long().sum() % 1000003)))
print("sum %.1f us   (CPFN_CSR_RADIX=%s, CPFN_CSR_THREADS=%s)" % (tot, os.environ.get("CPFN_CSR_RADIX", "1"), os.environ.get("CPFN_CSR_THREADS", "-8 (ordered, 8 waves)")))
