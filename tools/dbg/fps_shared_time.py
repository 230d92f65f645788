"""FPS 131072 -> 512 (several workgroups per cloud): median time of cpfn_fps, indices checked against the streaming kernel's
(env switches of the experiment: CPFN_FPS_RIDE, CPFN_FPS_XCD, CPFN_FPS_PPT)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cpfn_amd import ops, synthetic
dev = torch.device("cuda:0")
for B, N in ((1, 131072), (4, 131072), (1, 524288)):
    P = synthetic.uniform_cloud(B, N, seed=3).to(dev)
    start = torch.zeros(B, dtype=torch.int32, device=dev)
    sel = ops.fps(P, 512, start); torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); sel = ops.fps(P, 512, start); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    print("B %d N %6d: %.3f ms  (checksum %d, faults %d)" % (B, N, float(np.median(ts)), int(sel.long().sum()), ops.fps_faults()))
