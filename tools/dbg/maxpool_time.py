"""bn_relu_maxpool at the step's three shapes: time (20 launches of a replayed graph) and a checksum of out / arg / yarg.
    python tools/dbg/maxpool_time.py"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from cpfn_amd import fused_mlp      # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
tot = 0.0
for name, G, Kn, C in (("sa1 8192 groups x 64 x 128", 8192, 64, 128), ("sa2 2048 x 64 x 256", 2048, 64, 256), ("sa3 16 x 128 x 1024", 16, 128, 1024),
                       ("ragged 3000 x 37 x 64", 3000, 37, 64), ("ties 4096 x 64 x 128", 4096, 64, 128)):
    Y = torch.randn(G * Kn, C, device=dev).to(torch.bfloat16)
    if name.startswith("ties"):
        Y = (Y * 2).round() / 2          # many equal maxima: the first row must win
    scale = torch.rand(C, device=dev) + 0.5
    scale[::7] *= -1
    shift = torch.randn(C, device=dev)
    out, arg, yarg = fused_mlp.bn_relu_maxpool(Y, scale, shift, Kn)
    torch.cuda.synchronize()
    h = hashlib.sha256(out.view(torch.int16).cpu().numpy().tobytes() + arg.cpu().numpy().tobytes() + yarg.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                fused_mlp.bn_relu_maxpool(Y, scale, shift, Kn)
    ts = []
    for _ in range(7):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        e.record()
        e.synchronize()
        ts.append(a.elapsed_time(e) / 20 * 1e3)
    ts.sort()
    tot += ts[3] if name[:2] == "sa" else 0
    print("%-28s %7.1f us  %6.2f TB/s   sha %s" % (name, ts[3], 2 * Y.numel() / ts[3] / 1e6, h))
print("sum of the step's three: %.1f us" % tot)
