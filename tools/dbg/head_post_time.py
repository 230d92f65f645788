"""head_post forward / backward stand-alone at the step's size (16 x 8192 x (7 + 28)), as 50 launches of a replayed graph each."""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from cpfn_amd import synthetic                      # noqa: E402
from cpfn_amd.SPFN import fused_losses as fl       # noqa: E402

dev = torch.device("cuda:0")
b = {k: v.to(dev) for k, v in synthetic.training_batch(16, 8192, 28, seed=3).items()}
Y = torch.randn(16, 8192, 35, device=dev, requires_grad=True)


def fwd():
    return fl.HeadPost.apply(Y, b["X_gt"], b["I_gt"], b["T_gt"], True)


out = fwd()
print([None if o is None else tuple(o.shape) for o in out], flush=True)
Xn, W = out[0], out[1]
gX, gW = torch.randn_like(Xn), torch.randn_like(W)


class _Ctx:          # HeadPost.backward called directly (the autograd engine's worker thread does not capture)
    handover = None


ctx = _Ctx()
with torch.no_grad():
    Yc = Y.detach().contiguous()
    ctx.saved_tensors = (Yc, b["X_gt"].contiguous().float(), b["I_gt"].contiguous(), b["T_gt"].contiguous(), W.detach(), torch.stack([out[2], out[3], torch.full_like(out[2], 8192.0)], 1).contiguous())
gl = torch.ones(2, 16, device=dev)
gS = torch.randn(16, 30, 28, device=dev)


def bwd():
    with torch.no_grad():
        fl.HeadPost.backward(ctx, gX, gW, gl[0], gl[1], gS)


for name, fn in (("forward (+ finish)", fwd), ("backward", bwd)):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(50):
                fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        e.record()
        e.synchronize()
        ts.append(a.elapsed_time(e) / 50 * 1e3)
    ts.sort()
    print("%-22s %.1f us per call" % (name, ts[3]))
