import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch, numpy as np
import test_gpu_fullsize as t
cfg = t.LOCAL if len(sys.argv) > 1 and sys.argv[1] == "local" else t.GLOBAL
model, tr, batch_cpu, batch = t._bench_trainer(cfg, seed=2000)
starts, out = t._replayed_step(tr, batch)
K = cfg["K"]
Yg = model.heads_packed.detach().float().clone()
with torch.no_grad():
    model(batch["P"], fps_start=starts)
    Ye = model.heads_packed.detach().float().clone()
    model.set_compute_dtype(torch.float32)
    X, T, W, _, _ = model(batch["P"], fps_start=starts)
    Y32 = torch.cat([X, T, W], 2)
    model.set_compute_dtype(torch.bfloat16)
st, ref, aux = t._oracle_step(model, batch_cpu, starts, cfg["mult"])
Yo = torch.cat([h.detach() for h in aux["heads"]], 2).cuda()
def rel(a, b): return float((a - b).norm() / b.norm())
for name, sl in (("X", slice(0, 3)), ("T", slice(3, 7)), ("W", slice(7, 7 + K))):
    print(name, "graph-vs-eager bf16 %.3e | eager bf16 vs fp32 mode %.3e | fp32 mode vs oracle %.3e | graph vs oracle %.3e | norms %.3e %.3e" % (
        rel(Yg[..., sl], Ye[..., sl]), rel(Ye[..., sl], Y32[..., sl]), rel(Y32[..., sl], Yo[..., sl]), rel(Yg[..., sl], Yo[..., sl]),
        float(Yg[..., sl].norm()), float(Yo[..., sl].norm())))
