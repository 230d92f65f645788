"""PatchSelection training step (PointNet2(output_sizes=[2]) + cross-entropy, 16 x 8192 points): replayed PatchSelectionTrainer
step vs the reference's loop on the eager modules (forward, F.cross_entropy, backward, torch.optim.Adam, loss.item())."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cpfn_amd import synthetic, training
from cpfn_amd.PointNet2 import pn2_network
dev = torch.device("cuda:0")
B, N = 16, 8192
c = synthetic.primitive_cloud(B, N, n_prims=10, seed=5)
batch = {"P": c["P"].to(dev), "labels": (c["P"][..., 0] > 0).long().to(dev)}


def model():
    torch.manual_seed(0)
    m = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[2]).to(dev)
    return m.set_compute_dtype(torch.bfloat16)


m = model()
tr = training.PatchSelectionTrainer(m, batch_size=B, use_graphs=True, require_graphs=True)
with torch.cuda.stream(tr.stream(dev)):
    for _ in range(6):
        tr.step(batch, next_batch=batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        out = tr.step(batch, next_batch=batch)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 300
print("PatchSelectionTrainer, replayed: %.3f ms per step, %.0f clouds/s (loss %.4f)" % (1e3 * t, B / t, float(out[0])))
m = model()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
m.train()


def eager():
    opt.zero_grad()
    heat = m(batch["P"])[0]
    loss = torch.nn.functional.cross_entropy(heat.contiguous().view(B * N, 2), batch["labels"].view(B * N))
    v = loss.item()
    loss.backward()
    opt.step()
    return v


for _ in range(3):
    eager()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40):
    v = eager()
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 40
print("reference's loop on the eager modules: %.3f ms per step, %.0f clouds/s (loss %.4f)" % (1e3 * t, B / t, v))
