import sys, contextlib, io; sys.path.insert(0, '.')
import torch
from cpfn_amd import synthetic, training
from cpfn_amd.PointNet2 import pn2_network
from cpfn_amd.SPFN import fitter_factory
dev = torch.device('cuda:0')
with contextlib.redirect_stdout(io.StringIO()):
    fitter_factory.register_primitives(training.GLOBAL_SPFN_CLASSES)
def run(dtype, fused, graphs, steps=80):
    torch.manual_seed(0)
    model = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 28]).to(dev)
    model.set_compute_dtype(dtype)
    tr = training.SPFNTrainer(model, batch_size=4, use_graphs=graphs)
    tr.fused_losses = fused
    batches = [{k: v.to(dev) for k, v in synthetic.training_batch(4, N=2048, n_prims=6, n_inst_points=128, seed=s).items()} for s in range(4)]
    torch.manual_seed(7)
    hist = []
    for i in range(steps):
        out = tr.step(batches[i % 4])
        hist.append([float(o) for o in out])
    return hist, tr
for name, args in [('fp32 torch-mlp unfused-loss', (torch.float32, False, False)), ('bf16 fused eager', (torch.bfloat16, True, False)), ('bf16 fused graph', (torch.bfloat16, True, True))]:
    h, tr = run(*args)
    avg = lambda a, b: [round(sum(x[j] for x in h[a:b])/(b-a), 4) for j in range(6)]
    print(name, '| first8', avg(0, 8), '| last8', avg(72, 80), '| skipped', tr.skipped_steps if tr._graph is None else float(tr._graph['skipped']))
