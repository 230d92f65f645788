# single-GPU smoke of the data-parallel graph path: NCCL(RCCL) group of size 1, world-size spoofed to 2
import os, sys, contextlib, io; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 400))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from cpfn_amd import synthetic, training
from cpfn_amd.PointNet2 import pn2_network
from cpfn_amd.SPFN import fitter_factory
real = dist.get_world_size
training.dist.get_world_size = lambda *a, **k: 2
dev = torch.device('cuda:0')
with contextlib.redirect_stdout(io.StringIO()):
    fitter_factory.register_primitives(training.GLOBAL_SPFN_CLASSES)
torch.manual_seed(0)
model = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 28]).to(dev)
model.set_compute_dtype(torch.bfloat16)
training.broadcast_parameters(model)
tr = training.SPFNTrainer(model, batch_size=8, use_graphs=True)
batch = {k: v.to(dev) for k, v in synthetic.training_batch(4, N=2048, n_prims=6, n_inst_points=128, seed=5).items()}
hist = []
for i in range(30):
    hist.append(float(tr.step(batch, next_batch=batch)[0]))
torch.cuda.synchronize()
c_us = tr.comm_us()
print('comm_us', c_us)
assert c_us is not None and 0.0 < c_us < 5000.0, c_us          # the two stamps around the exchange of the last replayed step
print('graph captured:', tr._graph is not None, 'world in graph:', tr._graph['world'], 'exchange in graph:', tr._graph.get('exchange_in_graph'), 'loss', round(hist[0], 3), '->', round(hist[-1], 3), 'skipped', float(tr._graph['skipped']))
torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group()
