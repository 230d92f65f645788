"""Host-side profile (cProfile) of eager training steps: where the CPU time of one step goes."""
import contextlib, cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.getcwd())
from cpfn_amd import synthetic, training
from cpfn_amd.PointNet2 import pn2_network
from cpfn_amd.SPFN import fitter_factory
with contextlib.redirect_stdout(io.StringIO()):
    fitter_factory.register_primitives(training.GLOBAL_SPFN_CLASSES)
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 28]).to(dev)
model.set_compute_dtype(torch.bfloat16)
tr = training.SPFNTrainer(model, batch_size=16, use_graphs=False)
batch = {k: v.to(dev) for k, v in synthetic.training_batch(16, 8192, 28, seed=1).items()}
for _ in range(5):
    tr.step(batch, next_batch=batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    tr.step(batch, next_batch=batch)
torch.cuda.synchronize()
print("eager: %.2f ms/step" % ((time.perf_counter() - t0) * 100))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    tr.step(batch, next_batch=batch)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:5000])
