"""ms per step over consecutive 20-step windows right after the warm-up (is the driver's 20-step run in steady state?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cpfn_amd import synthetic, training
from cpfn_amd.PointNet2 import pn2_network
from cpfn_amd.SPFN import fitter_factory
import contextlib, io
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(io.StringIO()):
    fitter_factory.register_primitives(training.GLOBAL_SPFN_CLASSES)
torch.manual_seed(0)
model = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 28]).to(dev)
model.set_compute_dtype(torch.bfloat16)
batch = {k: v.to(dev) for k, v in synthetic.training_batch(16, 8192, 28, seed=1000).items()}
tr = training.SPFNTrainer(model, batch_size=16, use_graphs=True, require_graphs=True)
torch.cuda.set_stream(tr.stream(dev))
for _ in range(5):
    tr.step(batch, next_batch=batch)
torch.cuda.synchronize()
for w in range(8):
    t0 = time.perf_counter()
    for _ in range(20):
        tr.step(batch, next_batch=batch)
    torch.cuda.synchronize()
    print("window %d: %.4f ms/step" % (w, (time.perf_counter() - t0) / 20 * 1e3))
