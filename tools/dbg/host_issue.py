"""Host time to ISSUE one replayed step (no back-pressure: the queue is drained before every sample).
    python tools/dbg/host_issue.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cpfn_amd import synthetic, training
from cpfn_amd.PointNet2 import pn2_network
from cpfn_amd.SPFN import fitter_factory
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    fitter_factory.register_primitives(training.GLOBAL_SPFN_CLASSES)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 28]).to(dev)
model.set_compute_dtype(torch.bfloat16)
tr = training.SPFNTrainer(model, batch_size=16, use_graphs=True, require_graphs=True)
batch = {k: v.to(dev) for k, v in synthetic.training_batch(16, 8192, 28, seed=1000).items()}
torch.cuda.set_stream(tr.stream(dev))
for _ in range(8):
    tr.step(batch, next_batch=batch)
torch.cuda.synchronize()
one, three = [], []
for _ in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(batch, next_batch=batch)
    t1 = time.perf_counter()
    tr.step(batch, next_batch=batch)
    tr.step(batch, next_batch=batch)
    t2 = time.perf_counter()
    one.append(t1 - t0)
    three.append((t2 - t0) / 3)
one.sort(); three.sort()
print("host time to issue one replayed step: median %.0f us (first after a sync), %.0f us (mean of three back to back)"
      % (1e6 * one[len(one) // 2], 1e6 * three[len(three) // 2]))
st = tr._graph
t0 = time.perf_counter()
for _ in range(50):
    st["g"].replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("g.replay() alone: %.0f us of host time each (50 in a row, includes back-pressure if any)" % (1e6 * (t1 - t0) / 50))
