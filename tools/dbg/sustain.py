"""Does the replayed step slow down under SUSTAINED load?  N back-to-back steps after 0.3 s of idle, for growing N (ms per step), and
20-step windows inside one long run.  Beside it, from the shell: rocm-smi sampling clocks / power (tools/dbg/sustain.sh)."""
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
for _ in range(30):
    tr.step(batch, next_batch=batch)
torch.cuda.synchronize()
for n in (3, 10, 30, 100, 300, 1000, 3000, 3, 10, 30):
    time.sleep(0.3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        tr.step(batch, next_batch=batch)
    e1.record()
    torch.cuda.synchronize()
    print("%5d steps back to back after idle: %.4f ms/step" % (n, e0.elapsed_time(e1) / n), flush=True)
time.sleep(0.3)
evs = []
for w in range(40):
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    for _ in range(50):
        tr.step(batch, next_batch=batch)
e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
torch.cuda.synchronize()
print("50-step windows of one 2000-step run:", " ".join("%.3f" % (evs[i].elapsed_time(evs[i + 1]) / 50) for i in range(40)))
