import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, contextlib, io
from cpfn_amd import synthetic, training
from cpfn_amd.PointNet2 import pn2_network
from cpfn_amd.SPFN import fitter_factory
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(io.StringIO()):
    fitter_factory.register_primitives(training.GLOBAL_SPFN_CLASSES)
torch.manual_seed(0)
model = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 28]).to(dev)
model.set_compute_dtype(torch.bfloat16)
batch = {k: v.to(dev) for k, v in synthetic.training_batch(16, 8192, 28, seed=1000).items()}
tr = training.SPFNTrainer(model, batch_size=16, use_graphs=True, require_graphs=True)
torch.cuda.set_stream(tr.stream(dev))
for _ in range(6):
    tr.step(batch, next_batch=batch)
torch.cuda.synchronize()
g = tr._graph["g"]
marks = []
orig = g.replay
class Wrap:
    def replay(self):
        t = time.perf_counter(); orig(); marks.append((t, time.perf_counter()))
tr._graph["g"] = Wrap()
res = []
for i in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(batch, next_batch=batch)
    t1 = time.perf_counter()
    a, b = marks[-1]
    res.append(((a - t0) * 1e6, (b - a) * 1e6, (t1 - b) * 1e6))
import statistics
print("before main replay %.0f us, main replay %.0f us, after %.0f us (medians, idle GPU each time)" % tuple(statistics.median(x[i] for x in res) for i in range(3)))
import cProfile, pstats
pr = cProfile.Profile()
for i in range(50):
    torch.cuda.synchronize()
    pr.enable(); tr.step(batch, next_batch=batch); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
