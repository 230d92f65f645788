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
# CPFN_SMOKE_FAULT=own|peer: the fault word on the REPLAYED path.  own: raise_fault() -> the step that carries the word
# runs, its optimizer is skipped, this rank raises after it.  peer: the reduced fault slot is non-zero although this rank raised
# nothing (what a peer's word looks like after the collective) -> the step skips, and the pinned host copy of the slot stops this rank
# at the head of its next step instead of letting it replay into a collective whose partner is gone.
mode = os.environ.get("CPFN_SMOKE_FAULT")
if mode:
    before = [p.detach().clone() for p in model.parameters()]
    moments = [t.clone() for t in (tr.optimizer.exp_avg, tr.optimizer.exp_avg_sq)] if hasattr(tr.optimizer, "exp_avg") else []
    sk0 = float(tr._graph['skipped'])
    err = None
    try:
        if mode == "own":
            tr.raise_fault("smoke test")
            tr.step(batch, next_batch=batch)
        else:
            tr.fault_word(dev).fill_(1.0)
            tr.step(batch, next_batch=batch)           # carries the word: skipped everywhere, returns normally
            torch.cuda.synchronize()
            assert float(tr._graph["fault_host"][0]) != 0.0, "the reduced fault slot did not reach the pinned host word"
            tr.step(batch, next_batch=batch)           # must stop at its head
    except RuntimeError as e:
        err = str(e)
    torch.cuda.synchronize()
    same = all(torch.equal(a, b.detach()) for a, b in zip(before, model.parameters()))
    same_m = all(torch.equal(a, b) for a, b in zip(moments, (tr.optimizer.exp_avg, tr.optimizer.exp_avg_sq))) if moments else True
    print('fault mode', mode, '| raised:', err is not None and ("PEER" in err if mode == "peer" else "this rank raised" in err),
          '| weights untouched:', same, '| moments untouched:', same_m, '| skipped +', float(tr._graph['skipped']) - sk0, '|', (err or '')[:90])
    sys.stdout.flush()
    os._exit(0)          # (the process group was aborted on purpose)
torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group()
