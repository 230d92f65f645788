import torch, sys
sys.path.insert(0, "/root/repo")
from cpfn_amd import fused_mlp, training, synthetic
from cpfn_amd.PointNet2 import pn2_network
from cpfn_amd.SPFN import fitter_factory
import contextlib, io
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(io.StringIO()):
    fitter_factory.register_primitives(training.GLOBAL_SPFN_CLASSES)
orig = fused_mlp._flush_reductions
def spy():
    print("flush:", [(e[2], e[3], e[4], e[5], e[7] is not None) for e in fused_mlp._pending_reduce])
    orig()
fused_mlp._flush_reductions = spy
od = fused_mlp._defer_reduction
def dspy(ws, out, n, splits, row_in=0, row_out=0, params=(), out_ld=0, coef=None):
    print("defer n=%d splits=%d coef=%s free=%s pending=%d" % (n, splits, coef is not None, all(p.grad is None for p in params), len(fused_mlp._pending_reduce)))
    return od(ws, out, n, splits, row_in, row_out, params, out_ld, coef)
fused_mlp._defer_reduction = dspy
torch.manual_seed(0)
model = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 28]).to(dev)
model.set_compute_dtype(torch.bfloat16)
tr = training.SPFNTrainer(model, batch_size=16, use_graphs=False)
batch = {k: v.to(dev) for k, v in synthetic.training_batch(16, N=8192, n_prims=6, n_inst_points=512, seed=5).items()}
tr.step(batch)
torch.cuda.synchronize()
