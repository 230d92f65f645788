import sys, contextlib, io; sys.path.insert(0, '.')
import torch
from cpfn_amd import synthetic, training
from cpfn_amd.PointNet2 import pn2_network
from cpfn_amd.SPFN import fitter_factory
dev = torch.device('cuda:0')
with contextlib.redirect_stdout(io.StringIO()):
    fitter_factory.register_primitives(training.GLOBAL_SPFN_CLASSES)
def run(dtype, fused_losses=True):
    torch.manual_seed(0)
    model = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 28]).to(dev)
    model.set_compute_dtype(dtype); model.dropout_p = 0.0
    tr = training.SPFNTrainer(model, batch_size=4)
    tr.fused_losses = fused_losses
    batch = {k: v.to(dev) for k, v in synthetic.training_batch(4, N=2048, n_prims=6, n_inst_points=128, seed=5).items()}
    p0 = {k: v.detach().clone() for k, v in model.named_parameters()}
    starts = (torch.tensor([1, 2, 3, 4]), torch.tensor([5, 6, 7, 8]))
    out = tr.step(batch, fps_start=starts)
    g = {k: (None if v.grad is None else v.grad.detach().clone()) for k, v in model.named_parameters()}
    d = {k: (v.detach() - p0[k]) for k, v in model.named_parameters()}
    out2 = tr.step(batch, fps_start=starts)
    return [float(o) for o in out], [float(o) for o in out2], g, d
l32, l32b, g32, d32 = run(torch.float32, False)
l16, l16b, g16, d16 = run(torch.bfloat16, True)
print('fp32  losses', [round(x, 4) for x in l32], '->', round(l32b[0], 4))
print('bf16  losses', [round(x, 4) for x in l16], '->', round(l16b[0], 4))
for k in g32:
    a, b = g16[k], g32[k]
    if a is None:
        print('%-32s grad None (ref norm %.2e)' % (k, float(b.norm()))); continue
    cos = float((a*b).sum() / (a.norm()*b.norm()).clamp_min(1e-20))
    dcos = float((d16[k]*d32[k]).sum() / (d16[k].norm()*d32[k].norm()).clamp_min(1e-20))
    print('%-32s grad cos %.3f  ratio %.3f   update cos %.3f' % (k, cos, float(a.norm()/b.norm().clamp_min(1e-20)), dcos))
