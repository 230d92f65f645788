"""Every cpfn_mlp_gemm / cpfn_mlp_wgrad launch of one training step: shape, kernel time (HIP events), TB/s."""
import contextlib, io, os, sys
import torch
sys.path.insert(0, os.getcwd())
from cpfn_amd import synthetic, training, fused_mlp, lib as _l
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
for _ in range(3):
    tr.step(batch)
log = []
orig = fused_mlp.gemm
def gemm(A, Wb, n_out=None, gidx=None, stats=False, bias=None, out_f32=False, n_store=None, P=None, w_trans=False, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(A, Wb, n_out=n_out, gidx=gidx, stats=stats, bias=bias, out_f32=out_f32, n_store=n_store, P=P, w_trans=w_trans, **kw)
    e1.record()
    K, N = (Wb.shape[0], Wb.shape[1]) if w_trans else (Wb.shape[1], Wb.shape[0])
    log.append((A.shape[0] if P is None else P, K, N, int(stats), int(w_trans), int(out_f32), int(kw.get('a_scale') is not None), e0, e1))
    return r
fused_mlp.gemm = gemm
tr.step(batch)
torch.cuda.synchronize()
tot = 0
for P, K, N, st, wt, f32, atr, e0, e1 in log:
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    by = (P * K + P * N * (2 if f32 else 1) + N * K) * 2
    print("P=%7d K=%5d N=%5d stats=%d w_trans=%d f32=%d atr=%d  %6.1f us  %5.2f TB/s" % (P, K, N, st, wt, f32, atr, us, by / us / 1e6))
print("total %.0f us over %d launches" % (tot, len(log)))
