"""Which lines of cpfn_amd issue the remaining framework glue ops (copy_, cat, fill_, zeros, dtype casts ...) in one
eager training step: a TorchDispatchMode logs every aten call with the innermost cpfn_amd source line."""
import collections, contextlib, io, os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
for _ in range(3):
    tr.step(batch)
SKIP = ("aten.view", "aten.reshape", "aten._unsafe_view", "aten.transpose", "aten.t.", "aten.select", "aten.slice", "aten.detach",
        "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.alias", "aten.as_strided", "aten.permute", "aten.empty", "aten.unbind",
        "aten.split", "aten.narrow", "aten.lift_fresh", "aten.sym_", "aten.is_", "aten.stride", "aten.size", "aten._local_scalar")
log = collections.Counter()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            where = "(no cpfn_amd frame: autograd engine / optimizer)"
            for fr in reversed(traceback.extract_stack()):
                if "cpfn_amd/" in fr.filename:
                    where = "%s:%d" % (fr.filename.split("cpfn_amd/")[-1], fr.lineno)
                    break
            shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), ())
            log[(name, where, shp)] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    tr.step(batch)
torch.cuda.synchronize()
for (name, where, shp), n in sorted(log.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print("%3d  %-34s %-22s %s" % (n, name, str(shp), where))
