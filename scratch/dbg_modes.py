import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from test_gpu_trainer import _run
ref, sd_ref, _ = _run("eager")
for mode in ["eager", "prefetch", "graph"]:
    l, sd, tr = _run(mode)
    print(mode, 'graph' if tr._graph is not None else '', [round(abs(a[0]-b[0])/b[0], 5) for a, b in zip(l, ref)], 'max param rel', max(float((sd[k]-sd_ref[k]).norm()/sd_ref[k].norm().clamp_min(1e-6)) for k in sd if 'num_batches' not in k))
print('losses', [round(a[0],4) for a in ref])
