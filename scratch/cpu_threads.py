import sys, time; sys.path.insert(0, '.')
import torch, numpy as np
from cpfn_amd import synthetic
from oracle import pn2 as opn2
import os
print('cpu_count', os.cpu_count())
state = synthetic.synthetic_state_dict(synthetic.pointnet2_state_shapes(), seed=0)
batch = synthetic.training_batch(2, 8192, 28, seed=123)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    st = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in state.items()}
    ts = []
    for it in range(3):
        t0 = time.time()
        out = opn2.training_step_losses(st, batch, (np.array([1, 2]), np.array([3, 4])))
        out[0].backward()
        ts.append(time.time() - t0)
    print(nt, 'threads: %.2f s/step -> %.2f clouds/s' % (min(ts[1:]), 2 / min(ts[1:])))
