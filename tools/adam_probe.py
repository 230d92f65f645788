import os, sys, torch
sys.path.insert(0, os.getcwd())
from cpfn_amd import lib as _l
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1400000
p, g, m, v = (torch.randn(n, device=dev) for _ in range(4))
v.abs_()
lr = torch.tensor(1e-3, device=dev); step = torch.zeros((), device=dev); coef = torch.zeros(3, device=dev); pows = torch.ones(2, dtype=torch.float64, device=dev)
h = _l.lib()
def run():
    rc = h.cpfn_adam_flat(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr.data_ptr(), 0.9, 0.999, 1e-8, 0.0, step.data_ptr(), pows.data_ptr(), None, coef.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): run()
e1.record(); torch.cuda.synchronize()
print("adam_flat n=%d: %.1f us per step (2 launches)" % (n, e0.elapsed_time(e1) * 10))
