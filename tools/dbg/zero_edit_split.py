"""Where the 11.8 ms of bench.py's zero_edit route go: the reference loop's sections timed with a device synchronisation after
each (so the sections add up to MORE than the un-synchronised loop), plus the host-side cost of each section without one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cpfn_amd.PointNet2 import pn2_network
from cpfn_amd.SPFN import losses_implementation as li

dev = torch.device("cuda:0")
conf = bench._RouteConf()
loader = bench._route_loader(8, 24, 0)
torch.manual_seed(0)
model = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, bench.N_INSTANCES]).to(dev)
model.set_compute_dtype(torch.bfloat16)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
model.train()
acc = {}


def lap(name, t0, sync=True):
    if sync:
        torch.cuda.synchronize()
    t = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t - t0)
    return t


for it, data in enumerate(loader):
    if it == 4:
        acc.clear()
    torch.cuda.synchronize()
    t = time.perf_counter()
    opt.zero_grad()
    P, X_gt, ppi = (data[i].type(torch.FloatTensor).to(dev) for i in (0, 1, 2))
    I_gt, T_gt = (data[i].type(torch.LongTensor).to(dev) for i in (3, 4))
    gt = {k: data[i].type(torch.FloatTensor).to(dev) for k, i in (("plane_normal", 5), ("cylinder_axis", 6), ("cone_axis", 7))}
    t = lap("1 zero_grad + .to(device)", t)
    X, T, W, _, _ = model(P, glob_features=None, loc_features=None)
    t = lap("2 forward", t)
    X = torch.nn.functional.normalize(X, p=2, dim=2, eps=1e-12)
    W = torch.softmax(W, dim=2)
    out = li.compute_all_losses(P, W, I_gt, X, X_gt, T, T_gt, gt, ppi, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, False, mode_seg='mIoU',
                                classes=conf.get_list_of_primitives())
    t = lap("3 normalise, soft-max, compute_all_losses", t)
    out[0].backward()
    t = lap("4 backward", t)
    bad = any(p.grad is not None and bool(torch.any(torch.isinf(p.grad)) or torch.any(torch.isnan(p.grad))) for p in model.parameters())
    t = lap("5 isinf / isnan scan", t)
    opt.step()
    t = lap("6 torch.optim.Adam", t)
    vals = [v.item() for v in out[:6]] + [out[0].item()]
    t = lap("7 seven .item()", t)
n = len(loader) - 4
for k in sorted(acc):
    print("%-45s %7.3f ms" % (k, 1e3 * acc[k] / n))
print("%-45s %7.3f ms" % ("sum", 1e3 * sum(acc.values()) / n))
