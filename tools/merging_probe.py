"""Patch merging at config-5 size (131072 points, 32 patches x 8192, 21 + 28 labels): the HIP path against the
reference's dense formulation run with PyTorch ops on the same GPU (debugging / documentation aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from cpfn_amd.Utils import merging_utils as mu
dev = torch.device("cuda:0")
N, nb, npp, Lp, Lo = 131072, 32, 8192, 21, 28
g = torch.Generator().manual_seed(5)
pidx = torch.empty(nb, npp, dtype=torch.int64)
for b in range(nb):
    start = int(torch.randint(0, N - 3 * npp, (1,), generator=g))
    pidx[b] = start + torch.randperm(3 * npp, generator=g)[:npp]
pred = torch.softmax(torch.randn(nb, npp, Lp, generator=g) * 2, dim=2).to(dev)
spfn = torch.eye(Lo, dtype=torch.int64)[torch.randint(0, Lo, (N,), generator=g)].to(dev)
pidx = pidx.to(dev)
C = nb * Lp + Lo

def dense():
    M = torch.zeros(N, C, device=dev)
    for b in range(nb):
        M[pidx[b], b * Lp:(b + 1) * Lp] += pred[b]
    M[:, nb * Lp:] = spfn
    return torch.mm(M.t(), M), M

def timeit(f, n):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

t_hip = timeit(lambda: mu.similarity_soft(spfn, pred, pidx), 20)
t_ref = timeit(lambda: dense(), 5)
_, M = dense()
labels = torch.randint(0, 60, (C,), generator=g); labels[:60] = torch.arange(60); labels = labels.to(dev)
onehot = torch.eye(60, device=dev)[labels]
t_pool = timeit(lambda: mu.get_point_final(M, labels), 20)
t_pool_ref = timeit(lambda: torch.mm(M, onehot / (onehot.sum(0, keepdim=True) + 1e-10)), 10)
print("similarity_soft: HIP %.0f us   dense torch formulation %.0f us" % (t_hip, t_ref))
print("get_point_final: HIP %.0f us (%.2f TB/s over M)   torch mm %.0f us" % (t_pool, N * C * 4 / t_pool / 1e6, t_pool_ref))
