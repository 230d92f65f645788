import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cpfn_amd import synthetic
from cpfn_amd.PointNet2 import pn2_network
dev = torch.device("cuda:0")
P = synthetic.primitive_cloud(1, 131072, n_prims=12, noise=0.002, seed=9)["P"].to(dev)
torch.manual_seed(0)
g = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 28]).to(dev).eval()
g.set_compute_dtype(torch.bfloat16); g.dropout_p = 0.0
with torch.no_grad():
    for _ in range(5):
        g(P, fps_start=(torch.tensor([0]), torch.tensor([0])))
torch.cuda.synchronize()
