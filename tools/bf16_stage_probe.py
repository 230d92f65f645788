"""Where the bf16 forward pass leaves the fp32 one: relative L2 deviation of every stage's output from the
all-fp32 forward, for (a) every stack in bf16 (the bench mode), (b) the three small stacks sa3 / sfp1 / sfp2 — 1 % of the
bytes, the stages where the deviation jumps — in fp32, (c) sa3 + sfp1 only.  Training-mode BatchNorm, dropout off, same
FPS seeds.  usage: python tools/bf16_stage_probe.py [B] [N] [init: default|synthetic|<checkpoint.pt>]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                    # noqa: E402
from cpfn_amd import fused_mlp, mlp, synthetic                  # noqa: E402
from cpfn_amd.PointNet2 import pn2_network                      # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
init = sys.argv[3] if len(sys.argv) > 3 else "default"
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 28]).to(dev)
if init == "synthetic":
    m.load_state_dict(synthetic.synthetic_state_dict(synthetic.pointnet2_state_shapes(), seed=0))
elif init != "default":
    m.load_state_dict(torch.load(init, map_location=dev))
m.dropout_p = 0.0
m.train()
P = synthetic.training_batch(B, N, 28, seed=1000)["P"].to(dev)
starts = (torch.randint(0, N, (B,)), torch.randint(0, 512, (B,)))
F32, BF = torch.float32, torch.bfloat16


def run(dtypes, perturb_l1=0.0, round_l1=False):
    """dtypes: module name -> compute dtype (default bf16).  perturb_l1: Gaussian noise of that relative L2 size added to
    sa1's output; round_l1: sa1's output rounded to bf16 once — two ways of asking what the REST of the network, in fp32,
    does to an error of bf16 size."""
    m.set_compute_dtype(BF)
    for name, cd in dtypes.items():
        if name != "heads":
            for sub in getattr(m, name).modules():
                sub.compute_dtype = cd
    cdh = dtypes.get("heads", BF)
    outs = {}
    with torch.no_grad():
        fused_mlp.refresh_weight_panels(m.parameters())
        xyz = P.contiguous().float()
        l1_xyz, l1, _ = m.sa1.forward_rows(xyz, None, starts[0])
        if perturb_l1:
            g = torch.Generator(device=dev).manual_seed(7)
            noise = torch.randn(l1.shape, generator=g, device=dev)
            l1 = l1 + noise * (perturb_l1 * l1.float().norm() / noise.norm())
        if round_l1:
            l1 = l1.to(BF).to(l1.dtype)
        outs["l1"] = l1.float()
        l2_xyz, l2, _ = m.sa2.forward_rows(l1_xyz, l1, starts[1]); outs["l2"] = l2.float()
        _, l3, _ = m.sa3.forward_rows(l2_xyz, l2); outs["l3"] = l3.float()
        l4, _ = m.sfp1.forward_rows(l2_xyz, None, l2, l3); outs["l4"] = l4.float()
        l5, _ = m.sfp2.forward_rows(l1_xyz, l2_xyz, l1, l4); outs["l5"] = l5.float()
        l6, _ = m.sfp3.forward_rows(xyz, l1_xyz, None, l5); outs["l6"] = l6.float()
        feat = mlp.run_stack(l6.reshape(B * N, -1), [m.fc1], [m.bn1], cdh); outs["feat"] = feat.float()
        hs = mlp.heads(feat, m.fc2, cdh)
        outs["X"], outs["T"], outs["W"] = [h.float() for h in hs]
    return outs


ALL = ("sa1", "sa2", "sa3", "sfp1", "sfp2", "sfp3", "heads")
ref = run({k: F32 for k in ALL})
modes = {"all bf16": {}, "sa3+sfp1+sfp2 fp32": {k: F32 for k in ("sa3", "sfp1", "sfp2")},
         "sa3+sfp1 fp32": {k: F32 for k in ("sa3", "sfp1")}, "sfp1 fp32": {"sfp1": F32},
         "all but sa1 fp32": {k: F32 for k in ALL if k != "sa1"}}
keys = list(ref)
bf = run({})
e1 = float((bf["l1"] - ref["l1"]).norm() / ref["l1"].norm())
modes["fp32, l1 rounded to bf16"] = None
modes["fp32, l1 + noise(%.1e)" % e1] = None
print("%-26s" % "rel L2 vs fp32" + "".join("%9s" % k for k in keys))
for name, d in modes.items():
    o = (run({k: F32 for k in ALL}, round_l1=True) if "rounded" in name else run({k: F32 for k in ALL}, perturb_l1=e1)) if d is None else run(d)
    print("%-26s" % name + "".join("%9.2e" % float((o[k] - ref[k]).norm() / ref[k].norm()) for k in keys))
