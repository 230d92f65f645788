"""Census of the small PyTorch kernels left in one eager training step: which aten op, how many launches,
GPU time, and the cpfn_amd source line that issued it.  Used to decide what to fuse next.
    python tools/op_census.py [--top 60]
"""
import argparse
import collections
import contextlib
import io
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=60)
    args = ap.parse_args()
    from cpfn_amd import synthetic, training
    from cpfn_amd.PointNet2 import pn2_network
    from cpfn_amd.SPFN import fitter_factory
    with contextlib.redirect_stdout(io.StringIO()):
        fitter_factory.register_primitives(training.GLOBAL_SPFN_CLASSES)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 28]).to(dev)
    model.set_compute_dtype(torch.bfloat16)
    trainer = training.SPFNTrainer(model, batch_size=16, use_graphs=False)
    batch = {k: v.to(dev) for k, v in synthetic.training_batch(16, 8192, 28, seed=1).items()}
    for _ in range(3):
        trainer.step(batch)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        trainer.step(batch)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
            continue
        t = sum(k.duration for k in ev.kernels)
        if not ev.kernels:
            continue
        where = "(autograd engine)"
        e = ev
        while e is not None and where == "(autograd engine)":
            for fr in e.stack or []:
                if "cpfn_amd/" in fr:
                    where = fr.split("cpfn_amd/")[-1][:70]
                    break
            if e.name.startswith("autograd::engine::evaluate_function") or "Backward" in e.name:
                where = "bwd of " + e.name.replace("autograd::engine::evaluate_function: ", "")[:50]
            e = e.cpu_parent
        key = (ev.name, where)
        agg[key][0] += len(ev.kernels)
        agg[key][1] += t
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for v in agg.values())
    print("aten ops with GPU kernels in one eager step: %d launches, %.1f us" % (sum(v[0] for v in agg.values()), tot))
    for (name, where), (n, t) in rows[:args.top]:
        print("%8.1f us %4d  %-28s %s" % (t, n, name, where))


if __name__ == "__main__":
    main()
