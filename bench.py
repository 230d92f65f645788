#!/usr/bin/env python
"""Headline benchmark: point-clouds/sec of one full GlobalSPFN training step
(forward, all five losses incl. the fused primitive fitters, backward, gradient
all-reduce, non-finite guard, Adam) on synthetic 8192-point clouds, 16 clouds per GPU
(BASELINE.json configs[1]; configs[3] is the same per-GPU work on 8 GPUs -> weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Inputs are resident in HBM before the timed region.
`roofline` is measured live with HIP events around the dominant entry point of
libcpfn_hip.so; `cpu_baseline` times the oracle (a port of the reference's CPU path) on the
host cores of this box, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH_PER_GPU = int(os.environ.get("CPFN_BENCH_BATCH", "16"))   # 16 = BASELINE.json configs[1]; override only for experiments
N_POINTS = 8192
N_INSTANCES = 28
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

# Dominant kernel family of the step (profiles/r01_*_kernel_stats.csv; DESIGN.md "Measurement"):
# the bf16 MFMA GEMM behind every 1x1 convolution, forward and data-gradient (31 launches per
# step).  Arithmetic intensity is 32-128 FLOP/B (K, N <= 256 for the layers that carry the
# bytes), far left of the ~300 FLOP/B bf16 ridge, so the bound is HBM.  The algorithmic bytes
# of each launch (read A[P,K] and W[N,K] once, write Y[P,N] once) are summed by the caller
# (cpfn_amd/fused_mlp.py:gemm) while the timed region runs.
ROOFLINE_SYMBOL = "cpfn_mlp_gemm"


def cpu_baseline():
    """Reference CPU path (oracle port): one GlobalSPFN training step (forward, all losses incl. the four
    fitters, backward, Adam) on a bounded sample.  torch-CPU does not scale to the 256 hardware threads of
    the GPU box (128 threads is 10x SLOWER than 16), so a short sweep picks the best thread count and the
    baseline is timed there — `cores` is the number of threads actually used."""
    import numpy as np
    from cpfn_amd import synthetic
    from oracle import pn2 as opn2
    Bc = 4
    prev_threads = torch.get_num_threads()
    torch.manual_seed(0)
    state = synthetic.synthetic_state_dict(synthetic.pointnet2_state_shapes(), seed=0)
    st = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v)
          for k, v in state.items()}
    leaves = [v for v in st.values() if v.requires_grad]
    opt = torch.optim.Adam(leaves, lr=1e-3)
    batch = synthetic.training_batch(Bc, N_POINTS, N_INSTANCES, seed=123)

    def one_step(it):
        starts = (np.random.RandomState(it).randint(0, N_POINTS, Bc), np.random.RandomState(it + 1).randint(0, 512, Bc))
        t0 = time.time()
        opt.zero_grad()
        out = opn2.training_step_losses(st, batch, starts)
        out[0].backward()
        opt.step()
        return time.time() - t0

    ncpu = os.cpu_count() or 8
    best_nt, best_t = None, None
    for nt in [n for n in (8, 16, 32, 64) if n <= ncpu] or [ncpu]:
        torch.set_num_threads(nt)
        one_step(0)
        t = min(one_step(1), one_step(2))
        if best_t is None or t < best_t:
            best_nt, best_t = nt, t
    torch.set_num_threads(best_nt)
    times = [one_step(10 + i) for i in range(6)]
    torch.set_num_threads(prev_threads)
    mean = sum(times) / len(times)
    return {"value": Bc / mean, "unit": "point-clouds/s", "cores": best_nt, "kind": "port",
            "sample": "%d timed GlobalSPFN training steps (fwd+losses+bwd+Adam) of %dx%d pts at the best of "
                      "8/16/32/64 threads (%d; host has %d hardware threads), oracle/ (torch-CPU + C geometry), "
                      "mean %.3f s/step" % (len(times), Bc, N_POINTS, best_nt, ncpu, mean)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graphs", action="store_true", help="launch every kernel eagerly (no hipGraph replay)")
    ap.add_argument("--workload", default="global", choices=["global", "local"],
                    help="global (default) = BASELINE.json configs[1], the config the metric is quoted on; local = "
                         "configs[2] (LocalSPFN: 32 patches/GPU, 21 instances, fitter losses off) as an extra data point")
    args = ap.parse_args()
    global BATCH_PER_GPU, N_INSTANCES
    mult = None
    if args.workload == "local":                        # Configs/config_localSPFN.yml:10-11, training_SPFN.py:69-71
        BATCH_PER_GPU, N_INSTANCES = 32, 21
        mult = dict(miou=1.0, normal=1.0, type=1.0, parameter=0.0, residue=0.0, total=1.0)
        args.no_cpu_baseline = True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from cpfn_amd import lib, synthetic, training
    from cpfn_amd.PointNet2 import pn2_network
    from cpfn_amd.SPFN import fitter_factory
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        fitter_factory.register_primitives(training.GLOBAL_SPFN_CLASSES)
    lib.lib()                                           # fail loudly if the HIP library is missing

    torch.manual_seed(0)                                # default PyTorch init, identical on every rank
    model = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, N_INSTANCES]).to(dev)
    model.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    training.broadcast_parameters(model)
    trainer = training.SPFNTrainer(model, batch_size=BATCH_PER_GPU * world, use_graphs=not args.no_graphs, multipliers=mult)
    batch = {k: v.to(dev) for k, v in
             synthetic.training_batch(BATCH_PER_GPU, N_POINTS, N_INSTANCES, seed=1000 + rank).items()}
    torch.manual_seed(1234 + rank)                      # per-rank FPS starts / dropout masks

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Each step also prefetches the NEXT step's geometry (FPS / ball query / 3-NN with fresh random
    # FPS starts) on a side stream: one geometry pass per step, software-pipelined across steps.
    for _ in range(args.warmup):
        trainer.step(batch, next_batch=batch)
    lib.time_symbols([ROOFLINE_SYMBOL])
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = trainer.step(batch, next_batch=batch)
    sync()
    elapsed = time.perf_counter() - t0
    calls, kernel_ms = lib.timed_report()[ROOFLINE_SYMBOL]
    roof_bytes = lib.timed_bytes(ROOFLINE_SYMBOL)
    roof_mode = "events around every launch inside the timed region"
    if calls == 0:
        # hipGraph replay: the launches of the timed region are graph nodes and cannot be bracketed one
        # by one, so the same step is re-run eagerly right here (same process, same tensors, same
        # stream) with an event pair around every launch of the kernel family.  rocprofv3 (which does
        # see the kernels inside a replay) gives the same per-launch durations: profiles/README.md.
        # Launched one by one the host is slower than the GPU (5.5 ms vs 2.5 ms per step), and an event pair would
        # also time the GPU waiting for the next launch.  A spin kernel queued first holds the stream back until the
        # host has queued the whole step, so every pair brackets a kernel executing back to back like in the replay.
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.cuda._sleep(2_000_000); e1.record(); sync()
        spin = int(2_000_000 * 8.0 / max(e0.elapsed_time(e1), 1e-3))      # ~8 ms of spinning
        lib.time_symbols([ROOFLINE_SYMBOL])
        for _ in range(3):
            torch.cuda._sleep(spin)
            trainer.step(batch, force_eager=True)
        sync()
        calls, kernel_ms = lib.timed_report()[ROOFLINE_SYMBOL]
        roof_bytes = lib.timed_bytes(ROOFLINE_SYMBOL)
        roof_mode = ("events around every launch in 3 eager re-runs of the step right after the timed region, each queued "
                     "behind a spin kernel so that the launches execute back to back (graph replays are not bracketable)")
    lib.time_symbols([])
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        clouds = BATCH_PER_GPU * world * args.steps
        per_launch_s = kernel_ms / max(calls, 1) / 1e3
        bytes_per_launch = roof_bytes / max(calls, 1)
        achieved = bytes_per_launch / per_launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_mlp_gemm_traffic.json")
        if os.path.exists(tpath):                       # PMC pass (rocprofv3 --pmc), see profiles/README.md
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        line = {
            "metric": "point-clouds/sec (8192 pts, %sSPFN fwd+bwd)" % ("Global" if args.workload == "global" else "Local"),
            "value": clouds / elapsed, "unit": "point-clouds/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("GlobalSPFN training step (fwd + 5 losses incl. fitters + bwd + Adam), "
                                    "%d clouds/GPU x %d pts, %d instances, 4 primitive types" if args.workload == "global" else
                                    "LocalSPFN training step (fwd + normal/type/mIoU losses + bwd + Adam; fitter losses "
                                    "switched off as in config_localSPFN.yml), %d patches/GPU x %d pts, %d instances, "
                                    "4 primitive types") % (BATCH_PER_GPU, N_POINTS, N_INSTANCES),
                       "global_batch": BATCH_PER_GPU * world, "points": N_POINTS,
                       "parallelism": "dp%d" % world, "loss_last": float(out[0]),
                       "launch": ("eager" if trainer._graph is None else
                                  "hipGraph replay (1 graph/step, device-side assignment)" if trainer._graph.get("single")
                                  else "hipGraph replay (3 graphs/step around the host-side assignment)")},
            "roofline": {"bound": "hbm", "kernel": ROOFLINE_SYMBOL, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "launches": calls, "avg_launch_us": 1e6 * per_launch_s,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "measured": roof_mode},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
