# Two REAL ranks (gloo backend, both on GPU 0) of the bf16 graph trainer: the data-parallel path with world size 2 on
# a one-GPU box.  RCCL refuses two ranks on one device, so the collective is gloo's (on device tensors) — which cannot
# be captured into a graph: this exercises the rank-agreed fall-back (graph up to the gradient packing, exchange +
# optimizer as eager launches) with the real kernels, and checks that the replicas stay bit-identical.
import os, sys, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist, torch.multiprocessing as mp


def worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import warnings
    warnings.simplefilter("ignore")
    from cpfn_amd import synthetic, training
    from cpfn_amd.PointNet2 import pn2_network
    from cpfn_amd.SPFN import fitter_factory
    dev = torch.device("cuda:0")
    with contextlib.redirect_stdout(io.StringIO()):
        fitter_factory.register_primitives(training.GLOBAL_SPFN_CLASSES)
    torch.manual_seed(100 + rank)                      # different init per rank: the broadcast must fix it
    model = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 28]).to(dev)
    model.set_compute_dtype(torch.bfloat16)
    model.dropout_p = 0.0
    training.broadcast_parameters(model)
    tr = training.SPFNTrainer(model, batch_size=4 * world, use_graphs=True)
    batch = {k: v.to(dev) for k, v in synthetic.training_batch(4, N=2048, n_prims=6, n_inst_points=128, seed=5 + rank).items()}
    torch.manual_seed(7 + rank)
    hist = [float(tr.step(batch, next_batch=batch)[0]) for _ in range(12)]
    torch.cuda.synchronize()
    torch.save({"params": {k: v.detach().cpu() for k, v in model.named_parameters()}, "hist": hist,
                "graph": tr._graph is not None, "in_graph": bool(tr._graph and tr._graph.get("exchange_in_graph")),
                "skipped": tr.skipped_steps}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket, tempfile
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    d = tempfile.mkdtemp()
    mp.spawn(worker, args=(2, port, d), nprocs=2, join=True)
    a, b = torch.load(os.path.join(d, "r0.pt")), torch.load(os.path.join(d, "r1.pt"))
    same = all(torch.equal(a["params"][k], b["params"][k]) for k in a["params"])
    print("replicas identical:", same, "graph:", a["graph"] and b["graph"], "exchange in graph:", a["in_graph"] or b["in_graph"],
          "skipped:", a["skipped"] + b["skipped"], "loss rank0 %.3f -> %.3f rank1 %.3f -> %.3f" % (a["hist"][0], a["hist"][-1], b["hist"][0], b["hist"][-1]))
