"""Which gradient differs first when two identical training runs part?
    python tools/dbg/step_repro.py [runs] [steps] [ps|spfn]
The same short run (fresh model, same seeds, same device-resident batches, replayed step) is repeated; after every step the flat
gradient bucket and all parameters are cloned on the device and compared with the first run's.  For the first step that differs it
prints the parameters whose GRADIENT differs (count, largest difference) — the weights before that step were still equal, so the
kernel that produced the top-most of them in backward order is where the run-to-run difference enters."""
import contextlib
import io
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from cpfn_amd import synthetic, training                      # noqa: E402
from cpfn_amd.PointNet2 import pn2_network                    # noqa: E402

B, N, K = 4, 2048, 28


def run(kind, steps, dev):
    torch.manual_seed(0)
    if kind == "ps":
        m = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[2]).to(dev)
        m.set_compute_dtype(torch.bfloat16)
        tr = training.PatchSelectionTrainer(m, batch_size=B, init_learning_rate=1e-3, decay_step=16, decay_rate=0.7, bn_decay_step=24,
                                            use_graphs=True)
        bs = []
        for i in range(steps):
            c = synthetic.primitive_cloud(B, N, n_prims=6, seed=600 + i)
            bs.append({"P": c["P"].to(dev), "labels": (c["I_gt"] % 2).long().to(dev)})
    else:
        from cpfn_amd.SPFN import fitter_factory
        with contextlib.redirect_stdout(io.StringIO()):
            fitter_factory.register_primitives(training.GLOBAL_SPFN_CLASSES)
        m = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, K]).to(dev)
        m.set_compute_dtype(torch.bfloat16)
        tr = training.SPFNTrainer(m, batch_size=B, init_learning_rate=1e-3, decay_step=12, decay_rate=0.7, bn_decay_step=20, use_graphs=True)
        bs = [{k: v.to(dev) for k, v in synthetic.training_batch(B, N=N, n_max_instances=K, n_prims=6, n_inst_points=128, seed=100 + i).items()}
              for i in range(steps)]
    torch.manual_seed(78)
    rec = []
    with torch.cuda.stream(tr.stream(dev)):
        m.train()
        for i, b in enumerate(bs):
            out = tr.step(b, next_batch=bs[i + 1] if i + 1 < len(bs) else None)
            st = tr._graph
            geo = [t.clone() for t in st["geomA_flat"]] + [st["batch"]["P"].clone(), st["start_dev"].clone()] if st else []
            rec.append((torch.stack([torch.as_tensor(o, device=dev).float().reshape(()) for o in out]).clone(), tr.bucket.flat.clone(),
                        torch.cat([p.detach().flatten() for p in tr.bucket.params]), geo))
    torch.cuda.synchronize()
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    sizes = [p.numel() for p in tr.bucket.params]
    return rec, names, sizes


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    kind = sys.argv[3] if len(sys.argv) > 3 else "ps"
    dev = torch.device("cuda:0")
    ref, names, sizes = run(kind, steps, dev)
    bad = 0
    for r in range(runs):
        rec, _, _ = run(kind, steps, dev)
        for s, ((l0, g0, w0, e0), (l1, g1, w1, e1)) in enumerate(zip(ref, rec)):
            if torch.equal(g0, g1) and torch.equal(w0, w1) and torch.equal(l0, l1):
                continue
            bad += 1
            print("run %d step %d: loss equal %s, gradients equal %s, weights after equal %s" %
                  (r, s, torch.equal(l0, l1), torch.equal(g0, g1), torch.equal(w0, w1)), flush=True)
            for j, (a, b) in enumerate(zip(e0, e1)):         # the geometry set the step read (A), its cloud, the NEXT batch's seeds
                if not torch.equal(a, b):
                    d = a != b
                    print("    step input %d of %d %s %s: %d of %d differ, first at %d" %
                          (j, len(e0), tuple(a.shape), a.dtype, int(d.sum()), a.numel(), int(d.flatten().nonzero()[0])), flush=True)
            if e0 and bad <= 8:              # the cloud and both index sequences, for an offline replay of the sampling (tools/dbg/fps_event.py)
                os.makedirs("gpurun_out", exist_ok=True)
                torch.save({"P": e0[-2].cpu(), "idx_ref": e0[0].cpu(), "idx_bad": e1[0].cpu(), "P_bad": e1[-2].cpu()},
                           "gpurun_out/fps_event_%d.pt" % bad)
            for tag, e in (("first run", e0), ("this run", e1)):     # did the sampling kernel hold the coordinates the step's cloud has?
                if e:
                    P, idx, cen = e[-2], e[0].long(), e[1]
                    want = torch.gather(P, 1, idx.unsqueeze(2).expand(-1, -1, 3))
                    d = (want != cen).any(2)
                    print("    %s: centres != cloud[fps_idx] at %d of %d samples%s" %
                          (tag, int(d.sum()), d.numel(), (", first (cloud, sample) %s" % (tuple(d.nonzero()[0].tolist()),)) if bool(d.any()) else ""))
                    if bool(d.any()):
                        b_, i_ = d.nonzero()[0].tolist()
                        print("      idx %d centre %s cloud %s" % (int(idx[b_, i_]), cen[b_, i_].tolist(), want[b_, i_].tolist()))
            off = 0
            for n, k in list(zip(names, sizes))[:6]:
                a, b = g0[off:off + k], g1[off:off + k]
                if not torch.equal(a, b):
                    d = (a != b)
                    print("    grad %-40s %7d of %7d differ, max |d| %.3e (max |g| %.3e) first at %d" %
                          (n, int(d.sum()), k, float((a - b).abs().max()), float(a.abs().max()), int(d.nonzero()[0])), flush=True)
                off += k
            break
    print("RESULT", kind, "%d of %d runs differ" % (bad, runs), flush=True)


if __name__ == "__main__":
    main()
