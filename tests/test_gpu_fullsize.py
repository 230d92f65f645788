"""GPU: BASELINE.json's configurations AT FULL SIZE and IN BENCH MODE against the oracle.

What `bench.py` times is the bf16 product path replayed as ONE hipGraph (network forward, fused losses, device-side
assignment, fits, backward, gradient packing, Adam || the next batch's geometry on the forked branch).  Here exactly
that trainer is built (same constructor arguments as bench.py; dropout neutralised and lr = 0 so that the oracle can
follow: reference caller Utils/training_utils.py:136-150), a REPLAYED step's index tensors are pulled out of the
graph's static geometry buffers and compared bit for bit with oracle/geometry, and its six losses / heads / flat
gradient with oracle/pn2.training_step_losses on the same 16 x 8192 (config 2) or 32 x 8192 (config 3) batch.

Comparisons, tolerances stated here, achieved figures printed:

 (A) fp32 compute mode of the product (same HIP geometry, fused losses and fitters; PyTorch fp32 MLPs), eager, against
     the fp32 oracle at full size: heads 1e-3 relative L2, each loss 1e-3 relative (+1e-5), flat gradient 2e-2
     relative L2, matching identical.  This pins everything but the bf16 MLP stacks to the reference at full size.
 (B) every fused bf16 MLP stack of the step AT ITS BENCH SHAPE (sa1: 524288 x 3 -> 64 -> 64 -> 128 with the 64-row
     max-pool ... fc1, heads) on exactly the inputs it sees in that step ("teacher forced": the inputs are recorded
     from the fp32-mode forward of (A)): forward within 3e-2 relative L2 of PyTorch fp32 and 1e-2 of the fp32
     evaluation with the bf16 storage roundings made explicit; input / parameter gradients within 4e-2 of the
     latter (12e-2 for the max-pooled stacks: ties).  This pins the bf16 kernels themselves (streaming GEMM, BN statistics, pooling, weight gradients) at the
     sizes bench.py runs them.
 (C) the replayed bf16 graph end to end against the fp32 oracle: each loss within 6 % (+3e-3), matching agreement
     >= 0.5, flat-gradient cosine > 0.2 — loose ON PURPOSE, and bit-identical to the eager bf16 step.  Training-mode
     BatchNorm amplifies bf16's 0.2 % OPERAND rounding (which any bf16 GEMM has, fused or not) layer by layer —
     measured here: l1 0.4 %, l2 1.3 %, l3 2.7 %, l4 12 %, heads 30 % against the fp32 mode, the same in eager mode and
     at 2 x 2048; a CPU emulation with ONLY the GEMM operands rounded to bf16 gives 25 % — while in evaluation mode
     (running statistics) the same kernels sit at 0.2-0.5 % (tests/test_gpu_config5.py).  End-to-end agreement of a
     randomly initialised network therefore says little about the kernels; (A) and (B) do, and
     tests/test_gpu_trainer.py shows the two modes train alike.
The integer outputs and the fp32 interpolation weights have NO tolerance: bit-exact."""
import contextlib
import io

import numpy as np
import pytest
import torch

from cpfn_amd import synthetic
from oracle import geometry as og
from oracle import pn2 as opn2

pytestmark = pytest.mark.gpu

# `c_floor`: absolute bounds on (C), the replayed bf16 step against the fp32 oracle, at TWICE the error achieved on the final build of
# round 6 (profiles/r06_fullsize_parity.txt: config 2 heads 0.31 / 0.33 / 0.26, agreement 0.66, gradient rel-L2 1.05, cosine 0.33, worst
# loss 2.3e-2 relative; config 3 heads 0.32 / 0.33 / 0.27, agreement 0.67, gradient 0.80, cosine 0.68): heads rel-L2, 1 - agreement,
# gradient rel-L2, 1 - cosine (capped at 0.85), worst relative loss error — VERDICT r5 #7
GLOBAL = dict(B=16, K=28, mult=dict(miou=1.0, normal=1.0, type=1.0, parameter=1.0, residue=1.0, total=1.0),
              c_floor=dict(head=0.66, match=0.33, grad_rel=2.1, grad_cos=0.15, loss_rel=4.6e-2))
LOCAL = dict(B=32, K=21, mult=dict(miou=1.0, normal=1.0, type=1.0, parameter=0.0, residue=0.0, total=1.0),
             c_floor=dict(head=0.66, match=0.34, grad_rel=1.6, grad_cos=0.37, loss_rel=4.6e-2))
N = 8192


def dev():
    return torch.device("cuda:0")


def _bench_trainer(cfg, seed):
    """bench.py's trainer: default PyTorch init under manual_seed(0), bf16, graphs, FlatAdam — plus dropout off, lr 0."""
    from cpfn_amd import training
    from cpfn_amd.PointNet2 import pn2_network
    from cpfn_amd.SPFN import fitter_factory
    with contextlib.redirect_stdout(io.StringIO()):
        fitter_factory.register_primitives(training.GLOBAL_SPFN_CLASSES)
    torch.manual_seed(0)
    model = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, cfg["K"]]).to(dev())
    model.set_compute_dtype(torch.bfloat16)
    model.dropout_p = 0.0
    tr = training.SPFNTrainer(model, batch_size=cfg["B"], use_graphs=True, require_graphs=True, init_learning_rate=0.0,
                              multipliers=cfg["mult"])
    batch_cpu = synthetic.training_batch(cfg["B"], N, cfg["K"], seed=seed)
    return model, tr, batch_cpu, {k: v.to(dev()) for k, v in batch_cpu.items()}


def _replayed_step(tr, batch):
    """Warm up, capture, then one more REPLAYED step whose FPS starts are known: returns (starts of the geometry that
    step consumed, its outputs)."""
    torch.manual_seed(77)
    for _ in range(4):                                   # 2 eager warm-ups, capture, first replay
        tr.step(batch, next_batch=batch)
    assert tr._graph is not None and tr._graph["single"], "the step must be ONE replayed graph"
    torch.cuda.synchronize()
    starts = tr._graph["start_dev"].clone().cpu()        # drawn for the geometry the NEXT replay consumes
    out = tr.step(batch, next_batch=batch)
    torch.cuda.synchronize()
    return (starts[0].long(), starts[1].long()), out


def _check_geometry(geomA, xyz, starts):
    """Static geometry buffers of the replayed graph vs the C oracle: everything bit-exact."""
    f1 = og.farthest_point_sample(xyz, 512, starts[0].numpy())
    assert np.array_equal(geomA["sa1"]["fps_idx"].cpu().numpy(), f1.astype(np.int32)), "sa1 FPS"
    l1 = np.take_along_axis(xyz, f1[:, :, None], axis=1)
    assert np.array_equal(geomA["sa1"]["new_xyz"].cpu().numpy(), l1)
    f2 = og.farthest_point_sample(l1, 128, starts[1].numpy())
    assert np.array_equal(geomA["sa2"]["fps_idx"].cpu().numpy(), f2.astype(np.int32)), "sa2 FPS"
    l2 = np.take_along_axis(l1, f2[:, :, None], axis=1)
    assert np.array_equal(geomA["sa1"]["scales"][0][0].cpu().numpy(), og.ball_query(0.2, 64, xyz, l1).astype(np.int32)), "sa1 ball"
    assert np.array_equal(geomA["sa2"]["scales"][0][0].cpu().numpy(), og.ball_query(0.4, 64, l1, l2).astype(np.int32)), "sa2 ball"
    for lvl, q, p in (("sfp3", xyz, l1), ("sfp2", l1, l2)):
        d, i = og.three_nn(q, p)
        assert np.array_equal(geomA[lvl]["nn_idx"].cpu().numpy(), i.astype(np.int32)), lvl
        assert np.array_equal(geomA[lvl]["nn_w"].cpu().numpy().view(np.uint32), og.three_weights(d).view(np.uint32)), lvl
    return f1, f2


def _oracle_step(model, batch_cpu, starts, mult):
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    st = {k: (v.requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in state.items()}
    prev = torch.get_num_threads()
    torch.set_num_threads(16)            # torch-CPU is slower with all 256 hardware threads of the GPU box (bench.py)
    try:
        out, aux = opn2.training_step_losses(st, batch_cpu, starts, multipliers=mult, return_aux=True)
        out[0].backward()
    finally:
        torch.set_num_threads(prev)
    return st, out, aux


def _flat_ref(model, tr, st):
    named = {id(p): n for n, p in model.named_parameters()}
    return torch.cat([(st[named[id(p)]].grad if st[named[id(p)]].grad is not None else torch.zeros_like(st[named[id(p)]])).reshape(-1)
                      for p in tr.bucket.params])


def _report(tag, got_losses, Y, match, flat, ref, aux, st, model, tr, batch_cpu, K):
    """-> dict of achieved errors of (losses, heads, matching, flat gradient) against one oracle run."""
    from oracle import spfn as ospfn
    got = np.array([float(v) for v in got_losses[:6]])
    want = np.array([float(v) for v in ref[:6]])
    res = {"loss_abs": np.abs(got - want), "loss_ref": np.abs(want)}
    print("[%s] losses product %s" % (tag, np.round(got, 5)))
    print("[%s] losses oracle  %s" % (tag, np.round(want, 5)))
    for name, a, b in (("X", Y[..., :3], aux["heads"][0]), ("T", Y[..., 3:7], aux["heads"][1]), ("W", Y[..., 7:7 + K], aux["heads"][2])):
        res["head_" + name] = float((a - b.detach()).norm() / b.detach().norm())
    m_ref = ospfn.hungarian_matching(torch.softmax(aux["heads"][2].detach(), 2), batch_cpu["I_gt"]).numpy()
    n_gt = (batch_cpu["I_gt"].max(1)[0] + 1).numpy()
    res["match"] = float(np.mean([np.mean(match[b, :n_gt[b]] == m_ref[b, :n_gt[b]]) for b in range(len(n_gt))]))
    fr = _flat_ref(model, tr, st)
    res["grad_rel"] = float((flat - fr).norm() / fr.norm())
    res["grad_cos"] = float(torch.dot(flat, fr) / (flat.norm() * fr.norm()))
    print("[%s] heads rel L2 X %.2e T %.2e W %.2e | matching agreement %.3f | flat gradient rel L2 %.3e cosine %.5f (|g| %.4e vs %.4e)"
          % (tag, res["head_X"], res["head_T"], res["head_W"], res["match"], res["grad_rel"], res["grad_cos"], float(flat.norm()),
             float(fr.norm())))
    return res


def _losses_within(res, rel, abs_):
    return bool(np.all(res["loss_abs"] <= rel * res["loss_ref"] + abs_))


def _rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def _teacher_forced_stacks(model, batch, starts, tag):
    """(B): record every run_stack call of an fp32-mode forward (its inputs at bench size), then run each stack alone
    in bf16 (the fused HIP path), in plain fp32 and in fp32-with-explicit-bf16-roundings on those inputs and one
    random upstream gradient."""
    from cpfn_amd import fused_mlp, mlp
    from test_gpu_fused_mlp import _emulated_stack
    calls, orig = [], mlp.run_stack

    def spy(x, convs, bns, dtype=torch.float32, pool_k=None, xyz_rows=None, dropout=None, xyz_tail=None, handover=None, gather=None,
            **side_channels):            # (join_out, top_ride, ...: objects of the bf16 path's backward; unused in fp32 mode)
        assert xyz_tail is None and gather is None          # (fp32 mode: the grouped rows arrive concatenated)
        calls.append((None if x is None else x.detach().clone(), convs, bns, pool_k, None if xyz_rows is None else xyz_rows.detach().clone()))
        return orig(x, convs, bns, dtype, pool_k=pool_k, xyz_rows=xyz_rows, dropout=dropout)

    model.set_compute_dtype(torch.float32)
    mlp.run_stack = spy
    try:
        with torch.no_grad():
            model(batch["P"], fps_start=starts)
    finally:
        mlp.run_stack = orig
        model.set_compute_dtype(torch.bfloat16)
    assert len(calls) == 7, len(calls)          # sa1, sa2, sa3, sfp1, sfp2, sfp3, fc1
    gen = torch.Generator().manual_seed(3)
    bad = []
    for idx, (x, convs, bns, pool_k, xyz) in enumerate(calls):
        widths = [c.weight.shape[0] for c in convs]
        rows = (xyz if x is None else x).shape[0]
        gout = torch.randn(rows // pool_k if pool_k else rows, widths[-1], generator=gen).to(batch["P"].device)
        params = [p for c in convs for p in (c.weight,)] + [p for b in bns for p in (b.weight, b.bias)]
        saved = [(b.running_mean.clone(), b.running_var.clone(), b.num_batches_tracked.clone()) for b in bns]

        def run(kind):
            for p in params:
                p.grad = None
            for b, (rm, rv, nb) in zip(bns, saved):      # every run starts from the same running statistics
                b.running_mean.copy_(rm); b.running_var.copy_(rv); b.num_batches_tracked.copy_(nb)
            xin = None if x is None else x.clone().requires_grad_(True)
            if kind == "emulated":
                y = _emulated_stack(xin, convs, bns, pool_k, xyz, pool_arg=kernel_arg.get("arg"))
            else:
                y = mlp.run_stack(xin, convs, bns, kind, pool_k=pool_k, xyz_rows=xyz)
            (y.float() * gout).sum().backward()
            return y.detach().float(), None if xin is None else xin.grad.float(), [p.grad.clone() for p in params]
        # the fused path first, with the arg-max rows its max-pool kernel chose recorded for the emulation
        # (round 6: the pooling of the large stacks starts in the last GEMM's epilogue and ends in bn_pool_finish)
        kernel_arg, orig_pool, orig_finish = {}, fused_mlp.bn_relu_maxpool, fused_mlp.bn_pool_finish

        def pool_spy(*a, _f=None, **k):
            res = _f(*a, **k)
            kernel_arg["arg"] = res[1].clone()
            return res
        fused_mlp.bn_relu_maxpool = lambda *a, **k: pool_spy(*a, _f=orig_pool, **k)
        fused_mlp.bn_pool_finish = lambda *a, **k: pool_spy(*a, _f=orig_finish, **k)
        try:
            y16, gx16, gp16 = run(torch.bfloat16)
        finally:
            fused_mlp.bn_relu_maxpool, fused_mlp.bn_pool_finish = orig_pool, orig_finish
        assert (pool_k is None) == ("arg" not in kernel_arg)
        y32, gx32, gp32 = run(torch.float32)
        yem, gxem, gpem = run("emulated")
        e32, eem = _rel(y16, y32), _rel(y16, yem)
        egx = _rel(gx16[:, :gxem.shape[1]], gxem) if gxem is not None else 0.0
        egp = max(_rel(a, b) for a, b in zip(gp16, gpem))
        print("[%s (B) stack %d: %d rows, %s -> %s%s] fwd vs fp32 %.2e, vs bf16-rounding emulation %.2e | dX %.2e | worst dparam %.2e"
              % (tag, idx, rows, "xyz" if x is None else x.shape[1], widths, " pool %d" % pool_k if pool_k else "", e32, eem, egx, egp))
        # (pooled stacks: the emulation pools the rows the kernel's arg-max chose — see _emulated_stack — so they meet the
        #  dense stacks' bound: 7.3e-2 / 7.7e-2 for sa2 before, when near-ties inside a 64-row group flipped the arg-max)
        gtol = 4e-2
        bad.append((idx, e32, eem, egx, egp)) if not (e32 < 3e-2 and eem < 1e-2 and egx < gtol and egp < gtol) else None
    assert not bad, bad
    for p in model.parameters():
        p.grad = None


def _compare(model, tr, batch_cpu, batch, starts, out, cfg, tag):
    from cpfn_amd.SPFN import fused_losses
    K = cfg["K"]
    Yg = model.heads_packed.detach().float().clone()
    match = tr._graph["match"].cpu().numpy().copy()
    flat = tr.bucket.flat.detach().cpu().clone()
    assert torch.isfinite(flat).all() and tr.skipped_steps == 0
    out = tuple(o.clone() for o in out)
    # the replayed graph is the eager bf16 step, bit for bit (same kernels, same order, no atomics)
    with torch.no_grad():
        model(batch["P"], fps_start=starts)
    assert torch.equal(model.heads_packed.detach().float(), Yg), "graph replay != eager bf16 forward"
    st, ref, aux = _oracle_step(model, batch_cpu, starts, cfg["mult"])
    # ---- (C) replayed bf16 graph vs the fp32 oracle
    rc = _report(tag + " (C) bf16 graph vs fp32 oracle", out, Yg.cpu(), match, flat, ref, aux, st, model, tr, batch_cpu, K)
    # ---- (A) fp32 compute mode of the product, eager, vs the fp32 oracle
    model.set_compute_dtype(torch.float32)
    for p in model.parameters():
        p.grad = None
    X, T, W, _, _ = model(batch["P"], fps_start=starts)
    Ya = torch.cat([X, T, W], 2)
    oa = fused_losses.fused_losses(batch["P"], Ya, batch, cfg["mult"], tr.classes)
    oa[0].backward()
    flat_a = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in tr.bucket.params]).cpu()
    S = fused_losses.SegStats.apply(torch.softmax(Ya.detach()[..., 7:], 2), batch["I_gt"])
    match_a = fused_losses.hungarian_device(S, fused_losses.count_gt(batch["I_gt"])).cpu().numpy()
    ra = _report(tag + " (A) fp32 mode vs fp32 oracle", oa, Ya.detach().cpu(), match_a, flat_a, ref, aux, st, model, tr, batch_cpu, K)
    # ---- (P) what the network itself does to an error of bf16 size: the product's fp32 mode again, with Gaussian noise of the
    #      size of sa1's bf16 error (measured here: ~0.45 % relative L2) added to sa1's output and NOTHING else changed.
    #      tools/bf16_stage_probe.py: with every stack after sa1 in fp32 the heads still sit 29 % from the all-fp32 heads — the
    #      randomly initialised network with batch statistics amplifies a perturbation ~65 x — so the end-to-end deviation of
    #      the bf16 step is held to THIS run's, not to an absolute figure.
    l1 = {}
    orig_rows = model.sa1.forward_rows

    def spy(*a, **k):
        xyz1, f1, aux1 = orig_rows(*a, **k)
        if l1.get("noise"):
            g = torch.Generator(device=f1.device).manual_seed(7)
            nz = torch.randn(f1.shape, generator=g, device=f1.device, dtype=torch.float32)
            f1 = (f1.float() + nz * (l1["noise"] * f1.float().norm() / nz.norm())).to(f1.dtype)
        l1["out"] = f1.detach().float().clone()
        return xyz1, f1, aux1
    model.sa1.forward_rows = spy
    try:
        with torch.no_grad():
            model(batch["P"], fps_start=starts)
            l1_fp32 = l1["out"]
            model.set_compute_dtype(torch.bfloat16)
            model(batch["P"], fps_start=starts)
            e1 = _rel(l1["out"], l1_fp32)
            model.set_compute_dtype(torch.float32)
        l1["noise"] = e1
        for p in model.parameters():
            p.grad = None
        X, T, W, _, _ = model(batch["P"], fps_start=starts)
        Yp = torch.cat([X, T, W], 2)
        op = fused_losses.fused_losses(batch["P"], Yp, batch, cfg["mult"], tr.classes)
        op[0].backward()
    finally:
        model.sa1.forward_rows = orig_rows
    flat_p = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in tr.bucket.params]).cpu()
    S = fused_losses.SegStats.apply(torch.softmax(Yp.detach()[..., 7:], 2), batch["I_gt"])
    match_p = fused_losses.hungarian_device(S, fused_losses.count_gt(batch["I_gt"])).cpu().numpy()
    rp = _report(tag + " (P) fp32 mode, sa1 output + %.2e noise, vs fp32 oracle" % e1, op, Yp.detach().cpu(), match_p, flat_p, ref, aux, st,
                 model, tr, batch_cpu, K)
    model.set_compute_dtype(torch.bfloat16)
    for p in model.parameters():
        p.grad = None
    assert _losses_within(ra, 1e-3, 1e-5) and max(ra["head_X"], ra["head_T"], ra["head_W"]) < 1e-3, ra
    assert ra["match"] == 1.0 and ra["grad_rel"] < 2e-2, ra
    # (C) against (P): the bf16 step is as far from the fp32 oracle as an fp32 step whose sa1 output carries one bf16 error
    assert e1 < 8e-3, e1
    for k in ("head_X", "head_T", "head_W"):
        assert rc[k] <= 1.3 * rp[k] + 1e-3, (k, rc[k], rp[k])
    # (the matching is discrete: one noise draw moves the agreement by ~0.1; config 2 measured 0.64 bf16 / 0.75 noise)
    assert rc["match"] >= rp["match"] - 0.15 and rc["grad_cos"] >= rp["grad_cos"] - 0.1, (rc, rp)
    # ... and loose ABSOLUTE floors beside the relative bounds: the reference point itself must be sane — a noise
    # run that had drifted to cosine ~0 / agreement ~0 would let a broken bf16 backward pass (measured: noise run cosine 0.22,
    # agreement 0.75; bf16 graph 0.32 / 0.64 at config 2)
    assert rp["grad_cos"] > 0.12 and rp["match"] >= 0.5 and max(rp["head_X"], rp["head_T"], rp["head_W"]) < 0.5, rp
    assert rc["grad_cos"] > 0.1 and rc["match"] >= 0.4, rc
    assert rc["grad_rel"] <= 1.3 * rp["grad_rel"] + 1e-2, (rc, rp)
    fl = cfg["c_floor"]
    assert _losses_within(rc, fl["loss_rel"], 3e-3), rc
    assert max(rc["head_X"], rc["head_T"], rc["head_W"]) < fl["head"] and rc["match"] >= fl["match"], (rc, fl)
    assert rc["grad_rel"] < fl["grad_rel"] and rc["grad_cos"] > fl["grad_cos"], (rc, fl)
    # ---- (B) the fused bf16 stacks one by one at bench size
    _teacher_forced_stacks(model, batch, starts, tag)


def test_config2_global_spfn_bench_mode_16x8192():
    model, tr, batch_cpu, batch = _bench_trainer(GLOBAL, seed=1000)          # bench.py's batch (seed 1000 + rank)
    starts, out = _replayed_step(tr, batch)
    _check_geometry(tr._graph["geomA"], batch_cpu["P"].numpy(), starts)
    _compare(model, tr, batch_cpu, batch, starts, out, GLOBAL, "config 2")


def test_config3_local_spfn_bench_mode_32x8192():
    model, tr, batch_cpu, batch = _bench_trainer(LOCAL, seed=2000)
    starts, out = _replayed_step(tr, batch)
    _check_geometry(tr._graph["geomA"], batch_cpu["P"].numpy(), starts)
    assert float(out[4]) == 0.0 and float(out[5]) == 0.0                     # fitter losses off (config_localSPFN.yml:10-11)
    _compare(model, tr, batch_cpu, batch, starts, out, LOCAL, "config 3")
