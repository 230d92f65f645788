"""GPU: BASELINE.json's configurations AT FULL SIZE and IN BENCH MODE against the oracle.

What `bench.py` times is the bf16 product path replayed as ONE hipGraph (network forward, fused losses, device-side
assignment, fits, backward, gradient packing, Adam || the next batch's geometry on the forked branch).  Here exactly
that trainer is built (same constructor arguments as bench.py; dropout neutralised and lr = 0 so that the oracle can
follow: reference caller Utils/training_utils.py:136-150), a REPLAYED step's index tensors are pulled out of the
graph's static geometry buffers and compared bit for bit with oracle/geometry, and its six losses / heads / flat
gradient with oracle/pn2.training_step_losses on the same 16 x 8192 (config 2) or 32 x 8192 (config 3) batch.

Stated bf16 tolerances (the MLP stacks run on bf16 operands with fp32 accumulation, activations are stored in bf16;
~0.3 % of the ReLU masks / arg-maxes flip against an fp32 evaluation, DESIGN.md "Numerics"): each loss within 2 % of
the oracle's (+1e-3 absolute), heads within 3e-2 relative L2, the flat gradient within 15 % relative L2 with cosine
> 0.99, and the Hungarian matching identical on >= 90 % of the GT instances.  The achieved figures are printed.
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

GLOBAL = dict(B=16, K=28, mult=dict(miou=1.0, normal=1.0, type=1.0, parameter=1.0, residue=1.0, total=1.0))
LOCAL = dict(B=32, K=21, mult=dict(miou=1.0, normal=1.0, type=1.0, parameter=0.0, residue=0.0, total=1.0))
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


def _compare(model, tr, batch_cpu, starts, out, cfg, tag):
    st, ref, aux = _oracle_step(model, batch_cpu, starts, cfg["mult"])
    got = np.array([float(v) for v in out[:6]])
    want = np.array([float(v) for v in ref[:6]])
    print("[%s] losses product %s" % (tag, np.round(got, 5)))
    print("[%s] losses oracle  %s" % (tag, np.round(want, 5)))
    assert np.all(np.abs(got - want) <= 2e-2 * np.abs(want) + 1e-3), (got, want)
    # heads
    Y = model.heads_packed.detach().float().cpu()
    K = cfg["K"]
    for name, a, b in (("X", Y[..., :3], aux["heads"][0]), ("T", Y[..., 3:7], aux["heads"][1]), ("W", Y[..., 7:7 + K], aux["heads"][2])):
        e = float((a - b.detach()).norm() / b.detach().norm())
        print("[%s] head %s rel L2 %.2e" % (tag, name, e))
        assert e < 3e-2, (name, e)
    # matching (the product's, inside the graph) vs the oracle's on its own fp32 memberships
    from oracle import spfn as ospfn
    m_ref = ospfn.hungarian_matching(torch.softmax(aux["heads"][2].detach(), 2), batch_cpu["I_gt"]).numpy()
    m_got = tr._graph["match"].cpu().numpy()
    n_gt = (batch_cpu["I_gt"].max(1)[0] + 1).numpy()
    agree = np.mean([np.mean(m_got[b, :n_gt[b]] == m_ref[b, :n_gt[b]]) for b in range(len(n_gt))])
    print("[%s] matching agreement %.3f" % (tag, agree))
    assert agree >= 0.9
    # flat gradient, in the bucket's parameter order
    named = {id(p): n for n, p in model.named_parameters()}
    flat_ref = torch.cat([(st[named[id(p)]].grad if st[named[id(p)]].grad is not None else torch.zeros_like(st[named[id(p)]])).reshape(-1)
                          for p in tr.bucket.params])
    flat = tr.bucket.flat.detach().cpu()
    assert torch.isfinite(flat).all()
    # (conv biases in front of a training-mode BatchNorm: exactly 0 here, rounding noise ~1e-7 in the reference)
    rel = float((flat - flat_ref).norm() / flat_ref.norm())
    cos = float(torch.dot(flat, flat_ref) / (flat.norm() * flat_ref.norm()))
    print("[%s] flat gradient: rel L2 %.3e, cosine %.5f, |g| product %.4e oracle %.4e" % (tag, rel, cos, float(flat.norm()),
                                                                                          float(flat_ref.norm())))
    assert rel < 0.15 and cos > 0.99, (rel, cos)
    assert tr.skipped_steps == 0


def test_config2_global_spfn_bench_mode_16x8192():
    model, tr, batch_cpu, batch = _bench_trainer(GLOBAL, seed=1000)          # bench.py's batch (seed 1000 + rank)
    starts, out = _replayed_step(tr, batch)
    _check_geometry(tr._graph["geomA"], batch_cpu["P"].numpy(), starts)
    _compare(model, tr, batch_cpu, starts, out, GLOBAL, "config 2")


def test_config3_local_spfn_bench_mode_32x8192():
    model, tr, batch_cpu, batch = _bench_trainer(LOCAL, seed=2000)
    starts, out = _replayed_step(tr, batch)
    _check_geometry(tr._graph["geomA"], batch_cpu["P"].numpy(), starts)
    assert float(out[4]) == 0.0 and float(out[5]) == 0.0                     # fitter losses off (config_localSPFN.yml:10-11)
    _compare(model, tr, batch_cpu, starts, out, LOCAL, "config 3")
