"""CPU, world_size 2, gloo: the data-parallel path of cpfn_amd.training (flat gradient bucket,
one all-reduce per step, rank-0 broadcast) with the LocalSPFN loss configuration
(residue / parameter multipliers 0: Configs/config_localSPFN.yml:10-11), whose losses are plain
torch ops and therefore run without a GPU.  Data parallelism over clouds must reproduce
single-process training on the concatenated batch (no BatchNorm in the stand-in network)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cpfn_amd import synthetic, training


class TinyHeads(torch.nn.Module):
    """Stand-in with PointNet2's forward contract: P [B,N,3] -> [X, T, W, l3, feat]."""

    def __init__(self, K=21):
        super().__init__()
        self.body = torch.nn.Linear(3, 32)
        self.hx, self.ht, self.hw = torch.nn.Linear(32, 3), torch.nn.Linear(32, 4), torch.nn.Linear(32, K)

    def forward(self, P, fps_start=None):
        f = torch.tanh(self.body(P))
        return [self.hx(f), self.ht(f), self.hw(f), None, None]


LOCAL_MULT = dict(miou=1.0, normal=1.0, type=1.0, parameter=0.0, residue=0.0, total=1.0)


def _batch(B, seed):
    b = synthetic.training_batch(B, N=256, n_max_instances=21, n_prims=4, n_inst_points=16, seed=seed)
    return b


def _cat(batches):
    return {k: torch.cat([b[k] for b in batches], 0) for k in batches[0]}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_RENDEZVOUS_ERRORS = ("EADDRINUSE", "ddress already in use", "Connection refused", "Connection reset", "connectFullMesh")


def _spawn(fn, world, *args):
    """mp.spawn with a fresh port; a port another process took between _free_port() and the store's bind (seen once in the
    round-5 runs) gets two more tries — any other failure is the test's."""
    for attempt in range(3):
        try:
            mp.spawn(fn, args=(world, _free_port()) + args, nprocs=world, join=True)
            return
        except Exception as e:          # noqa: BLE001
            if attempt == 2 or not any(s in str(e) for s in _RENDEZVOUS_ERRORS):
                raise


def _worker(rank, world, port, out_dir, collective="all_reduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                 # different init per rank: broadcast must fix it
    model = TinyHeads()
    training.broadcast_parameters(model)
    tr = training.SPFNTrainer(model, batch_size=2 * world, multipliers=LOCAL_MULT, fused_adam=False)
    tr.bucket.collective = collective             # (what CPFN_DP_COLLECTIVE selects)
    assert tr.bucket.padded.numel() % world == 0 and tr.bucket.flat.data_ptr() == tr.bucket.padded.data_ptr()
    for step in range(3):
        out = tr.step(_batch(2, seed=10 * step + rank))
    assert tr.skipped_steps == 0 and tr.global_step == 3
    assert not tr.bucket.padded[tr.bucket.flat.numel():].any()          # the padding never picks anything up
    torch.save({k: v.clone() for k, v in model.state_dict().items()}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_data_parallel_matches_single_process(tmp_path):
    world = 2
    _spawn(_worker, world, str(tmp_path))
    sd = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    for k in sd[0]:
        assert torch.equal(sd[0][k], sd[1][k]), "replicas diverged: %s" % k
    # single process on the concatenated batches, same initial weights as rank 0
    torch.manual_seed(100)
    model = TinyHeads()
    tr = training.SPFNTrainer(model, batch_size=4, multipliers=LOCAL_MULT, fused_adam=False)
    for step in range(3):
        tr.step(_cat([_batch(2, seed=10 * step + r) for r in range(world)]))
    for k, v in model.state_dict().items():
        torch.testing.assert_close(v, sd[0][k], rtol=2e-4, atol=2e-6)


@pytest.mark.timeout(300)
def test_reduce_scatter_all_gather_layout_gives_the_same_replicas(tmp_path):
    """CPFN_DP_COLLECTIVE=rs_ag (reduce-scatter + all-gather on the padded flat bucket) against the default single
    all-reduce: replicas identical across the ranks in both layouts, and — two ranks: a + b is one rounding either way —
    identical between the layouts."""
    world = 2
    res = {}
    for mode in ("all_reduce", "rs_ag"):
        d = tmp_path / mode
        d.mkdir()
        _spawn(_worker, world, str(d), mode)
        sd = [torch.load(os.path.join(str(d), "rank%d.pt" % r)) for r in range(world)]
        for k in sd[0]:
            assert torch.equal(sd[0][k], sd[1][k]), "replicas diverged (%s): %s" % (mode, k)
        res[mode] = sd[0]
    for k in res["all_reduce"]:
        assert torch.equal(res["all_reduce"][k], res["rs_ag"][k]), k


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [3, 4])
def test_reduce_scatter_all_gather_layout_at_three_and_four_ranks(tmp_path, world):
    """The flat bucket is padded to multiples of 3360 elements (+ the fault slot) so that it splits into 16-byte-aligned
    shards for EVERY world size up to 8; round 3 only ran two ranks.  Three and four gloo ranks, both collective layouts:
    replicas bit-identical across the ranks, the two layouts equal up to the order of the cross-rank additions."""
    res = {}
    for mode in ("all_reduce", "rs_ag"):
        d = tmp_path / mode
        d.mkdir()
        _spawn(_worker, world, str(d), mode)
        sd = [torch.load(os.path.join(str(d), "rank%d.pt" % r)) for r in range(world)]
        for r in range(1, world):
            for k in sd[0]:
                assert torch.equal(sd[0][k], sd[r][k]), "replicas diverged (%s, rank %d): %s" % (mode, r, k)
        res[mode] = sd[0]
    for k in res["all_reduce"]:
        torch.testing.assert_close(res["all_reduce"][k], res["rs_ag"][k], rtol=1e-5, atol=1e-7)
    b = training.FlatGradBucket(TinyHeads())
    assert all(b.padded.numel() % w == 0 and (b.padded.numel() // w * 4) % 16 == 0 for w in range(1, 9))
    assert b.fault_slot.data_ptr() == b.flat.data_ptr() + 4 * b.flat.numel() and b.padded.numel() > b.flat.numel()


def _fault_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    model = TinyHeads()
    training.broadcast_parameters(model)
    tr = training.SPFNTrainer(model, batch_size=2 * world, multipliers=LOCAL_MULT, fused_adam=False)
    snap = lambda: {k: v.clone() for k, v in model.state_dict().items()}
    for step in range(2):
        tr.step(_batch(2, seed=10 * step + rank))
    before = snap()
    moments_before = [v.clone() for st_ in tr.optimizer.state.values() for v in st_.values() if torch.is_tensor(v)]
    if rank == 1:
        tr.raise_fault("injected by the test on step 3")
    err = None
    try:
        tr.step(_batch(2, seed=20 + rank))                       # step 3: the fault word rides on the all-reduce
    except RuntimeError as e:
        err = str(e)
    after = snap()
    moments_after = [v.clone() for st_ in tr.optimizer.state.values() for v in st_.values() if torch.is_tensor(v)]
    torch.save({"before": before, "after": after, "err": err, "skipped": tr.skipped_steps, "fault_slot": float(tr.bucket.fault_slot),
                "moments_equal": all(torch.equal(a, b) for a, b in zip(moments_before, moments_after)),
                "grad_norm": float(tr.bucket.flat.norm())}, os.path.join(out_dir, "rank%d.pt" % rank))
    try:
        dist.destroy_process_group()
    except Exception:
        pass


@pytest.mark.timeout(300)
def test_one_rank_s_fault_skips_the_step_on_every_rank_and_both_fail_fast(tmp_path):
    """VERDICT r4 #8 / ADVICE r3: the fault slot's BEHAVIOUR, not its layout.  Rank 1 raises its fault word before step 3; the word
    rides on the step's one all-reduce in the bucket's fault slot; BOTH ranks skip the optimizer on that step (weights and Adam
    moments bit-identical before / after, on both ranks, replicas still equal), the raising rank errors with its reason and the
    peer errors too — at once, not after a collective time-out (the test's own time-out would catch a hang)."""
    world = 2
    _spawn(_fault_worker, world, str(tmp_path))
    r = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % k)) for k in range(world)]
    for k in range(world):
        assert r[k]["skipped"] == 1 and r[k]["fault_slot"] == 0.5 and r[k]["moments_equal"], (k, r[k]["skipped"], r[k]["fault_slot"])
        assert r[k]["grad_norm"] > 0                               # a real averaged gradient WAS there to be applied
        for name in r[k]["before"]:
            assert torch.equal(r[k]["before"][name], r[k]["after"][name]), "rank %d applied the step: %s" % (k, name)
            assert torch.equal(r[0]["after"][name], r[k]["after"][name]), "replicas diverged: %s" % name
    assert r[1]["err"] is not None and "this rank raised its fault word (injected by the test on step 3)" in r[1]["err"]
    assert r[0]["err"] is not None and "PEER rank raised its fault word" in r[0]["err"]


# ---- the REAL network: PointNet2 (fp32 compute mode) + SPFNTrainer + FlatGradBucket on two gloo ranks ------------------
# The device kernels of the geometry path are replaced by oracle-backed CPU stand-ins for the duration of the test
# (tests/cpu_standins.py: test infrastructure); everything else — module tree, per-replica BatchNorm, flat bucket, the
# one all-reduce per step, Adam — is the product's host code.
def _real_batch(rank, step):
    return synthetic.training_batch(2, N=1024, n_max_instances=21, n_prims=5, n_inst_points=32, seed=100 * step + rank)


def _real_starts(rank, step):
    g = torch.Generator().manual_seed(1000 * step + rank)
    return (torch.randint(0, 1024, (2,), generator=g), torch.randint(0, 512, (2,), generator=g))


def _real_model():
    from cpfn_amd.PointNet2 import pn2_network
    m = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 21])
    m.load_state_dict(synthetic.synthetic_state_dict(synthetic.pointnet2_state_shapes(output_sizes=(3, 4, 21)), seed=5), strict=True)
    m.dropout_p = 0.0
    return m


def _real_worker(rank, world, port, out_dir):
    import cpu_standins
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)      # (same summation order as the in-process emulation: the Hungarian matching of a randomly
    #                                initialised network is discontinuous in rounding noise)
    with cpu_standins.installed():
        model = _real_model()
        if rank != 0:
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(1.0)               # a different start on rank 1: the broadcast must undo it
        training.broadcast_parameters(model)
        tr = training.SPFNTrainer(model, batch_size=2 * world, multipliers=LOCAL_MULT, fused_adam=False)
        for step in range(2):
            out = tr.step(_real_batch(rank, step), fps_start=_real_starts(rank, step))
            assert all(torch.isfinite(v) for v in out)
        assert tr.skipped_steps == 0
    torch.save({k: v.clone() for k, v in model.state_dict().items()}, os.path.join(out_dir, "real%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_data_parallel_real_pointnet2(tmp_path):
    """Two gloo ranks train the real PointNet2 for two steps; checked against an in-process emulation of the same
    semantics: one model copy per rank (its own BatchNorm statistics: per-replica BN, as SURVEY 8e prescribes), gradients
    averaged over the copies, one Adam step each."""
    import copy
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cpu_standins
    world = 2
    _spawn(_real_worker, world, str(tmp_path))
    sd = [torch.load(os.path.join(str(tmp_path), "real%d.pt" % r)) for r in range(world)]
    param_keys = [k for k, _ in _real_model().named_parameters()]
    for k in param_keys:
        assert torch.equal(sd[0][k], sd[1][k]), "replicas diverged: %s" % k
    assert any(not torch.equal(sd[0][k], sd[1][k]) for k in sd[0] if "running_mean" in k)      # BatchNorm stays per replica
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    with cpu_standins.installed():
        copies = [_real_model() for _ in range(world)]
        trainers = [training.SPFNTrainer(m, batch_size=2 * world, multipliers=LOCAL_MULT, fused_adam=False) for m in copies]
        for step in range(2):
            grads = []
            for r, (m, tr) in enumerate(zip(copies, trainers)):
                m.train()
                tr.bucket.zero()
                tr.losses(_real_batch(r, step), _real_starts(r, step))[0].backward()
                tr.bucket.collect()
                grads.append(tr.bucket.flat.clone())
            mean = sum(grads) / world
            for tr in trainers:
                tr.bucket.flat.copy_(mean)
                tr.optimizer.step()
    torch.set_num_threads(prev_threads)
    for r in range(world):
        ref = copies[r].state_dict()
        for k in param_keys:
            torch.testing.assert_close(sd[r][k], ref[k], rtol=1e-4, atol=1e-6, msg=lambda m, k=k: "%s: %s" % (k, m))
        for k in ref:
            if "running" in k:
                torch.testing.assert_close(sd[r][k], ref[k], rtol=1e-4, atol=1e-6)


def test_schedules_match_reference_formulas():
    # Utils/training_utils.py:9-30 with the GlobalSPFN config (bs 16, steps of 200000 samples)
    assert training.get_batch_norm_decay(0, 16, 200000) == 0.5
    assert training.get_batch_norm_decay(12500, 16, 200000) == 0.25
    assert training.get_batch_norm_decay(10 ** 7, 16, 200000) == pytest.approx(0.01)
    assert training.get_learning_rate(1e-3, 12499, 16, 200000, 0.7) == 1e-3
    assert training.get_learning_rate(1e-3, 25000, 16, 200000, 0.7) == pytest.approx(1e-3 * 0.49)
    m = torch.nn.Sequential()
    m.add_module("bn1", torch.nn.BatchNorm1d(4))
    m.add_module("fc", torch.nn.Linear(4, 4))
    training.update_momentum(m, 0.123)
    assert m.bn1.momentum == 0.123


def test_flat_bucket_views_and_finite_check():
    model = TinyHeads()
    b = training.FlatGradBucket(model)
    assert b.flat.numel() == sum(p.numel() for p in model.parameters())
    model(torch.randn(2, 8, 3))[2].sum().backward()
    b.collect()
    assert float(b.flat.abs().sum()) > 0            # the fresh gradients were packed into the flat buffer
    used = [p for p in model.parameters() if p.grad is not None]
    assert used and all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(b.params, b.views) if p.grad is not None)
    assert bool(b.finite())
    b.flat[3] = float("nan")
    assert not bool(b.finite())
    b.zero()
    assert all(p.grad is None for p in model.parameters())


def test_bench_scale_fields_shape():
    """The fields bench.py adds to a multi-GPU line (per-rank step times, stamped exchange times, collective layout) — the
    assembly is a pure function, so the first SCALE run cannot die in it."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    f = bench.scale_fields("rs_ag", 5625228, True, [1.81, 1.83, 1.9, 1.82], [41.0, 39.5, float("nan"), 44.0], 21)
    assert "NaN" not in json.dumps(f) and f["comm_us_per_step"]["all"][2] is None          # strict JSON
    assert f["ms_per_step_ranks"] == {"min": 1.81, "max": 1.9, "all": [1.81, 1.83, 1.9, 1.82]}
    assert f["comm_us_per_step"]["median_over_ranks"] == 41.0 and f["comm_us_per_step"]["max"] == 44.0
    assert "rs_ag" in f["collective"] and "inside the step's graph" in f["collective"]
    g = bench.scale_fields("all_reduce", 4, False, [2.0], [float("nan")], 1)
    assert g["comm_us_per_step"]["median_over_ranks"] is None and "after the graph" in g["collective"]
