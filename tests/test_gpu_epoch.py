"""GPU: `cpfn_amd.training.spfn_train_val_epoch` (the reference's epoch loop, Utils/training_utils.py:84-176, on the replayed
step) against `SPFNTrainer.step` / `eval_losses` called by hand on the same batches already resident on the device: the
look-ahead staging (pinned and pageable host tensors, dtype casts, three rotating slots, copy stream), the announcement of
the next batch, the ragged last batch, a BatchNorm-momentum change inside an epoch (re-capture) and the deferred logging
must not change a single loss value."""
import contextlib
import io

import pytest
import torch

from cpfn_amd import synthetic

pytestmark = pytest.mark.gpu

B, N, K = 4, 2048, 28


class Conf:
    def get_batch_size(self): return B
    def get_bn_decay_step(self): return 20          # momentum 0.5 -> 0.25 at step 5: inside the first epoch
    def get_decay_step(self): return 12             # learning rate x 0.7 at step 3
    def get_decay_rate(self): return 0.7
    def get_init_learning_rate(self): return 1e-3
    def get_miou_loss_multiplier(self): return 1.0
    def get_normal_loss_multiplier(self): return 1.0
    def get_type_loss_multiplier(self): return 1.0
    def get_parameter_loss_multiplier(self): return 1.0
    def get_residue_loss_multiplier(self): return 1.0
    def get_total_loss_multiplier(self): return 1.0
    def get_list_of_primitives(self): return ['sphere', 'plane', 'cylinder', 'cone']


class Args:
    network = 'GlobalSPFN'


class Visualiser:
    def __init__(self):
        self.calls = []

    def log_loss(self, value, name):
        self.calls.append((name, value))

    def update(self):
        self.calls.append(("update",))


ORDER = ("P", "X_gt", "points_per_instance", "I_gt", "T_gt", "plane_n_gt", "cylinder_axis_gt", "cone_axis_gt")


def _host_batches(n, seed0, ragged_last):
    out = []
    for i in range(n):
        b = synthetic.training_batch(B // 2 if (ragged_last and i == n - 1) else B, N=N, n_max_instances=K, n_prims=6,
                                     n_inst_points=128, seed=seed0 + i)
        out.append(b)
    return out


def _as_loader(batches):
    """Tuples like the reference's DataLoader yields; every second batch pinned (DataLoader(pin_memory=True)), the others
    pageable, P in double / I_gt in int32 on some (the loop casts with .type(FloatTensor) / .type(LongTensor))."""
    out = []
    for i, b in enumerate(batches):
        t = [b[k].clone() for k in ORDER]
        if i % 3 == 1:
            t[0], t[3] = t[0].double(), t[3].int()
        if i % 2 == 0:
            t = [x.pin_memory() for x in t]
        out.append(tuple(t))
    return out


def _model(dev):
    from cpfn_amd import training
    from cpfn_amd.PointNet2 import pn2_network
    from cpfn_amd.SPFN import fitter_factory
    with contextlib.redirect_stdout(io.StringIO()):
        fitter_factory.register_primitives(training.GLOBAL_SPFN_CLASSES)
    torch.manual_seed(0)
    m = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, K]).to(dev)
    m.set_compute_dtype(torch.bfloat16)
    return m


def test_epoch_function_equals_trainer_steps_by_hand():
    from cpfn_amd import training
    dev = torch.device("cuda:0")
    conf = Conf()
    train_b, val_b = _host_batches(9, 300, True), _host_batches(2, 400, False)

    # ---- by hand: the trainer on device-resident batches --------------------------------------------------------------
    model = _model(dev)
    tr = training.SPFNTrainer(model, batch_size=B, init_learning_rate=conf.get_init_learning_rate(), decay_step=conf.get_decay_step(),
                              decay_rate=conf.get_decay_rate(), bn_decay_step=conf.get_bn_decay_step(), use_graphs=True)
    tb = [{k: v.to(dev) for k, v in b.items()} for b in train_b]
    vb = [{k: v.to(dev) for k, v in b.items()} for b in val_b]
    torch.manual_seed(77)
    hand = []
    with torch.cuda.stream(tr.stream(dev)):
        model.train()
        for i, b in enumerate(tb):
            ragged = b["P"].shape[0] != B
            nxt = tb[i + 1] if i + 1 < len(tb) and tb[i + 1]["P"].shape == b["P"].shape and not ragged else None
            hand.append([float(o) for o in tr.step(b, next_batch=nxt, force_eager=ragged)])
        assert tr._graph is not None
        model.eval()
        for i, b in enumerate(vb):
            hand.append([float(o) for o in tr.eval_losses(b, next_batch=vb[i + 1] if i + 1 < len(vb) else None)])
    torch.cuda.synchronize()
    w_hand = {k: v.detach().clone() for k, v in model.state_dict().items()}
    assert tr.skipped_steps == 0

    # ---- the epoch function on host batches --------------------------------------------------------------------------
    model2 = _model(dev)
    opt = torch.optim.Adam(model2.parameters(), lr=conf.get_init_learning_rate())
    vis = Visualiser()
    torch.manual_seed(77)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        gs, tot_train = training.spfn_train_val_epoch(_as_loader(train_b), model2, 0, opt, 0, vis, Args(), conf, dev, network_mode='train')
        with torch.no_grad():
            gs2, tot_val = training.spfn_train_val_epoch(_as_loader(val_b), model2, 0, opt, gs, vis, Args(), conf, dev, network_mode='val')
    torch.cuda.synchronize()
    runner = model2.__dict__["_cpfn_epoch_runner"]
    assert runner.trainer._graph is not None and runner.trainer.skipped_steps == 0
    # the validation pass was replayed too (the evaluation-mode twin of the step's graph), by hand and inside the epoch function
    assert False in runner.trainer._graph.get("val", {}) and False in tr._graph.get("val", {})
    assert gs == len(train_b) and gs2 == gs
    logged = [c for c in vis.calls if len(c) > 1]
    names = [n for n, _ in logged]
    assert names[:6] == ['train_loss', 'train_normal_loss', 'train_type_loss', 'train_miou_loss', 'train_residue_loss', 'train_parameter_loss']
    assert names[-6:] == ['val_loss', 'val_normal_loss', 'val_type_loss', 'val_miou_loss', 'val_residue_loss', 'val_parameter_loss']
    assert sum(1 for c in vis.calls if c == ("update",)) == len(train_b) + len(val_b)
    vals = [v for _, v in logged]
    epoch_rows = [vals[6 * i:6 * i + 6] for i in range(len(train_b) + len(val_b))]
    for i, (a, b) in enumerate(zip(epoch_rows, hand)):
        assert a == b, (i, a, b)                                    # bit for bit: the same kernels on the same bytes
    sizes = [b["P"].shape[0] for b in train_b]
    assert tot_train == pytest.approx(sum(s * r[0] for s, r in zip(sizes, hand[:len(train_b)])), rel=1e-12)
    assert tot_val == pytest.approx(sum(B * r[0] for r in hand[len(train_b):]), rel=1e-12)
    for k, v in model2.state_dict().items():
        assert torch.equal(v, w_hand[k]), k
    # the staircases reached the module, the flat optimizer and the caller's optimizer
    assert model2.bn1.momentum == 0.25 and model2.sa1.bn_blocks[0][0].momentum == 0.25
    assert opt.param_groups[0]['lr'] == pytest.approx(1e-3 * 0.7 ** 2)
    # ... whose state ARE the flat moments (what optimizer.state_dict() would save)
    p0 = next(model2.parameters())
    assert float(opt.state[p0]["step"]) == len(train_b) and opt.state[p0]["exp_avg"].abs().sum() > 0
    assert "[train][Epoch 0 - Iteration 0]" in buf.getvalue() and "Parameter Loss" in buf.getvalue()


def test_epoch_function_fp32_model_runs_eagerly():
    """An fp32 model (dropin.install() without compute_dtype) takes the same function: eager trainer steps with the staging and
    the deferred logging."""
    from cpfn_amd import training
    dev = torch.device("cuda:0")
    model = _model(dev)
    model.set_compute_dtype(torch.float32)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    vis = Visualiser()
    with contextlib.redirect_stdout(io.StringIO()):
        gs, tot = training.spfn_train_val_epoch(_as_loader(_host_batches(3, 500, False)), model, 0, opt, 0, vis, Args(), Conf(), dev)
    assert gs == 3 and tot == tot and tot > 0
    assert model.__dict__["_cpfn_epoch_runner"].trainer._graph is None


# ---- PatchSelection (Utils/training_utils.py:33-82; training_PatchSelection.py:79-86) ----------------------------------------------
class PsConf(Conf):
    def get_bn_decay_step(self): return 24          # momentum 0.5 -> 0.25 at step 6
    def get_decay_step(self): return 16             # learning rate x 0.7 at step 4


def _ps_batches(n, seed0, ragged_last):
    out = []
    for i in range(n):
        c = synthetic.primitive_cloud(B // 2 if (ragged_last and i == n - 1) else B, N, n_prims=6, seed=seed0 + i)
        out.append({"P": c["P"], "labels": (c["I_gt"] % 2).long()})
    return out


def _ps_model(dev):
    from cpfn_amd.PointNet2 import pn2_network
    torch.manual_seed(0)
    m = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[2]).to(dev)
    m.set_compute_dtype(torch.bfloat16)
    return m


def test_patch_selection_epoch_equals_trainer_steps_by_hand():
    """`patch_selection_train_val_epoch` (the reference's signature) on the replayed PatchSelectionTrainer step against the same
    trainer driven by hand on device-resident batches: every loss and every weight bit for bit, through a re-capture (momentum
    change) and the ragged last batch; the validation call runs in TRAINING mode like the reference's (:45-50) and moves the
    running statistics; the training reduces the loss."""
    from cpfn_amd import training
    dev = torch.device("cuda:0")
    conf = PsConf()
    train_b, val_b = _ps_batches(10, 600, True), _ps_batches(2, 700, False)
    model = _ps_model(dev)
    tr = training.PatchSelectionTrainer(model, batch_size=B, init_learning_rate=conf.get_init_learning_rate(),
                                        decay_step=conf.get_decay_step(), decay_rate=conf.get_decay_rate(),
                                        bn_decay_step=conf.get_bn_decay_step(), use_graphs=True)
    tb = [{k: v.to(dev) for k, v in b.items()} for b in train_b]
    vb = [{k: v.to(dev) for k, v in b.items()} for b in val_b]
    torch.manual_seed(78)
    hand = []
    with torch.cuda.stream(tr.stream(dev)):
        model.train()
        for i, b in enumerate(tb):
            ragged = b["P"].shape[0] != B
            nxt = tb[i + 1] if i + 1 < len(tb) and tb[i + 1]["P"].shape == b["P"].shape and not ragged else None
            hand.append(float(tr.step(b, next_batch=nxt, force_eager=ragged)[0]))
        assert tr._graph is not None and tr._graph["single"]
        rm_before = model.bn1.running_mean.clone()
        for i, b in enumerate(vb):
            hand.append(float(tr.eval_losses(b, next_batch=vb[i + 1] if i + 1 < len(vb) else None)[0]))
        assert not torch.equal(model.bn1.running_mean, rm_before)            # the "validation" pass ran on batch statistics
    torch.cuda.synchronize()
    w_hand = {k: v.detach().clone() for k, v in model.state_dict().items()}
    assert tr.skipped_steps == 0 and all(h == h and 0.0 < h < 2.0 for h in hand), hand
    # ... and the objective is learnable through this step: 60 replayed steps on one batch whose labels are a function of the
    # coordinates (x > 0) bring the cross-entropy well below its start
    fixed = dict(tb[0], labels=(tb[0]["P"][..., 0] > 0).long())
    model_l = _ps_model(dev)
    tr_l = training.PatchSelectionTrainer(model_l, batch_size=B, use_graphs=True)
    with torch.cuda.stream(tr_l.stream(dev)):
        curve = [float(tr_l.step(fixed, next_batch=fixed)[0]) for _ in range(60)]
    assert tr_l._graph is not None and curve[-1] < 0.6 * curve[0], (curve[0], curve[-1])

    model2 = _ps_model(dev)
    opt = torch.optim.Adam(model2.parameters(), lr=conf.get_init_learning_rate())
    vis = Visualiser()
    torch.manual_seed(78)
    loader = lambda bs: [(b["P"].clone().pin_memory() if i % 2 else b["P"].double(), b["labels"].int() if i % 3 == 0 else b["labels"],
                          torch.arange(N)) for i, b in enumerate(bs)]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        gs, tot = training.patch_selection_train_val_epoch(loader(train_b), model2, 0, opt, 0, vis, Args(), conf, dev, network_mode='train')
        with torch.no_grad():
            gs2, tot_v = training.patch_selection_train_val_epoch(loader(val_b), model2, 0, opt, gs, vis, Args(), conf, dev, network_mode='val')
    torch.cuda.synchronize()
    assert gs == len(train_b) and gs2 == gs and model2.training
    assert True in model2.__dict__["_cpfn_epoch_runner"].trainer._graph.get("val", {})       # (training-mode twin: the quirk)
    logged = [c for c in vis.calls if len(c) > 1]
    assert [n for n, _ in logged] == ['train_loss'] * len(train_b) + ['val_loss'] * len(val_b)
    assert [v for _, v in logged] == hand
    sizes = [b["P"].shape[0] for b in train_b]
    assert tot == pytest.approx(sum(s * l for s, l in zip(sizes, hand)), rel=1e-12)
    assert tot_v == pytest.approx(sum(B * l for l in hand[len(train_b):]), rel=1e-12)
    for k, v in model2.state_dict().items():
        assert torch.equal(v, w_hand[k]), k
    lines = buf.getvalue().splitlines()
    assert lines[0].startswith("[train][Epoch 0 - Iteration 0] Loss: ") and lines[-1].startswith("[val][Epoch 0 - Iteration 0] Loss: ")
    assert model2.bn1.momentum == 0.25 and opt.param_groups[0]['lr'] == pytest.approx(1e-3 * 0.7 ** 2)


def test_heat_cross_entropy_matches_torch_and_its_hint_is_bit_identical(monkeypatch):
    """cpfn_ce2 (loss + gradient + the heads' padded gradient rows / column sums in one launch) against F.cross_entropy
    (Utils/training_utils.py:66-68), and a whole PatchSelection step with / without the hand-over to the heads' backward."""
    from cpfn_amd import lib as _l, training
    from cpfn_amd.SPFN import fused_losses as fl
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    for P_ in (256 * 512, 1000):                                   # (a ragged size: no hint, partial last block)
        Y = (torch.randn(1, P_, 2, generator=g) * 3).to(dev).requires_grad_(True)
        lab = torch.randint(0, 2, (1, P_), generator=g).to(dev)
        with fl.unit_loss_gradient():
            loss = fl.HeatCrossEntropy.apply(Y, lab, None)
        loss.backward()
        Yt = Y.detach().clone().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(Yt.view(P_, 2), lab.view(P_))
        ref.backward()
        assert abs(float(loss) - float(ref)) < 1e-6 * max(1.0, abs(float(ref)))
        assert float((Y.grad - Yt.grad).abs().max()) < 1e-6 * float(Yt.grad.abs().max()) + 1e-12
    # labels outside {0, 1} (F.cross_entropy raises / ignores them): the fused pass poisons loss and gradient instead of
    # counting the row as class 1
    for bad in (2, -100):
        Y = (torch.randn(1, 1000, 2, generator=g) * 3).to(dev).requires_grad_(True)
        lab = torch.randint(0, 2, (1, 1000), generator=g).to(dev)
        lab[0, 123] = bad
        with fl.unit_loss_gradient():
            loss = fl.HeatCrossEntropy.apply(Y, lab, None)
        loss.backward()
        assert torch.isnan(loss) and torch.isnan(Y.grad[0, 123]).all() and torch.isfinite(Y.grad[0, :123]).all()
    c = synthetic.primitive_cloud(B, N, n_prims=6, seed=9)
    batch = {"P": c["P"].to(dev), "labels": (c["I_gt"] % 2).long().to(dev)}
    starts = (torch.arange(B), torch.arange(B) + 3)
    res = {}
    for hint in (True, False):
        monkeypatch.setattr(fl, "HEADS_HINT", hint)
        model = _ps_model(dev)
        model.dropout_p = 0.0
        tr = training.PatchSelectionTrainer(model, batch_size=B)
        tr.bucket.zero()
        _l.byte_census(True)
        out = tr.losses(batch, fps_start=starts)
        out[0].backward()
        census = _l.byte_census(False)
        assert "cpfn_ce2" in census and ("cpfn_colsum_f32" in census) == (not hint), sorted(census)
        res[hint] = (float(out[0]), [None if p.grad is None else p.grad.clone() for p in model.parameters()])
    assert res[True][0] == res[False][0]
    for a, b in zip(res[True][1], res[False][1]):
        assert (a is None and b is None) or torch.equal(a, b)


def test_replayed_validation_pass_agrees_with_the_eager_one():
    """`eval_losses` replays the validation twin of the step's graph when the trainer has one; called from another stream (or
    before any training step) it runs eager launches.  Same weights, same batches, dropout off: the two forms differ only in the
    FPS seeds they draw (order of the CPU-generator draws), i.e. by sampling noise."""
    from cpfn_amd import training
    dev = torch.device("cuda:0")
    model = _model(dev)
    model.dropout_p = 0.0
    tr = training.SPFNTrainer(model, batch_size=B, init_learning_rate=0.0, use_graphs=True)
    batches = [{k: v.to(dev) for k, v in b.items()} for b in _host_batches(4, 800, False)]
    with torch.cuda.stream(tr.stream(dev)):
        for i in range(4):
            tr.step(batches[i], next_batch=batches[(i + 1) % 4])
        model.eval()
        replayed = [[float(o) for o in tr.eval_losses(batches[i], next_batch=batches[i + 1] if i < 3 else None)] for i in range(4)]
        assert False in tr._graph["val"]
    torch.cuda.synchronize()
    eager = [[float(o) for o in tr.eval_losses(batches[i])] for i in range(4)]          # (default stream: eager launches)
    for a, b in zip(replayed, eager):
        for x, y in zip(a, b):
            assert abs(x - y) <= 0.08 * abs(y) + 2e-3, (a, b)
    assert tr.global_step == 4


class _CloudSet(torch.utils.data.Dataset):
    """A stand-in for Dataset_GlobalSPFN: one cloud per item, the eight arrays in the order the loop reads them (numpy, like
    the reference's h5 reader)."""

    def __init__(self, n):
        b = synthetic.training_batch(n, N=N, n_max_instances=K, n_prims=6, n_inst_points=128, seed=900)
        self.items = [tuple(b[k][i].numpy() for k in ORDER) for i in range(n)]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def test_epoch_function_over_a_real_dataloader():
    """`torch.utils.data.DataLoader(..., num_workers=2, pin_memory=True)` as training_SPFN.py:78 builds it: the default collate
    hands the loop a LIST of pinned tensors per batch (copied from where they lie), the last batch is ragged (drop_last=False)."""
    from cpfn_amd import training
    dev = torch.device("cuda:0")
    loader = torch.utils.data.DataLoader(_CloudSet(4 * 7 + 2), batch_size=B, num_workers=2, pin_memory=True, shuffle=False)
    model = _model(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    vis = Visualiser()
    class Flat(Conf):                                 # (no staircase change in these 16 steps: the graph stays)
        def get_bn_decay_step(self): return 10 ** 6
        def get_decay_step(self): return 10 ** 6
    with contextlib.redirect_stdout(io.StringIO()):
        gs, tot = training.spfn_train_val_epoch(loader, model, 0, opt, 0, vis, Args(), Flat(), dev)
        gs, tot2 = training.spfn_train_val_epoch(loader, model, 1, opt, gs, vis, Args(), Flat(), dev)
    runner = model.__dict__["_cpfn_epoch_runner"]
    assert gs == 16 and runner.trainer._graph is not None and runner.trainer.skipped_steps == 0
    assert tot == tot and tot2 == tot2 and tot2 < tot                       # finite, and the second pass over the data is better
    assert sum(1 for c in vis.calls if c == ("update",)) == 16


def test_local_spfn_with_feature_inputs_through_the_epoch_function():
    """`args.network == 'LocalSPFN'` with a network built with use_glob_features / use_loc_features (pn2_network.py:22-27, 51-54):
    data[8] / data[9] (Utils/training_utils.py:136-137) are staged with the batch, enter the captured step as static inputs and
    change the result; a network built WITHOUT them ignores them like the reference's (training_SPFN.py:69-71)."""
    from cpfn_amd import training
    from cpfn_amd.PointNet2 import pn2_network
    dev = torch.device("cuda:0")

    class LocalArgs:
        network = 'LocalSPFN'

    class LocalConf(Conf):
        def get_bn_decay_step(self): return 10 ** 6
        def get_decay_step(self): return 10 ** 6
        def get_parameter_loss_multiplier(self): return 0.0      # Configs/config_localSPFN.yml:10-11
        def get_residue_loss_multiplier(self): return 0.0

    g = torch.Generator().manual_seed(1)
    host = _host_batches(5, 950, False)
    feats = [(torch.randn(B, 1024, generator=g), torch.randn(B, 128, generator=g)) for _ in host]

    def loader(scale):
        return [tuple(b[k] for k in ORDER) + (f[0] * scale, f[1] * scale) for b, f in zip(host, feats)]

    def run(use, scale):
        torch.manual_seed(0)
        m = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, K], use_glob_features=use, use_loc_features=use).to(dev)
        m.set_compute_dtype(torch.bfloat16)
        m.dropout_p = 0.0
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        vis = Visualiser()
        torch.manual_seed(5)
        with contextlib.redirect_stdout(io.StringIO()):
            gs, tot = training.spfn_train_val_epoch(loader(scale), m, 0, opt, 0, vis, LocalArgs(), LocalConf(), dev)
        tr = m.__dict__["_cpfn_epoch_runner"].trainer
        assert gs == 5 and tr._graph is not None and tr.skipped_steps == 0
        assert ("glob_features" in tr._graph["batch"]) == use
        return tot, [c[1] for c in vis.calls if c[0] == "train_loss"]

    t1, l1 = run(True, 1.0)
    t2, l2 = run(True, 3.0)
    assert l1 != l2 and all(v == v for v in l1 + l2)               # the feature inputs reach the network inside the replayed graph
    t3, l3 = run(False, 1.0)
    t4, l4 = run(False, 3.0)
    assert l3 == l4                                                 # ... and are ignored by a network built without them
    assert t1 == t1 and t2 == t2 and t3 == t4
