"""CPU: the host logic of `cpfn_amd.epoch.spfn_train_val_epoch` — the reference's epoch loop
(Utils/training_utils.py:84-176) on the trainer: look-ahead over the loader, dtype casts, the ragged last batch, deferred
logging in the original order, the `total_loss_` sum, the staircases, train and val.  The network is a stand-in with
PointNet2's forward contract and the LocalSPFN loss configuration (fitter losses off: plain torch ops, no GPU needed).
In the build container (needs /root/reference) the same loader is also run through the REFERENCE's own function and the
two must agree call for call."""
import io
import os
import sys
from contextlib import redirect_stdout

import pytest
import torch

from cpfn_amd import synthetic

REF = "/root/reference"


class Tiny(torch.nn.Module):
    def __init__(self, K=21):
        super().__init__()
        self.body = torch.nn.Linear(3, 16)
        self.bn = torch.nn.BatchNorm1d(16)                       # (name contains 'bn': update_momentum reaches it)
        self.hx, self.ht, self.hw = torch.nn.Linear(16, 3), torch.nn.Linear(16, 4), torch.nn.Linear(16, K)

    def forward(self, P, glob_features=None, loc_features=None, fps_start=None, geometry=None):
        B, N, _ = P.shape
        f = torch.tanh(self.bn(self.body(P).reshape(B * N, -1))).reshape(B, N, -1)
        return [self.hx(f), self.ht(f), self.hw(f), None, None]


class Conf:
    """The getters the loop reads (Utils/config_loader.py), LocalSPFN multipliers, staircases that change inside the run."""

    def __init__(self, batch_size=2):
        self.bs = batch_size

    def get_batch_size(self): return self.bs
    def get_bn_decay_step(self): return 6
    def get_decay_step(self): return 4
    def get_decay_rate(self): return 0.7
    def get_init_learning_rate(self): return 1e-2
    def get_miou_loss_multiplier(self): return 1.0
    def get_normal_loss_multiplier(self): return 1.0
    def get_type_loss_multiplier(self): return 1.0
    def get_parameter_loss_multiplier(self): return 0.0
    def get_residue_loss_multiplier(self): return 0.0
    def get_total_loss_multiplier(self): return 1.0
    def get_list_of_primitives(self): return ['sphere', 'plane', 'cylinder', 'cone']


class Args:
    network = 'GlobalSPFN'


class Visualiser:
    def __init__(self):
        self.calls = []

    def log_loss(self, value, name):
        self.calls.append((name, float(value)))

    def update(self):
        self.calls.append(("update",))


def _loader(n_batches, B=2, ragged=True):
    """What a DataLoader over the reference's dataset yields: a tuple per batch, dtypes the loop has to cast."""
    out = []
    for i in range(n_batches):
        b = synthetic.training_batch(B if not (ragged and i == n_batches - 1) else 1, N=128, n_max_instances=21, n_prims=4,
                                     n_inst_points=8, seed=50 + i)
        out.append((b["P"].double(), b["X_gt"], b["points_per_instance"], b["I_gt"].int(), b["T_gt"], b["plane_n_gt"],
                    b["cylinder_axis_gt"].double(), b["cone_axis_gt"]))
    return out


def _restated_reference_loop(loader, model, epoch, optimizer, global_step, vis, conf, mode):
    """The sequence of Utils/training_utils.py:84-176 stated plainly on the product's own pieces (the checker of the
    look-ahead / deferred-logging version; the reference's own file is used instead where it exists, below)."""
    from cpfn_amd import training as tr
    from cpfn_amd.SPFN import losses_implementation as li
    bs = conf.get_batch_size()
    old_m = tr.get_batch_norm_decay(global_step, bs, conf.get_bn_decay_step())
    old_lr = tr.get_learning_rate(conf.get_init_learning_rate(), global_step, bs, conf.get_decay_step(), conf.get_decay_rate())
    total = 0
    model.train() if mode == 'train' else model.eval()
    for i, d in enumerate(loader):
        if i % 100 == 0:
            print('[%s][Epoch %d - Iteration %d]' % (mode, epoch, i))
        optimizer.zero_grad()
        m = tr.get_batch_norm_decay(global_step, bs, conf.get_bn_decay_step())
        if m != old_m:
            tr.update_momentum(model, m)
            old_m = m
        lr = tr.get_learning_rate(conf.get_init_learning_rate(), global_step, bs, conf.get_decay_step(), conf.get_decay_rate())
        if lr != old_lr:
            for g in optimizer.param_groups:
                g['lr'] = lr
            old_lr = lr
        P, X_gt, ppi = d[0].float(), d[1].float(), d[2].float()
        I_gt, T_gt = d[3].long(), d[4].long()
        gt = {'plane_normal': d[5].float(), 'cylinder_axis': d[6].float(), 'cone_axis': d[7].float()}
        X, T, W, _, _ = model(P)
        X = torch.nn.functional.normalize(X, p=2, dim=2, eps=1e-12)
        W = torch.softmax(W, dim=2)
        out = li.compute_all_losses(P, W, I_gt, X, X_gt, T, T_gt, gt, ppi, 1.0, 1.0, 1.0, 0.0, 0.0, 1.0, False,
                                    mode_seg='mIoU', classes=conf.get_list_of_primitives())[:6]
        total += P.shape[0] * out[0].item()
        if mode == 'train':
            out[0].backward()
            if all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None):
                optimizer.step()
            global_step += 1
        if i % 100 == 0:
            for label, v in zip(('Loss Value: ', 'Normal Loss', 'Type Loss', 'mIoU Loss', 'Residue Loss', 'Parameter Loss'), out):
                print(label, v.item())
        for v, name in zip(out, ("loss", "normal_loss", "type_loss", "miou_loss", "residue_loss", "parameter_loss")):
            vis.log_loss(v.item(), '%s_%s' % (mode, name))
        vis.update()
    return global_step, total


def _run(fn, seed=0, epochs=2):
    torch.manual_seed(seed)
    model = Tiny()
    conf = Conf()
    opt = torch.optim.Adam(model.parameters(), lr=conf.get_init_learning_rate())
    vis = Visualiser()
    gs, log, buf = 0, [], io.StringIO()
    with redirect_stdout(buf):
        for e in range(epochs):
            gs, tot = fn(_loader(5), model, e, opt, gs, vis, conf, 'train')
            log.append((gs, tot))
            with torch.no_grad():
                log.append(fn(_loader(3, ragged=False), model, e, opt, gs, vis, conf, 'val'))
    return model, opt, vis, log, buf.getvalue()


def _assert_same(a, b):
    (ma, oa, va, la, pa), (mb, ob, vb, lb, pb) = a, b
    assert [g for g, _ in la] == [g for g, _ in lb]                                  # global_step after every call
    assert la == pytest.approx(lb, rel=1e-6)
    assert [c[0] for c in va.calls] == [c[0] for c in vb.calls]                      # same visualiser calls, same order
    assert [c[1] for c in va.calls if len(c) > 1] == pytest.approx([c[1] for c in vb.calls if len(c) > 1], rel=1e-5)
    assert pa.splitlines()[0] == pb.splitlines()[0] == '[train][Epoch 0 - Iteration 0]'
    assert [l.split()[:2] for l in pa.splitlines()] == [l.split()[:2] for l in pb.splitlines()]
    for (k, x), (_, y) in zip(ma.state_dict().items(), mb.state_dict().items()):
        torch.testing.assert_close(x, y, rtol=1e-5, atol=1e-7, msg=k)
    assert ma.bn.momentum == mb.bn.momentum != 0.1                                   # the staircase reached the module
    assert oa.param_groups[0]['lr'] == pytest.approx(ob.param_groups[0]['lr']) and oa.param_groups[0]['lr'] < 1e-2


def test_epoch_loop_matches_the_reference_s_sequence():
    from cpfn_amd import training
    fast = lambda dl, m, e, o, g, v, c, mode: training.spfn_train_val_epoch(dl, m, e, o, g, v, Args(), c, 'cpu', network_mode=mode)
    a = _run(fast)
    b = _run(lambda dl, m, e, o, g, v, c, mode: _restated_reference_loop(dl, m, e, o, g, v, c, mode))
    _assert_same(a, b)
    assert a[3][0][0] == 5 and a[3][1][0] == 5 and a[3][2][0] == 10                  # val leaves global_step alone
    # one runner per (network, optimizer): the trainer and its optimizer state persist over the epochs
    assert a[0].__dict__["_cpfn_epoch_runner"].trainer.global_step == 10


def test_other_optimizers_fall_back_to_the_reference_loop(tmp_path):
    """Anything but the plain Adam of training_SPFN.py:90 is not taken over silently: the reference's own function runs."""
    (tmp_path / "Utils").mkdir()
    (tmp_path / "Utils" / "training_utils.py").write_text(
        "def spfn_train_val_epoch(*a, **k):\n    return 'the reference loop'\n"
        "def patch_selection_train_val_epoch(*a, **k):\n    return 'patch selection of the reference'\n")
    import cpfn_amd.Utils.training_utils as tu
    from cpfn_amd import training
    sys.path.insert(0, str(tmp_path))
    tu._reference_module = None
    try:
        model = Tiny()
        sgd = torch.optim.SGD(model.parameters(), lr=0.1)
        with pytest.warns(UserWarning, match="plain torch.optim.Adam"):
            assert training.spfn_train_val_epoch([], model, 0, sgd, 0, Visualiser(), Args(), Conf(), 'cpu') == 'the reference loop'
        with pytest.warns(UserWarning, match="plain torch.optim.Adam"):
            assert training.patch_selection_train_val_epoch([], model, 0, sgd, 0, Visualiser(), Args(), Conf(), 'cpu') == \
                'patch selection of the reference'
        assert tu.spfn_train_val_epoch.__module__ == tu.patch_selection_train_val_epoch.__module__ == "cpfn_amd.epoch"
    finally:
        sys.path.remove(str(tmp_path))
        tu._reference_module = None


@pytest.mark.skipif(not os.path.isdir(REF), reason="build container only: needs the reference checkout")
def test_fast_epoch_dropin_against_the_reference_s_own_function():
    """`dropin.install(fast_epoch=True)` + `from Utils import training_utils` as training_SPFN.py:14 does: the epoch function
    is ours, `patch_selection_train_val_epoch` and the schedule helpers are the reference's own — and the reference's OWN
    `spfn_train_val_epoch`, run on the same loader behind the same aliases, gives the same steps, sums, prints, visualiser
    calls and weights."""
    import cpfn_amd.dropin as d
    import cpfn_amd.Utils.training_utils as tu
    saved = {k: sys.modules.get(k) for k in d.alias_names() + ["Utils", "Utils.training_utils", "SPFN.primitives"]}
    sys.path.insert(0, REF)
    tu._reference_module = None
    try:
        d.install()                                    # (fast_epoch is the default since round 5)
        from Utils import training_utils
        assert training_utils is tu
        assert training_utils.spfn_train_val_epoch.__module__ == "cpfn_amd.epoch"
        ref = tu._load_reference_module()
        assert os.path.samefile(ref.__file__, os.path.join(REF, "Utils", "training_utils.py"))
        assert training_utils.patch_selection_train_val_epoch.__module__ == "cpfn_amd.epoch"
        assert training_utils.get_batch_norm_decay is ref.get_batch_norm_decay and training_utils.update_momentum is ref.update_momentum
        ours = _run(lambda dl, m, e, o, g, v, c, mode: training_utils.spfn_train_val_epoch(dl, m, e, o, g, v, Args(), c, 'cpu', network_mode=mode))
        theirs = _run(lambda dl, m, e, o, g, v, c, mode: ref.spfn_train_val_epoch(dl, m, e, o, g, v, Args(), c, torch.device('cpu'), network_mode=mode))
        _assert_same(ours, theirs)
    finally:
        sys.path.remove(REF)
        tu._reference_module = None
        import cpfn_amd.SPFN._reference as r
        r.reset()
        for k in [k for k in sys.modules if k == "Utils" or k.startswith("Utils.") or k == "SPFN" or k.startswith("SPFN.")]:
            del sys.modules[k]
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v


# ---- the PatchSelection loop (Utils/training_utils.py:33-82) ----------------------------------------------------------------
class TinyHeat(torch.nn.Module):
    """PointNet2(output_sizes=[2])'s forward contract: points [B,N,3] -> [heat-map logits [B,N,2], l3, features]."""

    def __init__(self):
        super().__init__()
        self.body = torch.nn.Linear(3, 16)
        self.bn = torch.nn.BatchNorm1d(16)
        self.head = torch.nn.Linear(16, 2)

    def forward(self, P, glob_features=None, loc_features=None, fps_start=None, geometry=None):
        B, N, _ = P.shape
        f = torch.tanh(self.bn(self.body(P).reshape(B * N, -1))).reshape(B, N, -1)
        return [self.head(f), None, None]


def _ps_loader(n_batches, ragged=True):
    out = []
    for i in range(n_batches):
        b = synthetic.training_batch(2 if not (ragged and i == n_batches - 1) else 1, N=128, n_max_instances=21, n_prims=4,
                                     n_inst_points=8, seed=70 + i)
        out.append((b["P"].double(), (b["I_gt"] % 2).int(), torch.arange(128)))          # (points, labels, shuffled indices)
    return out


def _restated_patch_selection_loop(loader, model, epoch, optimizer, global_step, vis, conf, mode):
    """The sequence of Utils/training_utils.py:33-82 stated plainly (incl. its quirk: training mode in every mode, :45-50)."""
    from cpfn_amd import training as tr
    bs = conf.get_batch_size()
    old_m = tr.get_batch_norm_decay(global_step, bs, conf.get_bn_decay_step())
    old_lr = tr.get_learning_rate(conf.get_init_learning_rate(), global_step, bs, conf.get_decay_step(), conf.get_decay_rate())
    total = 0
    model.train()
    for i, d in enumerate(loader):
        optimizer.zero_grad()
        m = tr.get_batch_norm_decay(global_step, bs, conf.get_bn_decay_step())
        if m != old_m:
            tr.update_momentum(model, m)
            old_m = m
        lr = tr.get_learning_rate(conf.get_init_learning_rate(), global_step, bs, conf.get_decay_step(), conf.get_decay_rate())
        if lr != old_lr:
            for g in optimizer.param_groups:
                g['lr'] = lr
            old_lr = lr
        pts, lab = d[0].float(), d[1].long()
        B, N, _ = pts.shape
        loss = torch.nn.functional.cross_entropy(model(pts)[0].contiguous().view(B * N, 2), lab.view(B * N))
        total += B * loss.item()
        if i % 100 == 0:
            print('[%s][Epoch %d - Iteration %d] Loss: %f' % (mode, epoch, i, loss.item()))
        if mode == 'train':
            loss.backward()
            optimizer.step()
            global_step += 1
        vis.log_loss(loss.item(), '%s_loss' % mode)
        vis.update()
    return global_step, total


def _run_ps(fn, epochs=2):
    torch.manual_seed(3)
    model = TinyHeat()
    conf = Conf()
    opt = torch.optim.Adam(model.parameters(), lr=conf.get_init_learning_rate())
    vis = Visualiser()
    gs, log, buf = 0, [], io.StringIO()
    with redirect_stdout(buf):
        for e in range(epochs):
            gs, tot = fn(_ps_loader(5), model, e, opt, gs, vis, conf, 'train')
            log.append((gs, tot))
            with torch.no_grad():
                log.append(fn(_ps_loader(3, ragged=False), model, e, opt, gs, vis, conf, 'val'))
    return model, opt, vis, log, buf.getvalue()


def _assert_same_ps(a, b):
    (ma, oa, va, la, pa), (mb, ob, vb, lb, pb) = a, b
    assert [g for g, _ in la] == [g for g, _ in lb] and la == pytest.approx(lb, rel=1e-6)
    assert [c[0] for c in va.calls] == [c[0] for c in vb.calls]
    assert [c[1] for c in va.calls if len(c) > 1] == pytest.approx([c[1] for c in vb.calls if len(c) > 1], rel=1e-5)
    la_, lb_ = pa.splitlines(), pb.splitlines()
    assert len(la_) == len(lb_) == 4 and [l.rsplit(' ', 1)[0] for l in la_] == [l.rsplit(' ', 1)[0] for l in lb_]
    assert [float(l.rsplit(' ', 1)[1]) for l in la_] == pytest.approx([float(l.rsplit(' ', 1)[1]) for l in lb_], rel=1e-4)
    for (k, x), (_, y) in zip(ma.state_dict().items(), mb.state_dict().items()):
        torch.testing.assert_close(x, y, rtol=1e-5, atol=1e-7, msg=k)        # incl. the running statistics the val pass moved
    assert ma.training and mb.training                                        # (:50: training mode even after a 'val' call)


def test_patch_selection_epoch_loop_matches_the_reference_s_sequence():
    from cpfn_amd import training
    fast = lambda dl, m, e, o, g, v, c, mode: training.patch_selection_train_val_epoch(dl, m, e, o, g, v, Args(), c, 'cpu', network_mode=mode)
    a = _run_ps(fast)
    b = _run_ps(_restated_patch_selection_loop)
    _assert_same_ps(a, b)
    assert a[3][0][0] == 5 and a[3][1][0] == 5 and a[3][2][0] == 10


@pytest.mark.skipif(not os.path.isdir(REF), reason="build container only: needs the reference checkout")
def test_patch_selection_epoch_against_the_reference_s_own_function():
    import cpfn_amd.Utils.training_utils as tu
    from cpfn_amd import training
    sys.path.insert(0, REF)
    tu._reference_module = None
    try:
        ref = tu._load_reference_module()
        ours = _run_ps(lambda dl, m, e, o, g, v, c, mode: training.patch_selection_train_val_epoch(dl, m, e, o, g, v, Args(), c, 'cpu', network_mode=mode))
        theirs = _run_ps(lambda dl, m, e, o, g, v, c, mode: ref.patch_selection_train_val_epoch(dl, m, e, o, g, v, Args(), c, torch.device('cpu'), network_mode=mode))
        _assert_same_ps(ours, theirs)
    finally:
        sys.path.remove(REF)
        tu._reference_module = None


def test_epoch_loop_edge_cases():
    """An empty loader, a one-batch loader and a validation call before any training step."""
    from cpfn_amd import training
    torch.manual_seed(0)
    model, conf, vis = Tiny(), Conf(), Visualiser()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    buf = io.StringIO()
    with redirect_stdout(buf):
        assert training.spfn_train_val_epoch([], model, 0, opt, 7, vis, Args(), conf, 'cpu') == (7, 0.0)
        with torch.no_grad():
            gs, tot = training.spfn_train_val_epoch(_loader(1, ragged=False), model, 0, opt, 7, vis, Args(), conf, 'cpu', network_mode='val')
        assert gs == 7 and tot > 0 and not model.training
        gs, tot = training.spfn_train_val_epoch(iter(_loader(1, ragged=False)), model, 1, opt, 7, vis, Args(), conf, 'cpu')      # (any iterable)
        assert gs == 8 and model.training
    assert vis.calls.count(("update",)) == 2
    assert buf.getvalue().count("[train][Epoch 1 - Iteration 0]") == 1 and buf.getvalue().count("[val][Epoch 0 - Iteration 0]") == 1
