"""cpfn_adam_flat / optim.FlatAdam against torch.optim.Adam (the reference's optimizer): same trajectory over
several steps, the skip-on-non-finite flag, a learning-rate change through the device scalar."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(dev):
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.ReLU(), torch.nn.Linear(13, 5), torch.nn.Linear(5, 3)).to(dev)


def test_flat_adam_matches_torch_adam():
    from cpfn_amd import training
    from cpfn_amd.optim import FlatAdam
    dev = torch.device("cuda:0")
    ma = _model(dev)
    mb = copy.deepcopy(ma)
    bucket = training.FlatGradBucket(ma)
    oa = FlatAdam(bucket, lr=1e-2)
    ob = torch.optim.Adam(mb.parameters(), lr=1e-2)
    assert all(p.data_ptr() >= oa.flat_p.data_ptr() for p in ma.parameters())          # parameters live in the flat buffer
    g = torch.Generator(device="cpu").manual_seed(0)
    for step in range(12):
        x = torch.randn(32, 7, generator=g).to(dev)
        if step == 6:                                  # staircase: new learning rate through the device scalar
            oa.param_groups[0]["lr"].fill_(3e-3)
            ob.param_groups[0]["lr"] = 3e-3
        for m, o in ((ma, oa), (mb, ob)):
            for p in m.parameters():
                p.grad = None
            (m(x) ** 2).sum().backward()
        bucket.collect()
        if step == 4:                                  # non-finite gradients: the whole step is skipped, count included
            oa.found_inf = torch.ones((), device=dev)
            oa.step()
            oa.found_inf = torch.zeros((), device=dev)
            assert float(oa.step_count) == 4.0
            continue
        oa.step()
        ob.step()
        for pa, pb in zip(ma.parameters(), mb.parameters()):
            torch.testing.assert_close(pa, pb, rtol=2e-6, atol=1e-7)
    assert float(oa.step_count) == 11.0
    sd = oa.state_dict()
    assert sd["flat"]["exp_avg"].numel() == bucket.flat.numel()


def test_nonfinite_flag():
    from cpfn_amd import training
    dev = torch.device("cuda:0")
    m = torch.nn.Linear(1000, 37).to(dev)            # 37037 gradients: ragged tail
    b = training.FlatGradBucket(m)
    b.flat.normal_()
    assert float(b.nonfinite_flag()) == 0.0
    for pos, val in ((0, float("nan")), (b.flat.numel() - 1, float("inf")), (12345, -float("inf"))):
        b.flat.normal_()
        b.flat[pos] = val
        assert float(b.nonfinite_flag()) == 1.0
    b.flat.normal_()
    b.flat[5] = 3.0e38                               # large but finite
    assert float(b.nonfinite_flag()) == 0.0


def test_checked_step_scans_gradients_and_counts_skips():
    """FlatAdam.step(check_gradients=True): the finite scan of the flat gradient, the skip decision and the
    skipped-step counter in the optimizer's own launches (what the graph-replayed trainer uses)."""
    from cpfn_amd import training
    from cpfn_amd.optim import FlatAdam
    dev = torch.device("cuda:0")
    m = torch.nn.Linear(1000, 37).to(dev)
    bucket = training.FlatGradBucket(m)
    opt = FlatAdam(bucket, lr=1e-2)
    skipped = torch.zeros((), device=dev)
    for pos, val, skip in ((None, 0.0, False), (0, float("nan"), True), (bucket.flat.numel() - 1, float("inf"), True),
                           (777, 3.0e38, False), (12345, -float("inf"), True)):
        bucket.flat.normal_()
        if pos is not None:
            bucket.flat[pos] = val
        before, steps, sk = opt.flat_p.clone(), float(opt.step_count), float(skipped)
        opt.step(check_gradients=True, skipped=skipped)
        assert (float(opt.step_count) == steps) == skip
        assert float(skipped) == sk + (1.0 if skip else 0.0)
        assert torch.equal(opt.flat_p, before) == skip
    assert float(skipped) == 3.0 and float(opt.step_count) == 2.0


def test_checked_packing_copy_feeds_the_optimizer():
    """FlatGradBucket.collect(check=True): the finite scan rides on the copy that packs the gradients; its flags make
    FlatAdam skip the step exactly when a gradient holds a NaN / inf (ragged sizes, unaligned slices)."""
    from cpfn_amd import training
    from cpfn_amd.optim import FlatAdam
    dev = torch.device("cuda:0")
    m = torch.nn.Sequential(torch.nn.Linear(1000, 37), torch.nn.Linear(37, 3), torch.nn.Linear(3, 129)).to(dev)
    bucket = training.FlatGradBucket(m)
    opt = FlatAdam(bucket, lr=1e-2)
    skipped = torch.zeros((), device=dev)
    params = list(m.parameters())
    for which, pos, val, skip in ((None, 0, 0.0, False), (0, 36999, float("nan"), True), (3, 2, float("inf"), True),
                                  (4, 100, 3.0e38, False), (5, 128, -float("inf"), True)):
        for p in params:
            p.grad = torch.randn_like(p)
        if which is not None:
            params[which].grad.view(-1)[pos] = val
        nf = bucket.collect(check=True)
        assert nf is not None and nf[1] > 0
        before, steps = opt.flat_p.clone(), float(opt.step_count)
        opt.step(skipped=skipped, nf_flags=nf)
        assert (float(opt.step_count) == steps) == skip
        assert torch.equal(opt.flat_p, before) == skip
        for p, v in zip(bucket.params, bucket.views):
            assert p.grad.data_ptr() == v.data_ptr()
    assert float(skipped) == 3.0


def test_gradients_written_in_place_and_the_sticky_word():
    """FlatGradBucket.sink + collect(sink=...): gradients written into the bucket's views by a checked launch are not copied,
    whatever else arrived as a fresh tensor still is (and is scanned by that copy); the writers' sticky word is flags[0], read
    and CLEARED by the optimizer's prepare kernel — a NaN stored in one step skips that step only."""
    from cpfn_amd import training, fused_mlp, lib as _l
    from cpfn_amd.optim import FlatAdam
    dev = torch.device("cuda:0")
    h = _l.lib()
    m = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.Linear(128, 35), torch.nn.Linear(35, 3)).to(dev)
    bucket = training.FlatGradBucket(m)
    opt = FlatAdam(bucket, lr=1e-2)
    skipped = torch.zeros((), device=dev)
    params = list(m.parameters())
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(4)
    for step, (poison_sink, poison_copy) in enumerate([(False, False), (True, False), (False, False), (False, True), (False, False)]):
        for p in params:
            p.grad = None
        with bucket.sink(check=True) as sink:
            assert sink is not None and sink.flag is not None
            # parameter 0 (a [128, 64] weight): written in place by a checked split reduction of 9 partial slabs
            ws = torch.randn(9, 128 * 64, generator=g).to(dev)
            if poison_sink:
                ws[4, 77] = float("nan")
            out = fused_mlp._grad_out(params[0], (128, 64), dev)
            arr = (fused_mlp._ReduceDesc * 1)(fused_mlp._ReduceDesc(ws.data_ptr(), out.data_ptr(), 128 * 64, 9, 0, 0))
            _l.check(h.cpfn_multi_split_reduce_checked(arr, 1, sink.flag.data_ptr(), st), "reduce")
            params[0].grad = out
            # the others: fresh tensors, as a framework op would leave them
            for p in params[1:]:
                p.grad = torch.randn(p.shape, generator=g).to(dev)
            if poison_copy:
                params[3].grad.view(-1)[1] = float("inf")
        nf = bucket.collect(check=True, sink=sink)
        assert nf is not None and nf[2] == 1 and nf[1] > 1
        assert params[0].grad.data_ptr() == bucket.views[[id(q) for q in bucket.params].index(id(params[0]))].data_ptr()
        torch.testing.assert_close(bucket.views[[id(q) for q in bucket.params].index(id(params[0]))].reshape(-1), ws.sum(0),
                                   rtol=1e-5, atol=1e-5, equal_nan=True)
        before, steps = opt.flat_p.clone(), float(opt.step_count)
        opt.step(skipped=skipped, nf_flags=nf)
        skip = poison_sink or poison_copy
        assert (float(opt.step_count) == steps) == skip, step
        assert torch.equal(opt.flat_p, before) == skip, step
        assert int(bucket._sink_flags[0]) == 0                       # consumed
    assert float(skipped) == 2.0
