"""GPU numerics of the fused MFMA MLP stacks (bf16 operands, fp32 accumulation) against a
plain PyTorch fp32 reference of the same op (conv1x1 -> batch_norm(batch stats) -> relu
[-> max over neighbours]), forward and backward.  Tolerances are bf16-level: activations and
their gradients are stored as bf16 between layers."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _stack(cin, widths, seed, conv1d=True):
    torch.manual_seed(seed)
    convs, bns = nn.ModuleList(), nn.ModuleList()
    c = cin
    for w in widths:
        convs.append(nn.Conv1d(c, w, 1) if conv1d else nn.Conv2d(c, w, 1))
        bn = nn.BatchNorm1d(w) if conv1d else nn.BatchNorm2d(w)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.weight[::7] *= -1            # negative gammas: the pooling must still take max of z, not of y
            bn.bias.uniform_(-0.3, 0.3)
        bns.append(bn)
        c = w
    return convs.to(dev()), bns.to(dev())


def _rel(a, b):
    """relative L2 error — the fused path stores activations/gradients as bf16 (8-bit mantissa)
    between layers and an arg-max can flip on a near-tie, so element-wise max error is noisy."""
    a, b = a.float(), b.float()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def _emulated_stack(x, convs, bns, pool_k, xyz_rows, pool_arg=None):
    """The same op in plain PyTorch fp32 with the fused path's storage roundings made explicit
    (bf16 operands, bf16-stored pre-activations / activations, fp32 statistics), so ReLU masks and
    pooling arg-maxes are decided on the same values — otherwise ~0.3 % of the masks flip and the
    L2 error of a gradient is dominated by those flips (sqrt(0.003) ~ 5 %), not by the kernels."""
    import torch.nn.functional as F
    r = lambda t: t.to(torch.bfloat16).float()
    a = xyz_rows.float() if x is None else r(x)
    for i, (conv, bn) in enumerate(zip(convs, bns)):
        W = conv.weight.reshape(conv.weight.shape[0], -1)
        W = W.float() if (x is None and i == 0) else r(W)
        y32 = a @ W.t()
        mean, var = y32.mean(0), y32.var(0, unbiased=False)
        with torch.no_grad():
            n = y32.shape[0]
            bn.running_mean.mul_(1 - bn.momentum).add_(bn.momentum * (mean + conv.bias))
            bn.running_var.mul_(1 - bn.momentum).add_(bn.momentum * var * n / (n - 1))
        y = y32 + (r(y32) - y32).detach()                      # stored as bf16, straight-through
        z = (y - mean) * torch.rsqrt(var + bn.eps) * bn.weight + bn.bias
        if pool_k and i == len(convs) - 1:
            # pool_arg [G,C]: the arg-max rows the KERNEL chose.  Two bf16 pipelines with different fp32 summation orders
            # round ~1 % of the stored pre-activations to neighbouring bf16 values, and among 64 rows the two largest are
            # often within one bf16 step: the arg-max then flips to a row with another input, which is a few per cent of a
            # gradient's L2 norm and says nothing about either pipeline.  Without pool_arg: the first row that attains the
            # maximum (the kernel's rule for exact ties).
            zz = z.reshape(-1, pool_k, z.shape[1])
            if pool_arg is None:
                pool_arg = (zz == zz.amax(dim=1, keepdim=True)).to(torch.uint8).argmax(dim=1)
            z = torch.gather(zz, 1, pool_arg.long().unsqueeze(1)).squeeze(1)
        a = F.relu(z)
        a = a + (r(a) - a).detach()
    return a


def _run(x, convs, bns, dtype, pool_k, xyz_rows, gout):
    from cpfn_amd import mlp
    for p in list(convs.parameters()) + list(bns.parameters()):
        p.grad = None
    for bn in bns:
        bn.running_mean.zero_(); bn.running_var.fill_(1.0); bn.num_batches_tracked.zero_()
    xin = None if x is None else x.clone().requires_grad_(True)
    if dtype == "emulated":
        y = _emulated_stack(xin, convs, bns, pool_k, xyz_rows)
    else:
        y = mlp.run_stack(xin, convs, bns, dtype, pool_k=pool_k, xyz_rows=xyz_rows)
    (y.float() * gout).sum().backward()
    grads = [p.grad.clone() if p.grad is not None else None for p in list(convs.parameters()) + list(bns.parameters())]
    stats = [(bn.running_mean.clone(), bn.running_var.clone()) for bn in bns]
    return y.detach().float(), None if xin is None else xin.grad.float(), grads, stats


@pytest.mark.parametrize("name,P,cin,widths,pool_k,use_xyz", [
    ("sa1-like", 2 * 40 * 16, 3, [64, 64, 128], 16, True),
    ("sa2-like", 2 * 24 * 64, 131, [128, 128, 256], 64, False),
    ("sa3-like", 3 * 128, 259, [256, 512, 1024], 128, False),
    ("sfp-like", 1000, 384, [256, 128], None, False),
    ("sfp3-like", 4096 + 77, 128, [128, 128, 128], None, False),
])
def test_fused_stack_matches_fp32_reference(name, P, cin, widths, pool_k, use_xyz):
    convs, bns = _stack(cin, widths, seed=len(name))
    g = torch.Generator().manual_seed(P)
    if use_xyz:
        xyz = (torch.rand(P, 3, generator=g) * 0.4 - 0.2).to(dev())
        x = None
    else:
        xyz = None
        x = torch.randn(P, cin, generator=g).to(dev())
    rows_out = P // pool_k if pool_k else P
    gout = torch.randn(rows_out, widths[-1], generator=g).to(dev())
    y_32, gx_32, gr_32, _ = _run(x, convs, bns, torch.float32, pool_k, xyz, gout)
    y_ref, gx_ref, gr_ref, st_ref = _run(x, convs, bns, "emulated", pool_k, xyz, gout)
    y, gx, gr, st = _run(x, convs, bns, torch.bfloat16, pool_k, xyz, gout)
    assert y.shape == y_ref.shape
    assert all(int(bn.num_batches_tracked) == 1 for bn in bns)     # advanced by each layer's finalize launch
    # (1) against true fp32: bf16-level agreement of the forward, gradients same direction
    assert _rel(y, y_32) < 3e-2, ("out vs fp32", _rel(y, y_32))
    # (2) against the rounding-emulated reference: the kernels themselves
    assert _rel(y, y_ref) < 1e-2, ("out", _rel(y, y_ref))
    if gx_ref is not None:
        assert _rel(gx[:, :cin], gx_ref) < 3e-2, ("gx", _rel(gx[:, :cin], gx_ref))
        assert _rel(gx[:, :cin], gx_32) < 0.35
    names = [n for n, _ in list(convs.named_parameters()) + list(bns.named_parameters())]
    gmax = max(float(t.abs().max()) for t in gr_ref if t is not None)
    for pos, (n, a, b) in enumerate(zip(names, gr, gr_ref)):
        is_conv_bias = n.endswith("bias") and pos < 2 * len(widths)
        if is_conv_bias:
            # exactly zero under training-mode batch-norm: not produced by the fused path
            assert a is None and (b is None or float(b.abs().max()) < 1e-3 * gmax), n
            continue
        assert _rel(a, b) < 3e-2, (n, _rel(a, b))
    for (rm, rv), (rm_r, rv_r) in zip(st, st_ref):
        assert _rel(rm, rm_r) < 1e-2 and _rel(rv, rv_r) < 1e-2


@pytest.mark.parametrize("name,P,cin,widths,pool_k,use_xyz", [
    ("sa1-like", 2 * 40 * 16, 3, [64, 64, 128], 16, True),
    ("sa3-like", 3 * 128, 259, [256, 512, 1024], 128, False),
    ("sfp3-like", 40000 + 77, 128, [128, 128, 128], None, False),
    ("k192-k256", 33000, 192, [256, 128], None, False),
])
def test_bn_apply_on_operand_load_is_bit_identical(name, P, cin, widths, pool_k, use_xyz, monkeypatch):
    """BN + ReLU of a hidden layer applied on the operand load of the next GEMM and of its weight gradient
    (default) against the variant that materialises the activated tensor: same bits, forward and backward."""
    from cpfn_amd import fused_mlp
    # (the seams of round 6 replace the ordered fp32 sums of a hidden layer's statistics by fixed-point ones — equal to ~1e-7, not
    #  bit for bit — and exist only WITH the operand-load apply: this test holds the two apply routes to each other on the ordered sums)
    monkeypatch.setattr(fused_mlp, "ATOMIC_SEAMS", False)
    convs, bns = _stack(cin, widths, seed=7)
    g = torch.Generator().manual_seed(P)
    xyz = (torch.rand(P, 3, generator=g) * 0.4 - 0.2).to(dev()) if use_xyz else None
    x = None if use_xyz else torch.randn(P, cin, generator=g).to(dev())
    gout = torch.randn(P // pool_k if pool_k else P, widths[-1], generator=g).to(dev())
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(fused_mlp, "BN_APPLY_FUSED", fused)
        res[fused] = _run(x, convs, bns, torch.bfloat16, pool_k, xyz, gout)
    (ya, gxa, gra, sta), (yb, gxb, grb, stb) = res[True], res[False]
    # bit-identical whenever both variants run the same GEMM kernel; with K = 256 the fused variant runs the generic
    # kernel and the other one the whole-K stream kernel: the BN statistics are then summed in a different order
    same = (lambda a, b: torch.equal(a, b)) if name != "k192-k256" else (lambda a, b: _rel(a, b) < 2e-3)
    assert same(ya, yb)
    assert (gxa is None and gxb is None) or same(gxa, gxb)
    for a, b in zip(gra, grb):
        assert (a is None and b is None) or same(a, b)
    for (rm, rv), (rm_r, rv_r) in zip(sta, stb):
        assert same(rm, rm_r) and same(rv, rv_r)


@pytest.mark.parametrize("name,P,cin,widths,pool_k,use_xyz", [
    ("sa1-like", 2 * 40 * 16, 3, [64, 64, 128], 16, True),
    ("sa1-size", 40 * 512 * 64, 3, [64, 64, 128], 64, True),          # 1.3 M rows: the streaming kernels, 8 replicas
    ("sa2-like", 2 * 24 * 64, 131, [128, 128, 256], 64, False),
    ("sa3-like", 3 * 128, 259, [256, 512, 1024], 128, False),
    ("sfp-like", 8192, 384, [256, 128], None, False),
    ("sfp3-like", 40000 + 77, 128, [128, 128, 128, 128], None, False),
    ("sfp3-size", 131072, 128, [128, 128, 128, 128], None, False),
])
def test_atomic_seams_against_the_finalize_launch(name, P, cin, widths, pool_k, use_xyz, monkeypatch):
    """a hidden layer's BatchNorm statistics as fixed-point atomics folded by the next layer's GEMM (csrc/seam.h)
    against the partial rows + cpfn_bn_finalize they replace.  (i) the statistics themselves — scale, shift, mean, rstd as the
    consumer's first workgroup leaves them, running mean / variance — within 1e-6 relative of the ordered sums' (the fixed point
    resolves 2.4e-7 per partial sum); (ii) outputs and every gradient within a bf16 rounding flip of the other route's (rel-L2
    < 2e-3: a 1e-7 change of a scale moves a handful of bf16 roundings); (iii) TWO seam runs bit-identical — integer sums;
    (iv) fewer cpfn_bn_finalize launches, the same step counters."""
    from cpfn_amd import fused_mlp, lib as _l
    convs, bns = _stack(cin, widths, seed=11)
    g = torch.Generator().manual_seed(P + 1)
    xyz = (torch.rand(P, 3, generator=g) * 0.4 - 0.2).to(dev()) if use_xyz else None
    x = None if use_xyz else torch.randn(P, cin, generator=g).to(dev())
    gout = torch.randn(P // pool_k if pool_k else P, widths[-1], generator=g).to(dev())
    res, launches, stats = {}, {}, {}
    for mode in ("finalize", "seam", "seam2"):
        monkeypatch.setattr(fused_mlp, "ATOMIC_SEAMS", mode != "finalize")
        grabbed = []
        orig = fused_mlp._FusedStack.forward

        def spy(ctx, *a, _orig=orig, _g=grabbed):
            out = _orig(ctx, *a)
            _g.append([t[3] for t in ctx.saved])                   # every layer's [4, C] scale | shift | mean | rstd
            return out
        monkeypatch.setattr(fused_mlp._FusedStack, "forward", staticmethod(spy))
        _l.byte_census(True)
        with fused_mlp.seam_pass(dev(), True):
            res[mode] = _run(x, convs, bns, torch.bfloat16, pool_k, xyz, gout)
        launches[mode] = _l.byte_census(False).get("cpfn_bn_finalize", (0, 0))[0]
        torch.cuda.synchronize()
        stats[mode] = [t.clone() for t in grabbed[0]]
        monkeypatch.setattr(fused_mlp._FusedStack, "forward", orig)
        assert all(int(bn.num_batches_tracked) == 1 for bn in bns), mode
    # only the stack's last layer keeps its launch — unless its pooling starts in the GEMM (large pooled stacks): then the
    # [G, C]-sized finish launch folds the last seam too
    pooled_in_gemm = bool(pool_k) and bool(_l.lib().cpfn_mlp_gemm_pool_ok(P, widths[-2], widths[-1], pool_k))
    assert launches["finalize"] == len(widths) and launches["seam"] == (0 if pooled_in_gemm else 1), launches
    rel = lambda a, b: float(((a.double() - b.double()).abs() / b.double().abs().clamp_min(1e-3)).max())
    for li, (sa, sb) in enumerate(zip(stats["seam"][:-1], stats["finalize"][:-1])):
        # (only the FIRST layer sees identical inputs on both routes; behind it the other route's own rounding flips move the sums)
        tol = 1e-6 if li == 0 else 5e-3
        d_mean = float(((sa[2].double() - sb[2].double()).abs() * sb[3].double()).max())      # in units of the channel's sigma
        d_shift = float(((sa[1].double() - sb[1].double()).abs() / (1.0 + sb[1].double().abs())).max())
        assert d_mean < tol and rel(sa[3], sb[3]) < tol and rel(sa[0], sb[0]) < tol and d_shift < tol, \
            (li, d_mean, rel(sa[3], sb[3]), rel(sa[0], sb[0]), d_shift)
    (ya, gxa, gra, sta), (yb, gxb, grb, stb), (yc, gxc, grc, stc) = res["seam"], res["finalize"], res["seam2"]
    assert _rel(ya, yb) < 2e-3, _rel(ya, yb)
    assert (gxa is None) or _rel(gxa, gxb) < 5e-3
    for a, b in zip(gra, grb):
        assert (a is None and b is None) or _rel(a, b) < 5e-3, _rel(a, b)
    assert float((sta[0][0] - stb[0][0]).abs().max()) < 1e-6 and rel(sta[0][1], stb[0][1]) < 1e-5   # running statistics, first layer
    assert torch.equal(ya, yc) and ((gxa is None) or torch.equal(gxa, gxc))
    for a, c in zip(gra, grc):
        assert (a is None and c is None) or torch.equal(a, c)
    for (rm, rv), (rm_c, rv_c) in zip(sta, stc):
        assert torch.equal(rm, rm_c) and torch.equal(rv, rv_c)


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), 3e22])
def test_atomic_seam_poison_becomes_nan_statistics(bad, monkeypatch):
    """A NaN / inf / absurdly large value in a hidden layer's output must reach the consumer as NaN statistics (the fp32 sums would
    have carried a NaN; an integer cannot): the producer raises the seam's poison word instead of adding an undefined integer, and the
    consumer then behaves exactly like the finalize route on NaN sums — same running statistics (NaN), same output bits."""
    from cpfn_amd import fused_mlp, mlp
    out = {}
    for seams in (True, False):
        monkeypatch.setattr(fused_mlp, "ATOMIC_SEAMS", seams)
        convs, bns = _stack(128, [128, 128], seed=3)
        x = torch.randn(40000, 128, generator=torch.Generator().manual_seed(5)).to(dev())
        x[17, 5] = bad
        with fused_mlp.seam_pass(dev(), True):
            y = mlp.run_stack(x, convs, bns, torch.bfloat16)
        torch.cuda.synchronize()
        out[seams] = (y.detach().float(), bns[0].running_mean.clone(), bns[0].running_var.clone())
    assert bool(torch.isnan(out[True][1]).all()) and bool(torch.isnan(out[True][2]).all())
    if bad != bad:          # (a NaN input: the ordered sums are NaN too — identical behaviour down to the output bits)
        assert bool(torch.isnan(out[False][1]).all())
        assert torch.equal(out[True][0].nan_to_num(7.0), out[False][0].nan_to_num(7.0))


@pytest.mark.parametrize("name,P,cin,widths,pool_k,use_xyz", [
    ("sa1-size", 40 * 512 * 64, 3, [64, 64, 128], 64, True),
    ("sa1-ragged-tiles", 643 * 64, 3, [64, 64, 128], 64, True),          # 41152 rows: the last 128-row tile holds one group
    ("sa2-size", 16 * 128 * 64, 131, [128, 128, 256], 64, False),
    ("pool32", 1300 * 32, 128, [128, 128], 32, False),
    ("pool128", 300 * 128, 128, [64, 128], 128, False),
])
@pytest.mark.parametrize("seams", [True, False])
def test_pooling_started_in_the_gemm_epilogue(name, P, cin, widths, pool_k, use_xyz, seams, monkeypatch):
    """Round 6: the max over neighbours of a set-abstraction stack taken from the last GEMM's tile on its way out (per-wave winners
    of max(sign(gamma) * y), before the batch statistics exist) + the [G, C]-sized cpfn_bn_pool_finish, against the stand-alone
    cpfn_bn_relu_maxpool pass over the stored output: the SAME pooled output bits (max_k fma(s, y_k, t) = fma(s, max_k +-y_k, t):
    rounding is monotone; the test stacks have negative gammas), the same arg-max rows except where two different y give the same z,
    gradients equal up to those rows."""
    from cpfn_amd import fused_mlp
    monkeypatch.setattr(fused_mlp, "ATOMIC_SEAMS", seams)
    convs, bns = _stack(cin, widths, seed=5)
    g = torch.Generator().manual_seed(P + 3)
    xyz = (torch.rand(P, 3, generator=g) * 0.4 - 0.2).to(dev()) if use_xyz else None
    x = None if use_xyz else torch.randn(P, cin, generator=g).to(dev())
    gout = torch.randn(P // pool_k, widths[-1], generator=g).to(dev())
    res, args = {}, {}
    for fused in (True, False):
        monkeypatch.setattr(fused_mlp, "POOL_IN_GEMM", fused)
        grabbed = []
        orig = fused_mlp._FusedStack.forward

        def spy(ctx, *a, _orig=orig, _g=grabbed):
            out = _orig(ctx, *a)
            _g.append((ctx.saved[-1][5].clone(), ctx.saved[-1][6].clone()))      # arg, yarg of the pooled layer
            return out
        monkeypatch.setattr(fused_mlp._FusedStack, "forward", staticmethod(spy))
        with fused_mlp.seam_pass(dev(), True):
            res[fused] = _run(x, convs, bns, torch.bfloat16, pool_k, xyz, gout)
        monkeypatch.setattr(fused_mlp._FusedStack, "forward", orig)
        args[fused] = grabbed[0]
    (ya, gxa, gra, sta), (yb, gxb, grb, stb) = res[True], res[False]
    # with the seams on, the LAST layer's statistics are fixed-point on one route and ordered sums on the other (the stand-alone
    # pooling pass cannot consume a seam): ~1e-7 apart, a handful of bf16 roundings; without them: the same bits
    same_stats = not seams
    if same_stats:
        assert torch.equal(ya, yb)
    else:
        assert _rel(ya, yb) < 1e-3
    differ = float((args[True][0] != args[False][0]).float().mean())
    assert differ < (1e-4 if same_stats else 2e-3), differ
    if same_stats:
        sel = args[True][0] == args[False][0]
        assert torch.equal(args[True][1][sel], args[False][1][sel])
    for a, b in zip(gra, grb):
        assert (a is None and b is None) or _rel(a, b) < (2e-3 if same_stats else 5e-3), _rel(a, b)
    assert (gxa is None) or _rel(gxa, gxb) < 5e-3


def test_heads_linear():
    from cpfn_amd import mlp
    torch.manual_seed(0)
    heads = nn.ModuleList([nn.Conv1d(128, o, 1) for o in (3, 4, 28)]).to(dev())
    feat = torch.randn(3000, 128, device=dev())
    f1 = feat.clone().requires_grad_(True)
    outs = mlp.heads(f1, heads, torch.float32)
    gouts = [torch.randn_like(o) for o in outs]
    sum((o * g).sum() for o, g in zip(outs, gouts)).backward()
    ref = [p.grad.clone() for p in heads.parameters()]
    gx_ref = f1.grad.clone()
    for p in heads.parameters():
        p.grad = None
    f2 = feat.clone().requires_grad_(True)
    outs2 = mlp.heads(f2, heads, torch.bfloat16)
    sum((o * g).sum() for o, g in zip(outs2, gouts)).backward()
    for o, o2 in zip(outs, outs2):
        assert o2.dtype == torch.float32 and _rel(o2, o) < 2e-2
    for p, r in zip(heads.parameters(), ref):
        assert _rel(p.grad, r) < 3e-2
    assert _rel(f2.grad, gx_ref) < 3e-2


@pytest.mark.parametrize("P", [131072, 40000 + 24])
def test_heads_one_pass_takes_the_dropout_stack_s_reduction(P, monkeypatch):
    """fc1 (BatchNorm + ReLU + fused dropout) followed by the packed heads: with HEADS_RIDE the heads' one-pass backward also
    leaves pass 1 of fc1's BatchNorm backward — mask recomputed from the seed on the data-gradient slab — and
    cpfn_bn_relu_bwd is not launched: the same data-gradient bits, the same sums in another order."""
    from cpfn_amd import fused_mlp, lib as _l
    C = 128
    convs, bns = _stack(C, [128], seed=5)
    torch.manual_seed(11)
    heads_w = [torch.nn.Parameter(torch.randn(n, C, 1, device=dev()) * 0.1) for n in (3, 4, 28)]
    heads_b = [torch.nn.Parameter(torch.randn(n, device=dev()) * 0.1) for n in (3, 4, 28)]
    g = torch.Generator().manual_seed(P)
    x = torch.randn(P, C, generator=g).to(dev()).to(torch.bfloat16)
    gout = torch.randn(P, 35, generator=g).to(dev())
    res = {}
    for ride in (True, False):
        monkeypatch.setattr(fused_mlp, "HEADS_RIDE", ride)
        params = [q for q in list(convs.parameters()) + list(bns.parameters()) + heads_w + heads_b]
        for q in params:
            q.grad = None
        counter = torch.zeros(1, dtype=torch.int64, device=dev())      # same counter, same seed: the same mask both times
        xin = x.clone().requires_grad_(True)
        _l.byte_census(True)
        ho = fused_mlp.HandOver()
        feats = fused_mlp.fused_mlp_stack(xin, convs, bns, dropout=(0.5, counter, 77), handover=ho)
        outs = fused_mlp.linear_heads(feats, heads_w, heads_b, handover=ho)
        (torch.cat(outs, 1) * gout).sum().backward()
        census = _l.byte_census(False)
        assert ("cpfn_bn_relu_bwd" in census) == (not ride), sorted(census)
        assert ho.top_offer is None and ho.top_result is None and ho.heads_hint is None
        res[ride] = (feats.detach().clone(), xin.grad.float().clone(), [q.grad.clone() for q in params if q.grad is not None])
    (fa, gxa, gpa), (fb, gxb, gpb) = res[True], res[False]
    assert torch.equal(fa, fb)
    assert _rel(gxa, gxb) < 2e-3, _rel(gxa, gxb)
    assert len(gpa) == len(gpb)
    for a, b in zip(gpa, gpb):
        assert _rel(a, b) < 2e-3, _rel(a, b)


@pytest.mark.parametrize("P", [131072, 40000 + 8])
def test_heads_backward_in_one_pass(P, monkeypatch):
    """The packed heads' backward through the one-pass kernel's 64 <- 128 shape (weight gradient partials and data gradient
    from ONE pass over the gradient rows) against cpfn_mlp_wgrad + the transposed GEMM: the same bits."""
    from cpfn_amd import fused_mlp, lib as _l
    torch.manual_seed(2)
    ws = [torch.nn.Parameter(torch.randn(n, 128, 1, device=dev()) * 0.1) for n in (3, 4, 28)]
    bs = [torch.nn.Parameter(torch.randn(n, device=dev()) * 0.1) for n in (3, 4, 28)]
    g = torch.Generator().manual_seed(P)
    feat = torch.randn(P, 128, generator=g).to(dev()).to(torch.bfloat16)
    gout = torch.randn(P, 35, generator=g).to(dev())
    res = {}
    for one in (True, False):
        monkeypatch.setattr(fused_mlp, "HEADS_ONE_PASS", one)
        for q in ws + bs:
            q.grad = None
        f = feat.clone().requires_grad_(True)
        _l.byte_census(True)
        outs = fused_mlp.linear_heads(f, ws, bs)
        (torch.cat(outs, 1) * gout).sum().backward()
        census = _l.byte_census(False)
        assert ("cpfn_mlp_bwd_fused" in census) == one and ("cpfn_mlp_wgrad" in census) == (not one), sorted(census)
        torch.cuda.synchronize()
        res[one] = (f.grad.clone(), [q.grad.clone() for q in ws + bs])
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("P,K,N", [
    (1, 32, 64), (31, 96, 64), (33, 320, 320), (2048, 1280, 256), (2048, 512, 1024), (4096, 64, 128), (4097, 256, 128),
    (8192, 384, 256), (16384, 128, 64), (16385, 128, 128), (5000, 1056, 192), (40000, 320, 128), (33000, 256, 256),
])
@pytest.mark.parametrize("w_trans", [False, True])
@pytest.mark.parametrize("mode", ["plain", "stats", "stats+atr", "gather"])
def test_gemm_every_dispatch_path(P, K, N, w_trans, mode):
    """cpfn_mlp_gemm through all of its kernels (small-P split-K, whole-K stream, generic chunked) against an fp32
    matmul of the same bf16 operands: output within one bf16 rounding, BatchNorm partial sums within fp32 noise."""
    from cpfn_amd import fused_mlp
    g = torch.Generator().manual_seed(P * 7 + K + N)
    rows = P + 50 if mode == "gather" else P
    A = torch.randn(rows, K, generator=g).to(dev()).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev()).to(torch.bfloat16)
    kw = {}
    a_eff = A.float()
    if mode == "gather":
        idx = torch.randint(0, rows, (P,), generator=g).to(dev()).int()
        kw["gidx"] = idx
        a_eff = a_eff[idx.long()]
    if mode == "stats+atr":
        sc = (torch.rand(K, generator=g) + 0.5).to(dev())
        sh = (torch.rand(K, generator=g) - 0.5).to(dev())
        kw.update(a_scale=sc, a_shift=sh)
        a_eff = torch.relu(a_eff * sc + sh).to(torch.bfloat16).float()
    Wk = W.t().contiguous() if w_trans else W
    Y, part, nblk = fused_mlp.gemm(A, Wk, stats=mode.startswith("stats"), w_trans=w_trans, **kw)
    ref = a_eff @ W.float().t()
    assert Y.shape == (P, N) and Y.dtype == torch.bfloat16
    err = (Y.float() - ref).abs()
    assert float((err - ref.abs() * 2.0 ** -8).max()) <= 1e-3 * float(ref.abs().max()), float(err.max())
    if mode.startswith("stats"):
        assert part.shape == (nblk, 2, N)
        s = part.double().sum(0)
        scale = float(ref.abs().max())
        # the statistics are those of the fp32 accumulators (small-P and generic kernels) or of the bf16 values that
        # were stored (stream kernel: what BatchNorm will normalise) — the test accepts either, each at fp32 noise
        yd = Y.double()
        e1 = min(float((s[0] - ref.double().sum(0)).abs().max()), float((s[0] - yd.sum(0)).abs().max()))
        e2 = min(float((s[1] - (ref.double() ** 2).sum(0)).abs().max()), float((s[1] - (yd ** 2).sum(0)).abs().max()))
        assert e1 <= 4e-6 * float(ref.abs().double().sum(0).max()) + 1e-3, e1          # fp32 summation noise
        assert e2 <= 2e-5 * float((ref.double() ** 2).sum(0).max()) + 1e-3, e2


@pytest.mark.parametrize("P", [20000, 40000])       # 40000 rows: the dropout mask is applied inside cpfn_mlp_bwd_fused
def test_fused_dropout_mask_statistics_and_backward(P):
    """Dropout fused into the last BatchNorm apply (fc1 of the network): Bernoulli(1-p) mask with 1/(1-p) scaling,
    a fresh mask per forward pass, and a backward pass that applies exactly the forward mask (the mask is
    regenerated from an 8-byte seed, never stored)."""
    from cpfn_amd import fused_mlp
    C = 128
    convs, bns = _stack(C, [128], seed=3)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(P, C, generator=g).to(dev()).to(torch.bfloat16)
    gout = torch.randn(P, 128, generator=g).to(dev())
    counter = torch.zeros(1, dtype=torch.int64, device=dev())

    def run(dropout, grad_out):
        for p in list(convs.parameters()) + list(bns.parameters()):
            p.grad = None
        xin = x.clone().requires_grad_(True)
        y = fused_mlp.fused_mlp_stack(xin, convs, bns, dropout=dropout)
        (y.float() * grad_out).sum().backward()
        return y.detach().float(), xin.grad.float(), [p.grad.clone() for p in convs.parameters() if p.grad is not None] + \
            [p.grad.clone() for p in bns.parameters()]

    for p in (0.5, 0.2):
        y0, _, _ = run(None, gout)
        y1, gx1, gp1 = run((p, counter, 1234), gout)
        y2, _, _ = run((p, counter, 1234), gout)
        assert int(counter) == 2 + (0 if p == 0.5 else 2)
        live = y0 > 0
        keep1 = (y1 != 0) & live
        n = int(live.sum())
        frac = float(keep1.sum()) / n
        assert abs(frac - (1 - p)) < 5 * (p * (1 - p) / n) ** 0.5, frac
        # kept values are the undropped ones scaled by 1/(1-p) (one bf16 rounding apart)
        ratio = (y1[keep1] / y0[keep1])
        assert float((ratio - 1 / (1 - p)).abs().max()) < 2e-2
        # a new mask every forward pass; no structure along rows or channels
        keep2 = (y2 != 0) & live
        agree = float((keep1 == keep2)[live].float().mean())
        assert abs(agree - (p * p + (1 - p) * (1 - p))) < 0.01
        per_ch = keep1.float().sum(0) / live.float().sum(0).clamp_min(1)
        assert float((per_ch - (1 - p)).abs().max()) < 0.03
        # backward: identical to the undropped stack fed with the masked, rescaled output gradient
        mask = torch.where(live, keep1.float() / (1 - p), torch.zeros_like(y0))
        # (elements with y0 == 0 have zero gradient through the ReLU either way)
        _, gx_ref, gp_ref = run(None, (gout * mask).to(torch.bfloat16).float())
        assert _rel(gx1, gx_ref) < 2e-2, _rel(gx1, gx_ref)
        for a, b in zip(gp1, gp_ref):
            assert _rel(a, b) < 2e-2


@pytest.mark.parametrize("name,P,cin,widths,pool_k", [
    ("sfp3-like", 40000 + 77, 128, [128, 128, 128], None),
    ("sa2-like", 2 * 300 * 64, 131, [128, 128, 256], 64),
    ("sa1-like", 40 * 1024, 64, [64, 64, 128], 16),
])
@pytest.mark.parametrize("bn_eval", [False, True])
def test_bn_backward_reduction_on_the_data_gradient_gemm(name, P, cin, widths, pool_k, bn_eval, monkeypatch):
    """BatchNorm-backward pass 1 of a hidden layer taken by the data-gradient GEMM that produces its gradient
    (default where the streaming kernel runs) against the separate cpfn_bn_relu_bwd pass: the same sums in a
    different order."""
    from cpfn_amd import fused_mlp
    convs, bns = _stack(cin, widths, seed=11)
    if bn_eval:                                        # running statistics: the gradient has no batch-statistics terms
        for bn in bns:
            bn.eval()
            with torch.no_grad():
                bn.running_var.uniform_(0.5, 2.0)
                bn.running_mean.normal_(0, 0.2)
    g = torch.Generator().manual_seed(P)
    x = torch.randn(P, cin, generator=g).to(dev())
    gout = torch.randn(P // pool_k if pool_k else P, widths[-1], generator=g).to(dev())
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(fused_mlp, "BWD_STATS_FUSED", fused)
        res[fused] = _run(x, convs, bns, torch.bfloat16, pool_k, None, gout)
    (ya, gxa, gra, _), (yb, gxb, grb, _) = res[True], res[False]
    assert torch.equal(ya, yb)
    assert _rel(gxa, gxb) < 2e-3, _rel(gxa, gxb)
    for a, b in zip(gra, grb):
        assert (a is None and b is None) or _rel(a, b) < 2e-3


@pytest.mark.parametrize("name,P,cin,widths,pool_k", [
    ("sfp3-like", 40000 + 77, 128, [128, 128, 128], None),
    ("sa2-like", 2 * 300 * 64, 131, [128, 128, 256], 64),
    ("fc1-like", 32768, 128, [128], None),
    ("sa1-like", 40 * 1024 + 16, 64, [64, 64, 128], 16),       # the 64 -> 64 and (pooled) 64 -> 128 shapes, ragged last split
    ("sa1-pool64", 643 * 64, 64, [64, 64, 128], 64),            # pooled top layer: the pooled apply pass inside the kernel
    ("sa1-xyz", 643 * 64, 3, [64, 64, 128], 64),                # fp32 xyz first layer: apply pass inside its weight gradient
    ("pool32", 1100 * 32, 128, [128, 128], 32),                 # 128 -> 128 pooled (32-row steps)
])
@pytest.mark.parametrize("stats_fused", [False, True])
@pytest.mark.parametrize("variant", ["one-pass+apply", "one-pass", "apply-only"])
def test_one_pass_weight_and_data_gradient(name, P, cin, widths, pool_k, stats_fused, variant, monkeypatch):
    """cpfn_mlp_bwd_fused (weight gradient + data gradient [+ BatchNorm-backward pass 1 of the layer below] [+ the
    BatchNorm-backward apply pass, dense or max-pooled] of a dense layer from one read of its gradient) and
    cpfn_smallk_wgrad_apply against the separate kernels: the same arithmetic on the same operands, so bit-identical;
    with the statistics riding along only their summation order differs."""
    from cpfn_amd import fused_mlp, lib as _l
    assert _l.lib().cpfn_mlp_bwd_fused_ok(P, 128, 128) == 1 and _l.lib().cpfn_mlp_bwd_fused_ok(P, 64, 64) == 1
    use_xyz = cin == 3
    convs, bns = _stack(cin, widths, seed=13)
    g = torch.Generator().manual_seed(P)
    xyz = (torch.rand(P, 3, generator=g) * 0.4 - 0.2).to(dev()) if use_xyz else None
    x = None if use_xyz else torch.randn(P, cin, generator=g).to(dev())
    gout = torch.randn(P // pool_k if pool_k else P, widths[-1], generator=g).to(dev())
    monkeypatch.setattr(fused_mlp, "BWD_STATS_FUSED", stats_fused)
    # (round 6's riding form of the xyz first layer's weight gradient is not the stand-alone launch's arithmetic — no bf16 rounding
    #  of g_y — and has its own test, test_xyz_weight_gradient_rides_on_the_layer_above: here the routes that ARE bit-identical)
    monkeypatch.setattr(fused_mlp, "XYZ_WGRAD_RIDE", False)
    res = {}
    for cfg in (variant, "separate"):
        one_pass, apply_fused = cfg in ("one-pass+apply", "one-pass"), cfg in ("one-pass+apply", "apply-only")
        monkeypatch.setattr(fused_mlp, "FUSED_BWD", one_pass)
        monkeypatch.setattr(fused_mlp, "FUSED_BWD_APPLY", apply_fused)
        _l.byte_census(True)
        res[cfg] = _run(x, convs, bns, torch.bfloat16, pool_k, xyz, gout)
        census = _l.byte_census(False)
        assert ("cpfn_mlp_bwd_fused" in census) == one_pass, sorted(census)
        assert (("cpfn_smallk_wgrad_apply" in census) or ("cpfn_smallk_wgrad_apply_xyz" in census)) == (apply_fused and use_xyz), sorted(census)
        if cfg == "one-pass+apply":
            n_apply = census.get("cpfn_bn_bwd_apply", (0, 0))[0]
            if name != "fc1-like":     # every eligible layer lost its stand-alone apply launch
                assert n_apply < len(widths) - (1 if pool_k else 0), (n_apply, sorted(census))
            if name in ("sa1-pool64", "sa1-xyz", "pool32"):
                assert "cpfn_bn_pool_bwd_apply" not in census and "cpfn_bn_bwd_apply" not in census, sorted(census)
            if name == "sa2-like":       # <128,128,32> and the 128 -> 256 pooled top layer <256,128,32>; the first layer on a
                # CONCATENATED operand (131 -> padded 192 columns) keeps the generic pair since round 3 — the product hands
                # that layer its coordinates as an fp32 tail instead (test_xyz_tail_first_layer_against_the_concatenated_operand)
                assert "cpfn_bn_pool_bwd_apply" not in census and census["cpfn_bn_bwd_apply"][0] == 1, sorted(census)
                assert census["cpfn_mlp_bwd_fused"][0] == 2, census["cpfn_mlp_bwd_fused"]
    (ya, gxa, gra, _), (yb, gxb, grb, _) = res[variant], res["separate"]
    same = (lambda a, b: _rel(a, b) < 2e-3) if stats_fused else torch.equal
    assert torch.equal(ya, yb)
    assert (gxa is None and gxb is None) or same(gxa, gxb)
    for a, b in zip(gra, grb):
        assert (a is None and b is None) or same(a, b)


@pytest.mark.parametrize("name,P,cin,widths,pool_k", [
    ("sa3-like", 16 * 128, 259, [256, 512, 1024], 128),
    ("sfp1-like", 2048, 1280, [256, 256], None),
    ("sfp2-like", 8192 - 40, 384, [256, 128], None),
    ("ragged", 1000, 384, [256, 128], None),
])
@pytest.mark.parametrize("stats_fused", [False, True])
def test_small_layer_backward_fusion(name, P, cin, widths, pool_k, stats_fused, monkeypatch):
    """Small layers: cpfn_mlp_dgrad_small (the reduction of the layer below riding on the stored tile of the data gradient)
    against wgrad + GEMM + bn_relu_bwd: the same gradient bits, the same sums in a different order."""
    from cpfn_amd import fused_mlp, lib as _l
    convs, bns = _stack(cin, widths, seed=17)
    g = torch.Generator().manual_seed(P)
    x = torch.randn(P, cin, generator=g).to(dev())
    gout = torch.randn(P // pool_k if pool_k else P, widths[-1], generator=g).to(dev())
    monkeypatch.setattr(fused_mlp, "BWD_STATS_FUSED", stats_fused)
    res = {}
    for fused in ("merged", True, False):
        monkeypatch.setattr(fused_mlp, "SMALL_BWD_FUSED", bool(fused))
        monkeypatch.setattr(fused_mlp, "SMALL_BWD_MERGED", fused == "merged")
        _l.byte_census(True)
        res[fused] = _run(x, convs, bns, torch.bfloat16, pool_k, None, gout)
        census = _l.byte_census(False)
        if fused == "merged":
            # round 3: weight gradient + data gradient of EVERY layer of a small stack as one launch each (the pooled top layer
            # included: its data gradient then also carries the reduction of the layer below)
            # (round 6: sa3's 512 -> 1024 top layer too — cpfn_mlp_dgrad_small_ok bounded the contraction length by the size of the
            #  operand transform's vectors, which a data gradient does not use; it ran as three launches until then)
            n_merged = len(widths)
            assert census["cpfn_mlp_bwd_small"][0] == n_merged and "cpfn_mlp_dgrad_small" not in census, sorted(census)
            assert census.get("cpfn_mlp_wgrad", (0, 0))[0] == len(widths) - n_merged
            if stats_fused:          # only the top layer (and the layer below an unmerged one) needs its own reduction pass
                assert census["cpfn_bn_relu_bwd"][0] == 1 + (len(widths) - n_merged), census["cpfn_bn_relu_bwd"]
            continue
        assert ("cpfn_mlp_dgrad_small" in census) == (fused and stats_fused), sorted(census)
        if fused and not pool_k:
            if stats_fused:      # only the top layer still needs its own reduction pass
                assert census["cpfn_bn_relu_bwd"][0] == 1, census["cpfn_bn_relu_bwd"]
    same = (lambda a, b: _rel(a, b) < 2e-3) if stats_fused else torch.equal
    (ym, gxm, grm, _), (yf, gxf, grf, _) = res["merged"], res[True]
    # the merged launch runs the two launches' bodies: the same bits — except that a riding reduction sums its 32-row tiles
    # (the two-launch form: 64-row tiles where there are enough rows) in another order
    same_m = same
    assert torch.equal(ym, yf) and same_m(gxm, gxf)
    for a_, b_ in zip(grm, grf):
        assert (a_ is None and b_ is None) or same_m(a_, b_)
    (ya, gxa, gra, _), (yb, gxb, grb, _) = res[True], res[False]
    assert torch.equal(ya, yb)
    assert same(gxa, gxb)
    for a, b in zip(gra, grb):
        assert (a is None and b is None) or same(a, b)


@pytest.mark.parametrize("P,widths,pool_k", [(643 * 64, [64, 64, 128], 64), (40000 + 16, [64, 64], None)])
def test_first_layer_output_recomputed_in_backward(P, widths, pool_k, monkeypatch):
    """sa1: the fp32-xyz first layer's pre-BN output, recomputed from the coordinates inside cpfn_smallk_wgrad_apply_xyz
    instead of read: every gradient bit-identical, run after run."""
    from cpfn_amd import fused_mlp, lib as _l
    convs, bns = _stack(3, widths, seed=19)
    g = torch.Generator().manual_seed(P)
    xyz = (torch.rand(P, 3, generator=g) * 0.4 - 0.2).to(dev())
    gout = torch.randn(P // pool_k if pool_k else P, widths[-1], generator=g).to(dev())
    monkeypatch.setattr(fused_mlp, "XYZ_WGRAD_RIDE", False)      # (round 6's form has its own test below: not the same bits)
    res = {}
    for rec in (True, False):
        monkeypatch.setattr(fused_mlp, "XYZ_RECOMPUTE", rec)
        _l.byte_census(True)
        res[rec] = [_run(None, convs, bns, torch.bfloat16, pool_k, xyz, gout) for _ in range(3)]
        census = _l.byte_census(False)
        assert ("cpfn_smallk_wgrad_apply_xyz" in census) == rec, sorted(census)
    ref = res[False][0]
    for key, runs in res.items():
        for (ya, _, gra, _) in runs:             # every run of every variant: the same bits
            assert torch.equal(ya, ref[0]), key
            for a, b in zip(gra, ref[2]):
                assert (a is None and b is None) or torch.equal(a, b), key


@pytest.mark.parametrize("P,widths,pool_k", [(643 * 64, [64, 64, 128], 64), (40000 + 16, [64, 64], None), (16 * 512 * 64, [64, 64, 128], 64),
                                             (40 * 512 * 64, [64, 64, 128], 64)])
def test_xyz_weight_gradient_rides_on_the_layer_above(P, widths, pool_k, monkeypatch):
    """Round 6: sa1's first-layer weight gradient dW0 = sum_p g_y x^T is linear in the BatchNorm coefficients (g_y = c0 g_z + c1 y +
    c2), so its sums ride on the one-pass launch of the SECOND layer and the batched split reduction finishes c0 S1 + c1 S2 + c2 S3:
    no cpfn_smallk_wgrad_apply_xyz launch, no [P, 64] gradient tensor.  Against the stand-alone launch: every other gradient has the
    same bits; dW0 agrees to fp32 summation order + the bf16 rounding of g_y that the stand-alone kernel applies before its
    products and this form does not (rel-L2 < 1.5e-2); four runs of the riding form are bit-identical."""
    from cpfn_amd import fused_mlp, lib as _l
    convs, bns = _stack(3, widths, seed=23)
    g = torch.Generator().manual_seed(P + 9)
    xyz = (torch.rand(P, 3, generator=g) * 0.4 - 0.2).to(dev())
    gout = torch.randn(P // pool_k if pool_k else P, widths[-1], generator=g).to(dev())
    res = {}
    for ride in (True, False):
        monkeypatch.setattr(fused_mlp, "XYZ_WGRAD_RIDE", ride)
        _l.byte_census(True)
        # (four runs: built with the SLP vectoriser's packed fp32 this kernel's sums differed from run to run at 1.3 M rows —
        #  cpfn_amd/build.py, mlp_bwd_fused.hip; the build's ISA scan now rejects packed fp32 in that file)
        res[ride] = [_run(None, convs, bns, torch.bfloat16, pool_k, xyz, gout) for _ in range(4 if ride else 1)]
        census = _l.byte_census(False)
        assert ("cpfn_smallk_wgrad_apply_xyz" in census) == (not ride), sorted(census)
    (ya, _, gra, _), (yb, _, grb, _) = res[True][0], res[False][0]
    assert torch.equal(ya, yb)
    names = [n for n, _ in list(convs.named_parameters()) + list(bns.named_parameters())]
    for n, a, b in zip(names, gra, grb):
        if a is None and b is None:
            continue
        if n == "0.weight":
            # (the stand-alone launch rounds every g_y to bf16 — 2^-9 relative — before its products, as the tensor it replaced
            #  was stored; the riding form never forms g_y: measured 3.6e-3 ... 7.7e-3 between the two, both 1.7e-2 from the emulation)
            assert _rel(a, b) < 1.5e-2, ("dW0", _rel(a, b))
        else:
            assert torch.equal(a, b), n
    for other in res[True][1:]:
        for a, b in zip(res[True][0][2], other[2]):
            assert (a is None and b is None) or torch.equal(a, b)
    # ... and against the fp32-with-explicit-roundings emulation, like every other gradient
    y_ref, _, gr_ref, _ = _run(None, convs, bns, "emulated", pool_k, xyz, gout)
    assert _rel(gra[0], gr_ref[0]) < 3e-2, _rel(gra[0], gr_ref[0])


@pytest.mark.parametrize("B,n_src,S,k", [(16, 512, 128, 64), (5, 200, 128, 64), (2, 512, 256, 128)])
def test_grouped_rows_gathered_while_loading(B, n_src, S, k, monkeypatch):
    """Round 6: sa2's grouped input rows feats[b, idx[b, s, k]] are not materialised — cpfn_mlp_gemm_xyz_gather and
    cpfn_mlp_bwd_fused_xt_gather read them out of the feature table while loading their operand (autograd_ops.GroupConcat, lazy).
    Against the materialised gather (cpfn_group_concat_bf16 + the plain kernels): the same operand values in the same MFMA order —
    outputs, the gradient w.r.t. the feature table and every parameter gradient bit-identical; no cpfn_group_concat_bf16 launch."""
    from cpfn_amd import autograd_ops, fused_mlp, lib as _l, mlp, ops
    D, P = 128, B * S * k
    assert fused_mlp.xyz_tail_ok(P, D, 128)
    convs, bns = _stack(D + 3, [128, 128, 256], seed=37)
    g = torch.Generator().manual_seed(P + 5)
    feats = torch.randn(B, n_src, D, generator=g).to(dev()).to(torch.bfloat16)
    nbr = torch.randint(0, n_src, (B, S, k), generator=g).to(torch.int32).to(dev())
    rel = (torch.rand(P, 3, generator=g) * 0.8 - 0.4).to(dev())
    gout = torch.randn(P // k, 256, generator=g).to(dev())
    inv = ops.csr_build(nbr.reshape(B, S * k), n_src)
    params = [p for c in convs for p in (c.weight,)] + [p for b in bns for p in (b.weight, b.bias)]
    res = {}
    for lazy in (True, False):
        monkeypatch.setattr(fused_mlp, "GATHER_ON_LOAD", lazy)
        for p in params:
            p.grad = None
        for bn in bns:
            bn.running_mean.zero_(); bn.running_var.fill_(1.0); bn.num_batches_tracked.zero_()
        f = feats.clone().requires_grad_(True)
        ok = fused_mlp.gather_on_load_ok(B, n_src, S * k, D)
        assert ok == lazy
        _l.byte_census(True)
        x = autograd_ops.GroupConcat.apply(f, None, nbr, D, inv[0], inv[1], None, ok)
        y = mlp.run_stack(x, convs, bns, torch.bfloat16, pool_k=k, xyz_tail=rel,
                          gather=(f.detach().reshape(B * n_src, D), nbr.reshape(-1), S * k, n_src) if ok else None)
        (y.float() * gout).sum().backward()
        census = _l.byte_census(False)
        assert ("cpfn_group_concat_bf16" in census) == (not lazy), sorted(census)
        res[lazy] = (y.detach().float(), f.grad.float(), [p.grad.clone() for p in params])
    (ya, ga, pa), (yb, gb, pb) = res[True], res[False]
    assert torch.equal(ya, yb) and torch.equal(ga, gb)
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)


def test_gradient_accumulation_over_two_backward_passes():
    """The weight-gradient split reductions are deferred to the end of a backward pass, which is only sound while
    AccumulateGrad steals the returned tensor (p.grad is None).  With accumulation (p.grad already set) they must run at
    once: two identical backward passes without clearing the gradients give exactly twice the gradient, for the stack's
    weights and for the packed heads (weights and biases)."""
    from cpfn_amd import fused_mlp, mlp
    P = 40000
    convs, bns = _stack(128, [128, 128], seed=23)
    heads = nn.ModuleList([nn.Conv1d(128, o, 1) for o in (3, 4, 28)]).to(dev())
    g = torch.Generator().manual_seed(1)
    x = torch.randn(P, 128, generator=g).to(dev())
    gout = torch.randn(P, 35, generator=g).to(dev())
    params = list(convs.parameters()) + list(bns.parameters()) + list(heads.parameters())

    def once():
        feat = mlp.run_stack(x, convs, bns, torch.bfloat16)
        out = torch.cat(mlp.heads(feat, heads, torch.bfloat16), dim=1)
        (out.float() * gout).sum().backward()

    for p in params:
        p.grad = None
    once()
    torch.cuda.synchronize()
    single = [None if p.grad is None else p.grad.clone() for p in params]
    once()                                    # accumulates into the existing .grad tensors
    torch.cuda.synchronize()
    for p, s in zip(params, single):
        if s is None:
            assert p.grad is None
        else:
            assert torch.isfinite(p.grad).all()
            assert torch.equal(p.grad, 2 * s)
    assert not fused_mlp._pending_reduce


@pytest.mark.parametrize("P,pool_k", [(131072, 64), (40000 + 64 * 3, None), (32768, 64)])
def test_xyz_tail_first_layer_against_the_concatenated_operand(P, pool_k):
    """sa2's first layer at >= 32768 rows: [128 gathered bf16 channels | 3 fp32 coordinates] as the split operand of
    cpfn_mlp_gemm_xyz (the coordinate term is one more MFMA k-step built in registers; its weight-gradient columns ride on
    the one-pass backward kernel and land in the same [N, 131] gradient through the strided split reduction) against the
    round-2 form of the same stack — the coordinates as three bf16 columns of a zero-padded K = 192 operand — and against
    plain PyTorch fp32.  The split form keeps the coordinates in (nearly) fp32, so it sits CLOSER to fp32 than the
    concatenated one; run to run it is bitwise reproducible."""
    from cpfn_amd import fused_mlp, lib as _l, mlp
    assert fused_mlp.xyz_tail_ok(P, 128, 128)
    convs, bns = _stack(131, [128, 128, 256], seed=31)
    g = torch.Generator().manual_seed(P)
    feats = torch.randn(P, 128, generator=g).to(dev()).to(torch.bfloat16)
    rel = (torch.rand(P, 3, generator=g) * 0.8 - 0.4).to(dev())
    gout = torch.randn(P // pool_k if pool_k else P, 256, generator=g).to(dev())
    params = [p for c in convs for p in (c.weight,)] + [p for b in bns for p in (b.weight, b.bias)]

    def run(kind):
        for p in params:
            p.grad = None
        for bn in bns:
            bn.running_mean.zero_(); bn.running_var.fill_(1.0); bn.num_batches_tracked.zero_()
        f = feats.clone().requires_grad_(True)
        if kind == "tail":
            _l.byte_census(True)
            y = mlp.run_stack(f, convs, bns, torch.bfloat16, pool_k=pool_k, xyz_tail=rel)
        elif kind == "concat":
            y = mlp.run_stack(torch.cat([f, rel.to(torch.bfloat16)], 1), convs, bns, torch.bfloat16, pool_k=pool_k)
        else:
            y = mlp.run_stack(torch.cat([f.float(), rel], 1), convs, bns, torch.float32, pool_k=pool_k)
        (y.float() * gout).sum().backward()
        census = _l.byte_census(False) if kind == "tail" else None
        return y.detach().float(), f.grad.float(), [p.grad.clone() for p in params], census

    yt, gxt, gpt, census = run("tail")
    assert census["cpfn_mlp_bwd_fused"][0] == 3 and "cpfn_mlp_wgrad" not in census, sorted(census)
    yc, gxc, gpc, _ = run("concat")
    y32, gx32, gp32, _ = run("fp32")
    e_fwd, e_fwd_c = _rel(yt, y32), _rel(yc, y32)
    print("forward vs fp32: tail %.2e, concatenated %.2e | tail vs concatenated %.2e" % (e_fwd, e_fwd_c, _rel(yt, yc)))
    assert e_fwd < 3e-2 and e_fwd <= e_fwd_c * 1.05 and _rel(yt, yc) < 1e-2
    # (pooled: arg-max flips between two bf16 pipelines move a few per cent of a gradient's norm, see _emulated_stack)
    tol = 0.1 if pool_k else 6e-2            # (unpooled: ReLU-mask flips; the concatenated form itself sits 9e-2 from fp32 here)
    print("dX: tail vs concatenated %.2e, vs fp32 %.2e (concatenated vs fp32 %.2e)" % (_rel(gxt, gxc), _rel(gxt, gx32), _rel(gxc, gx32)))
    assert _rel(gxt, gxc) < tol and _rel(gxt, gx32) < 1.1 * _rel(gxc, gx32) + 1e-2
    w_t, w_c, w_32 = gpt[0].reshape(128, 131), gpc[0].reshape(128, 131), gp32[0].reshape(128, 131)
    print("dW feature columns: tail vs concatenated %.2e | coordinate columns %.2e (vs fp32: %.2e / %.2e)"
          % (_rel(w_t[:, :128], w_c[:, :128]), _rel(w_t[:, 128:], w_c[:, 128:]), _rel(w_t[:, 128:], w_32[:, 128:]), _rel(w_c[:, 128:], w_32[:, 128:])))
    assert _rel(w_t[:, :128], w_c[:, :128]) < tol and _rel(w_t[:, 128:], w_c[:, 128:]) < tol
    assert _rel(w_t, w_32) < 1.1 * _rel(w_c, w_32) + 1e-2
    for a, b in zip(gpt[1:], gpc[1:]):
        assert _rel(a, b) < tol
    y2, gx2, gp2, _ = run("tail")
    assert torch.equal(y2, yt) and torch.equal(gx2, gxt) and all(torch.equal(a, b) for a, b in zip(gp2, gpt))


@pytest.mark.parametrize("G,N,lda,off,ldb", [(2048, 256, 320, 3, 1280), (100, 64, 64, 0, 64), (4096, 128, 192, 5, 136)])
def test_pooled_pass1_on_the_sum_of_two_strided_gradients(G, N, lda, off, ldb):
    """cpfn_bn_relu_bwd_join: BatchNorm-backward pass 1 of a pooled layer on g = bf16(ga + gb), ga a 2-byte-aligned column slice of a
    wider gradient (sa3's input rows, columns 3..), gb the leading columns of another (sfp1's skip): the sum has the bits of the
    framework add autograd would have launched, the partial sums the bits of cpfn_bn_relu_bwd on that sum."""
    from cpfn_amd import lib as _l
    dev = torch.device("cuda:0")
    h = _l.lib()
    gen = torch.Generator(device="cpu").manual_seed(G + N)
    wide_a = torch.randn(G, lda, generator=gen).to(dev).to(torch.bfloat16)
    wide_b = torch.randn(G, ldb, generator=gen).to(dev).to(torch.bfloat16)
    wide_b[3, 1] = float("inf")
    y = torch.randn(G, N, generator=gen).to(dev).to(torch.bfloat16)
    scale = (torch.rand(N, generator=gen) - 0.3).to(dev)
    shift = (torch.randn(N, generator=gen) * 0.2).to(dev)
    ga = wide_a[:, off:off + N]
    ref_sum = (ga + wide_b[:, :N]).contiguous()
    nblk = h.cpfn_bn_bwd_blocks(G)
    part_ref = torch.empty(nblk, 2, N, device=dev)
    part = torch.full((nblk, 2, N), -7.0, device=dev)
    gsum = torch.empty(G, N, dtype=torch.bfloat16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _l.check(h.cpfn_bn_relu_bwd(ref_sum.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), G, N, None, part_ref.data_ptr(),
                                None, 0.0, st), "cpfn_bn_relu_bwd")
    _l.check(h.cpfn_bn_relu_bwd_join(ga.data_ptr(), lda, wide_b.data_ptr(), ldb, y.data_ptr(), scale.data_ptr(), shift.data_ptr(), G, N,
                                     gsum.data_ptr(), part.data_ptr(), st), "cpfn_bn_relu_bwd_join")
    assert torch.equal(gsum.view(torch.int16), ref_sum.view(torch.int16))
    assert torch.equal(part.view(torch.int32), part_ref.view(torch.int32))
    # refused: an odd address for ga, a misaligned gb, a row stride below the width
    assert h.cpfn_bn_relu_bwd_join(ga.data_ptr() + 1, lda, wide_b.data_ptr(), ldb, y.data_ptr(), scale.data_ptr(), shift.data_ptr(), G, N,
                                   gsum.data_ptr(), part.data_ptr(), st) != 0
    assert h.cpfn_bn_relu_bwd_join(ga.data_ptr(), lda, wide_b.data_ptr() + 2, ldb, y.data_ptr(), scale.data_ptr(), shift.data_ptr(), G, N,
                                   gsum.data_ptr(), part.data_ptr(), st) != 0
    assert h.cpfn_bn_relu_bwd_join(ga.data_ptr(), N - 8, wide_b.data_ptr(), ldb, y.data_ptr(), scale.data_ptr(), shift.data_ptr(), G, N,
                                   gsum.data_ptr(), part.data_ptr(), st) != 0
