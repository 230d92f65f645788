"""GPU: the small bookkeeping kernels behind the C ABI that replaced framework glue ops, each against the PyTorch
expression it replaced (ragged sizes, padding, alignment cases)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _call(name, *args):
    from cpfn_amd import lib as _l
    _l.check(getattr(_l.lib(), name)(*args), name)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def test_multi_cast_matches_per_tensor_copies():
    from cpfn_amd import fused_mlp
    g = torch.Generator().manual_seed(0)
    jobs = []
    for rows, cols, ld, f32 in [(64, 3, 64, 0), (128, 131, 192, 0), (1, 1, 8, 0), (256, 259, 320, 0), (1, 35, 35, 1),
                                (37, 128, 128, 0), (1000, 129, 136, 1)]:
        src = torch.randn(rows, cols, generator=g).to(dev())
        dst = torch.full((rows, ld), 7.0, dtype=torch.float32 if f32 else torch.bfloat16, device=dev())
        jobs.append((src, dst, rows, cols, ld, f32))
    arr = (fused_mlp._CastDesc * len(jobs))(*[fused_mlp._CastDesc(s.data_ptr(), d.data_ptr(), r, c, ld, f) for s, d, r, c, ld, f in jobs])
    _call("cpfn_multi_cast", arr, len(jobs), _stream())
    for src, dst, rows, cols, ld, f32 in jobs:
        want = src if f32 else src.to(torch.bfloat16)
        assert torch.equal(dst[:, :cols], want)
        assert bool((dst[:, cols:] == 7.0).all())               # padding columns untouched


def test_count_labels():
    from cpfn_amd.SPFN import fused_losses as fl
    g = torch.Generator().manual_seed(1)
    for B, N, hi in [(1, 1, 1), (3, 1000, 28), (16, 8192, 21), (5, 4097, 200)]:
        I = torch.randint(0, hi, (B, N), generator=g)
        I[0, -1] = hi - 1
        got = fl.count_gt(I.to(dev()))
        assert torch.equal(got.cpu(), I.max(dim=1)[0] + 1)


def test_concat_pos_feats_forward_and_backward():
    from cpfn_amd import autograd_ops
    g = torch.Generator().manual_seed(2)
    for R, C, cpad in [(1, 8, 64), (2048, 256, 320), (77, 128, 192)]:
        xyz = torch.randn(R, 3, generator=g).to(dev())
        feats = torch.randn(R, C, generator=g).to(dev()).to(torch.bfloat16).requires_grad_(True)
        out = autograd_ops.ConcatPosFeats.apply(xyz, feats, cpad)
        want = torch.cat([xyz.to(torch.bfloat16), feats.detach(), torch.zeros(R, cpad - C - 3, dtype=torch.bfloat16, device=dev())], 1)
        assert torch.equal(out, want)
        gout = torch.randn(R, cpad, generator=g).to(dev()).to(torch.bfloat16)
        out.backward(gout)
        assert torch.equal(feats.grad, gout[:, 3:3 + C])


def test_compacting_split_reduce():
    """cpfn_multi_split_reduce with row_in / row_out: the weight gradient of a zero-padded K, summed over the splits
    in subset order and compacted to [N, cin] by the same launch."""
    from cpfn_amd import fused_mlp
    g = torch.Generator().manual_seed(3)
    N, Kp, cin, splits = 96, 192, 131, 7
    ws = torch.randn(splits, N, Kp, generator=g).to(dev())
    flat = torch.randn(5, 1000, generator=g).to(dev())
    out_c = torch.empty(N, cin, device=dev())
    out_f = torch.empty(1000, device=dev())
    arr = (fused_mlp._ReduceDesc * 2)(fused_mlp._ReduceDesc(ws.data_ptr(), out_c.data_ptr(), N * Kp, splits, Kp, cin),
                                      fused_mlp._ReduceDesc(flat.data_ptr(), out_f.data_ptr(), 1000, 5, 0, 0))
    _call("cpfn_multi_split_reduce", arr, 2, _stream())
    torch.testing.assert_close(out_c, ws.sum(0)[:, :cin], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out_f, flat.sum(0), rtol=1e-5, atol=1e-5)


def test_wide_split_reduce_fixed_addition_orders():
    """Large buffers take a float4 layout with a FIXED order of additions per element (bitwise reproducible, checked here against
    the same order spelled out in torch): fewer than 64 partial rows — 256 elements x 4 split-subsets per workgroup, the order of
    the 64 x 4 layout: subset r adds rows r, r + 4, ... one by one, then (s0 + s1) + (s2 + s3); 64 rows or more (round 4) —
    64 elements x 16 subsets: subset r adds rows r, r + 16, ..., then s0 + s1 + ... + s15 in that order.  Padded rows are
    compacted by the same launch."""
    from cpfn_amd import fused_mlp
    g = torch.Generator().manual_seed(5)
    for n_rows, Kp, cin, splits in [(128, 128, 128, 256), (256, 128, 128, 37), (128, 192, 131, 256), (64, 64, 64, 3),
                                    (65, 64, 64, 5), (1030, 4, 4, 33), (67, 64, 35, 9), (128, 64, 64, 64), (100, 128, 128, 100)]:
        ws = torch.randn(splits, n_rows, Kp, generator=g).to(dev())
        out = torch.empty(n_rows, cin, device=dev())
        arr = (fused_mlp._ReduceDesc * 1)(fused_mlp._ReduceDesc(ws.data_ptr(), out.data_ptr(), n_rows * Kp, splits,
                                                                 0 if cin == Kp else Kp, 0 if cin == Kp else cin))
        _call("cpfn_multi_split_reduce", arr, 1, _stream())
        wide_deep = splits >= 64 and n_rows * Kp >= 4096 and (n_rows * Kp) % 4 == 0
        nsub = 16 if wide_deep else 4
        sub = []
        for r in range(nsub):
            acc = torch.zeros(n_rows, Kp, device=dev())
            for i in range(r, splits, nsub):
                acc = acc + ws[i]
            sub.append(acc)
        if wide_deep:
            want = sub[0]
            for q in range(1, 16):
                want = want + sub[q]
        else:
            want = (sub[0] + sub[1]) + (sub[2] + sub[3])
        assert torch.equal(out, want[:, :cin]), (n_rows, Kp, cin, splits)


def test_colsum_with_padded_bf16_copy():
    g = torch.Generator().manual_seed(4)
    for P, C in [(1, 35), (131072, 35), (5000, 64), (257, 3)]:
        X = torch.randn(P, C, generator=g).to(dev())
        ws = torch.empty(((P + 255) // 256) * C, device=dev())
        out = torch.empty(C, device=dev())
        pad = torch.full((P, 64), 5.0, dtype=torch.bfloat16, device=dev())
        _call("cpfn_colsum_f32", X.data_ptr(), P, C, ws.data_ptr(), out.data_ptr(), pad.data_ptr(), _stream())
        torch.testing.assert_close(out, X.sum(0), rtol=1e-4, atol=1e-3)
        assert torch.equal(pad[:, :C], X.to(torch.bfloat16)) and bool((pad[:, C:] == 0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("mode,B,M,N,C1,C2", [("interp", 3, 128, 512, 128, 256), ("interp", 2, 512, 2000, 64, 128),
                                               ("broadcast", 4, 1, 128, 256, 1024)])
def test_concat_interp_matches_interp_plus_cat(mode, B, M, N, C1, C2):
    """cpfn_concat_interp_bf16 (skip | interpolation or broadcast, one launch) against the two-launch form: forward bit for
    bit; adjoints equal (interpolation: same inverse-index kernel on a strided view; broadcast: fp32 column sums)."""
    from cpfn_amd import autograd_ops, ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * N + C2)
    skip = torch.randn(B, N, C1, generator=g).to(dev).to(torch.bfloat16)
    feats = torch.randn(B, M, C2, generator=g).to(dev).to(torch.bfloat16)
    gout = torch.randn(B, N, C1 + C2, generator=g).to(dev).to(torch.bfloat16)
    if mode == "interp":
        idx = torch.randint(0, M, (B, N, 3), generator=g).to(dev).to(torch.int32)
        w = torch.rand(B, N, 3, generator=g).to(dev)
        w = (w / w.sum(2, keepdim=True)).contiguous()
        inv = ops.csr_build(idx, M)
    else:
        idx = w = inv = None
    res = []
    for fused in (True, False):
        s_, f_ = skip.clone().requires_grad_(True), feats.clone().requires_grad_(True)
        if fused:
            assert autograd_ops.concat_interp_ok(s_, f_, idx)
            out = autograd_ops.concat_interp(s_, f_, idx, w, inv)
        else:
            it = f_.expand(B, N, C2) if idx is None else autograd_ops.interp_rows(f_, idx, w, inv)
            out = torch.cat([s_, it], dim=2)
        out.backward(gout)
        res.append((out.detach(), s_.grad, f_.grad))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    if mode == "interp":
        assert torch.equal(res[0][2], res[1][2])
    else:
        a, b = res[0][2].float(), res[1][2].float()
        assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max())


def test_checked_reductions_and_finalize_raise_the_sticky_word_only_for_what_they_store():
    """cpfn_multi_split_reduce_checked / cpfn_bn_bwd_finalize(_ride)_checked: the finite check of a step's gradients rides on the
    launches that write them.  One word, OR-ed with 1 by a workgroup that STORED a NaN / inf: every reduction layout (flat, compact,
    deep, wide, wide-deep, the xyz coefficient form), a NaN in a padding column that is never stored must NOT raise it, and the
    outputs are those of the plain entries, bit for bit."""
    from cpfn_amd import fused_mlp, lib as _l
    h = _l.lib()
    g = torch.Generator().manual_seed(11)
    flag = torch.zeros(1, dtype=torch.int32, device=dev())
    cases = [(96, 192, 131, 7), (128, 128, 128, 256), (256, 128, 128, 37), (30, 16, 16, 200), (1, 35, 35, 512), (64, 8192 // 64, 70, 16)]
    for poison in (None, "stored", "padding"):
        for n_rows, Kp, cin, splits in cases:
            if poison == "padding" and cin == Kp:
                continue
            ws = torch.randn(splits, n_rows, Kp, generator=g).to(dev())
            if poison == "stored":
                ws[splits // 2, n_rows - 1, cin - 1] = float("inf")
            elif poison == "padding":
                ws[splits // 2, 0, Kp - 1] = float("nan")
            out = torch.empty(n_rows, cin, device=dev())
            ref = torch.empty(n_rows, cin, device=dev())
            d = lambda o: (fused_mlp._ReduceDesc * 1)(fused_mlp._ReduceDesc(ws.data_ptr(), o.data_ptr(), n_rows * Kp, splits,
                                                                           0 if cin == Kp else Kp, 0 if cin == Kp else cin))
            flag.zero_()
            _call("cpfn_multi_split_reduce_checked", d(out), 1, flag.data_ptr(), _stream())
            _call("cpfn_multi_split_reduce", d(ref), 1, _stream())
            assert torch.equal(out.view(torch.int32), ref.view(torch.int32))
            assert int(flag) == (1 if poison == "stored" else 0), (poison, n_rows, Kp, cin, splits)
    # the xyz coefficient form: out[c][j] = c0 S1 + c1 S2 + c2 S3
    C, splits = 64, 40
    for poison in (False, True):
        part = torch.randn(splits, 7, C, generator=g).to(dev())
        coef = torch.randn(3, C, generator=g).to(dev())
        if poison:
            part[3, 1, 5] = float("nan")
        out = torch.empty(C, 3, device=dev())
        arr = (fused_mlp._ReduceDesc * 1)(fused_mlp._ReduceDesc(part.data_ptr(), out.data_ptr(), 3 * C, splits, C, 0, 0, coef.data_ptr()))
        flag.zero_()
        _call("cpfn_multi_split_reduce_checked", arr, 1, flag.data_ptr(), _stream())
        assert int(flag) == int(poison)
    # finalize (+ riders): dgamma / dbeta of a layer, one NaN partial
    for ride in (False, True):
        for poison in (False, True):
            C, nblk, P = 128, 37, 4096
            part = torch.randn(nblk, 2, C, generator=g).to(dev())
            if poison:
                part[7, 1, 100] = float("inf")
            gamma, mean, rstd = (torch.rand(C, generator=g).to(dev()) + 0.5 for _ in range(3))
            dg, db, coef = torch.empty(C, device=dev()), torch.empty(C, device=dev()), torch.empty(3, C, device=dev())
            ws = torch.randn(5, 1000, generator=g).to(dev())
            o2 = torch.empty(1000, device=dev())
            arr = (fused_mlp._ReduceDesc * 1)(fused_mlp._ReduceDesc(ws.data_ptr(), o2.data_ptr(), 1000, 5, 0, 0))
            flag.zero_()
            if ride:
                _call("cpfn_bn_bwd_finalize_ride_checked", part.data_ptr(), nblk, C, float(P), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                      1, dg.data_ptr(), db.data_ptr(), coef.data_ptr(), arr, 1, flag.data_ptr(), _stream())
                torch.testing.assert_close(o2, ws.sum(0), rtol=1e-5, atol=1e-5)
            else:
                _call("cpfn_bn_bwd_finalize_checked", part.data_ptr(), nblk, C, float(P), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                      1, dg.data_ptr(), db.data_ptr(), coef.data_ptr(), flag.data_ptr(), _stream())
            assert int(flag) == int(poison), (ride, poison)
            assert bool(torch.isfinite(dg).all()) == (not poison)
