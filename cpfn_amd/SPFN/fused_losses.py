"""The loss section of the SPFN training step on the fused HIP kernels of csrc/losses.hip.

Same mathematics and return values as `losses_implementation.compute_all_losses`
(reference: SPFN/losses_implementation.py:675-720 with the normalise / soft-max of
Utils/training_utils.py:141-142 folded in), organised as four launches instead of ~500 ops:

  heads Y[B,N,7+K] --HeadPost--> X̂, W (soft-max), normal loss[B], type loss[B]
  W, I_gt          --SegStats--> S[B,K+2,K]  -> Hungarian cost (host SciPy, ONE device->host copy)
                                             -> relaxed IoU of the matched pairs ([B,K] algebra)
  P, X̂, W          --fitters---> 22 parameters / instance           (moments + algebra + cone pass)
  params, match    --Residue---> residue loss[B,K], axis loss[B,K]
"""
import ctypes
import os

import torch
from scipy.optimize import linear_sum_assignment

from .. import lib as _l
from ..ops import _ptr, _stream
from . import fitters_common as _fc

# CPFN_HOST_ASSIGNMENT=1: solve the assignment with SciPy on the host like the reference (one device->host->device
# round trip per step) instead of cpfn_hungarian_match.
HOST_ASSIGNMENT = os.environ.get("CPFN_HOST_ASSIGNMENT", "0") == "1"
# (Round 2 measured and dropped: the assignment branch and the fitter branch on two forked streams — the fork / join
#  pair, forward AND backward, costs more than the 45 us it hides: 2.40 ms serial against 2.51 ms forked.)
# The label-segmented membership sums ride on the heads post-processing launch (K <= 31; wider label sets: SegStats' own
# pass), their adjoint is added to gW inside cpfn_head_post_bwd, and the assignment rides as extra workgroups on the fits'
# first launch (cpfn_fit_moments_fwd_match).

HEADS_HINT = True        # (module attribute for tests: False = the heads' backward makes its own pass over gY, cpfn_colsum_f32)

PARAM_LAYOUT = (("plane_normal", 3), ("plane_center", 1), ("sphere_center", 3), ("sphere_radius_squared", 1),
                ("cylinder_axis", 3), ("cylinder_center", 3), ("cylinder_radius_squared", 1),
                ("cone_apex", 3), ("cone_axis", 3), ("cone_half_angle", 1))


class HeadPost(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Y, X_gt, I_gt, T_gt, with_seg=False, handover=None):
        ctx.handover = handover
        B, N, C = Y.shape
        K = C - 7
        Yc = Y.detach().contiguous().float()
        Xg, Ig, Tg = X_gt.contiguous().float(), I_gt.contiguous(), T_gt.contiguous()
        dev = Y.device
        Xn = torch.empty(B, N, 3, dtype=torch.float32, device=dev)
        W = torch.empty(B, N, K, dtype=torch.float32, device=dev)
        stats = torch.empty(B, 3, dtype=torch.float32, device=dev)
        h = _l.lib()
        chunks = h.cpfn_head_post_chunks(N)
        ws = torch.empty(B * chunks * 3, dtype=torch.float32, device=dev)
        # the label-segmented membership sums ride on this launch (K <= 31): SegStats then has nothing to
        # compute in its forward pass
        seg_ws = S = None
        if with_seg and K <= 31:
            seg_ws = torch.empty(B * chunks * (K + 2) * K, dtype=torch.float32, device=dev)
            S = torch.empty(B, K + 2, K, dtype=torch.float32, device=dev)
        # ... and so does the number of GT instances per cloud (count_gt picks it up: no cpfn_count_labels launch)
        lab_ws = n_gt = None
        if with_seg:
            lab_ws = torch.empty(B * chunks, dtype=torch.int32, device=dev)
            n_gt = torch.empty(B, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            _l.check(h.cpfn_head_post_fwd(_ptr(Yc), _ptr(Xg), _ptr(Ig), _ptr(Tg), B, N, K, _ptr(Xn), _ptr(W), _ptr(ws),
                                          _ptr(stats), _ptr(seg_ws), _ptr(S), _ptr(lab_ws), _ptr(n_gt), _stream()),
                     "cpfn_head_post_fwd")
        global _n_gt_of_last_heads_pass
        _n_gt_of_last_heads_pass = None if n_gt is None else (I_gt.data_ptr(), I_gt._version, tuple(I_gt.shape), n_gt)
        _l.add_bytes("cpfn_head_post_fwd", 4 * B * N * (C + 3 + 3 + K) + 8 * B * N + 8 * B * K)
        ctx.save_for_backward(Yc, Xg, Ig, Tg, W, stats)
        ctx.set_materialize_grads(False)          # (an unused output costs no zero-fill launch)
        if with_seg:
            # S is a DIFFERENTIABLE output of this node — its adjoint (what SegStats.backward computes,
            # cpfn_seg_stats_bwd) is added to gW inside cpfn_head_post_bwd: one launch instead of three (the adjoint, the
            # framework's accumulation add of the two [B,N,K] gradients of W, the heads' backward)
            if S is None:
                S = torch.empty(0, device=dev)
                ctx.mark_non_differentiable(S)
            return Xn, W, stats[:, 0], stats[:, 1], S
        return Xn, W, stats[:, 0], stats[:, 1]

    @staticmethod
    def backward(ctx, gXn, gW, gnl, gtl, gS=None):
        Yc, Xg, Ig, Tg, W, stats = ctx.saved_tensors
        B, N, C = Yc.shape
        dev = Yc.device
        # dL/d(normal loss), dL/d(type loss): LossTail leaves them one after the other in one buffer ([2,B] planar),
        # which the kernel reads as is; anything else is interleaved into [B,2] first
        planar = (gnl is not None and gtl is not None and gnl.dtype == gtl.dtype == torch.float32 and gnl.is_contiguous()
                  and gtl.is_contiguous() and gtl.data_ptr() == gnl.data_ptr() + 4 * B)
        if planar:
            gl = gnl
        else:
            gl = torch.stack([gnl if gnl is not None else torch.zeros(B, device=dev),
                              gtl if gtl is not None else torch.zeros(B, device=dev)], dim=1).contiguous().float()
        gXn = None if gXn is None else gXn.contiguous().float()
        gW = None if gW is None else gW.contiguous().float()
        gS = None if gS is None else gS.contiguous().float()
        gY = torch.empty_like(Yc)
        # what the fc2 heads' backward makes of gY first — its rows as zero-padded bf16 (the operand of their two GEMMs) and the
        # per-256-row column sums (their bias gradient) — leaves this launch too, from the tile while it is in LDS, and is handed
        # to fused_mlp._Linear.backward through the forward pass's HandOver, for exactly this tensor (no cpfn_colsum_f32 launch,
        # no second pass over gY)
        gb = csp = None
        ho = ctx.handover
        if HEADS_HINT and ho is not None and N % 256 == 0 and C <= 64:
            gb = torch.empty(B * N, 64, dtype=torch.bfloat16, device=dev)
            csp = torch.empty((B * N // 256) * C, dtype=torch.float32, device=dev)
            ho.heads_hint = (gY.data_ptr(), gY._version, B * N, C, gb, csp)
        with torch.cuda.device(dev):
            _l.check(_l.lib().cpfn_head_post_bwd(_ptr(Yc), _ptr(Xg), _ptr(Ig), _ptr(Tg), _ptr(W), _ptr(stats), _ptr(gXn),
                                                 _ptr(gW), _ptr(gl), 1 if planar else 0, B, N, C - 7, _ptr(gY), _ptr(gS),
                                                 _ptr(gb), _ptr(csp), _stream()), "cpfn_head_post_bwd")
        _l.add_bytes("cpfn_head_post_bwd", 4 * B * N * (2 * C + 3 + (C - 7) + (3 if gXn is not None else 0) + (C - 7 if gW is not None else 0))
                     + 8 * B * N)
        return gY, None, None, None, None, None


class SegStats(torch.autograd.Function):
    @staticmethod
    def forward(ctx, W, I_gt, S_pre=None):
        B, N, K = W.shape
        Ig = I_gt.contiguous()
        ctx.save_for_backward(Ig)
        ctx.shape = (B, N, K)
        if S_pre is not None and S_pre.numel() == B * (K + 2) * K:
            return S_pre.view(B, K + 2, K)          # computed by the heads post-processing launch (HeadPost)
        Wc = W.detach().contiguous().float()
        h = _l.lib()
        chunks = h.cpfn_seg_stats_chunks(B, N)
        ws = torch.empty(B * chunks * (K + 2) * K, dtype=torch.float32, device=W.device)
        S = torch.empty(B, K + 2, K, dtype=torch.float32, device=W.device)
        with torch.cuda.device(W.device):
            _l.check(h.cpfn_seg_stats_fwd(_ptr(Wc), _ptr(Ig), B, N, K, _ptr(ws), _ptr(S), _stream()), "cpfn_seg_stats_fwd")
        _l.add_bytes("cpfn_seg_stats_fwd", 4 * B * N * K + 8 * B * N + 4 * B * (K + 2) * K)
        return S

    @staticmethod
    def backward(ctx, gS):
        (Ig,) = ctx.saved_tensors
        B, N, K = ctx.shape
        g = gS.contiguous().float()
        dW = torch.empty(B, N, K, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _l.check(_l.lib().cpfn_seg_stats_bwd(_ptr(g), _ptr(Ig), B, N, K, _ptr(dW), _stream()), "cpfn_seg_stats_bwd")
        _l.add_bytes("cpfn_seg_stats_bwd", 4 * B * N * K + 8 * B * N + 4 * B * (K + 2) * K)
        return dW, None, None


class ResidueLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, params, match, T_gt, pts, gt_axes, type_ids):
        B, K, _ = params.shape
        NP = pts.shape[2]
        pc = params.detach().contiguous().float()
        mc, tc = match.contiguous(), T_gt.contiguous()
        out = torch.empty(B, K, 2, dtype=torch.float32, device=pc.device)
        dout = torch.empty(B, K, 10, dtype=torch.float32, device=pc.device)
        ids = (ctypes.c_int * 4)(*type_ids)
        with torch.cuda.device(pc.device):
            _l.check(_l.lib().cpfn_residue_fwd(_ptr(pc), _ptr(mc), _ptr(tc), _ptr(pts.contiguous().float()),
                                               _ptr(gt_axes.contiguous().float()), B, K, NP, ids, _ptr(out), _ptr(dout),
                                               _stream()), "cpfn_residue_fwd")
        _l.add_bytes("cpfn_residue_fwd", 12 * B * K * NP + 4 * B * K * (22 + 9 + 2 + 10) + 16 * B * K)
        ctx.save_for_backward(dout, mc, tc)
        ctx.type_ids = tuple(type_ids)
        return out

    @staticmethod
    def backward(ctx, g):
        dout, mc, tc = ctx.saved_tensors
        B, K, _ = dout.shape
        gp = torch.empty(B, K, 22, dtype=torch.float32, device=dout.device)       # (every element written by the kernel)
        ids = (ctypes.c_int * 4)(*ctx.type_ids)
        with torch.cuda.device(dout.device):
            _l.check(_l.lib().cpfn_residue_bwd(_ptr(g.contiguous().float()), _ptr(dout), _ptr(mc), _ptr(tc), B, K, ids,
                                               _ptr(gp), _stream()), "cpfn_residue_bwd")
        return gp, None, None, None, None, None


class LossTail(torch.autograd.Function):
    """(S, rp, nl, tl, match, n_gt) -> total [] and parts [5] = normal, type, miou, residue, parameter
    (cpfn_loss_tail: one launch; the gradients of the total are produced by the same launch)."""
    unit_grad = False      # set by a trainer that always back-propagates the plain total (saves one scaling kernel)

    @staticmethod
    def forward(ctx, S, rp, nl, tl, match, n_gt, mult6):
        B, K2, K = S.shape
        dev = S.device
        Sc = S.detach().contiguous().float()
        rpc = None if rp is None else rp.detach().contiguous().float()
        assert nl.stride(0) == tl.stride(0) and nl.dtype == tl.dtype == torch.float32
        out = torch.empty(6, dtype=torch.float32, device=dev)
        nS, nR = B * K2 * K, B * K * 2
        flat = torch.empty(nS + nR + 2 * B, dtype=torch.float32, device=dev)
        gS, grp = flat[:nS].view(B, K2, K), flat[nS:nS + nR].view(B, K, 2)
        gnl, gtl = flat[nS + nR:nS + nR + B], flat[nS + nR + B:]
        mu = (ctypes.c_float * 6)(*[float(v) for v in mult6])
        with torch.cuda.device(dev):
            _l.check(_l.lib().cpfn_loss_tail(_ptr(Sc), _ptr(rpc), _ptr(nl), _ptr(tl), nl.stride(0), _ptr(match.contiguous()),
                                             _ptr(n_gt.contiguous()), B, K, mu, _ptr(out), _ptr(gS), _ptr(grp), _ptr(gnl),
                                             _ptr(gtl), _stream()), "cpfn_loss_tail")
        ctx.save_for_backward(flat)
        ctx.dims = (B, K2, K, rp is not None)
        ctx.unit_grad = LossTail.unit_grad
        parts = out[1:]
        ctx.mark_non_differentiable(parts)
        ctx.set_materialize_grads(False)
        return out[0], parts

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        (flat,) = ctx.saved_tensors
        B, K2, K, has_rp = ctx.dims
        nS, nR = B * K2 * K, B * K * 2
        # (unit_grad: the caller promises to call total.backward() with the default gradient 1 — the trainer does)
        f = flat if ctx.unit_grad else flat * g_total
        return (f[:nS].view(B, K2, K), f[nS:nS + nR].view(B, K, 2) if has_rp else None, f[nS + nR:nS + nR + B],
                f[nS + nR + B:], None, None, None)


class HeatCrossEntropy(torch.autograd.Function):
    """The PatchSelection objective (Utils/training_utils.py:66-68): mean two-class cross-entropy of the heat-map logits
    Y [B,N,2] (the packed fp32 heads) against labels [B,N], with d loss / d Y formed in the same launch (cpfn_ce2) and, for the
    heads' backward, its padded bf16 rows / column sums handed over like HeadPost does."""

    @staticmethod
    def forward(ctx, Y, labels, handover=None):
        B, N, C = Y.shape
        if C != 2:
            raise RuntimeError("HeatCrossEntropy: two-class logits expected")
        Yc = Y.detach().contiguous().float()
        lab = labels.contiguous().long()
        P = B * N
        dev = Y.device
        h = _l.lib()
        ws = torch.empty(h.cpfn_ce2_blocks(P), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dY = torch.empty_like(Yc)
        gb = csp = None
        if HEADS_HINT and handover is not None and P % 256 == 0:
            gb = torch.empty(P, 64, dtype=torch.bfloat16, device=dev)
            csp = torch.empty((P // 256) * 2, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _l.check(h.cpfn_ce2(_ptr(Yc), _ptr(lab), P, _ptr(ws), _ptr(loss), _ptr(dY), _ptr(gb), _ptr(csp), _stream()), "cpfn_ce2")
        _l.add_bytes("cpfn_ce2", 4 * P * 4 + 8 * P + (128 * P if gb is not None else 0))
        ctx.save_for_backward(dY)
        ctx.hint = (handover, P, gb, csp)
        ctx.unit_grad = LossTail.unit_grad
        return loss

    @staticmethod
    def backward(ctx, g):
        (dY,) = ctx.saved_tensors
        ho, P, gb, csp = ctx.hint
        if not ctx.unit_grad:                 # (a caller that scales the loss: the hint describes the unscaled gradient)
            return dY * g, None, None
        if ho is not None and gb is not None:
            ho.heads_hint = (dY.data_ptr(), dY._version, P, 2, gb, csp)
        return dY, None, None


class unit_loss_gradient:
    """Context manager for a trainer that back-propagates the plain total (`total.backward()`): LossTail then
    hands its stored gradients on as they are instead of multiplying them by the incoming 1.0."""

    def __enter__(self):
        self._prev, LossTail.unit_grad = LossTail.unit_grad, True

    def __exit__(self, *exc):
        LossTail.unit_grad = self._prev
        return False


_n_gt_of_last_heads_pass = None      # (labels' data_ptr, version, shape, n_gt) left by the last HeadPost.forward


def count_gt(I_gt):
    """[B] int64: number of GT instances per cloud (max label + 1; reference lines 603-606).  When the heads
    post-processing launch has just counted them for the same label tensor, that result is returned."""
    if not I_gt.is_cuda:
        return I_gt.max(dim=1)[0] + 1
    global _n_gt_of_last_heads_pass
    c, _n_gt_of_last_heads_pass = _n_gt_of_last_heads_pass, None       # one-shot: a later call counts again
    if c is not None and c[0] == I_gt.data_ptr() and c[1] == I_gt._version and c[2] == tuple(I_gt.shape):
        return c[3]
    Ig = I_gt.contiguous()
    n_gt = torch.empty(Ig.shape[0], dtype=torch.int64, device=Ig.device)
    with torch.cuda.device(Ig.device):
        _l.check(_l.lib().cpfn_count_labels(_ptr(Ig), Ig.shape[0], Ig.shape[1], _ptr(n_gt), _stream()), "cpfn_count_labels")
    return n_gt


def hungarian_cost_pack(S, I_gt, n_gt=None):
    """Device part of the assignment: relaxed-IoU cost of every (GT label, prediction) pair from the
    segmented sums plus the number of GT labels, packed as one [B, K*K+1] tensor (reference lines 19-25).
    Capturable; `hungarian_from_pack` is the host part."""
    B, K2, K = S.shape
    D, col, cnt = S[:, :K], S[:, K], S[:, K + 1]
    den = cnt.unsqueeze(2) + col.unsqueeze(1) - D
    cost = D / den.clamp(min=1e-10)
    if n_gt is None:
        n_gt = count_gt(I_gt)
    _drop_pending_n_gt()
    return torch.cat([cost.reshape(B, -1), n_gt.unsqueeze(1).to(cost.dtype)], dim=1)


def hungarian_host(h, K, out=None):
    """Host part on a host array h [B, K*K+1] (numpy or CPU tensor): SciPy assignment per cloud (reference :27).
    Returns (or fills `out`, e.g. a pinned buffer) the [B, K] int64 matching."""
    h = h.numpy() if isinstance(h, torch.Tensor) else h
    B = h.shape[0]
    match = torch.zeros(B, K, dtype=torch.long) if out is None else out.zero_()
    for b in range(B):
        n = int(h[b, -1])
        _, c = linear_sum_assignment(-h[b, :-1].reshape(K, K)[:n])
        match[b, :n] = torch.from_numpy(c)
    return match


def hungarian_from_pack(pack, K):
    """Host part: ONE device->host copy for the whole batch, SciPy assignment per cloud (reference :27)."""
    return hungarian_host(pack.cpu(), K).to(pack.device)


def _drop_pending_n_gt():
    """The count left by a heads pass belongs to THAT batch's loss computation: any later stage of it drops an unclaimed one,
    so that it can never answer a count_gt() of a different batch that happens to live at the same address."""
    global _n_gt_of_last_heads_pass
    _n_gt_of_last_heads_pass = None


def hungarian_device(S, n_gt):
    """The assignment on the device (cpfn_hungarian_match): S [B,K+2,K] from SegStats, n_gt [B] int64 ->
    match [B,K] int64.  Same solver and tie-breaking as the SciPy call of the reference
    (losses_implementation.py:27), no host round trip, capturable."""
    _drop_pending_n_gt()
    B, K2, K = S.shape
    Sc = S.detach().contiguous().float()
    match = torch.empty(B, K, dtype=torch.long, device=S.device)
    with torch.cuda.device(S.device):
        _l.check(_l.lib().cpfn_hungarian_match(_ptr(Sc), _ptr(n_gt.contiguous()), B, K, _ptr(match), _stream()),
                 "cpfn_hungarian_match")
    return match


def fit_params_and_match(P, W, Xn, multipliers, S, n_gt):
    """fit_params + hungarian_device with the assignment riding on the fits' first launch (cpfn_fit_moments_fwd_match:
    the two are independent; one launch and the shorter of the two durations less on the chain).  Returns (params, match);
    falls back to the two separate launches when there are no fits to ride on."""
    if not (multipliers["residue"] > 0 or multipliers["parameter"] > 0):
        return fit_params(P, W, Xn, multipliers), hungarian_device(S, n_gt)
    from . import moments as _m
    _drop_pending_n_gt()
    B, K2, K = S.shape
    Sc = S.detach().contiguous().float()
    match = torch.empty(B, K, dtype=torch.long, device=S.device)
    _m.set_match_rider(Sc, n_gt.contiguous(), match)
    try:
        params = fit_params(P, W, Xn, multipliers)
    finally:
        left = _m.pending_match_rider()
    if left is not None:                 # nobody took it (shapes beyond the fused kernels)
        match = hungarian_device(S, n_gt)
    return params, match


def hungarian_from_stats(S, I_gt):
    return hungarian_from_pack(hungarian_cost_pack(S, I_gt), S.shape[2])


def pre_match(Y, batch, handover=None):
    """Everything before the host-side assignment: unit normals, memberships, per-cloud normal /
    type losses and the label-segmented sums S.  (Capturable: no host synchronisation.)
    handover: the fused_mlp.HandOver of the forward pass that produced Y (the model's `handover` attribute): the heads'
    backward then gets its padded gradient rows / column sums from this section's backward launch."""
    Xn, W, nl, tl, S_pre = HeadPost.apply(Y, batch["X_gt"], batch["I_gt"], batch["T_gt"], True, handover)
    if S_pre.numel() == W.shape[0] * (W.shape[2] + 2) * W.shape[2]:
        return Xn, W, nl, tl, S_pre                   # computed AND differentiated by the heads post-processing node
    return Xn, W, nl, tl, SegStats.apply(W, batch["I_gt"], S_pre)


def fit_params(P, W, Xn, multipliers):
    """The match-independent part of the post-assignment work (all four fits of every instance); None when
    neither the residue nor the parameter loss is switched on.  A trainer can run it while the host solves the
    assignment."""
    if multipliers["residue"] > 0 or multipliers["parameter"] > 0:
        return _fc.fit_params(P, W, Xn)
    return None


def post_match(P, Xn, W, nl, tl, S, match, batch, multipliers, classes, n_gt=None, params=None):
    """Everything after the assignment: relaxed IoU of the matched pairs, the fitters, residue and
    axis losses, the weighted total.  (Capturable.)  Four autograd nodes: FitParams -> ResidueLoss ->
    LossTail (<- SegStats, HeadPost)."""
    m = multipliers
    T_gt = batch["T_gt"]
    if n_gt is None:
        n_gt = count_gt(batch["I_gt"])
    _drop_pending_n_gt()
    rp = None
    if m["residue"] > 0 or m["parameter"] > 0:
        if params is None:
            params = _fc.fit_params(P, W, Xn)
        gt_axes = batch.get("gt_axes")          # [3,B,K,3]; a trainer with static input buffers keeps them stacked
        if gt_axes is None:
            gt_axes = torch.stack([batch["plane_n_gt"], batch["cylinder_axis_gt"], batch["cone_axis_gt"]], 0)
        ids = [classes.index(c) for c in ("plane", "sphere", "cylinder", "cone")]
        rp = ResidueLoss.apply(params, match, T_gt, batch["points_per_instance"], gt_axes, ids)
    mult6 = [m["normal"], m["type"], m["miou"], m["residue"], m["parameter"], m["total"]]
    total, parts = LossTail.apply(S, rp, nl, tl, match, n_gt, mult6)
    return total, parts[0], parts[1], parts[2], parts[3], parts[4]


def fused_losses(P, Y, batch, multipliers, classes, handover=None):
    """P [B,N,3]; Y [B,N,7+K] = packed fp32 heads (normal | type logits | membership logits).
    Returns the reference's (total, normal, type, miou, residue, parameter) scalars.
    More than 32 instance columns (beyond the fused kernels' tile): the op-by-op twin
    `losses_implementation.compute_all_losses` on the same heads (HIP fitters, stock reductions)."""
    if Y.shape[2] - 7 > 32:
        from . import losses_implementation as li
        X = torch.nn.functional.normalize(Y[..., :3], p=2, dim=2, eps=1e-12)
        W = torch.softmax(Y[..., 7:], dim=2)
        gt = {"plane_normal": batch["plane_n_gt"], "cylinder_axis": batch["cylinder_axis_gt"], "cone_axis": batch["cone_axis_gt"]}
        m = multipliers
        return li.compute_all_losses(P, W, batch["I_gt"], X, batch["X_gt"], Y[..., 3:7], batch["T_gt"], gt,
                                     batch["points_per_instance"], m["normal"], m["type"], m["miou"], m["residue"],
                                     m["parameter"], m["total"], False, mode_seg='mIoU', classes=classes)[:6]
    if HOST_ASSIGNMENT:
        Xn, W, nl, tl, S = pre_match(Y, batch, handover)
        n_gt = count_gt(batch["I_gt"])
        match = hungarian_from_pack(hungarian_cost_pack(S.detach(), batch["I_gt"], n_gt), S.shape[2])
        return post_match(P, Xn, W, nl, tl, S, match, batch, multipliers, classes, n_gt)
    Xn, W, nl, tl, S = pre_match(Y, batch, handover)
    n_gt = count_gt(batch["I_gt"])
    params, match = fit_params_and_match(P, W, Xn, multipliers, S, n_gt)
    return post_match(P, Xn, W, nl, tl, S, match, batch, multipliers, classes, n_gt, params)
