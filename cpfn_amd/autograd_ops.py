"""torch.autograd wrappers around the points-major HIP data movers."""
import torch

from . import lib as _l
from . import ops
from .ops import _ptr, _stream


class GatherRows(torch.autograd.Function):
    """rows [B,N,C] , idx [B,...] i32 -> [B,...,C]; adjoint = fp32 scatter-add of rows."""

    @staticmethod
    def forward(ctx, rows, idx):
        ctx.save_for_backward(idx)
        ctx.n = rows.shape[1]
        ctx.in_dtype = rows.dtype
        return ops.gather_rows(rows.contiguous(), idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g32 = g.contiguous().float()
        return ops.scatter_add_rows(g32, idx, ctx.n).to(ctx.in_dtype), None


class InterpRows(torch.autograd.Function):
    """feats [B,M,C] f32, idx/w [B,N,3] -> [B,N,C] (three_weighted_sum in row layout)."""

    @staticmethod
    def forward(ctx, feats, idx, w):
        ctx.save_for_backward(idx, w)
        ctx.m = feats.shape[1]
        ctx.in_dtype = feats.dtype
        return ops.interp_rows_fwd(feats.contiguous().float(), idx, w).to(feats.dtype)

    @staticmethod
    def backward(ctx, g):
        idx, w = ctx.saved_tensors
        return ops.interp_rows_bwd(g.contiguous().float(), idx, w, ctx.m).to(ctx.in_dtype), None, None


def _scatter_bf16(g, ldg, idx, w, T, B, R, M, C):
    """LDS-privatised scatter-add of bf16 rows into a zero-filled fp32 [B,M,C] target (M <= 1024)."""
    out = torch.zeros(B, M, C, dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        _l.check(_l.lib().cpfn_scatter_rows_bf16(_ptr(g), ldg, _ptr(idx), _ptr(w), T, B, R, M, C, _ptr(out), _stream()),
                 "cpfn_scatter_rows_bf16")
    _l.add_bytes("cpfn_scatter_rows_bf16", 2 * B * R * C + 4 * B * R * T + 8 * B * M * C)
    return out


def _csr_sum_bf16(g, ldg, inv, w, T, B, R, M, C, addend=None, ld_add=0):
    """Atomic-free adjoint through the inverse index `inv` = (offsets, entries): bf16 [B,M,C].
    addend (bf16 rows [B*M, ld_add]): the other gradient of a two-consumer tensor, added in the same launch (SkipJoin)."""
    out = torch.empty(B, M, C, dtype=torch.bfloat16, device=g.device)
    with torch.cuda.device(g.device):
        if addend is None:
            _l.check(_l.lib().cpfn_csr_gather_sum_bf16(_ptr(g), ldg, _ptr(inv[0]), _ptr(inv[1]), _ptr(w), T, B, R, M, C, _ptr(out),
                                                       _stream()), "cpfn_csr_gather_sum_bf16")
        else:
            _l.check(_l.lib().cpfn_csr_gather_sum_add_bf16(_ptr(g), ldg, _ptr(inv[0]), _ptr(inv[1]), _ptr(w), T, B, R, M, C,
                                                           _ptr(addend), ld_add, _ptr(out), _stream()), "cpfn_csr_gather_sum_add_bf16")
    # compulsory: every source row once, the inverse index (+ weights), one bf16 row per target (the kernel re-reads a
    # source row once per (target, entry) pair it appears in: T times — that shows as traffic above this figure)
    _l.add_bytes("cpfn_csr_gather_sum_bf16", 2 * B * R * C + 4 * B * R * T * (2 if w is not None else 1) + 4 * B * (M + 1) + 2 * B * M * C
                 + (2 * B * M * C if addend is not None else 0))
    return out


# A tensor with two consumers inside ONE forward pass gets its two gradients added by autograd's input buffer: a framework
# bf16 add (5 us + a kernel boundary on the step's chain) between the two backward nodes.  Where the network's topology fixes
# the order of those nodes — sa1's features feed sa2's grouping (GroupConcat) and sfp2's skip concatenation (ConcatInterp),
# PointNet2/pn2_network.py:45-46,55, and sa2 is an ancestor of sfp2: ConcatInterp's backward always runs first — the first
# node hands its gradient to the second through a SkipJoin (an object of that one forward pass, like fused_mlp.HandOver) and
# returns None; the second adds it inside its own launch with the framework add's roundings (cpfn_csr_gather_sum_add_bf16:
# bit-identical gradients, tests/test_gpu_network.py).  Armed only when both nodes saw the SAME tensor (address, in-place
# version, shape); a gradient handed over and never picked up — a partial backward pass that stops above sa2 — raises at the
# end of that backward pass instead of being lost.
# Round 6, the second such tensor: sa2's pooled features feed sa3's input rows (ConcatPosFeats) and sfp1's skip (pn2_network.py:
# 48-49,56).  There the node that takes the handed-over gradient is the PRODUCER's backward — sa2's fused stack, whose first launch
# (BatchNorm-backward pass 1 over the pooled gradient) forms the sum on load (cpfn_bn_relu_bwd_join): the step's last framework add.
SKIP_JOIN = True
OUTPUT_JOIN = __import__("os").environ.get("CPFN_OUTPUT_JOIN", "1") == "1"      # the round-6 form (sa2's output)


class SkipJoin:
    __slots__ = ("src", "armed", "addend")

    def __init__(self):
        self.src, self.armed, self.addend = None, False, None

    @staticmethod
    def key(t):
        return (t.data_ptr(), t._version, tuple(t.shape), t.dtype)

    @staticmethod
    def flat_key(t):
        """(for a producer that sees its output as rows [G, C] while the consumer sees [B, S, C]: same storage, same version)"""
        return (t.data_ptr(), t._version, t.numel(), t.dtype)

    def _unclaimed(self):
        if self.addend is not None:
            self.addend = None
            raise RuntimeError("cpfn_amd: a skip connection's gradient was handed to the grouping adjoint (autograd_ops.SkipJoin) "
                               "but that node did not run in this backward pass; set cpfn_amd.autograd_ops.SKIP_JOIN = False "
                               "for partial backward passes")


class InterpRowsBf16(torch.autograd.Function):
    """bf16 feats [B,M,C] (C % 8 == 0, M <= 1024), idx/w [B,N,3] -> bf16 [B,N,C].
    `inv` = optional inverse index of idx (ops.csr_build): deterministic, atomic-free adjoint."""

    @staticmethod
    def forward(ctx, feats, idx, w, inv_off=None, inv_ent=None):
        B, M, C = feats.shape
        N = idx.shape[1]
        f = feats.contiguous()
        out = torch.empty(B, N, C, dtype=torch.bfloat16, device=f.device)
        with torch.cuda.device(f.device):
            _l.check(_l.lib().cpfn_interp_rows_bf16(_ptr(f), _ptr(idx), _ptr(w), B, M, N, C, _ptr(out), _stream()),
                     "cpfn_interp_rows_bf16")
        _l.add_bytes("cpfn_interp_rows_bf16", 2 * B * M * C + 24 * B * N + 2 * B * N * C)
        ctx.save_for_backward(idx, w)
        ctx.inv = None if inv_off is None else (inv_off, inv_ent)
        ctx.dims = (B, M, N, C)
        return out

    @staticmethod
    def backward(ctx, g):
        idx, w = ctx.saved_tensors
        B, M, N, C = ctx.dims
        g = g.contiguous().to(torch.bfloat16)
        if ctx.inv is not None:
            return _csr_sum_bf16(g, C, ctx.inv, w, 3, B, N, M, C), None, None, None, None
        return _scatter_bf16(g, C, idx, w, 3, B, N, M, C).to(torch.bfloat16), None, None, None, None


class GroupConcat(torch.autograd.Function):
    """Grouped set-abstraction input rows in one pass (modules/pointset_abstraction.py:62-66):
    out[p] = [feats[b, idx[p], :C] | rel[p, :3] | zeros] as bf16 [B*S*K, Cpad]; rel None (cpad == C): the gather alone —
    the coordinates then reach the first layer as its fp32 "xyz tail" (fused_mlp)."""

    @staticmethod
    def forward(ctx, feats, rel, idx, cpad, inv_off=None, inv_ent=None, join=None, lazy=False):
        """lazy (round 6, rel None): the rows are NOT gathered — the consumer (fused_mlp: cpfn_mlp_gemm_xyz_gather and the backward
        kernel's twin) reads them out of `feats` through `idx` while loading its operand; the returned tensor is uninitialised
        storage that only carries the autograd edge (its gradient arrives here as before)."""
        B, N, C = feats.shape
        R = idx[0].numel()
        f = feats.contiguous()
        ctx.join = None
        if join is not None and SKIP_JOIN and inv_off is not None:
            join.src, join.armed, join.addend = SkipJoin.key(f), False, None
            ctx.join = join
        out = torch.empty(B * R, cpad, dtype=torch.bfloat16, device=f.device)
        if not (lazy and rel is None):
            with torch.cuda.device(f.device):
                _l.check(_l.lib().cpfn_group_concat_bf16(_ptr(f), None if rel is None else _ptr(rel.contiguous()), _ptr(idx), B, N, R, C, cpad, _ptr(out),
                                                         _stream()), "cpfn_group_concat_bf16")
            _l.add_bytes("cpfn_group_concat_bf16", 2 * B * N * C + 16 * B * R + 2 * B * R * cpad)
        ctx.save_for_backward(idx)
        ctx.inv = None if inv_off is None else (inv_off, inv_ent)
        ctx.dims = (B, N, R, C, cpad)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        B, N, R, C, cpad = ctx.dims
        g = g.contiguous().to(torch.bfloat16)
        if ctx.inv is not None:
            addend, ld_add = None, 0
            j = ctx.join
            if j is not None and j.addend is not None:
                addend, ld_add = j.addend
                j.addend = None
            return _csr_sum_bf16(g, cpad, ctx.inv, None, 1, B, R, N, C, addend, ld_add), None, None, None, None, None, None, None
        return _scatter_bf16(g, cpad, idx, None, 1, B, R, N, C).to(torch.bfloat16), None, None, None, None, None, None, None


class ConcatPosFeats(torch.autograd.Function):
    """Group-all set-abstraction input rows in one pass (modules/pointset_abstraction.py:56, positions FIRST):
    out = [bf16(xyz) | feats | zeros] as bf16 [R, Cpad]; the positions carry no gradient."""

    @staticmethod
    def forward(ctx, xyz_rows, feats_rows, cpad):
        R, C = feats_rows.shape
        out = torch.empty(R, cpad, dtype=torch.bfloat16, device=feats_rows.device)
        with torch.cuda.device(feats_rows.device):
            _l.check(_l.lib().cpfn_concat_pos_feats_bf16(_ptr(xyz_rows.contiguous().float()), _ptr(feats_rows.contiguous()),
                                                         R, C, cpad, _ptr(out), _stream()), "cpfn_concat_pos_feats_bf16")
            _l.add_bytes("cpfn_concat_pos_feats_bf16", 12 * R + 2 * R * C + 2 * R * cpad)
        ctx.C = C
        return out

    @staticmethod
    def backward(ctx, g):
        return None, g[:, 3:3 + ctx.C], None


class ConcatInterp(torch.autograd.Function):
    """[ skip [B,N,C1] | three-NN interpolation of feats [B,M,C2] ] -> bf16 [B,N,C1+C2] in one launch (idx = w = None and
    M = 1: the global feature vector broadcast to every point).  Adjoint: the skip part is a view of the incoming gradient,
    the interpolated part goes through the inverse index (or the LDS scatter) straight from the gradient's column block —
    no slice copy —, the broadcast part is a column sum (cpfn_colsum_rows_bf16)."""

    @staticmethod
    def forward(ctx, skip, feats, idx, w, inv_off=None, inv_ent=None, join=None, top_ride=None):
        """top_ride (broadcast form): the fused_mlp.TopRide of the pooled stack that produced `feats` — this node's backward then
        also leaves BatchNorm-backward pass 1 of that stack's last layer (cpfn_colsum_rows_pass1_bf16)."""
        B, N, C1 = skip.shape
        M, C2 = feats.shape[1], feats.shape[2]
        sk, f = skip.contiguous(), feats.contiguous()
        ctx.join = None
        if (join is not None and SKIP_JOIN and join.src is not None and join.src in (SkipJoin.key(sk), SkipJoin.flat_key(sk))
                and (C1 + C2) % 8 == 0):
            join.armed = True
            ctx.join = join
        out = torch.empty(B, N, C1 + C2, dtype=torch.bfloat16, device=f.device)
        with torch.cuda.device(f.device):
            _l.check(_l.lib().cpfn_concat_interp_bf16(_ptr(sk), C1, _ptr(f), _ptr(idx), _ptr(w), B, M, N, C2, _ptr(out), _stream()),
                     "cpfn_concat_interp_bf16")
        _l.add_bytes("cpfn_concat_interp_bf16", 2 * B * N * C1 + 2 * B * M * C2 + (24 * B * N if idx is not None else 0)
                     + 2 * B * N * (C1 + C2))
        ctx.save_for_backward(idx, w)
        ctx.inv = None if inv_off is None else (inv_off, inv_ent)
        ctx.dims = (B, M, N, C1, C2)
        o_ = None if top_ride is None else top_ride.offer
        ctx.top_ride = top_ride if (o_ is not None and idx is None and o_[0] == f.data_ptr() and o_[1] == B and o_[2] == C2) else None
        return out

    @staticmethod
    def backward(ctx, g):
        idx, w = ctx.saved_tensors
        B, M, N, C1, C2 = ctx.dims
        g = g.contiguous().to(torch.bfloat16)
        g_skip, g_int = g[:, :, :C1], g[:, :, C1:]            # views: the consumers take the row stride
        j = ctx.join
        if j is not None and j.armed and ctx.needs_input_grad[0]:
            # the skip tensor's other consumer (an ancestor: its backward node runs later) adds this gradient in its own launch
            j.addend = (g, C1 + C2)
            torch.autograd.Variable._execution_engine.queue_callback(j._unclaimed)
            g_skip = None
        if idx is None:
            gf = torch.empty(B, 1, C2, dtype=torch.bfloat16, device=g.device)
            tr = ctx.top_ride
            with torch.cuda.device(g.device):
                if tr is not None and tr.offer is not None:
                    _, _, _, yarg, sc_, sh_ = tr.offer
                    part = torch.empty(B, 2, C2, dtype=torch.float32, device=g.device)
                    _l.check(_l.lib().cpfn_colsum_rows_pass1_bf16(_ptr(g_int), C1 + C2, B, N, C2, _ptr(gf), _ptr(yarg), _ptr(sc_),
                                                                  _ptr(sh_), _ptr(part), _stream()), "cpfn_colsum_rows_pass1_bf16")
                    tr.result = (gf.data_ptr(), gf._version, part, B)
                else:
                    _l.check(_l.lib().cpfn_colsum_rows_bf16(_ptr(g_int), C1 + C2, B, N, C2, _ptr(gf), _stream()), "cpfn_colsum_rows_bf16")
            _l.add_bytes("cpfn_colsum_rows_bf16", 2 * B * N * C2 + 2 * B * C2)
        elif ctx.inv is not None:
            gf = _csr_sum_bf16(g_int, C1 + C2, ctx.inv, w, 3, B, N, M, C2)
        else:
            gf = _scatter_bf16(g_int, C1 + C2, idx, w, 3, B, N, M, C2).to(torch.bfloat16)
        return g_skip, gf, None, None, None, None, None, None


def concat_interp_ok(skip, feats, idx):
    """The one-launch form needs bf16 rows on the GPU, channel counts in multiples of 8, and (interpolation) <= 1024 coarse
    points for the LDS scatter fallback / (broadcast) exactly one."""
    return (skip is not None and feats.is_cuda and skip.dtype == feats.dtype == torch.bfloat16 and skip.shape[2] % 8 == 0
            and feats.shape[2] % 8 == 0 and ((idx is None and feats.shape[1] == 1) or (idx is not None and feats.shape[1] <= 1024)))


def concat_interp(skip, feats, idx=None, w=None, inv=None, join=None, top_ride=None):
    if inv is not None:
        return ConcatInterp.apply(skip, feats, idx, w, inv[0], inv[1], join, top_ride)
    return ConcatInterp.apply(skip, feats, idx, w, None, None, join, top_ride)


def interp_rows(feats, idx, w, inv=None):
    if feats.dtype == torch.bfloat16 and feats.is_cuda and feats.shape[2] % 8 == 0 and feats.shape[1] <= 1024:
        if inv is not None:
            return InterpRowsBf16.apply(feats, idx, w, inv[0], inv[1])
        return InterpRowsBf16.apply(feats, idx, w)
    return InterpRows.apply(feats, idx, w)


gather_rows = GatherRows.apply
