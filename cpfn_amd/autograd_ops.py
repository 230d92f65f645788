"""torch.autograd wrappers around the points-major HIP data movers."""
import torch

from . import ops


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


gather_rows = GatherRows.apply
interp_rows = InterpRows.apply
