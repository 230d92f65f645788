"""Drop-in for PointNet2/pointnet2_ops/modules/geometry_utils.py: same function names,
argument order and channel-major [B, C, N] tensor conventions.

Every function runs a HIP kernel from libcpfn_hip.so.  By default the kernels reproduce the
arithmetic of the reference's `fast=False` (CPU) route bit for bit whatever `fast` says
(BASELINE.json's parity target).  With the process-wide opt-in `cuda_ops.CUDA_ROUTE`
(CPFN_CUDA_ROUTE=1), `fast=True` selects what the reference's compiled CUDA ops return instead:
FPS from index 0 skipping near-origin points, direct-distance ball query and 3-NN, and the
SQUARE ROOTS of the 3-NN distances (reference line 184).  CPU tensors raise "CPU not supported"
like the reference's native ops (cuda_ops/src/sampling.cpp:33-35); there is no fallback.
"""
import torch

from .. import cuda_ops
from .... import ops as _ops


def _rows(t_bcn):
    """[B,C,N] -> contiguous [B,N,C] (the layout the kernels take)."""
    return t_bcn.transpose(1, 2).contiguous()


def pairwise_squared_distance(src, dst):
    """src [B,C,N], dst [B,C,M] -> [B,N,M]: -2 srcᵀdst + ‖src‖² + ‖dst‖² in that order
    (reference lines 4-23).  Materialised only for API parity."""
    return _ops.pairwise_sqdist(_rows(src), _rows(dst))


def select_point_subset(points, idx):
    """points [B,C,N], idx [B,*] (long or int) -> [B,C,*]   (reference lines 26-44)."""
    lead = idx.shape[1:]
    flat = idx.reshape(idx.shape[0], -1).to(torch.int32).contiguous()
    return _GroupPoints.apply(points.contiguous(), flat).reshape(*points.shape[:2], *lead)


class _GroupPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        ctx.save_for_backward(idx)
        ctx.n = points.shape[2]
        return cuda_ops.gather_points(points, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return cuda_ops.gather_points_grad(g.contiguous(), idx, ctx.n), None


def farthest_point_sample(point_pos, num_point, fast=True, start_idx=None):
    """point_pos [B,3,N] -> indices [B,num_point] (long).

    Like the reference's CPU route (lines 88-101) the first sample is drawn with
    `torch.randint(0, N, (B,))` from the default CPU generator unless `start_idx`
    is given, so `torch.manual_seed(s)` selects the same points as the reference."""
    if point_pos.shape[1] != 3:
        raise ValueError('Points must have exactly three position dimensions when using the fast method.')
    B, _, N = point_pos.shape
    if cuda_ops.CUDA_ROUTE and fast:
        return cuda_ops.farthest_point_sampling(_rows(point_pos), num_point, start_idx=start_idx,
                                                cuda_compat=True).to(dtype=torch.long)
    if start_idx is None:
        start_idx = torch.randint(0, N, (B,), dtype=torch.long)
    return cuda_ops.farthest_point_sampling(_rows(point_pos), num_point, start_idx=start_idx,
                                            cuda_compat=False).to(dtype=torch.long)


def ball_query(radius, num_samples, point_pos, query_pos, fast=True):
    """point_pos [B,3,N], query_pos [B,3,S] -> [B,S,num_samples] (long)   (reference lines 133-161)."""
    if point_pos.shape[1] != 3:
        raise ValueError('Points must have exactly three position dimensions when using the fast method.')
    return cuda_ops.ball_query(_rows(query_pos), _rows(point_pos), radius, num_samples,
                               cuda_compat=bool(cuda_ops.CUDA_ROUTE and fast)).to(dtype=torch.long)


def three_nn(point_pos, query_pos, fast=True):
    """point_pos [B,3,N] (known), query_pos [B,3,S] -> (squared dists [B,S,3], idx [B,S,3] long).
    Squared distances, as the reference's CPU route returns (lines 212-215); on the opt-in CUDA route their
    square roots, as `_FastThreeNN` returns (line 184)."""
    if point_pos.shape[1] != 3:
        raise ValueError('Points must have exactly three position dimensions when using the fast method.')
    cr = bool(cuda_ops.CUDA_ROUTE and fast)
    d, i = _ops.three_nn(_rows(query_pos), _rows(point_pos), cuda_route=cr, sqrt=cr)
    return d, i.to(dtype=torch.long)


class _ThreeWeightedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, idx, weight):
        ctx.save_for_backward(idx, weight)
        ctx.m = feats.shape[2]
        return cuda_ops.three_weighted_sum(feats, idx, weight)

    @staticmethod
    def backward(ctx, g):
        idx, weight = ctx.saved_tensors
        return cuda_ops.three_weighted_sum_grad(g.contiguous(), idx, weight, ctx.m), None, None


def three_weighted_sum(point_feats, indices, weights, fast=True):
    """point_feats [B,C,N], indices/weights [B,S,3] -> [B,C,S]   (reference lines 267-283)."""
    return _ThreeWeightedSum.apply(point_feats.contiguous(), indices.to(torch.int32).contiguous(),
                                   weights.contiguous())
