"""The nine functions the reference binds as its `cuda_ops` extension
(PointNet2/pointnet2_ops/cuda_ops/src/bindings.cpp:6-19), re-created on top of
libcpfn_hip.so with the same names, argument order, dtypes (fp32 features, int32
indices), output shapes and error behaviour (bad arguments -> RuntimeError; CPU
tensors -> "CPU not supported").

Two deliberate differences, both so that results equal the reference's *CPU* route
(the parity target, BASELINE.json): `farthest_point_sampling` takes an optional
`start_idx` (the CPU route starts from a random index, the CUDA kernel from 0) and
does not skip near-origin points unless `cuda_compat=True`; `three_nn` and
`ball_query` use the expanded ‖q‖²+‖p‖²−2q·p distance the CPU route uses unless
`cuda_compat=True` (then: the CUDA kernels' direct (q−p)² distance, `d2 < r*r`).
`cuda_compat` defaults to the process-wide switch `cpfn_amd.cuda_ops.CUDA_ROUTE`
(environment CPFN_CUDA_ROUTE=1), off unless asked for.
"""
import os

import torch

from . import ops

# Process-wide default of `cuda_compat`: reproduce what the reference's compiled CUDA extension returns
# (`fast=True`) instead of its CPU route.  Checkpoints trained by the reference on a GPU saw these semantics.
CUDA_ROUTE = os.environ.get("CPFN_CUDA_ROUTE", "0") == "1"


def gather_points(points, idx):
    """points [b,c,n] f32, idx [b,m] i32 -> [b,c,m]   (sampling.cpp:14-39)."""
    return ops.group_fwd(points, idx)


def gather_points_grad(grad_out, idx, n):
    """grad_out [b,c,m], idx [b,m] -> [b,c,n]   (sampling.cpp:41-63)."""
    return ops.group_bwd(grad_out, idx, n)


def farthest_point_sampling(points, nsamples, start_idx=None, cuda_compat=None):
    """points [b,n,3] f32 -> [b,nsamples] i32   (sampling.cpp:64-86)."""
    cuda_compat = CUDA_ROUTE if cuda_compat is None else cuda_compat
    if start_idx is not None:
        start_idx = start_idx.to(device=points.device, dtype=torch.int32).contiguous()
    return ops.fps(points, nsamples, start_idx, skip_near_origin=cuda_compat)


def three_nn(unknowns, knows, cuda_compat=None):
    """unknown [b,n,3], known [b,m,3] -> [dist2 [b,n,3] f32, idx [b,n,3] i32]   (interpolate.cpp).
    Squared distances on both routes, as the bound function returns them (the sqrt of the CUDA route is
    taken by the Python wrapper, modules/geometry_utils.py:184)."""
    cuda_compat = CUDA_ROUTE if cuda_compat is None else cuda_compat
    d, i = ops.three_nn(unknowns, knows, cuda_route=cuda_compat)
    return [d, i]


def three_weighted_sum(points, idx, weight):
    """points [b,c,m], idx [b,n,3] i32, weight [b,n,3] -> [b,c,n]   (interpolate.cpp)."""
    return ops.three_interp_fwd(points, idx, weight)


def three_weighted_sum_grad(grad_out, idx, weight, m):
    """grad_out [b,c,n] -> [b,c,m]   (interpolate.cpp)."""
    return ops.three_interp_bwd(grad_out, idx, weight, m)


def ball_query(new_xyz, xyz, radius, nsample, cuda_compat=None):
    """new_xyz [b,m,3], xyz [b,n,3] -> [b,m,nsample] i32   (ball_query.cpp)."""
    cuda_compat = CUDA_ROUTE if cuda_compat is None else cuda_compat
    return ops.ball_query(new_xyz, xyz, radius, nsample, cuda_route=cuda_compat)


def group_points(points, idx):
    """points [b,c,n], idx [b,np,ns] i32 -> [b,c,np,ns]   (group_points.cpp)."""
    return ops.group_fwd(points, idx)


def group_points_grad(grad_out, idx, n):
    """grad_out [b,c,np,ns] -> [b,c,n]   (group_points.cpp)."""
    return ops.group_bwd(grad_out, idx, n)
