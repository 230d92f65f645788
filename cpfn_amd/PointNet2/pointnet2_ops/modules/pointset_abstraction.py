"""Set-abstraction level of PointNet++ (drop-in for the reference module of the same
name, modules/pointset_abstraction.py:7-77): FPS -> ball query -> neighbourhood gather
(centred on the sampled point) -> shared MLP -> max over the neighbourhood.

Same constructor signature, parameter names (`conv_blocks.i.j`, `bn_blocks.i.j`) and
`forward(pos [B,C,N], feats [B,D,N]) -> (new_pos [B,C,S], new_feats [B,D',S])` contract.
Internally everything is points-major: `forward_rows` is what PointNet2 calls.
"""
from collections.abc import Sequence

import torch
import torch.nn as nn

from .... import autograd_ops, mlp, ops


class PointsetAbstraction(nn.Module):
    def __init__(self, num_points, dim_pos, dim_feats, radius_list, num_samples_list, mlp_list, group_all=False):
        super().__init__()
        seq = lambda v: list(v) if isinstance(v, Sequence) else [v]
        self.num_points = num_points
        self.group_all = group_all
        self.has_feats = dim_feats > 0
        self.radius_list = seq(radius_list)
        self.num_samples_list = seq(num_samples_list)
        self.mlp_list = list(mlp_list) if isinstance(mlp_list[0], Sequence) else [list(mlp_list)]
        if not (len(self.radius_list) == len(self.num_samples_list) == len(self.mlp_list)):
            raise ValueError('Radius, number of samples and mlps lists must have the same number of entries.')
        self.conv_blocks = nn.ModuleList()
        self.bn_blocks = nn.ModuleList()
        for widths in self.mlp_list:
            convs, bns, c_in = nn.ModuleList(), nn.ModuleList(), dim_pos + dim_feats
            for c_out in widths:
                convs.append(nn.Conv2d(c_in, c_out, 1))
                bns.append(nn.BatchNorm2d(c_out))
                c_in = c_out
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)

    # ---------------------------------------------------------------- native layout
    def compute_geometry(self, xyz, start_idx=None, cuda_route=False, need_inverse=True):
        """Everything of this level that depends only on coordinates (no weights, no gradients):
        FPS indices, sampled centres, ball-query neighbours and the centred neighbour coordinates.
        Can be run ahead of time on a side stream (PointNet2.compute_geometry).
        cuda_route: the semantics of the reference's compiled CUDA ops (`fast=True`): FPS from index 0 skipping
        near-origin points (sampling_gpu.cu:76-91), direct-distance ball query (ball_query_gpu.cu:21-31).
        = sample() + neighbours(); a caller may interleave the two halves of several levels (PointNet2.compute_geometry
        runs both levels' samplings first)."""
        return self.neighbours(xyz, self.sample(xyz, start_idx, cuda_route), cuda_route, need_inverse)

    def sample(self, xyz, start_idx=None, cuda_route=False):
        """FPS + the sampled centres: (fps_idx [B,S] i32, new_xyz [B,S,3])."""
        B, N, _ = xyz.shape
        if cuda_route:          # always from index 0 (sampling_gpu.cu:76-77); a caller's start indices do not apply
            return ops.fps_centres(xyz, self.num_points, None, skip_near_origin=True)
        if start_idx is None:   # the reference's CPU route draws the start here (geometry_utils.py:92)
            start_idx = torch.randint(0, N, (B,), dtype=torch.long)
        start_idx = start_idx.to(device=xyz.device, dtype=torch.int32)
        return ops.fps_centres(xyz, self.num_points, start_idx)      # (indices + the centres themselves: one launch)

    def neighbours(self, xyz, sampled, cuda_route=False, need_inverse=True):
        """Ball-query neighbours, centred neighbour coordinates and the inverse index for sample()'s centres."""
        B, N, _ = xyz.shape
        sel, new_xyz = sampled
        scales = []
        for r, k in zip(self.radius_list, self.num_samples_list):
            # neighbours [B,S,K] i32 and their centred coordinates rel [B,S,K,3] fp32 (one launch beside a training step)
            scales.append(ops.ball_query_rel(new_xyz, xyz, r, k, cuda_route=cuda_route))
        out = {"fps_idx": sel, "new_xyz": new_xyz, "scales": scales}
        if need_inverse and self.has_feats and N <= 2048 and len(scales) == 1:
            # inverse of the neighbour index: atomic-free, deterministic adjoint of the feature gather
            out["inv"] = ops.csr_build(scales[0][0], N)
        return out

    def forward_rows(self, xyz, feats, start_idx=None, geom=None, cuda_route=False, join=None, join_out=None, top_ride=None):
        """xyz [B,N,3] f32, feats [B,N,D] or None -> (new_xyz [B,S,3] | None, new_feats [B,S,D'], aux).
        join: an autograd_ops.SkipJoin shared with the OTHER consumer of `feats` (a later feature-propagation level's skip).
        join_out: the same for this level's OUTPUT, where it has two consumers (single-scale levels only).
        top_ride: a fused_mlp.TopRide shared with the consumer whose backward node produces the gradient of this level's output."""
        B, N, _ = xyz.shape
        aux = {}
        cd = getattr(self, "compute_dtype", torch.float32)
        if self.group_all:
            new_xyz = None
            if feats is not None and cd == torch.bfloat16 and feats.is_cuda and feats.dtype == torch.bfloat16:
                # positions + features + zero padding to the GEMM's K, one kernel (pos FIRST, ref :56)
                D = feats.shape[2]
                g = autograd_ops.ConcatPosFeats.apply(xyz.reshape(B * N, 3), feats.reshape(B * N, D), (D + 3 + 63) // 64 * 64)
            else:
                g = xyz if feats is None else torch.cat([xyz.to(feats.dtype), feats], dim=2)  # pos FIRST (ref :56)
            groups = [(g.reshape(B * N, -1), None, 1, N)] * len(self.mlp_list)
        else:
            if geom is None:
                geom = self.compute_geometry(xyz, start_idx, cuda_route, need_inverse=self.training and torch.is_grad_enabled())
            new_xyz = geom["new_xyz"]
            aux["fps_idx"] = geom["fps_idx"]
            groups = []
            S = self.num_points
            for (nbr, rel), k, convs_of_scale in zip(geom["scales"], self.num_samples_list, self.conv_blocks):
                aux["ball_idx"] = nbr
                if feats is not None:
                    D = feats.shape[2]
                    if cd == torch.bfloat16 and feats.is_cuda and feats.dtype == torch.bfloat16 and D % 8 == 0 and N <= 1024:
                        # gather + centred coordinates + zero padding to the GEMM's K, one kernel (feats FIRST, ref :66)
                        inv = geom.get("inv") or (None, None)
                        from .... import fused_mlp
                        if fused_mlp.xyz_tail_ok(B * S * k, D, convs_of_scale[0].weight.shape[0]):
                            # long layers: the coordinates do not become three bf16 columns of a K = 192 operand — they
                            # reach the first layer as its fp32 xyz tail (cpfn_mlp_gemm_xyz) and the rows are the gather alone
                            # (round 6: ... and not even that — the first layer's GEMM and its backward kernel read the rows out of
                            #  `feats` through `nbr` while loading their operand; `x` only carries the autograd edge)
                            lazy = fused_mlp.gather_on_load_ok(B, N, S * k, D)
                            x = autograd_ops.GroupConcat.apply(feats, None, nbr, D, inv[0], inv[1], join, lazy)
                            groups.append((x, None, S, k, rel.reshape(B * S * k, 3),
                                           (feats.contiguous().reshape(B * N, D), nbr.reshape(-1), S * k, N) if lazy else None))
                            continue
                        x = autograd_ops.GroupConcat.apply(feats, rel, nbr, (D + 3 + 63) // 64 * 64, inv[0], inv[1], join)
                    else:
                        gf = autograd_ops.gather_rows(feats, nbr)                         # [B,S,K,D]
                        x = torch.cat([gf, rel.to(gf.dtype)], dim=3).reshape(B * S * k, -1)
                    groups.append((x, None, S, k))
                else:
                    groups.append((None, rel.reshape(B * S * k, 3), S, k))
        outs = []
        for grp, convs, bns in zip(groups, self.conv_blocks, self.bn_blocks):
            x, xyz_rows, S, k = grp[:4]
            y = mlp.run_stack(x, convs, bns, cd, pool_k=k, xyz_rows=xyz_rows, xyz_tail=grp[4] if len(grp) > 4 else None,
                              gather=grp[5] if len(grp) > 5 else None,  # max over the K neighbours (ref :74)
                              join_out=join_out if len(self.conv_blocks) == 1 else None,
                              top_ride=top_ride if len(self.conv_blocks) == 1 else None)
            outs.append(y.reshape(B, S, -1))
        return new_xyz, torch.cat(outs, dim=2) if len(outs) > 1 else outs[0], aux

    # ---------------------------------------------------------------- reference layout
    def forward(self, pos, feats, fast=True):
        """`fast` selects the CUDA-route semantics only when the process-wide switch cuda_ops.CUDA_ROUTE is on;
        by default both values give the reference's CPU-route results."""
        from .... import cuda_ops as _co
        xyz = pos.transpose(1, 2).contiguous()
        f = None if feats is None else feats.transpose(1, 2).contiguous()
        new_xyz, new_feats, _ = self.forward_rows(xyz, f, cuda_route=bool(_co.CUDA_ROUTE and fast))
        return (None if new_xyz is None else new_xyz.transpose(1, 2)), new_feats.transpose(1, 2)
