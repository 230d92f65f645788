"""Feature-propagation level of PointNet++ (drop-in for the reference module of the same
name, modules/pointset_feature_propagation.py:6-52): 3-NN inverse-distance interpolation
of the coarse features onto the dense points, skip concatenation, shared MLP.

Same constructor, parameter names (`mlp_convs.i`, `mlp_bns.i`) and
`forward(pos1, pos2, feats1, feats2)` contract; `forward_rows` is the native entry.
"""

import torch
import torch.nn as nn

from .... import autograd_ops, mlp, ops


class PointsetFeaturePropagation(nn.Module):
    def __init__(self, dim_feats, mlp):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        c_in = dim_feats
        for c_out in mlp:
            self.mlp_convs.append(nn.Conv1d(c_in, c_out, 1))
            self.mlp_bns.append(nn.BatchNorm1d(c_out))
            c_in = c_out

    @staticmethod
    def compute_geometry(xyz1, xyz2, cuda_route=False, need_inverse=True):
        """3-NN indices and inverse-distance weights (coordinates only; prefetchable).
        cuda_route: what the reference's `fast=True` gives — direct distances and their SQUARE ROOTS
        (geometry_utils.py:184), so the weights are 1/(d + 1e-8) instead of the CPU route's 1/(d² + 1e-8)."""
        # CPU route: squared distances; the weights 1/(d+1e-8), normalised (ref :40-42), leave the same launch
        _, nn_idx, nn_w = ops.three_nn_weights(xyz1, xyz2, cuda_route=cuda_route, sqrt=cuda_route)
        out = {"nn_idx": nn_idx, "nn_w": nn_w}
        if need_inverse and xyz2.shape[1] <= 2048:                          # (evaluation never runs the adjoint: one
            out["inv"] = ops.csr_build(nn_idx, xyz2.shape[1])               #  393216-entry sort per 131072-point cloud, 3.2 ms)
        #                                                                     for the atomic-free interpolation adjoint
        return out

    def forward_rows(self, xyz1, xyz2, feats1, feats2, geom=None, cuda_route=False, tail=None, join=None, top_ride=None):
        """xyz1 [B,N,3] dense, xyz2 [B,S,3] coarse or None, feats1 [B,N,D1] or None,
        feats2 [B,S,D2] -> [B,N,D'].
        tail = (convs, bns, dropout[, handover]) (bf16 HIP path only): more (conv, bn, relu) layers run as part of the SAME fused stack
        — the caller's next per-point layers (GlobalSPFN's fc1 + bn1 + dropout): the stack's own last activation is then
        never materialised and its BatchNorm-backward reduction rides on the next layer's data gradient.
        join: an autograd_ops.SkipJoin shared with the earlier consumer of `feats1` (a set-abstraction level's grouping).
        top_ride: the fused_mlp.TopRide of the pooled stack that produced `feats2` (the broadcast form: sa3's global vector)."""
        B, N, _ = xyz1.shape
        aux = {}
        if xyz2 is not None and geom is None:
            geom = self.compute_geometry(xyz1, xyz2, cuda_route, need_inverse=self.training and torch.is_grad_enabled())
        idx = None if xyz2 is None else geom["nn_idx"]
        if feats1 is not None and autograd_ops.concat_interp_ok(feats1, feats2, idx):
            # bf16 HIP path: [feats1 | interpolation (or the broadcast global vector)] written by ONE launch
            x = autograd_ops.concat_interp(feats1, feats2, idx, None if xyz2 is None else geom["nn_w"],
                                           None if xyz2 is None else geom.get("inv"), join, top_ride if xyz2 is None else None)
            aux = {} if xyz2 is None else geom
        else:
            if xyz2 is None:
                interp = feats2.expand(B, N, feats2.shape[2])                  # broadcast the global vector (ref :33-34)
            else:
                interp = autograd_ops.interp_rows(feats2, geom["nn_idx"], geom["nn_w"], geom.get("inv"))
                aux = geom
            x = interp if feats1 is None else torch.cat([feats1.to(interp.dtype), interp], dim=2)   # feats1 FIRST (ref :46)
        convs, bns, dropout, handover = list(self.mlp_convs), list(self.mlp_bns), None, None
        if tail is not None:
            convs, bns, dropout = convs + list(tail[0]), bns + list(tail[1]), tail[2]
            handover = tail[3] if len(tail) > 3 else None
        y = mlp.run_stack(x.reshape(B * N, -1), convs, bns, getattr(self, "compute_dtype", torch.float32), dropout=dropout,
                          handover=handover)
        return y.reshape(B, N, -1), aux

    def forward(self, pos1, pos2, feats1, feats2, fast=True):
        from .... import cuda_ops as _co
        t = lambda a: None if a is None else a.transpose(1, 2).contiguous()
        out, _ = self.forward_rows(t(pos1), t(pos2), t(feats1), t(feats2), cuda_route=bool(_co.CUDA_ROUTE and fast))
        return out.transpose(1, 2)
