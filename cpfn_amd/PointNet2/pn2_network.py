"""PointNet++ encoder/decoder used by GlobalSPFN, LocalSPFN and PatchSelection — drop-in
for the reference's PointNet2/pn2_network.py (same constructor arguments, state_dict keys
and `forward(x [B,N,C]) -> [head_0 [B,N,o0], ..., l3_feats [B,1024,1], output_feat [B,128,N]]`).

Data stay points-major ([B, N, C] rows) from input to heads; the two channel-major
outputs are transposed views.  Quirk kept on purpose: dropout(p=0.5) is applied in every
mode, as in the reference (pn2_network.py:63 calls F.dropout with its default
training=True); set `self.dropout_p = 0.0` to neutralise it for parity tests.
"""
import os

import torch
import torch.nn.functional as F

from .. import autograd_ops, mlp
from .pointnet2_ops.modules.pointset_abstraction import PointsetAbstraction
from .pointnet2_ops.modules.pointset_feature_propagation import PointsetFeaturePropagation


# CPFN_FUSED_DROPOUT=0: F.dropout as separate PyTorch kernels (mask tensor) instead of the mask generated inside the
# BatchNorm apply / backward kernels
FUSED_DROPOUT = os.environ.get("CPFN_FUSED_DROPOUT", "1") != "0"


class PointNet2(torch.nn.Module):
    def __init__(self, dim_input=3, dim_pos=3, output_sizes=[16], use_glob_features=False,
                 use_loc_features=False, features_extractor=False):
        super().__init__()
        self.dim_pos = dim_pos
        self.use_glob_features = use_glob_features
        self.use_loc_features = use_loc_features
        self.features_extractor = features_extractor
        self.dropout_p = 0.5
        self._dropout_counter = None          # device step counter of the fused dropout (bf16 HIP path)
        self._dropout_base = 0
        extra = (1024 if use_glob_features else 0) + (128 if use_loc_features else 0)
        self.sa1 = PointsetAbstraction(512, dim_pos, dim_input - dim_pos, [0.2], [64], [[64, 64, 128]])
        self.sa2 = PointsetAbstraction(128, dim_pos, 128, [0.4], [64], [[128, 128, 256]])
        self.sa3 = PointsetAbstraction(None, dim_pos, 256, None, None, [256, 512, 1024], group_all=True)
        self.sfp1 = PointsetFeaturePropagation(1024 + extra + 256, [256, 256])
        self.sfp2 = PointsetFeaturePropagation(256 + 128, [256, 128])
        self.sfp3 = PointsetFeaturePropagation(128 + dim_input - dim_pos, [128, 128, 128])
        self.fc1 = torch.nn.Conv1d(128, 128, 1)
        if not features_extractor:
            self.bn1 = torch.nn.BatchNorm1d(128)
            self.fc2 = torch.nn.ModuleList(torch.nn.Conv1d(128, o, 1) for o in output_sizes)

    def packed_parameter_groups(self):
        """Parameters whose gradients the fused path produces as ONE packed block each: the fc2 heads' weights ([sum o_i, 128], in
        head order) and their biases.  training.FlatGradBucket keeps each group adjacent so that the block is written in place."""
        if self.features_extractor:
            return []
        return [[h.weight for h in self.fc2], [h.bias for h in self.fc2]]

    def set_compute_dtype(self, dtype):
        """GEMM operand type of every per-point MLP (torch.bfloat16 on MI355X; fp32 for parity)."""
        self.compute_dtype = dtype
        for m in self.modules():
            m.compute_dtype = dtype
        return self

    @torch.no_grad()
    def compute_geometry(self, x, fps_start=None, cuda_route=None):
        """All index tensors of a forward pass (FPS / ball query of sa1, sa2; 3-NN of sfp2, sfp3).
        They depend on the input coordinates only, so a trainer can compute them for batch t+1 on a
        side stream while batch t is still in its backward pass (training.SPFNTrainer.prefetch)."""
        from .. import cuda_ops as _co
        cr = bool(_co.CUDA_ROUTE) if cuda_route is None else bool(cuda_route)
        xyz = x[:, :, :self.dim_pos].contiguous().float()
        s1, s2 = fps_start if fps_start is not None else (None, None)
        inv = self.training           # the inverse indices serve the backward adjoints only
        # both levels' FPS chains first (sa2's needs sa1's centres only), then every chip-wide neighbourhood query (ball
        # queries, 3-NN), the inverse-index builds (one workgroup per cloud, hundreds of microseconds of mostly waiting
        # lanes) last: beside a training step the wide kernels then land in the loss section, where the chip is nearly
        # idle, instead of beside the GEMMs of the forward / backward pass (same kernels, same results; -4 and -9 us per
        # step against level-by-level order)
        from ..ops import csr_build as _csr
        a1 = self.sa1.sample(xyz, s1, cr)
        a2 = self.sa2.sample(a1[1], s2, cr)
        g1 = self.sa1.neighbours(xyz, a1, cr, False)
        g2 = self.sa2.neighbours(a1[1], a2, cr, False)
        f2 = self.sfp2.compute_geometry(a1[1], a2[1], cr, False)
        f3 = self.sfp3.compute_geometry(xyz, a1[1], cr, False)
        if inv:
            for g, sa, n_src in ((g1, self.sa1, xyz.shape[1]), (g2, self.sa2, a1[1].shape[1])):
                if sa.has_feats and n_src <= 2048 and len(g["scales"]) == 1:
                    g["inv"] = _csr(g["scales"][0][0], n_src)
            for f, m in ((f2, a2[1].shape[1]), (f3, a1[1].shape[1])):
                if m <= 2048:
                    f["inv"] = _csr(f["nn_idx"], m)
        return {"sa1": g1, "sa2": g2, "sfp2": f2, "sfp3": f3}

    def forward(self, x, glob_features=None, loc_features=None, fast=True, fps_start=None, geometry=None):
        """`fast`: the reference switches between its compiled CUDA ops (default) and its PyTorch CPU route, which
        give DIFFERENT results (FPS start / near-origin skip, ball-query distance, sqrt vs squared 3-NN distances in
        the interpolation weights).  Here the CPU route's results are the default for both values (BASELINE.json's
        parity target); with the process-wide opt-in `cuda_ops.CUDA_ROUTE` (CPFN_CUDA_ROUTE=1) `fast=True` selects
        the CUDA route's semantics — what a checkpoint trained by the reference on a GPU saw."""
        from .. import cuda_ops as _co, fused_mlp
        bf16 = getattr(self, "compute_dtype", torch.float32) == torch.bfloat16 and x.is_cuda
        # An evaluation forward under no_grad (evaluation_globalSPFN.py:84-85; evaluation_localSPFN.py:95 calls the module with
        # gradients ENABLED and therefore stays on eager launches: its outputs must carry autograd history) is ~100 launches
        # whose host cost exceeds their run time: from the SECOND time a shape is seen it is captured once and replayed as one
        # hipGraph (inference.GraphedForward; bit-identical, same CPU-generator draws for the FPS starts; outputs are copies).
        # The first sighting runs eager launches: a caller whose shapes never repeat (full-resolution clouds with a per-shape N)
        # then pays one eager forward per call, not a warm-up + a capture + a replay and a churned graph pool.
        # `model.auto_graph = False` opts out.
        if (bf16 and not self.training and not torch.is_grad_enabled() and geometry is None and getattr(self, "auto_graph", True)
                and (fast or not _co.CUDA_ROUTE) and not self.__dict__.get("_graph_busy")
                and not torch.cuda.is_current_stream_capturing()):
            seen = self.__dict__.setdefault("_auto_seen", {})
            skey = (tuple(x.shape), None if glob_features is None else tuple(glob_features.shape),
                    None if loc_features is None else tuple(loc_features.shape))
            if skey in seen:
                auto = self.__dict__.get("_auto_graph")
                if auto is None:
                    from ..inference import GraphedForward
                    auto = self.__dict__["_auto_graph"] = GraphedForward(self, max_shapes=4, clone_outputs=True, weak=True)
                return auto(x, glob_features=glob_features, loc_features=loc_features, fps_start=fps_start)
            if len(seen) >= 64:
                seen.clear()
            seen[skey] = True
        if bf16:
            # one multi-tensor fp32 -> bf16 conversion; when sa1's input is coordinates only, its fp32 first layer is the
            # forward pass's first launch and the conversion rides on it (cpfn_smallk_fwd_cast)
            first_is_xyz = (not self.sa1.has_feats and not self.sa1.group_all and
                            getattr(self.sa1, "compute_dtype", torch.float32) == torch.bfloat16)
            fused_mlp.refresh_weight_panels(self.parameters(), defer=first_is_xyz)
        try:
            # (seam_pass: the accumulators of this pass's BatchNorm seams — statistics that reach their consumer without a finalize
            #  launch, csrc/seam.h — are zeroed by ONE fill up front)
            with fused_mlp.deferred_bn_counters(), fused_mlp.seam_pass(x.device, bf16 and self.training):
                return self._forward(x, glob_features, loc_features, fps_start, geometry, bool(_co.CUDA_ROUTE and fast))
        finally:
            fused_mlp.flush_pending_cast()       # (never leave a refresh queued behind an exception or an unusual model)

    def _fused_dropout(self, device):
        """(p, step counter, base seed) of the dropout fused into fc1's BatchNorm apply (cpfn_amd/fused_mlp.py)."""
        if self._dropout_counter is None or self._dropout_counter.device != device:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("run one eager forward pass before capturing a graph (the dropout step counter "
                                   "must exist outside the captured region)")
            self._dropout_counter = torch.zeros(1, dtype=torch.int64, device=device)
            self._dropout_base = torch.initial_seed() ^ 0x5DEECE66D
        return (self.dropout_p, self._dropout_counter, self._dropout_base)

    def _forward(self, x, glob_features, loc_features, fps_start, geometry, cuda_route=False):
        """`fps_start` = optional (start_sa1 [B], start_sa2 [B]) FPS seeds; by default each SA
        level draws its own from the CPU generator like the reference's CPU route."""
        B, N, _ = x.shape
        xyz = x[:, :, :self.dim_pos].contiguous().float()
        feats0 = x[:, :, self.dim_pos:].contiguous() if x.shape[2] > self.dim_pos else None
        s1, s2 = fps_start if fps_start is not None else (None, None)
        gm = geometry if geometry is not None else {}
        cr = cuda_route
        l1_xyz, l1, self.aux_sa1 = self.sa1.forward_rows(xyz, feats0, s1, gm.get("sa1"), cr)
        # sa1's features have two consumers, sa2's grouping and sfp2's skip: their two gradients are added inside the grouping
        # adjoint's launch instead of by a framework add between the two backward nodes (autograd_ops.SkipJoin)
        join1 = autograd_ops.SkipJoin() if (l1.is_cuda and l1.dtype == torch.bfloat16 and torch.is_grad_enabled()) else None
        # ... and so have sa2's (sa3's input rows, sfp1's skip): that sum is formed by sa2's own backward, in its first launch
        join2 = autograd_ops.SkipJoin() if (join1 is not None and autograd_ops.OUTPUT_JOIN) else None
        l2_xyz, l2, self.aux_sa2 = self.sa2.forward_rows(l1_xyz, l1, s2, gm.get("sa2"), cr, join=join1, join_out=join2)
        # (sa3's global vector is read by sfp1's broadcast alone: pass 1 of sa3's last BatchNorm backward rides on that adjoint)
        ride3 = None
        if join1 is not None:
            from .. import fused_mlp
            ride3 = fused_mlp.TopRide()
        _, l3, _ = self.sa3.forward_rows(l2_xyz, l2, top_ride=ride3)       # [B,1,1024]
        if self.use_glob_features:
            l3 = torch.cat([l3, glob_features.unsqueeze(1).to(l3.dtype)], dim=2)
        if self.use_loc_features:
            l3 = torch.cat([l3, loc_features.unsqueeze(1).to(l3.dtype)], dim=2)
        l4, _ = self.sfp1.forward_rows(l2_xyz, None, l2, l3, join=join2, top_ride=ride3)
        l5, _ = self.sfp2.forward_rows(l1_xyz, l2_xyz, l1, l4, gm.get("sfp2"), cr, join=join1)
        cd = getattr(self, "compute_dtype", torch.float32)
        # bf16 HIP path: fc1 + bn1 + relu + dropout run as the last layer of sfp3's fused stack
        chain = (cd == torch.bfloat16 and x.is_cuda and self.dropout_p > 0.0 and FUSED_DROPOUT
                 and not self.features_extractor and getattr(self.sfp3, "compute_dtype", torch.float32) == torch.bfloat16)
        feat = None
        # side results between the backward nodes of THIS forward pass (fc1's stack <-> packed heads <-> loss section)
        ho = self.handover = None
        if cd == torch.bfloat16 and x.is_cuda:
            from .. import fused_mlp
            ho = self.handover = fused_mlp.HandOver()
        if chain:
            feat, self.aux_sfp3 = self.sfp3.forward_rows(xyz, l1_xyz, feats0, l5, gm.get("sfp3"), cr,
                                                         tail=([self.fc1], [self.bn1], self._fused_dropout(x.device), ho))
            feat = feat.reshape(B * N, -1)
            l6 = None
        else:
            l6, self.aux_sfp3 = self.sfp3.forward_rows(xyz, l1_xyz, feats0, l5, gm.get("sfp3"), cr)
        l3_out = l3.transpose(1, 2)                                         # [B,1024(+extra),1]
        if getattr(self, "return_point_features", True) or self.features_extractor:
            l3_out = l3_out.float()            # (a trainer that only consumes the heads skips this conversion too)
        if self.features_extractor:
            feat = mlp.conv_as_linear(l6.reshape(B * N, -1).to(cd), self.fc1).float()
            return l3_out, feat.reshape(B, N, -1).transpose(1, 2)
        if feat is not None:
            pass
        elif cd == torch.bfloat16 and l6.is_cuda and self.dropout_p > 0.0 and FUSED_DROPOUT:
            # fc1 + bn1 + relu + the always-on dropout (ref :60-63) with the mask generated inside the BN apply kernel
            # and regenerated in the backward passes (no mask tensor, no separate dropout kernels)
            feat = mlp.run_stack(l6.reshape(B * N, -1), [self.fc1], [self.bn1], cd, dropout=self._fused_dropout(l6.device),
                                 handover=ho)
        else:
            feat = mlp.run_stack(l6.reshape(B * N, -1), [self.fc1], [self.bn1], cd)       # fc1 + bn1 + relu (ref :60-62)
            feat = F.dropout(feat, p=self.dropout_p, training=True)                        # always on (ref :63)
        results = [r.reshape(B, N, -1) for r in mlp.heads(feat, self.fc2, cd, handover=ho)]
        self.heads_packed = None
        if cd == torch.bfloat16 and feat.is_cuda:
            from .. import fused_mlp
            self.heads_packed = fused_mlp.linear_heads.last_packed.reshape(B, N, -1)       # [B,N,3+4+K] fp32
        results.append(l3_out)
        # per-point features [B,128,N] fp32 like the reference's fifth output; a trainer that only consumes the heads
        # sets `return_point_features = False` and gets the bf16 rows as a view instead (saves a 100 MB conversion)
        if getattr(self, "return_point_features", True):
            results.append(feat.float().reshape(B, N, -1).transpose(1, 2))
        else:
            results.append(feat.reshape(B, N, -1).transpose(1, 2))
        return results
