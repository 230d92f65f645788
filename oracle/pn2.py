"""torch-CPU restatement of the reference's PointNet++ encoder/decoder and of one
GlobalSPFN training step (the "reference CPU path").

TEST INFRASTRUCTURE ONLY — see ``oracle/__init__.py``.  Pinned against the
reference by ``tests/golden/network_*.npz`` and ``tests/golden/step_*.npz``.

Functional style: the network is a function of a ``state`` dict carrying the
reference's state_dict keys (``sa1.conv_blocks.0.0.weight`` … ``fc2.2.bias``;
PointNet2/pn2_network.py:11-36), so the same tensors can be loaded into the
reference model, into this oracle and into the HIP-backed product model.
Index tensors (FPS, ball-query, 3-NN) come from the C oracle (bit-exact integer
work); everything differentiable is plain fp32 torch.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import geometry as og
from . import spfn as ospfn

BN_EPS = 1e-5


def _np_bn3(pos_bcn):
    """[B,3,N] torch -> [B,N,3] contiguous numpy."""
    return np.ascontiguousarray(pos_bcn.detach().transpose(1, 2).numpy())


def _gather(points, idx):
    """select_point_subset (modules/geometry_utils.py:26-44).
    points [B,C,N]; idx [B,S] or [B,S,K] long -> [B,C,S(,K)]."""
    B, C, N = points.shape
    flat = idx.reshape(B, 1, -1).expand(B, C, -1)
    return torch.gather(points, 2, flat).reshape(B, C, *idx.shape[1:])


def _mlp(state, conv_fmt, bn_fmt, n_layers, x, training, conv):
    """Shared 1x1-conv + BatchNorm(batch statistics) + ReLU stack
    (pointset_abstraction.py:70-73, pointset_feature_propagation.py:49-51)."""
    for j in range(n_layers):
        w, b = state[conv_fmt % j + ".weight"], state[conv_fmt % j + ".bias"]
        x = conv(x, w, b)
        rm, rv = state.get(bn_fmt % j + ".running_mean"), state.get(bn_fmt % j + ".running_var")
        x = F.batch_norm(x, None if training else rm, None if training else rv,
                         state[bn_fmt % j + ".weight"], state[bn_fmt % j + ".bias"],
                         training=training, eps=BN_EPS)
        x = F.relu(x)
    return x


def _count(state, fmt):
    j = 0
    while (fmt % j + ".weight") in state:
        j += 1
    return j


def set_abstraction(state, name, pos, feats, num_points, radius, num_samples, start,
                    training=True):
    """PointsetAbstraction.forward (modules/pointset_abstraction.py:38-77), one scale.
    pos [B,3,N]; feats [B,D,N] or None; `start` = FPS start indices [B]
    (the reference draws them with torch.randint, geometry_utils.py:92).
    num_points None -> group_all."""
    B, C, N = pos.shape
    conv_fmt, bn_fmt = name + ".conv_blocks.0.%d", name + ".bn_blocks.0.%d"
    if num_points is None:                                                  # :52-57 group_all
        g = pos.view(B, C, 1, N)
        if feats is not None:
            g = torch.cat([g, feats.view(B, -1, 1, N)], dim=1)              # pos FIRST
        new_pos, aux = None, {}
    else:
        xyz = _np_bn3(pos)
        sel = torch.from_numpy(og.farthest_point_sample(xyz, num_points, np.asarray(start)))
        new_pos = _gather(pos, sel)                                         # :50
        grp = torch.from_numpy(og.ball_query(radius, num_samples, xyz, _np_bn3(new_pos)))
        g = _gather(pos, grp) - new_pos.unsqueeze(-1)                       # :62-63
        if feats is not None:
            g = torch.cat([_gather(feats, grp), g], dim=1)                  # :66 feats FIRST
        aux = {"fps_idx": sel, "ball_idx": grp}
    g = _mlp(state, conv_fmt, bn_fmt, _count(state, conv_fmt), g, training,
             lambda x, w, b: F.conv2d(x, w, b))
    return new_pos, g.max(dim=3)[0], aux                                    # :74


def feature_propagation(state, name, pos1, pos2, feats1, feats2, training=True):
    """PointsetFeaturePropagation.forward (modules/pointset_feature_propagation.py:20-52)."""
    B, _, N = pos1.shape
    aux = {}
    if pos2 is None:
        interp = feats2.repeat(1, 1, N)                                     # :33-34
    else:
        d, i = og.three_nn(_np_bn3(pos1), _np_bn3(pos2))                    # :38 (squared dists)
        d, i = torch.from_numpy(d), torch.from_numpy(i)
        recip = 1.0 / (d + 1e-8)                                            # :40
        w = recip / recip.sum(dim=2, keepdim=True)                          # :41-42
        interp = (_gather(feats2, i) * w.unsqueeze(1)).sum(-1)              # :44
        aux = {"nn_idx": i, "nn_w": w}
    x = interp if feats1 is None else torch.cat([feats1, interp], dim=1)    # :45-48
    conv_fmt, bn_fmt = name + ".mlp_convs.%d", name + ".mlp_bns.%d"
    x = _mlp(state, conv_fmt, bn_fmt, _count(state, conv_fmt), x, training,
             lambda x, w, b: F.conv1d(x, w, b))
    return x, aux


def pointnet2_forward(state, x, fps_starts, training=True, dropout_mask=None, glob_features=None, loc_features=None):
    """PointNet2.forward (PointNet2/pn2_network.py:38-73) for dim_input == dim_pos == 3.
    x [B,N,3]; fps_starts = (start_sa1 [B], start_sa2 [B]).
    `dropout_mask` [B,128,N] multiplies the fc1 activations (the reference applies
    F.dropout(p=0.5) unconditionally, :63); None = dropout neutralised.
    glob_features [B,1024] / loc_features [B,128]: the use_glob_features / use_loc_features variant (:51-54) — the state's
    sfp1 is then 1024 / 128 channels wider (:22-27).  A state without `bn1.*` is the features_extractor variant (:31, :70-71):
    returns ([], l3_feats, fc1 output, aux).
    Returns ([heads...], l3_feats [B,1024(+extra),1], output_feat [B,128,N], aux)."""
    pos = x.transpose(2, 1)
    l1_pos, l1_f, a1 = set_abstraction(state, "sa1", pos, None, 512, 0.2, 64, fps_starts[0], training)
    l2_pos, l2_f, a2 = set_abstraction(state, "sa2", l1_pos, l1_f, 128, 0.4, 64, fps_starts[1], training)
    _, l3_f, _ = set_abstraction(state, "sa3", l2_pos, l2_f, None, None, None, None, training)
    if glob_features is not None:
        l3_f = torch.cat([l3_f, glob_features.unsqueeze(2)], dim=1)                     # :51-52
    if loc_features is not None:
        l3_f = torch.cat([l3_f, loc_features.unsqueeze(2)], dim=1)                      # :53-54
    l4, _ = feature_propagation(state, "sfp1", l2_pos, None, l2_f, l3_f, training)
    l5, a5 = feature_propagation(state, "sfp2", l1_pos, l2_pos, l1_f, l4, training)
    l6, a6 = feature_propagation(state, "sfp3", pos, l1_pos, None, l5, training)
    feat = F.conv1d(l6, state["fc1.weight"], state["fc1.bias"])                       # :60
    aux = {"sa1": a1, "sa2": a2, "sfp2": a5, "sfp3": a6}
    if "bn1.weight" not in state:                                                       # features_extractor (:70-71)
        return [], l3_f, feat, aux
    feat = F.relu(F.batch_norm(feat, None if training else state["bn1.running_mean"],
                               None if training else state["bn1.running_var"],
                               state["bn1.weight"], state["bn1.bias"], training=training, eps=BN_EPS))
    if dropout_mask is not None:
        feat = feat * dropout_mask                                                      # :63
    heads = []
    j = 0
    while "fc2.%d.weight" % j in state:
        heads.append(F.conv1d(feat, state["fc2.%d.weight" % j], state["fc2.%d.bias" % j]).transpose(1, 2))
        j += 1
    return heads, l3_f, feat, aux


def patch_selection_loss(state, points, labels, fps_starts, training=True):
    """Forward + loss of patch_selection_train_val_epoch (Utils/training_utils.py:62-68): cross-entropy of the two-class
    heat-map logits [B,N,2] against labels [B,N].  Returns (loss, logits)."""
    heads, _, _, _ = pointnet2_forward(state, points, fps_starts, training)
    logits = heads[0]
    B, N, _ = logits.shape
    return F.cross_entropy(logits.contiguous().view(B * N, 2), labels.view(B * N)), logits


def training_step_losses(state, batch, fps_starts, classes=("sphere", "plane", "cylinder", "cone"),
                         dropout_mask=None, match=None, multipliers=None, return_aux=False):
    """Forward + all losses of spfn_train_val_epoch (Utils/training_utils.py:140-146).
    `batch` carries the tensors of cpfn_amd.synthetic.training_batch().  `multipliers`: the six loss multipliers
    of the config (default all 1.0 = GlobalSPFN; LocalSPFN switches residue / parameter off,
    Configs/config_localSPFN.yml:10-11).  return_aux: also the index tensors of the forward pass and the heads."""
    heads, _, _, aux = pointnet2_forward(state, batch["P"], fps_starts, True, dropout_mask)
    X, T, W = heads
    raw_heads = heads
    X = F.normalize(X, p=2, dim=2, eps=1e-12)                                          # :141
    W = torch.softmax(W, dim=2)                                                         # :142
    gt = {"plane_normal": batch["plane_n_gt"], "cylinder_axis": batch["cylinder_axis_gt"],
          "cone_axis": batch["cone_axis_gt"]}
    out = ospfn.compute_all_losses(batch["P"], W, batch["I_gt"], X, batch["X_gt"], T, batch["T_gt"],
                                   gt, batch["points_per_instance"], classes=classes, match=match,
                                   multipliers=multipliers)
    if return_aux:
        aux = dict(aux)
        aux["heads"] = tuple(raw_heads)            # RAW network outputs (before normalise / soft-max)
        return out, aux
    return out
