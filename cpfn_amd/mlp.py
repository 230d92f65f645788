"""Shared per-point MLP stacks (1x1 conv + BatchNorm + ReLU) on points-major rows.

The reference expresses these as Conv2d/Conv1d + BatchNorm2d/1d + ReLU on
channel-major tensors (modules/pointset_abstraction.py:70-73,
modules/pointset_feature_propagation.py:49-51).  Here a stack consumes rows
`[P, C_in]` (P = every point / neighbour of the batch) and produces rows `[P, C_out]`;
a 1x1 convolution is a GEMM over rows.  The nn.Conv*/nn.BatchNorm* modules are kept
only as parameter containers so state_dict keys, shapes and the trainer's
`'bn' in name -> module.momentum` updates (Utils/training_utils.py:19-22) keep working.
"""
import torch
import torch.nn.functional as F


def _bn_rows(y, bn):
    """BatchNorm over rows [P, C] with the module's parameters / running statistics."""
    training = bn.training or not bn.track_running_stats
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    momentum = 0.0 if bn.momentum is None else bn.momentum
    return F.batch_norm(y, bn.running_mean if bn.track_running_stats else None,
                        bn.running_var if bn.track_running_stats else None,
                        bn.weight, bn.bias, training, momentum, bn.eps)


# Experiment switch (tools/bf16_vs_fp32_training.py --arms fp32r; NOTEBOOK R5.5): the fp32 PyTorch path with the bf16 path's STORAGE
# roundings — GEMM operands (activations, weights) and the stored pre-BN output rounded to bf16, everything else fp32 — to tell
# "rounding noise" from "anything else the fused path does differently" when bf16 and fp32 trainings are compared.
ROUND_STORAGE = False


def _r(t):
    return t.to(torch.bfloat16).to(t.dtype) if ROUND_STORAGE else t


def conv_as_linear(x, conv):
    """1x1 Conv1d/Conv2d applied to rows [P, C_in] -> [P, C_out]."""
    w = conv.weight.reshape(conv.weight.shape[0], -1)
    return _r(F.linear(_r(x), _r(w.to(x.dtype)), None if conv.bias is None else conv.bias.to(x.dtype)))


def shared_mlp(x, convs, bns, dtype=torch.float32):
    """x [P, C_in] -> relu(bn(conv(.))) for every (conv, bn) pair.  `dtype` is the GEMM
    operand type (bf16 for training on MI355X, fp32 for parity tests); BatchNorm statistics
    and normalisation are always fp32."""
    for conv, bn in zip(convs, bns):
        x = F.relu(_bn_rows(conv_as_linear(x.to(dtype), conv).float(), bn))
    return x


def run_stack(x, convs, bns, dtype=torch.float32, pool_k=None, xyz_rows=None, dropout=None, xyz_tail=None, handover=None, gather=None,
              join_out=None, top_ride=None):
    """Run a whole (conv, bn, relu)* stack on rows and optionally max-pool every `pool_k`
    consecutive rows.  `x` [P, C_in] (any float dtype) — or None with `xyz_rows` [P, 3] fp32
    when the stack's only input is relative coordinates (sa1).

    xyz_tail [P,3] fp32 (bf16 HIP path only): three more input channels of the first layer, BEHIND x's (sa2: gathered
    features first, then the centred coordinates) — kept in fp32 beside the bf16 rows instead of concatenated.

    join_out (bf16 HIP path, pooled stacks): an autograd_ops.SkipJoin through which the backward node of the output's OTHER consumer
    hands its gradient to this stack's backward (ignored elsewhere: the join then never arms).

    dtype == torch.bfloat16 on a HIP device: the fused MFMA path (cpfn_amd/fused_mlp.py).
    dtype == torch.float32: plain PyTorch ops — the fp32 reference the fused path is tested
    against, and the path used for tight parity against the reference's goldens."""
    src = x if x is not None else xyz_rows
    if dtype == torch.bfloat16 and src.is_cuda:
        from . import fused_mlp
        if x is None:
            return fused_mlp.fused_mlp_stack(xyz_rows.float().contiguous(), convs, bns, pool_k=pool_k, first_fp32=True)
        if xyz_tail is not None:
            return fused_mlp.fused_mlp_stack(x.contiguous(), convs, bns, pool_k=pool_k, dropout=dropout, xyz_tail=xyz_tail.float().contiguous(),
                                             gather=gather, join_out=join_out)
        P, C = x.shape
        Cp = (C + 63) // 64 * 64
        if Cp != C or x.dtype != torch.bfloat16 or not x.is_contiguous():
            xp = torch.zeros(P, Cp, dtype=torch.bfloat16, device=x.device) if Cp != C else None
            if xp is None:
                xp = x.to(torch.bfloat16).contiguous()
            else:
                xp[:, :C] = x
            x = xp
        return fused_mlp.fused_mlp_stack(x, convs, bns, pool_k=pool_k, dropout=dropout, handover=handover, join_out=join_out,
                                         top_ride=top_ride)
    if dropout is not None or xyz_tail is not None:
        raise ValueError("fused dropout / the xyz tail exist on the bf16 HIP path only")
    y = shared_mlp(src, convs, bns, dtype)
    if pool_k:
        y = y.reshape(-1, pool_k, y.shape[1]).max(dim=1)[0]
    return y


def heads(feat, head_convs, dtype=torch.float32, handover=None):
    """fc2 heads on rows [P,128] -> list of fp32 [P, o_i]."""
    if dtype == torch.bfloat16 and feat.is_cuda:
        from . import fused_mlp
        return fused_mlp.linear_heads(feat.to(torch.bfloat16).contiguous(), [h.weight for h in head_convs],
                                      [h.bias for h in head_convs], handover=handover)
    return [conv_as_linear(feat.to(dtype), h).float() for h in head_convs]
