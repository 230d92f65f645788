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


def conv_as_linear(x, conv):
    """1x1 Conv1d/Conv2d applied to rows [P, C_in] -> [P, C_out]."""
    w = conv.weight.reshape(conv.weight.shape[0], -1)
    return F.linear(x, w.to(x.dtype), None if conv.bias is None else conv.bias.to(x.dtype))


def shared_mlp(x, convs, bns):
    """x [P, C_in] -> relu(bn(conv(.))) for every (conv, bn) pair."""
    for conv, bn in zip(convs, bns):
        x = F.relu(_bn_rows(conv_as_linear(x, conv).float(), bn))
    return x
