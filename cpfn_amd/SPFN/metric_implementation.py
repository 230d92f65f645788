"""Evaluation metrics of SPFN (drop-in names for SPFN/metric_implementation.py).

Same functions, arguments and return values as the reference, evaluated on the device with the kernels of
the training path: the assignment runs on the device (cpfn_hungarian_match on the label-segmented sums), the
four fits are one fused pass (`losses_implementation.compute_parameters`), and P coverage — the one
heavy metric: every point of the cloud against every instance slot — is one streaming kernel
(cpfn_p_coverage) instead of `[B,K,N,3]` / `[B,K,N,4]` expansions (77 MB per 131072-point cloud at K = 49).
The `[B,K]`-sized bookkeeping stays stock PyTorch, as in the reference.
"""
import ctypes

import torch

from .. import lib as _l
from ..ops import _ptr, _stream
from . import fused_losses as _fl
from . import losses_implementation

PARAM_ORDER = [k for k, _ in _fl.PARAM_LAYOUT]


def hungarian_matching(W_pred, I_gt):
    """-> (matching_indices [B,K] long, mask [B,K] bool: True for the n_gt existing GT instances)
    (reference lines 9-30)."""
    B, N, K = W_pred.shape
    S = _fl.SegStats.apply(W_pred.detach().float(), I_gt)
    n_gt = _fl.count_gt(I_gt)
    if _fl.HOST_ASSIGNMENT or K > 64:       # (the device solver: one lane per column)
        match = _fl.hungarian_from_pack(_fl.hungarian_cost_pack(S, I_gt, n_gt), K)
    else:
        match = _fl.hungarian_device(S, n_gt)
    mask = torch.arange(K, device=W_pred.device).unsqueeze(0) < n_gt.unsqueeze(1)
    return match, mask


def hard_W_encoding(W):
    """One-hot of the arg-max membership (reference lines 33-37)."""
    return torch.nn.functional.one_hot(torch.argmax(W, dim=2), W.shape[2]).to(W.dtype)


def get_instance_type(T, W):
    """Per-instance type = arg-max of the membership-weighted per-point type scores (reference lines 52-55)."""
    return torch.argmax(torch.bmm(W.transpose(1, 2), T), dim=2)


def sqrt_safe(x):
    return torch.sqrt(torch.abs(x) + 1e-10)


def get_residual_loss(parameters, matching_indices, points_per_instance, T, classes=['plane', 'sphere', 'cylinder', 'cone']):
    """sqrt_safe of the per-point residue of the primitive type T[b,k], matched prediction (reference lines 69-75)."""
    B, K, Np, _ = points_per_instance.shape
    _, per_point = losses_implementation.compute_residue_loss(parameters, matching_indices, points_per_instance,
                                                              torch.gather(T, 1, matching_indices), classes=classes)
    per_point = torch.gather(per_point, 3, T.view(B, K, 1, 1).expand(B, K, Np, 1)).squeeze(3)
    return sqrt_safe(per_point)


def acos_safe(x):
    return torch.acos(torch.clamp(x, min=-1.0 + 1e-6, max=1.0 - 1e-6))


def compute_segmentation_iou(W, I_gt, matching_indices, mask):
    mIoU = 1 - losses_implementation.compute_miou_loss(W, I_gt, matching_indices)[0]
    return torch.sum(mask * mIoU, dim=1) / torch.sum(mask, dim=1)


def compute_type_accuracy(T, T_gt, matching_indices, mask):
    T_reordered = torch.gather(T, 1, matching_indices)
    return torch.sum(mask * (T_reordered == T_gt), dim=1) / torch.sum(mask, dim=1)


def compute_normal_difference(X, X_gt):
    return torch.mean(acos_safe(torch.abs(torch.sum(X * X_gt, dim=2))), dim=1)


def compute_axis_difference(predicted_parameters, gt_parameters, matching_indices, T, T_gt, mask,
                            classes=['plane', 'sphere', 'cylinder', 'cone'], div_eps=1e-10):
    mask = mask * (T == T_gt).float()
    parameter_loss = losses_implementation.compute_parameter_loss(predicted_parameters, gt_parameters, matching_indices,
                                                                  T_gt, is_eval=True, classes=classes)
    return torch.sum(mask * parameter_loss, dim=1) / torch.clamp(torch.sum(parameter_loss, dim=1), min=div_eps, max=None)


def compute_meanstd_Sk_residual(residue_loss, mask):
    mean_residual = torch.sum(mask * torch.mean(residue_loss, dim=2), dim=1) / torch.sum(mask, dim=1)
    std_residual = torch.sum(mask * torch.std(residue_loss, dim=2), dim=1) / torch.sum(mask, dim=1)
    return mean_residual, std_residual


def compute_Sk_coverage(residue_loss, epsilon, mask):
    residue_loss = torch.mean((residue_loss < epsilon).float(), dim=2)
    return torch.sum(mask * residue_loss, dim=1) / torch.sum(mask, dim=1)


def pack_parameters(predicted_parameters):
    """dict of the 10 fitted tensors -> [B,K,22] fp32 in the cpfn_fit_pack_fwd layout."""
    cols = [predicted_parameters[k] if predicted_parameters[k].dim() == 3 else predicted_parameters[k].unsqueeze(-1)
            for k in PARAM_ORDER]
    return torch.cat(cols, dim=-1).float().contiguous()


def compute_P_coverages(P, T, matching_indices, predicted_parameters, list_epsilon, classes=['plane', 'sphere', 'cylinder', 'cone']):
    """P coverage for up to four epsilons in ONE pass over the cloud -> [n_eps, B] (cpfn_p_coverage)."""
    B, N, _ = P.shape
    K = T.shape[1]
    if not P.is_cuda:
        raise RuntimeError("compute_P_coverage: CPU not supported")
    params = pack_parameters(predicted_parameters)
    slot_type = torch.gather(T, 1, matching_indices).contiguous()          # reference line 412: gather(T, 1, matching)
    n_eps = len(list_epsilon)
    h = _l.lib()
    ws = torch.empty(B * ((N + 255) // 256) * n_eps, dtype=torch.float32, device=P.device)
    out = torch.empty(B, n_eps, dtype=torch.float32, device=P.device)
    ids = (ctypes.c_int * 4)(*[classes.index(c) for c in ("plane", "sphere", "cylinder", "cone")])
    eps = (ctypes.c_float * n_eps)(*[float(e) for e in list_epsilon])
    with torch.cuda.device(P.device):
        _l.check(h.cpfn_p_coverage(_ptr(P.contiguous().float()), _ptr(params), _ptr(matching_indices.contiguous()),
                                   _ptr(slot_type), B, N, K, ids, eps, n_eps, _ptr(ws), _ptr(out), _stream()),
                 "cpfn_p_coverage")
    return out.t()


def compute_P_coverage(P, T, matching_indices, predicted_parameters, epsilon, classes=['plane', 'sphere', 'cylinder', 'cone']):
    """Fraction of the cloud's points within epsilon of some fitted primitive -> [B] (reference lines 409-415)."""
    return compute_P_coverages(P, T, matching_indices, predicted_parameters, [epsilon], classes)[0]


def compute_all_metrics(P, X, X_gt, W, I_gt, T, T_gt, points_per_instance, gt_parameters, list_epsilon=[0.01, 0.02],
                        classes=['plane', 'sphere', 'cylinder', 'cone']):
    """Same 11-tuple as the reference (lines 485-514): mIoU, type accuracy, normal difference, axis difference,
    mean / std Sk residual, Sk coverage per epsilon, P coverage per epsilon, hard W, fitted parameters, instance types."""
    W = hard_W_encoding(W)
    T = get_instance_type(T, W)
    diff = T.size(1) - T_gt.size(1)
    if diff > 0:
        T_gt = torch.cat((T_gt, torch.zeros_like(T_gt[:, 0:1]).expand(-1, diff)), dim=1)
    elif diff < 0:
        W = torch.cat((W, torch.zeros_like(W[:, :, 0:1]).expand(-1, -1, -diff)), dim=2)
        T = torch.cat((T, torch.zeros_like(T[:, 0:1]).expand(-1, -diff)), dim=1)
    matching_indices, mask = hungarian_matching(W, I_gt)
    mask = mask.float()
    mIoU = compute_segmentation_iou(W, I_gt, matching_indices, mask)
    type_accuracy = compute_type_accuracy(T, T_gt, matching_indices, mask)
    normal_difference = compute_normal_difference(X, X_gt)
    predicted_parameters = losses_implementation.compute_parameters(P, W, X)
    if diff > 0:
        gt_parameters = dict(gt_parameters)
        for key in ('plane_normal', 'cylinder_axis', 'cone_axis'):
            g = gt_parameters[key]
            gt_parameters[key] = torch.cat((g, torch.zeros_like(g[:, 0:1]).expand(-1, diff, 3)), dim=1)
        points_per_instance = torch.cat((points_per_instance, torch.zeros_like(points_per_instance[:, 0:1]).expand(
            -1, diff, points_per_instance.shape[2], 3)), dim=1)
    axis_difference = compute_axis_difference(predicted_parameters, gt_parameters, matching_indices, T, T_gt, mask, classes=classes)
    residue_loss = get_residual_loss(predicted_parameters, matching_indices, points_per_instance, T_gt, classes=classes)
    mean_residual, std_residual = compute_meanstd_Sk_residual(residue_loss, mask)
    Sk_coverage = [compute_Sk_coverage(residue_loss, epsilon, mask) for epsilon in list_epsilon]
    pc = compute_P_coverages(P, T, matching_indices, predicted_parameters, list_epsilon, classes=classes)
    P_coverage = [pc[i] for i in range(len(list_epsilon))]
    return (mIoU, type_accuracy, normal_difference, axis_difference, mean_residual, std_residual, Sk_coverage, P_coverage,
            W, predicted_parameters, T)


# Names the device path does not define (host-side GT parsing / JSON export, the TensorFlow twins) come from the
# reference's own SPFN/metric_implementation.py, found on sys.path (_reference.py): nothing of it is restated here.
from . import _reference as _ref  # noqa: E402

__getattr__ = _ref.module_fallback("metric_implementation")
