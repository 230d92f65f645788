"""Evaluation metrics of SPFN (drop-in names for SPFN/metric_implementation.py).

Same functions, arguments and return values as the reference, evaluated on the device.  `compute_all_metrics`
(reference :485-514) is a chain of ten launches (csrc/metrics.hip + the training path's kernels), not ~200 framework
launches on [B,N,K] / [B,K,N',4] expansions:

  cpfn_metrics_points     one pass over the points: hard one-hot W, the (GT label x predicted label) histogram that is the
                          assignment's cost input for one-hot memberships, per-instance types, the normal difference
  cpfn_hungarian_match    the assignment on the device (K <= 64; SciPy on the host beyond, as in the reference)
  FitParams               the four fits of every instance (moments -> algebra -> cone pass -> pack: 4 launches)
  cpfn_metrics_tail       matched IoU, type accuracy, axis difference and the per-instance residue statistics (mean, std,
                          Sk coverage), reduced to the per-cloud figures
  cpfn_p_coverage         every point of the cloud against every instance slot (2 launches)

Label sets of different widths (K predictions vs K_gt ground-truth slots, :487-492, :505-508) are handled INSIDE the
kernels — a slot beyond either table reads as zeros — so nothing is padded with concatenations.  The stand-alone helpers
(`compute_type_accuracy`, `compute_Sk_coverage`, ...) keep the reference's names and results for callers that use them one
by one; they are written on one shared masked-mean helper.
"""
import ctypes

import torch

from .. import lib as _l
from ..ops import _ptr, _stream
from . import fused_losses as _fl
from . import losses_implementation

PARAM_ORDER = [k for k, _ in _fl.PARAM_LAYOUT]
_CANON = ("plane", "sphere", "cylinder", "cone")         # order of the kernels' type-id argument


def hungarian_matching(W_pred, I_gt):
    """-> (matching_indices [B,K] long, mask [B,K] bool: True for the n_gt existing GT instances)
    (reference lines 9-30)."""
    B, N, K = W_pred.shape
    S = _fl.SegStats.apply(W_pred.detach().float(), I_gt)
    n_gt = _fl.count_gt(I_gt)
    match = _assign(S, I_gt, n_gt)
    return match, _slot_mask(n_gt, K)


def _assign(S, I_gt, n_gt):
    K = S.shape[2]
    if _fl.HOST_ASSIGNMENT or K > 64:       # (the device solver: one lane per column)
        return _fl.hungarian_from_pack(_fl.hungarian_cost_pack(S, I_gt, n_gt), K)
    return _fl.hungarian_device(S, n_gt)


def _slot_mask(n_gt, K):
    return torch.arange(K, device=n_gt.device).unsqueeze(0) < n_gt.unsqueeze(1)


def _instance_mean(per_slot, mask):
    """Average of a [B,K] quantity over the slots that hold a GT instance."""
    m = mask.to(per_slot.dtype)
    return (per_slot * m).sum(dim=1) / m.sum(dim=1)


def hard_W_encoding(W):
    """One-hot of the arg-max membership (reference lines 33-37)."""
    return torch.nn.functional.one_hot(torch.argmax(W, dim=2), W.shape[2]).to(W.dtype)


def get_instance_type(T, W):
    """Per-instance type = arg-max of the membership-weighted per-point type scores (reference lines 52-55)."""
    return torch.argmax(torch.bmm(W.transpose(1, 2), T), dim=2)


def sqrt_safe(x):
    return torch.sqrt(torch.abs(x) + 1e-10)


def get_residual_loss(parameters, matching_indices, points_per_instance, T, classes=['plane', 'sphere', 'cylinder', 'cone']):
    """sqrt_safe of the per-point residue of the primitive type T[b,k], matched prediction (reference lines 76-81)."""
    B, K, Np, _ = points_per_instance.shape
    _, per_point = losses_implementation.compute_residue_loss(parameters, matching_indices, points_per_instance,
                                                              torch.gather(T, 1, matching_indices), classes=classes)
    per_point = torch.gather(per_point, 3, T.view(B, K, 1, 1).expand(B, K, Np, 1)).squeeze(3)
    return sqrt_safe(per_point)


def acos_safe(x):
    return torch.acos(torch.clamp(x, min=-1.0 + 1e-6, max=1.0 - 1e-6))


def compute_segmentation_iou(W, I_gt, matching_indices, mask):
    """(reference lines 119-121)"""
    iou_loss, _ = losses_implementation.compute_miou_loss(W, I_gt, matching_indices)
    return _instance_mean(1 - iou_loss, mask)


def compute_type_accuracy(T, T_gt, matching_indices, mask):
    """(reference lines 142-144)"""
    return _instance_mean((T.gather(1, matching_indices) == T_gt).to(torch.float32), mask)


def compute_normal_difference(X, X_gt):
    """(reference lines 170-172)"""
    return acos_safe((X * X_gt).sum(dim=2).abs()).mean(dim=1)


def compute_axis_difference(predicted_parameters, gt_parameters, matching_indices, T, T_gt, mask,
                            classes=['plane', 'sphere', 'cylinder', 'cone'], div_eps=1e-10):
    """Angle between matched and GT axis, counted where the slot's OWN predicted type equals the GT type and normalised by
    the sum of the angles over all slots (the reference's definition, lines 189-193)."""
    angle = losses_implementation.compute_parameter_loss(predicted_parameters, gt_parameters, matching_indices, T_gt,
                                                         is_eval=True, classes=classes)
    counted = mask.to(angle.dtype) * (T == T_gt).to(angle.dtype)
    return (counted * angle).sum(dim=1) / angle.sum(dim=1).clamp(min=div_eps)


def compute_meanstd_Sk_residual(residue_loss, mask):
    """(reference lines 257-260; torch.std: unbiased)"""
    return _instance_mean(residue_loss.mean(dim=2), mask), _instance_mean(residue_loss.std(dim=2), mask)


def compute_Sk_coverage(residue_loss, epsilon, mask):
    """(reference lines 332-335)"""
    return _instance_mean((residue_loss < epsilon).to(torch.float32).mean(dim=2), mask)


def pack_parameters(predicted_parameters):
    """dict of the 10 fitted tensors -> [B,K,22] fp32 in the cpfn_fit_pack_fwd layout."""
    cols = [predicted_parameters[k] if predicted_parameters[k].dim() == 3 else predicted_parameters[k].unsqueeze(-1)
            for k in PARAM_ORDER]
    return torch.cat(cols, dim=-1).float().contiguous()


def unpack_parameters(params22):
    """[B,K,22] (cpfn_fit_pack_fwd layout) -> dict of the 10 fitted tensors, as views."""
    out, o = {}, 0
    for key, width in _fl.PARAM_LAYOUT:
        out[key] = params22[..., o:o + width] if width > 1 else params22[..., o]
        o += width
    return out


def _type_ids(classes):
    return (ctypes.c_int * 4)(*[classes.index(c) if c in classes else -1 for c in _CANON])


def _p_coverage(P, params22, matching_indices, slot_type, list_epsilon, classes):
    B, N, _ = P.shape
    K = slot_type.shape[1]
    n_eps = len(list_epsilon)
    h = _l.lib()
    ws = torch.empty(B * ((N + 255) // 256) * n_eps, dtype=torch.float32, device=P.device)
    out = torch.empty(B, n_eps, dtype=torch.float32, device=P.device)
    eps = (ctypes.c_float * n_eps)(*[float(e) for e in list_epsilon])
    with torch.cuda.device(P.device):
        _l.check(h.cpfn_p_coverage(_ptr(P), _ptr(params22), _ptr(matching_indices), _ptr(slot_type), B, N, K, _type_ids(classes),
                                   eps, n_eps, _ptr(ws), _ptr(out), _stream()), "cpfn_p_coverage")
    return out


def compute_P_coverages(P, T, matching_indices, predicted_parameters, list_epsilon, classes=['plane', 'sphere', 'cylinder', 'cone']):
    """P coverage for up to four epsilons in ONE pass over the cloud -> [n_eps, B] (cpfn_p_coverage)."""
    if not P.is_cuda:
        raise RuntimeError("compute_P_coverage: CPU not supported")
    slot_type = torch.gather(T, 1, matching_indices).contiguous()          # reference line 412: gather(T, 1, matching)
    return _p_coverage(P.contiguous().float(), pack_parameters(predicted_parameters), matching_indices.contiguous(), slot_type,
                       list_epsilon, classes).t()


def compute_P_coverage(P, T, matching_indices, predicted_parameters, epsilon, classes=['plane', 'sphere', 'cylinder', 'cone']):
    """Fraction of the cloud's points within epsilon of some fitted primitive -> [B] (reference lines 409-415)."""
    return compute_P_coverages(P, T, matching_indices, predicted_parameters, [epsilon], classes)[0]


FUSED_MAX_K = 128         # label-set width up to which the point pass keeps its histogram on chip (cpfn_metrics_points)


def compute_all_metrics(P, X, X_gt, W, I_gt, T, T_gt, points_per_instance, gt_parameters, list_epsilon=[0.01, 0.02],
                        classes=['plane', 'sphere', 'cylinder', 'cone']):
    """Same 11-tuple as the reference (lines 485-514): mIoU, type accuracy, normal difference, axis difference,
    mean / std Sk residual, Sk coverage per epsilon, P coverage per epsilon, hard W, fitted parameters, instance types —
    the last three Kp = max(K, K_gt) wide, like the reference's padded ones."""
    if not P.is_cuda:
        raise RuntimeError("compute_all_metrics: CPU not supported (cpfn_amd runs on the HIP path only)")
    from .. import ops as _ops
    _ops.check_fps_faults("compute_all_metrics")             # (degenerate sampling upstream must not become a metric)
    B, N, K = W.shape
    Kgt, Np = T_gt.shape[1], points_per_instance.shape[2]
    Kp, NT = max(K, Kgt), T.shape[2]
    dev = P.device
    f32 = lambda t: t.contiguous().float()
    P, X, X_gt, W, T = f32(P), f32(X), f32(X_gt), f32(W), f32(T)
    I_gt, T_gt = I_gt.contiguous().long(), T_gt.contiguous().long()
    h = _l.lib()
    hardW = torch.empty(B, N, Kp, dtype=torch.float32, device=dev)
    head = torch.empty(B, 1, dtype=torch.float32, device=dev)                 # normal difference
    if Kp <= FUSED_MAX_K and NT <= 8:
        S = torch.empty(B, Kp + 2, Kp, dtype=torch.float32, device=dev)
        n_gt = torch.empty(B, dtype=torch.int64, device=dev)
        T_inst = torch.empty(B, Kp, dtype=torch.int64, device=dev)
        ws = torch.empty(h.cpfn_metrics_workspace(B, N, Kp, NT), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _l.check(h.cpfn_metrics_points(_ptr(W), _ptr(T), _ptr(X), _ptr(X_gt), _ptr(I_gt), B, N, K, Kp, NT, _ptr(hardW), _ptr(ws),
                                           _ptr(S), _ptr(n_gt), _ptr(T_inst), _ptr(head), _stream()), "cpfn_metrics_points")
    else:               # label sets wider than the on-chip histogram: the same quantities from the generic kernels
        hardW.zero_()
        hardW[:, :, :K] = hard_W_encoding(W)
        T_inst = torch.zeros(B, Kp, dtype=torch.int64, device=dev)
        T_inst[:, :K] = get_instance_type(T, hardW[:, :, :K])
        S = _fl.SegStats.apply(hardW, I_gt)
        n_gt = _fl.count_gt(I_gt)
        head[:, 0] = compute_normal_difference(X, X_gt)
    match = _assign(S, I_gt, n_gt)
    with torch.no_grad():
        params22 = _fitters_packed(P, hardW, X)
    n_eps = len(list_epsilon)
    tail = torch.empty(B, 5 + min(n_eps, 4), dtype=torch.float32, device=dev)
    slot_type = torch.empty(B, Kp, dtype=torch.int64, device=dev)
    axes = [f32(gt_parameters[k]) for k in ('plane_normal', 'cylinder_axis', 'cone_axis')]
    ppi = f32(points_per_instance)
    Sk, Pc = [], []
    for e0 in range(0, max(n_eps, 1), 4):                                     # (the kernels take four thresholds at a time)
        eps = [float(e) for e in list_epsilon[e0:e0 + 4]]
        ceps = (ctypes.c_float * max(len(eps), 1))(*eps)
        if e0 > 0:
            tail = torch.empty(B, 5 + len(eps), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _l.check(h.cpfn_metrics_tail(_ptr(S), _ptr(match), _ptr(n_gt), _ptr(T_inst), _ptr(T_gt), _ptr(params22), _ptr(ppi),
                                         _ptr(axes[0]), _ptr(axes[1]), _ptr(axes[2]), B, Kp, Kgt, Np, _type_ids(classes), ceps,
                                         len(eps), _ptr(tail), _ptr(slot_type), _stream()), "cpfn_metrics_tail")
        if e0 == 0:
            first = tail
        if eps:
            Sk += [tail[:, 5 + i] for i in range(len(eps))]
            pc = _p_coverage(P, params22, match, slot_type, eps, classes)
            Pc += [pc[:, i] for i in range(len(eps))]
    # The caller reads these on the host next (evaluation_*.py: `.item()` / `.cpu()` on every entry), so the stream is drained
    # HERE and the sampling fault word checked behind it: a several-workgroups FPS of the forward pass that was still queued
    # when this function started cannot hide behind the check at its head any more.
    if not torch.cuda.is_current_stream_capturing():
        torch.cuda.current_stream(dev).synchronize()
        _ops.check_fps_faults("the end of compute_all_metrics")
    return (first[:, 0], first[:, 1], head[:, 0], first[:, 2], first[:, 3], first[:, 4], Sk, Pc, hardW,
            unpack_parameters(params22), T_inst)


def _fitters_packed(P, W, X):
    from . import fitters_common as _fc
    return _fc.fit_params(P, W, X)


# Names the device path does not define (host-side GT parsing / JSON export, the TensorFlow twins) come from the
# reference's own SPFN/metric_implementation.py, found on sys.path (_reference.py): nothing of it is restated here.
from . import _reference as _ref  # noqa: E402

__getattr__ = _ref.module_fallback("metric_implementation")
