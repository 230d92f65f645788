"""Cone fitter (drop-in names for SPFN/cone_fitter.py)."""
import numpy as np
import torch

from . import fitters_common as _fc
from . import plane_fitter as _plane


def acos_safe(x):
    """acos clamped to ±(1 − 1e-6)   (reference lines 9-10)."""
    return _plane.acos_safe(x)


def compute_parameters(P, W, X, div_eps=1e-10):
    """P, X [B,N,3], W [B,N,K] -> apex [B,K,3], axis [B,K,3], half_angle [B,K]   (reference lines 12-36)."""
    apex, axis, half = _fc.cone_from_moments(_fc.moments(P, W, X), P, W, div_eps)
    return apex.to(P.dtype), axis.to(P.dtype), half.to(P.dtype)


def compute_residue_single(apex, axis, half_angle, p):
    """sin²(min(|∠(p−apex, axis) − half_angle|, π/2)) · ‖p − apex‖²   (reference lines 98-103)."""
    v = p - apex
    vn = torch.nn.functional.normalize(v, p=2, dim=-1, eps=1e-12)
    alpha = acos_safe(torch.sum(vn * axis, dim=-1))
    return torch.sin(torch.clamp(torch.abs(alpha - half_angle), max=np.pi / 2)) ** 2 * torch.sum(v * v, dim=-1)


def compute_parameter_loss(predicted_axis, gt_axis, matching_indices, angle_diff):
    """1 − |axis_pred·axis_gt| (or its angle) of the matched instances   (reference lines 138-148: the plane's loss on the axis)."""
    return _plane.compute_parameter_loss(predicted_axis, gt_axis, matching_indices, angle_diff)


# Names the device path does not define (host-side GT parsing / JSON export, the TensorFlow twins) come from the
# reference's own SPFN/cone_fitter.py, found on sys.path (_reference.py): nothing of it is restated here.
from . import _reference as _ref  # noqa: E402

__getattr__ = _ref.module_fallback("cone_fitter")
