"""Cone fitter (drop-in names for SPFN/cone_fitter.py)."""
import numpy as np
import torch

from . import fitters_common as _fc
from .plane_fitter import acos_safe, compute_parameter_loss  # noqa: F401 (reference lines 9-10, 105-115)


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


# Names the device path does not define (host-side GT parsing / JSON export, the TensorFlow twins) come from the
# reference's own SPFN/cone_fitter.py, found on sys.path (_reference.py): nothing of it is restated here.
from . import _reference as _ref  # noqa: E402

__getattr__ = _ref.module_fallback("cone_fitter")
