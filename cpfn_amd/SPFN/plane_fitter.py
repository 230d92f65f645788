"""Plane fitter (drop-in names for SPFN/plane_fitter.py; GT/JSON helpers are host-side
data plumbing and out of scope)."""
import torch

from . import fitters_common as _fc


def compute_parameters(P, W):
    """P [B,N,3], W [B,N,K] -> n [B,K,3], c [B,K]   (reference lines 9-17)."""
    n, c = _fc.plane_from_moments(_fc.moments(P, W))
    return n.to(P.dtype), c.to(P.dtype)


def compute_residue_single(n, c, p):
    """(p·n − c)²   (reference lines 54-55)."""
    return (torch.sum(p * n, dim=-1) - c) ** 2


def acos_safe(x):
    return torch.acos(torch.clamp(x, min=-1.0 + 1e-6, max=1.0 - 1e-6))


def compute_parameter_loss(predicted_n, gt_n, matching_indices, angle_diff):
    """1 − |n_pred·n_gt| (or its angle) of the matched instances   (reference lines 87-97)."""
    B, Kgt, _ = gt_n.size()
    pred = torch.gather(predicted_n, 1, matching_indices.unsqueeze(2).expand(B, Kgt, 3))
    dot_abs = torch.abs(torch.sum(pred * gt_n, dim=2))
    return acos_safe(dot_abs) if angle_diff else 1.0 - dot_abs


# Names the device path does not define (host-side GT parsing / JSON export, the TensorFlow twins) come from the
# reference's own SPFN/plane_fitter.py, found on sys.path (_reference.py): nothing of it is restated here.
from . import _reference as _ref  # noqa: E402

__getattr__ = _ref.module_fallback("plane_fitter")
