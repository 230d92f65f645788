"""Cylinder fitter (drop-in names for SPFN/cylinder_fitter.py)."""
import torch

from . import fitters_common as _fc
from . import plane_fitter as _plane


def compute_parameters(P, W, X):
    """P, X [B,N,3], W [B,N,K] -> axis [B,K,3], centre [B,K,3], radius² [B,K]   (reference lines 10-28)."""
    n, c, r2 = _fc.cylinder_from_moments(_fc.moments(P, W, X))
    return n.to(P.dtype), c.to(P.dtype), r2.to(P.dtype)


def sqrt_safe(x):
    return torch.sqrt(torch.abs(x) + 1e-10)


def compute_residue_single(axis, center, radius_squared, p):
    """(dist(p, axis line) − r)²   (reference lines 85-89)."""
    d = p - center
    d2 = torch.sum(d ** 2, dim=-1)
    along = torch.sum(d * axis, dim=-1)
    return (sqrt_safe(d2 - along ** 2) - sqrt_safe(radius_squared)) ** 2


def acos_safe(x):
    """acos clamped to ±(1 − 1e-6)   (reference lines 126-127)."""
    return _plane.acos_safe(x)


def compute_parameter_loss(predicted_axis, gt_axis, matching_indices, angle_diff):
    """1 − |axis_pred·axis_gt| (or its angle) of the matched instances   (reference lines 129-139: the plane's loss on the axis)."""
    return _plane.compute_parameter_loss(predicted_axis, gt_axis, matching_indices, angle_diff)


# Names the device path does not define (host-side GT parsing / JSON export, the TensorFlow twins) come from the
# reference's own SPFN/cylinder_fitter.py, found on sys.path (_reference.py): nothing of it is restated here.
from . import _reference as _ref  # noqa: E402

__getattr__ = _ref.module_fallback("cylinder_fitter")
