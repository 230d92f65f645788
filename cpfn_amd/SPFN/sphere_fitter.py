"""Sphere fitter (drop-in names for SPFN/sphere_fitter.py)."""
import torch

from . import fitters_common as _fc


def compute_parameters(P, W):
    """P [B,N,3], W [B,N,K] -> centre [B,K,3], radius² [B,K]   (reference lines 9-19)."""
    c, r2 = _fc.sphere_from_moments(_fc.moments(P, W))
    return c.to(P.dtype), r2.to(P.dtype)


def sqrt_safe(x):
    return torch.sqrt(torch.abs(x) + 1e-10)


def compute_residue_single(center, radius_squared, p):
    """(‖p − c‖ − r)²   (reference lines 61-62)."""
    return (sqrt_safe(torch.sum((p - center) ** 2, dim=-1)) - sqrt_safe(radius_squared)) ** 2


# Names the device path does not define (host-side GT parsing / JSON export, the TensorFlow twins) come from the
# reference's own SPFN/sphere_fitter.py, found on sys.path (_reference.py): nothing of it is restated here.
from . import _reference as _ref  # noqa: E402

__getattr__ = _ref.module_fallback("sphere_fitter")
