"""Weighted total least squares (drop-in names for SPFN/differentiable_tls.py).

`min_x Σ_n w_n (a_n·x)²  s.t. ‖x‖ = 1` is the eigenvector of M = Σ w a aᵀ with the
smallest eigenvalue.  The reference builds M from a [B,N,3,3] temporary and takes
torch.svd (lines 200-209, 123-127); here M comes out of the fused moment kernel and a
batched fp64 Jacobi kernel diagonalises it.  The backward pass is the reference's
hand-written SVD adjoint (lines 131-143) specialised to the last column of V.
"""
import torch

from . import moments as _m

_SYM = ((0, 1, 2), (1, 3, 4), (2, 4, 5))


_index_cache = {}


def cached_index(name, nested, device):
    """Small constant index tensors, created once per device (never inside a graph capture)."""
    key = (name, str(device))
    t = _index_cache.get(key)
    if t is None:
        t = torch.tensor(nested, device=device)
        _index_cache[key] = t
    return t


def sym3(v6):
    """[...,6] (xx xy xz yy yz zz) -> symmetric [...,3,3]."""
    return v6[..., cached_index("sym3", _SYM, v6.device)]


def guard_one_over_matrix(M, min_abs_value=1e-10):
    """Reference lines 8-17: reciprocal of the off-diagonal entries with the upper triangle
    pushed to >= +eps and the lower triangle to <= -eps; zero diagonal."""
    n = M.shape[-1]
    iu = torch.triu(torch.ones(n, n, dtype=torch.bool, device=M.device), 1)
    il = iu.transpose(0, 1)
    out = torch.zeros_like(M)
    out = torch.where(iu, 1.0 / M.clamp(min=min_abs_value), out)
    return torch.where(il, 1.0 / M.clamp(max=-min_abs_value), out)


def compute_svd_K(s):
    """Reference lines 45-53: K[i,j] = 1/(s_i² − s_j²), guarded, for s sorted descending."""
    s2 = s * s
    return guard_one_over_matrix(s2.unsqueeze(-1) - s2.unsqueeze(-2))


class Custom_svd_v_colum(torch.autograd.Function):
    """Last right-singular vector of a symmetric PSD 3x3 matrix given as 6 unique entries.

    forward : fp64 Jacobi eigen-decomposition (HIP), v = eigenvector of the smallest eigenvalue.
    backward: the reference's adjoint (lines 131-143).  With G = Vᵀ·grad_V non-zero only in
              its last column it collapses to
                 dL/dM = Σ_{i<2} (v_i·g) K[2,i] ( s_i v_i v_2ᵀ + s_2 v_2 v_iᵀ ),
              K[2,i] = 1/min(s_2² − s_i², −1e-10), s descending, U = V for a PSD matrix.
    """

    @staticmethod
    def forward(ctx, S6):
        lam, V = _m.eigh3(S6)
        ctx.save_for_backward(lam, V)
        return V[..., :, 0].to(S6.dtype)

    @staticmethod
    def backward(ctx, g):
        lam, V = ctx.saved_tensors
        g = g.double()
        s = lam.abs().flip(-1)                       # descending singular values
        Vd = V.flip(-1)                              # columns ordered like s: v_0, v_1, v_2(smallest)
        K = compute_svd_K(s)                         # [...,3,3]
        v2 = Vd[..., :, 2]
        gM = torch.zeros_like(V)
        for i in (0, 1):
            vi = Vd[..., :, i]
            coef = ((vi * g).sum(-1) * K[..., 2, i]).unsqueeze(-1).unsqueeze(-1)
            gM = gM + coef * (s[..., i, None, None] * vi.unsqueeze(-1) * v2.unsqueeze(-2)
                              + s[..., 2, None, None] * v2.unsqueeze(-1) * vi.unsqueeze(-2))
        # adjoint of the 6 unique entries of the symmetric matrix
        g6 = torch.stack([gM[..., 0, 0], gM[..., 0, 1] + gM[..., 1, 0], gM[..., 0, 2] + gM[..., 2, 0],
                          gM[..., 1, 1], gM[..., 1, 2] + gM[..., 2, 1], gM[..., 2, 2]], dim=-1)
        return g6


def smallest_eigvec(S6):
    return Custom_svd_v_colum.apply(S6)


def solve_weighted_tls(A, W):
    """A [G,N,3], W [G,N] -> x [G,3]  (reference lines 200-209)."""
    M = _m.FitMoments.apply(A, A, W.unsqueeze(2))            # instance axis of size 1
    return smallest_eigvec(M[:, 0, _m.AXX]).to(A.dtype)      # x⊗x slots: differentiable in A


# Names the device path does not define (host-side GT parsing / JSON export, the TensorFlow twins) come from the
# reference's own SPFN/differentiable_tls.py, found on sys.path (_reference.py): nothing of it is restated here.
from . import _reference as _ref  # noqa: E402

__getattr__ = _ref.module_fallback("differentiable_tls")
