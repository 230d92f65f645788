"""autograd bindings of the fused moment kernels (cpfn_amd/csrc/fitters.hip)."""
import torch

from .. import lib as _l
from ..ops import _chk, _ptr, _stream

SLOTS = 52
# slot map of include/cpfn_hip.h
A0, AP, APP, AX, AXX = 0, slice(1, 4), slice(4, 10), slice(10, 13), slice(13, 19)
B0, BP, BPP, BPPP, BXX, BXPX = 20, slice(21, 24), slice(24, 30), slice(30, 40), slice(40, 46), slice(46, 49)


def _f32c(t):
    return t.detach().contiguous().float()


class FitMoments(torch.autograd.Function):
    """(P [B,N,3], X [B,N,3], W [B,N,K]) -> M [B,K,52] float64.  Differentiable in W and X
    (P is data: Utils/training_utils.py:122 never asks for its gradient)."""

    @staticmethod
    def forward(ctx, P, X, W):
        P, X, W = _f32c(P), _f32c(X), _f32c(W)
        for t, n in ((P, "P"), (X, "X"), (W, "W")):
            _chk(t, n, torch.float32)
        B, N, K = W.shape
        h = _l.lib()
        chunks = h.cpfn_fit_num_chunks(B, N)
        ws = torch.empty(chunks * B * K * SLOTS, dtype=torch.float64, device=W.device)
        M = torch.empty(B, K, SLOTS, dtype=torch.float64, device=W.device)
        with torch.cuda.device(W.device):
            _l.check(h.cpfn_fit_moments_fwd(_ptr(P), _ptr(X), _ptr(W), B, N, K, _ptr(ws), _ptr(M), _stream()),
                     "cpfn_fit_moments_fwd")
        ctx.save_for_backward(P, X, W)
        return M

    @staticmethod
    def backward(ctx, G):
        P, X, W = ctx.saved_tensors
        B, N, K = W.shape
        G32 = G.contiguous().float()
        dW = torch.empty_like(W)
        dX = torch.empty_like(X)
        with torch.cuda.device(W.device):
            _l.check(_l.lib().cpfn_fit_moments_bwd(_ptr(P), _ptr(X), _ptr(W), _ptr(G32), B, N, K, _ptr(dW), _ptr(dX),
                                                   _stream()), "cpfn_fit_moments_bwd")
        return None, dX, dW


class ConePass(torch.autograd.Function):
    """(P, W, apex [B,K,3], axis [B,K,3]) -> [B,K,2] float64:
    Σ_n W·(axis·normalize(p−apex)) and Σ_n W·acos_safe(|·|)   (SPFN/cone_fitter.py:25-34)."""

    @staticmethod
    def forward(ctx, P, W, apex, axis):
        P, W = _f32c(P), _f32c(W)
        ap32, ax32 = _f32c(apex), _f32c(axis)
        B, N, K = W.shape
        h = _l.lib()
        chunks = h.cpfn_fit_num_chunks(B, N)
        ws = torch.empty(chunks * B * K * 2, dtype=torch.float64, device=W.device)
        out = torch.empty(B, K, 2, dtype=torch.float64, device=W.device)
        with torch.cuda.device(W.device):
            _l.check(h.cpfn_cone_pass_fwd(_ptr(P), _ptr(W), _ptr(ap32), _ptr(ax32), B, N, K, _ptr(ws), _ptr(out),
                                          _stream()), "cpfn_cone_pass_fwd")
        ctx.save_for_backward(P, W, ap32, ax32)
        ctx.out_dtypes = (apex.dtype, axis.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        P, W, ap32, ax32 = ctx.saved_tensors
        B, N, K = W.shape
        g_acos = g[..., 1].contiguous().float()      # the Σ W·dot column only feeds sign(): zero adjoint
        h = _l.lib()
        chunks = h.cpfn_fit_num_chunks(B, N)
        ws = torch.empty(chunks * B * K * 6, dtype=torch.float64, device=W.device)
        dW = torch.empty_like(W)
        d6 = torch.empty(B, K, 6, dtype=torch.float64, device=W.device)
        with torch.cuda.device(W.device):
            _l.check(h.cpfn_cone_pass_bwd(_ptr(P), _ptr(W), _ptr(ap32), _ptr(ax32), _ptr(g_acos), B, N, K, _ptr(dW),
                                          _ptr(ws), _ptr(d6), _stream()), "cpfn_cone_pass_bwd")
        return None, dW, d6[..., :3].to(ctx.out_dtypes[0]), d6[..., 3:].to(ctx.out_dtypes[1])


def eigh3(S6):
    """S6 [...,6] float64 (xx xy xz yy yz zz) -> (lam [...,3] ascending, V [...,3,3] eigenvectors in columns)."""
    S = S6.detach().contiguous().double()
    G = S.numel() // 6
    lam = torch.empty(S.shape[:-1] + (3,), dtype=torch.float64, device=S.device)
    V = torch.empty(S.shape[:-1] + (3, 3), dtype=torch.float64, device=S.device)
    with torch.cuda.device(S.device):
        _l.check(_l.lib().cpfn_eigh3(_ptr(S), G, _ptr(lam), _ptr(V), _stream()), "cpfn_eigh3")
    return lam, V


class FitAlgebra(torch.autograd.Function):
    """M [B,K,52] float64 -> out [B,K,21] float64 (one lane per instance; backward = Jᵀg by
    forward-mode AD inside the kernel, 52 lanes per instance)."""

    @staticmethod
    def forward(ctx, M):
        Mc = M.detach().contiguous().double()
        G = Mc.numel() // SLOTS
        out = torch.empty(Mc.shape[:-1] + (21,), dtype=torch.float64, device=Mc.device)
        with torch.cuda.device(Mc.device):
            _l.check(_l.lib().cpfn_fit_algebra_fwd(_ptr(Mc), G, _ptr(out), _stream()), "cpfn_fit_algebra_fwd")
        ctx.save_for_backward(Mc)
        return out

    @staticmethod
    def backward(ctx, g):
        (Mc,) = ctx.saved_tensors
        G = Mc.numel() // SLOTS
        gc = g.contiguous().double()
        gM = torch.empty_like(Mc)
        with torch.cuda.device(Mc.device):
            _l.check(_l.lib().cpfn_fit_algebra_bwd(_ptr(Mc), _ptr(gc), G, _ptr(gM), _stream()), "cpfn_fit_algebra_bwd")
        return gM
