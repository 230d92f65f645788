"""autograd bindings of the fused moment kernels (cpfn_amd/csrc/fitters.hip)."""
import torch

from .. import lib as _l
from ..ops import _chk, _ptr, _stream

SLOTS = 52
# FitParams.backward as 3 launches (cpfn_fit_params_bwd_cone / _algebra, cpfn_fit_moments_bwd) instead of 5; False: the
# separate pack adjoint and chunk reduction (kept for the bit-identity test)
PARAMS_BWD_FUSED = True
# FitParams.forward: the moments' chunk reduction folded into the algebra launch (cpfn_fit_moments_algebra_fwd)
PARAMS_FWD_FUSED = True
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
            _l.check(_l.lib().cpfn_fit_moments_bwd(_ptr(P), _ptr(X), _ptr(W), _ptr(G32), B, N, K, None, _ptr(dW), _ptr(dX),
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
                                          _ptr(ws), _ptr(d6), 6, 0, _stream()), "cpfn_cone_pass_bwd")
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
            _l.check(_l.lib().cpfn_fit_algebra_fwd(_ptr(Mc), G, _ptr(out), None, _stream()), "cpfn_fit_algebra_fwd")
        ctx.save_for_backward(Mc)
        return out

    @staticmethod
    def backward(ctx, g):
        (Mc,) = ctx.saved_tensors
        G = Mc.numel() // SLOTS
        gc = g.contiguous().double()
        gM = torch.empty_like(Mc)
        with torch.cuda.device(Mc.device):
            _l.check(_l.lib().cpfn_fit_algebra_bwd(_ptr(Mc), _ptr(gc), None, G, _ptr(gM), None, _stream()),
                     "cpfn_fit_algebra_bwd")
        return gM


# The assignment of the loss section (fused_losses.hungarian_device) can ride on the next FitParams forward launch: the
# caller leaves (S [B,K+2,K] fp32 contiguous, n_gt [B] int64, match [B,K] int64 to fill) here; FitParams.forward takes it
# if the shapes fit (K <= 32), else the caller finds it still pending and solves the assignment with its own launch.
_match_rider = None


def set_match_rider(S, n_gt, match):
    global _match_rider
    _match_rider = (S, n_gt, match)


def pending_match_rider():
    """The rider if no FitParams launch took it (and forget it)."""
    global _match_rider
    r, _match_rider = _match_rider, None
    return r


def _take_match_rider(B, K):
    global _match_rider
    r = _match_rider
    if r is None:
        return None
    S, n_gt, match = r
    if K > 32 or tuple(S.shape) != (B, K + 2, K) or tuple(match.shape) != (B, K) or not (S.is_contiguous() and S.dtype == torch.float32):
        return None
    _match_rider = None
    return r


class FitParams(torch.autograd.Function):
    """(P [B,N,3], X [B,N,3] unit normals, W [B,N,K]) -> params [B,K,22] fp32: all four fits of every instance
    in the layout of include/cpfn_hip.h (cpfn_fit_pack_fwd).  The same kernels as FitMoments -> FitAlgebra ->
    ConePass -> (sign fix, half angle, concatenation) chained by hand in both directions, so no framework op
    runs between them: 6 launches forward, 6 backward (the autograd-glued chain was ~60).
    Differentiable in W and X (SPFN/{plane,sphere,cylinder,cone}_fitter.compute_parameters)."""

    @staticmethod
    def forward(ctx, P, X, W):
        P, X, W = _f32c(P), _f32c(X), _f32c(W)
        for t, n in ((P, "P"), (X, "X"), (W, "W")):
            _chk(t, n, torch.float32)
        B, N, K = W.shape
        dev = W.device
        h = _l.lib()
        chunks = h.cpfn_fit_num_chunks(B, N)
        ws = torch.empty(chunks * B * K * SLOTS, dtype=torch.float64, device=dev)
        M = torch.empty(B, K, SLOTS, dtype=torch.float64, device=dev)
        alg = torch.empty(B, K, 21, dtype=torch.float64, device=dev)
        cone_in = torch.empty(2, B, K, 3, dtype=torch.float32, device=dev)      # apex, axis (fp32) for the cone pass
        sums = torch.empty(B, K, 2, dtype=torch.float64, device=dev)
        params = torch.empty(B, K, 22, dtype=torch.float32, device=dev)
        G = B * K
        with torch.cuda.device(dev):
            st = _stream()
            rider = _take_match_rider(B, K)      # the loss section's assignment as extra workgroups of the moments launch
            S, n_gt, match = rider if rider is not None else (None, None, None)
            if PARAMS_FWD_FUSED:        # the moments' chunk reduction inside the algebra launch
                _l.check(h.cpfn_fit_moments_algebra_fwd(_ptr(P), _ptr(X), _ptr(W), B, N, K, _ptr(ws), _ptr(M), _ptr(alg),
                                                        _ptr(cone_in), _ptr(S), _ptr(n_gt), _ptr(match), st),
                         "cpfn_fit_moments_algebra_fwd")
            else:
                if rider is not None:
                    _l.check(h.cpfn_fit_moments_fwd_match(_ptr(P), _ptr(X), _ptr(W), B, N, K, _ptr(ws), _ptr(M), _ptr(S),
                                                          _ptr(n_gt), _ptr(match), st), "cpfn_fit_moments_fwd_match")
                else:
                    _l.check(h.cpfn_fit_moments_fwd(_ptr(P), _ptr(X), _ptr(W), B, N, K, _ptr(ws), _ptr(M), st), "cpfn_fit_moments_fwd")
                _l.check(h.cpfn_fit_algebra_fwd(_ptr(M), G, _ptr(alg), _ptr(cone_in), st), "cpfn_fit_algebra_fwd")
            # (the cone pass leaves its per-chunk partials in ws; the pack launch sums them: 3 launches, not 4)
            _l.check(h.cpfn_cone_pass_fwd(_ptr(P), _ptr(W), _ptr(cone_in[0]), _ptr(cone_in[1]), B, N, K, _ptr(ws), None, st),
                     "cpfn_cone_pass_fwd")
            _l.check(h.cpfn_fit_pack_fwd_partials(_ptr(alg), _ptr(ws), B, N, K, _ptr(M), _ptr(sums), _ptr(params), st),
                     "cpfn_fit_pack_fwd_partials")
        _l.add_bytes("cpfn_fit_moments_fwd", 4 * B * N * (6 + K) + 8 * (chunks + 1) * B * K * SLOTS)
        _l.add_bytes("cpfn_fit_algebra_fwd", 8 * G * (SLOTS + 21) + 24 * G)
        _l.add_bytes("cpfn_cone_pass_fwd", 4 * B * N * (3 + K) + 24 * G + 16 * (chunks + 1) * G)
        _l.add_bytes("cpfn_fit_pack_fwd_partials", 8 * G * (21 + 2 + SLOTS) + 88 * G)
        ctx.save_for_backward(P, X, W, M, cone_in, sums)
        return params

    @staticmethod
    def backward(ctx, g):
        P, X, W, M, cone_in, sums = ctx.saved_tensors
        B, N, K = W.shape
        dev = W.device
        G = B * K
        h = _l.lib()
        gp = g.contiguous().float()
        chunks = h.cpfn_fit_num_chunks(B, N)
        ws = torch.empty(chunks * B * K * 6, dtype=torch.float64, device=dev)
        dWc = torch.empty_like(W)
        gM32 = torch.empty(B, K, SLOTS, dtype=torch.float32, device=dev)
        dW = torch.empty_like(W)
        dX = torch.empty_like(X)
        with torch.cuda.device(dev):
            st = _stream()
            if PARAMS_BWD_FUSED:
                # three launches: the cone pass adjoint derives g_acos from gp itself, the algebra adjoint sums the cone
                # pass's per-chunk partials of d(apex, axis) into its own copy of the algebra's adjoint
                _l.check(h.cpfn_fit_params_bwd_cone(_ptr(P), _ptr(W), _ptr(cone_in[0]), _ptr(cone_in[1]), _ptr(gp), _ptr(sums),
                                                    _ptr(M), B, N, K, _ptr(dWc), _ptr(ws), st), "cpfn_fit_params_bwd_cone")
                _l.check(h.cpfn_fit_params_bwd_algebra(_ptr(M), _ptr(gp), _ptr(sums), _ptr(ws), chunks, B, K, _ptr(gM32), st),
                         "cpfn_fit_params_bwd_algebra")
            else:
                g_alg = torch.empty(B, K, 21, dtype=torch.float64, device=dev)
                g_acos = torch.empty(B, K, dtype=torch.float32, device=dev)
                gA0 = torch.empty(B, K, dtype=torch.float64, device=dev)
                _l.check(h.cpfn_fit_pack_bwd(_ptr(gp), _ptr(sums), _ptr(M), G, _ptr(g_alg), _ptr(g_acos), _ptr(gA0), st),
                         "cpfn_fit_pack_bwd")
                # cone pass adjoint: dW term, and d(apex, axis) accumulated into columns 15..20 of g_alg
                _l.check(h.cpfn_cone_pass_bwd(_ptr(P), _ptr(W), _ptr(cone_in[0]), _ptr(cone_in[1]), _ptr(g_acos), B, N, K,
                                              _ptr(dWc), _ptr(ws), g_alg.data_ptr() + 15 * 8, 21, 1, st), "cpfn_cone_pass_bwd")
                _l.check(h.cpfn_fit_algebra_bwd(_ptr(M), _ptr(g_alg), _ptr(gA0), G, None, _ptr(gM32), st), "cpfn_fit_algebra_bwd")
            _l.check(h.cpfn_fit_moments_bwd(_ptr(P), _ptr(X), _ptr(W), _ptr(gM32), B, N, K, _ptr(dWc), _ptr(dW), _ptr(dX), st),
                     "cpfn_fit_moments_bwd")
        if not PARAMS_BWD_FUSED:
            _l.add_bytes("cpfn_fit_pack_bwd", 88 * G + 8 * G * (2 + SLOTS + 21 + 1) + 4 * G)
        _l.add_bytes("cpfn_cone_pass_bwd", 4 * B * N * (3 + 2 * K) + 28 * G + 48 * (chunks + 1) * G)
        _l.add_bytes("cpfn_fit_algebra_bwd", 8 * G * (SLOTS + 22) + 4 * G * SLOTS + 48 * chunks * G)
        _l.add_bytes("cpfn_fit_moments_bwd", 4 * B * N * (6 + 2 * K) + 4 * G * SLOTS + 4 * B * N * (K + 3))
        return None, dX, dW
