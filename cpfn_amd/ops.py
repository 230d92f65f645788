"""Tensor-level front-end of the HIP kernels: argument checks (the reference's
CHECK_CONTIGUOUS / CHECK_IS_FLOAT / CHECK_IS_INT / CHECK_CUDA macros,
cuda_ops/include/utils.h:5-25, become RuntimeErrors here), output allocation with
torch (the C ABI never allocates) and launch on torch's current stream.
"""
import numpy as np

import torch

from . import lib as _l


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name, dtype):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError("%s must be a tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA(HIP) tensor: CPU not supported" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be a %s tensor" % (name, str(dtype).replace("torch.", "")))
    return t


_background = [False]


class background_geometry:
    """Context: the FPS / ball-query / 3-NN calls issued inside run on a side stream beside other work (the next batch's
    geometry beside a training step) and use the kernel shapes that disturb their neighbours least
    (cpfn_set_background_geometry) instead of the fastest ones; same results, bit for bit."""

    def __enter__(self):
        self._was = _l.lib().cpfn_set_background_geometry(1)
        self._was_py, _background[0] = _background[0], True
        return self

    def __exit__(self, *exc):
        _l.lib().cpfn_set_background_geometry(self._was)
        _background[0] = self._was_py
        return False


def ball_query_threshold(radius):
    """f32(radius**2 in double): what `sqrdists > radius ** 2` compares against
    (modules/geometry_utils.py:156)."""
    return float(np.float32(float(radius) ** 2))


def fps_faults():
    """Sampling faults since the library was loaded (cpfn_fps_faults): clouds whose several-workgroups FPS (N > 8192) gave up on a
    sibling workgroup (their remaining samples are index 0) + samples whose update was lost on the lane that owns them (the
    tripwire of csrc/sampling.hip: caught and counted; that launch's indices are wrong).  0 in a healthy process; a pinned host word: no synchronisation."""
    n = _l.lib().cpfn_fps_faults()
    if n < 0:
        raise RuntimeError("cpfn_fps_faults failed")
    return n


_fps_faults_seen = 0


def check_fps_faults(where):
    """Raise if the sampling kernels have reported a fault since the last check: a several-workgroups FPS that gave up on a sibling
    (the samples of such a cloud are index 0 from there on — plausible-looking, degenerate geometry), or the tripwire: a sample's
    own min-distance was not zeroed by its update (a lost update: round 4's packed-fp32 fault; that launch's indices are wrong,
    and lanes that do not own a sample may have lost theirs unseen).  Called where the host synchronises anyway (the end of
    `compute_all_metrics`, `get_point_final`, the epoch loop's periodic loss read, the trainer's periodic flag check)."""
    global _fps_faults_seen
    if torch.cuda.is_current_stream_capturing():
        return
    n = fps_faults()
    if n > _fps_faults_seen:
        new, _fps_faults_seen = n - _fps_faults_seen, n
        raise RuntimeError("cpfn_amd: farthest-point sampling reported %d fault(s) before %s: either the workgroups of one large "
                           "cloud were not co-resident (a sibling never arrived; that cloud's remaining samples are index 0), or a "
                           "sample's distance update was LOST on the lane that owns it (the tripwire of csrc/sampling.hip: packed "
                           "fp32 beside a weight-gradient workgroup on some MI355X boxes, DESIGN.md section 4) — results since the "
                           "last check are suspect" % (new, where))



def fps(xyz, num_samples, start=None, skip_near_origin=False):
    """xyz [B,N,3] f32, start [B] i32 or None (-> index 0) -> idx [B,S] i32."""
    _chk(xyz, "xyz", torch.float32)
    B, N, _ = xyz.shape
    if start is not None:
        _chk(start, "start", torch.int32)
    out = torch.empty(B, num_samples, dtype=torch.int32, device=xyz.device)
    scratch = torch.empty(B, N, dtype=torch.float32, device=xyz.device) if N > 8192 else None
    with torch.cuda.device(xyz.device):
        _l.check(_l.lib().cpfn_fps(_ptr(xyz), B, N, int(num_samples), _ptr(start),
                                   1 if skip_near_origin else 0, _ptr(out), _ptr(scratch), _stream()), "cpfn_fps")
    _l.add_bytes("cpfn_fps", 12 * B * N + 4 * B * int(num_samples))
    return out


def fps_centres(xyz, num_samples, start=None, skip_near_origin=False):
    """fps + the sampled centres xyz[b, idx] (what select_point_subset gathers right after the sampling,
    pointset_abstraction.py:50) -> (idx [B,S] i32, centres [B,S,3] f32).  One launch for clouds the one-workgroup-per-cloud
    kernels take (N <= 8192: they hold every sample's coordinates anyway); sampling + gather_rows beyond."""
    _chk(xyz, "xyz", torch.float32)
    B, N, _ = xyz.shape
    if N > _l.lib().cpfn_fps_max_resident():
        idx = fps(xyz, num_samples, start, skip_near_origin)
        return idx, gather_rows(xyz, idx)
    if start is not None:
        _chk(start, "start", torch.int32)
    out = torch.empty(B, num_samples, dtype=torch.int32, device=xyz.device)
    ctr = torch.empty(B, num_samples, 3, dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _l.check(_l.lib().cpfn_fps_centres(_ptr(xyz), B, N, int(num_samples), _ptr(start), 1 if skip_near_origin else 0, _ptr(out),
                                           _ptr(ctr), _stream()), "cpfn_fps_centres")
    _l.add_bytes("cpfn_fps", 12 * B * N + 16 * B * int(num_samples))
    return out, ctr


def ball_query_rel(new_xyz, xyz, radius, nsample, cuda_route=False):
    """ball_query + group_xyz_centered -> (idx [B,S,K] i32, rel [B,S,K,3] f32 = xyz[idx] - new_xyz).  Beside a training step
    (background_geometry; the wave-per-query kernel on the packed cloud) ONE launch: a lane that keeps a neighbour holds its
    coordinates."""
    _chk(new_xyz, "new_xyz", torch.float32)
    _chk(xyz, "xyz", torch.float32)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    if cuda_route or not (_background[0] and N >= 512):
        idx = ball_query(new_xyz, xyz, radius, nsample, cuda_route=cuda_route)
        return idx, group_xyz_centered(xyz, new_xyz, idx)
    out = torch.empty(B, S, int(nsample), dtype=torch.int32, device=xyz.device)
    rel = torch.empty(B, S, int(nsample), 3, dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        packed = torch.empty(B, N, 4, dtype=torch.float32, device=xyz.device)
        _l.check(_l.lib().cpfn_pack_xyzn(_ptr(xyz), B, N, _ptr(packed), _stream()), "cpfn_pack_xyzn")
        _l.check(_l.lib().cpfn_ball_query_packed_rel(_ptr(packed), _ptr(new_xyz), B, N, S, ball_query_threshold(radius),
                                                     int(nsample), _ptr(out), _ptr(rel), _stream()), "cpfn_ball_query_packed_rel")
    _l.add_bytes("cpfn_ball_query", 12 * B * (N + S) + 16 * B * S * int(nsample))
    return out, rel


def ball_query(new_xyz, xyz, radius, nsample, cuda_route=False):
    """new_xyz [B,S,3], xyz [B,N,3] -> idx [B,S,K] i32 (argument order of the
    reference's cuda_ops.ball_query, ball_query.cpp).  cuda_route: the CUDA kernel's direct
    (q-p)^2 < radius*radius test (ball_query_gpu.cu:21-31) instead of the CPU route's expanded distance."""
    _chk(new_xyz, "new_xyz", torch.float32)
    _chk(xyz, "xyz", torch.float32)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    out = torch.empty(B, S, int(nsample), dtype=torch.int32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        if cuda_route:
            _l.check(_l.lib().cpfn_ball_query_direct(_ptr(xyz), _ptr(new_xyz), B, N, S, float(radius), int(nsample),
                                                     _ptr(out), _stream()), "cpfn_ball_query_direct")
        elif _background[0] and N >= 512:
            # beside a training step: the wave-per-query kernel on the cloud packed with its norms (less work per point)
            pk = torch.empty(B, N, 4, dtype=torch.float32, device=xyz.device)
            _l.check(_l.lib().cpfn_pack_xyzn(_ptr(xyz), B, N, _ptr(pk), _stream()), "cpfn_pack_xyzn")
            _l.check(_l.lib().cpfn_ball_query_packed(_ptr(pk), _ptr(new_xyz), B, N, S, ball_query_threshold(radius),
                                                     int(nsample), _ptr(out), _stream()), "cpfn_ball_query_packed")
        else:
            _l.check(_l.lib().cpfn_ball_query(_ptr(xyz), _ptr(new_xyz), B, N, S, ball_query_threshold(radius),
                                              int(nsample), _ptr(out), _stream()), "cpfn_ball_query")
    _l.add_bytes("cpfn_ball_query", 12 * B * (N + S) + 4 * B * S * int(nsample))
    return out


def three_nn(unknown, known, cuda_route=False, sqrt=False):
    """unknown [B,N,3] queries, known [B,M,3] -> (dist2 [B,N,3] f32, idx [B,N,3] i32).
    cuda_route: the CUDA kernel's direct distance (interpolate_gpu.cu:34); sqrt (cuda_route only): the square
    roots, which is what the reference's fast=True wrapper returns (modules/geometry_utils.py:184)."""
    _chk(unknown, "unknown", torch.float32)
    _chk(known, "known", torch.float32)
    B, N, _ = unknown.shape
    M = known.shape[1]
    d = torch.empty(B, N, 3, dtype=torch.float32, device=unknown.device)
    i = torch.empty(B, N, 3, dtype=torch.int32, device=unknown.device)
    with torch.cuda.device(unknown.device):
        if cuda_route:
            _l.check(_l.lib().cpfn_three_nn_direct(_ptr(unknown), _ptr(known), B, N, M, 1 if sqrt else 0, _ptr(d), _ptr(i),
                                                   _stream()), "cpfn_three_nn_direct")
        else:
            if sqrt:
                raise RuntimeError("sqrt distances exist on the CUDA route only")
            _l.check(_l.lib().cpfn_three_nn(_ptr(unknown), _ptr(known), B, N, M, _ptr(d), _ptr(i), _stream()),
                     "cpfn_three_nn")
    _l.add_bytes("cpfn_three_nn", 12 * B * (N + M) + 24 * B * N)
    return d, i


def three_nn_weights(unknown, known, cuda_route=False, sqrt=False):
    """three_nn + three_weights of the distances it returns in ONE launch -> (dist [B,N,3], idx [B,N,3] i32, w [B,N,3])."""
    _chk(unknown, "unknown", torch.float32)
    _chk(known, "known", torch.float32)
    if sqrt and not cuda_route:
        raise RuntimeError("sqrt distances exist on the CUDA route only")
    B, N, _ = unknown.shape
    M = known.shape[1]
    d = torch.empty(B, N, 3, dtype=torch.float32, device=unknown.device)
    i = torch.empty(B, N, 3, dtype=torch.int32, device=unknown.device)
    w = torch.empty(B, N, 3, dtype=torch.float32, device=unknown.device)
    with torch.cuda.device(unknown.device):
        _l.check(_l.lib().cpfn_three_nn_weights(_ptr(unknown), _ptr(known), B, N, M, 1 if cuda_route else 0, 1 if sqrt else 0,
                                                _ptr(d), _ptr(i), _ptr(w), _stream()), "cpfn_three_nn_weights")
    _l.add_bytes("cpfn_three_nn", 12 * B * (N + M) + 36 * B * N)
    return d, i, w


def pairwise_sqdist(src, dst):
    """src [B,N,3], dst [B,M,3] -> [B,N,M] f32."""
    _chk(src, "src", torch.float32); _chk(dst, "dst", torch.float32)
    B, N, _ = src.shape
    M = dst.shape[1]
    out = torch.empty(B, N, M, dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        _l.check(_l.lib().cpfn_pairwise_sqdist(_ptr(src), _ptr(dst), B, N, M, _ptr(out), _stream()),
                 "cpfn_pairwise_sqdist")
    return out


def three_weights(dist):
    _chk(dist, "dist", torch.float32)
    w = torch.empty_like(dist)
    with torch.cuda.device(dist.device):
        _l.check(_l.lib().cpfn_three_weights(_ptr(dist), dist.numel() // 3, _ptr(w), _stream()), "cpfn_three_weights")
    _l.add_bytes("cpfn_three_weights", 8 * dist.numel())
    return w


# ----------------------------------------------------------- channel-major (drop-in) ops
def three_interp_fwd(feats, idx, w):
    _chk(feats, "points", torch.float32); _chk(idx, "idx", torch.int32); _chk(w, "weight", torch.float32)
    B, C, M = feats.shape
    N = idx.shape[1]
    out = torch.empty(B, C, N, dtype=torch.float32, device=feats.device)
    with torch.cuda.device(feats.device):
        _l.check(_l.lib().cpfn_three_interp_fwd(_ptr(feats), _ptr(idx), _ptr(w), B, C, M, N, _ptr(out), _stream()),
                 "cpfn_three_interp_fwd")
    return out


def three_interp_bwd(grad_out, idx, w, M):
    _chk(grad_out, "grad_out", torch.float32); _chk(idx, "idx", torch.int32); _chk(w, "weight", torch.float32)
    B, C, N = grad_out.shape
    out = torch.zeros(B, C, int(M), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _l.check(_l.lib().cpfn_three_interp_bwd(_ptr(grad_out), _ptr(idx), _ptr(w), B, C, N, int(M), _ptr(out),
                                                _stream()), "cpfn_three_interp_bwd")
    return out


def group_fwd(points, idx):
    """points [B,C,N], idx [B,S,K] or [B,S] -> [B,C,S,K] / [B,C,S]."""
    _chk(points, "points", torch.float32); _chk(idx, "idx", torch.int32)
    B, C, N = points.shape
    S = idx.shape[1]
    K = idx.shape[2] if idx.dim() == 3 else 1
    out = torch.empty((B, C, S, K) if idx.dim() == 3 else (B, C, S), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        _l.check(_l.lib().cpfn_group_fwd(_ptr(points), _ptr(idx), B, C, N, S, K, _ptr(out), _stream()),
                 "cpfn_group_fwd")
    return out


def group_bwd(grad_out, idx, N):
    _chk(grad_out, "grad_out", torch.float32); _chk(idx, "idx", torch.int32)
    B, C = grad_out.shape[:2]
    S = idx.shape[1]
    K = idx.shape[2] if idx.dim() == 3 else 1
    out = torch.zeros(B, C, int(N), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _l.check(_l.lib().cpfn_group_bwd(_ptr(grad_out), _ptr(idx), B, C, int(N), S, K, _ptr(out), _stream()),
                 "cpfn_group_bwd")
    return out


# ----------------------------------------------------------- points-major (native) ops
def gather_rows(rows, idx):
    """rows [B,N,C] (any 2/4-byte dtype with C*itemsize % 4 == 0), idx [B,...] i32 -> [B,...,C]."""
    _chk(idx, "idx", torch.int32)
    if not (rows.is_cuda and rows.is_contiguous()):
        raise RuntimeError("rows must be a contiguous CUDA(HIP) tensor")
    B, N, C = rows.shape
    R = idx[0].numel()
    out = torch.empty((B,) + tuple(idx.shape[1:]) + (C,), dtype=rows.dtype, device=rows.device)
    with torch.cuda.device(rows.device):
        _l.check(_l.lib().cpfn_gather_rows(_ptr(rows), _ptr(idx), B, N, R, C * rows.element_size(), _ptr(out),
                                           _stream()), "cpfn_gather_rows")
    _l.add_bytes("cpfn_gather_rows", rows.element_size() * (B * N * C + B * R * C) + 4 * B * R)
    return out


def scatter_add_rows(grad_out, idx, N):
    """grad_out [B,...,C] f32, idx [B,...] -> [B,N,C] f32."""
    _chk(grad_out, "grad_out", torch.float32); _chk(idx, "idx", torch.int32)
    B = grad_out.shape[0]
    C = grad_out.shape[-1]
    R = idx[0].numel()
    out = torch.zeros(B, int(N), C, dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _l.check(_l.lib().cpfn_scatter_add_rows_f32(_ptr(grad_out), _ptr(idx), B, int(N), R, C, _ptr(out), _stream()),
                 "cpfn_scatter_add_rows_f32")
    return out


def group_xyz_centered(xyz, new_xyz, idx):
    """xyz [B,N,3], new_xyz [B,S,3], idx [B,S,K] -> [B,S,K,3]."""
    _chk(xyz, "xyz", torch.float32); _chk(new_xyz, "new_xyz", torch.float32); _chk(idx, "idx", torch.int32)
    B, N, _ = xyz.shape
    S, K = idx.shape[1:]
    out = torch.empty(B, S, K, 3, dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _l.check(_l.lib().cpfn_group_xyz_centered(_ptr(xyz), _ptr(new_xyz), _ptr(idx), B, N, S, K, _ptr(out),
                                                  _stream()), "cpfn_group_xyz_centered")
    _l.add_bytes("cpfn_group_xyz_centered", 12 * B * (N + S) + 16 * B * S * K)
    return out


def interp_rows_fwd(feats, idx, w):
    """feats [B,M,C] f32, idx/w [B,N,3] -> [B,N,C]."""
    _chk(feats, "feats", torch.float32); _chk(idx, "idx", torch.int32); _chk(w, "weight", torch.float32)
    B, M, C = feats.shape
    N = idx.shape[1]
    out = torch.empty(B, N, C, dtype=torch.float32, device=feats.device)
    with torch.cuda.device(feats.device):
        _l.check(_l.lib().cpfn_interp_rows_fwd(_ptr(feats), _ptr(idx), _ptr(w), B, M, N, C, _ptr(out), _stream()),
                 "cpfn_interp_rows_fwd")
    return out


def interp_rows_bwd(grad_out, idx, w, M):
    _chk(grad_out, "grad_out", torch.float32); _chk(idx, "idx", torch.int32); _chk(w, "weight", torch.float32)
    B, N, C = grad_out.shape
    out = torch.zeros(B, int(M), C, dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _l.check(_l.lib().cpfn_interp_rows_bwd(_ptr(grad_out), _ptr(idx), _ptr(w), B, int(M), N, C, _ptr(out),
                                               _stream()), "cpfn_interp_rows_bwd")
    return out


# The inverse index (csrc/gather.hip), three builds of the same ascending lists:
#   "ordered" (CPFN_CSR_THREADS < 0, the default since round 5): count / scan / in-order LDS-atomic scatter, verified in the kernel;
#   the stable radix sort (CPFN_CSR_THREADS = 0 | 512 | 1024; also where the ordered form's LDS slab does not fit);
#   count + scatter + per-list sort (rounds 1-4: CPFN_CSR_RADIX=0, and E > 32768).
# The step's three launches, stand-alone: 47 us (8 waves per cloud) / 106 us / 367 us; the step itself: -20 us / 0 / 0 (NOTEBOOK
# R5.4, R5.7).
CSR_RADIX = __import__("os").environ.get("CPFN_CSR_RADIX", "1") != "0"
CSR_THREADS = int(__import__("os").environ.get("CPFN_CSR_THREADS", "-8"))      # (ordered: -1 = 4 waves per cloud, -8 = 8, -16 = 16)


def _csr_ordered_fits(E, M):
    waves = {-8: 8, -16: 16}.get(CSR_THREADS, 4)
    return 4 * ((waves + 1) * M + 1) + 4 * E + 64 <= 150 * 1024 and M <= 65536 and E <= 65536


_CSR_FALLBACKS = {}


def _csr_fallback_word(device):
    w = _CSR_FALLBACKS.get(device)
    if w is None:
        w = _CSR_FALLBACKS[device] = torch.zeros(1, dtype=torch.int32, device=device)
    return w


def csr_fallbacks(device=None):
    """Clouds whose "ordered" inverse index (CPFN_CSR_THREADS < 0) failed its in-kernel verification and was sorted by the
    fall-back loop — 0 on every MI355X seen so far; the result is ascending either way (csrc/gather.hip)."""
    return sum(int(w.item()) for d, w in _CSR_FALLBACKS.items() if device is None or d == device)


def csr_build(idx, M):
    """idx [B, ...] i32 with values in [0, M) -> (offsets [B, M+1] i32, entries [B, E] i32): for every
    target m the ascending list of flattened source positions that reference it."""
    _chk(idx, "idx", torch.int32)
    B = idx.shape[0]
    E = idx[0].numel()
    off = torch.empty(B, M + 1, dtype=torch.int32, device=idx.device)
    ent = torch.empty(B, E, dtype=torch.int32, device=idx.device)
    with torch.cuda.device(idx.device):
        if CSR_RADIX and 0 < E <= 32768:
            threads = CSR_THREADS
            if threads < 0 and _csr_ordered_fits(E, int(M)):      # no scratch; the word counts the clouds whose scatter had to be sorted after all
                ws = _csr_fallback_word(idx.device)
            else:
                threads = 1024 if threads < 0 else threads
                ws = torch.empty(B, E, dtype=torch.int32, device=idx.device)       # (scratch of the radix passes)
            _l.check(_l.lib().cpfn_csr_build_ws(_ptr(idx), B, E, int(M), _ptr(off), _ptr(ent), _ptr(ws), threads, _stream()),
                     "cpfn_csr_build_ws")
        else:
            _l.check(_l.lib().cpfn_csr_build(_ptr(idx), B, E, int(M), _ptr(off), _ptr(ent), _stream()), "cpfn_csr_build")
    _l.add_bytes("cpfn_csr_build", 8 * B * E + 4 * B * (M + 1))
    return off, ent
