"""ctypes front-end of ``cpfn_oracle.c`` (numpy in, numpy out).

TEST INFRASTRUCTURE ONLY — see ``oracle/__init__.py``.
Each wrapper names the reference lines its C body restates.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcpfn_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile the C oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "cpfn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcpfn_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _pf(a):
    return a.ctypes.data_as(_f32p)


def _pi(a):
    return a.ctypes.data_as(_i64p)


def pairwise_squared_distance(src, dst):
    """modules/geometry_utils.py:4-23.  src [B,N,3], dst [B,M,3] -> [B,N,M]."""
    src, dst = _f(src), _f(dst)
    B, N, _ = src.shape
    M = dst.shape[1]
    out = np.empty((B, N, M), np.float32)
    lib().orc_pairwise_sqdist(_pf(src), _pf(dst), B, N, M, _pf(out))
    return out


def farthest_point_sample(xyz, num_point, start):
    """modules/geometry_utils.py:88-101.  xyz [B,N,3], start [B] -> idx [B,S] int64."""
    xyz, start = _f(xyz), _i(start)
    B, N, _ = xyz.shape
    out = np.empty((B, num_point), np.int64)
    lib().orc_fps(_pf(xyz), B, N, int(num_point), _pi(start), _pi(out))
    return out


def ball_query_threshold(radius):
    """The fp32 value the reference's `sqrdists > radius ** 2` compares against
    (modules/geometry_utils.py:156): the Python double radius**2 rounded to f32."""
    return np.float32(float(radius) ** 2)


def ball_query(radius, num_samples, xyz, new_xyz):
    """modules/geometry_utils.py:151-161.  xyz [B,N,3], new_xyz [B,S,3] -> [B,S,K] int64."""
    xyz, new_xyz = _f(xyz), _f(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    out = np.empty((B, S, num_samples), np.int64)
    lib().orc_ball_query(_pf(xyz), _pf(new_xyz), B, N, S,
                         ctypes.c_float(ball_query_threshold(radius)), int(num_samples), _pi(out))
    return out


def three_nn(unknown, known):
    """modules/geometry_utils.py:212-215.  unknown [B,N,3] (queries), known [B,M,3]
    -> (squared dist [B,N,3] f32, idx [B,N,3] int64)."""
    unknown, known = _f(unknown), _f(known)
    B, N, _ = unknown.shape
    M = known.shape[1]
    d = np.empty((B, N, 3), np.float32)
    i = np.empty((B, N, 3), np.int64)
    lib().orc_three_nn(_pf(unknown), _pf(known), B, N, M, _pf(d), _pi(i))
    return d, i


def three_weights(dist):
    """modules/pointset_feature_propagation.py:40-42."""
    dist = _f(dist)
    w = np.empty_like(dist)
    lib().orc_three_weights(_pf(dist), ctypes.c_int64(dist.size // 3), _pf(w))
    return w


def three_weighted_sum(feats, idx, w):
    """modules/geometry_utils.py:281-283.  feats [B,C,M], idx/w [B,N,3] -> [B,C,N]."""
    feats, idx, w = _f(feats), _i(idx), _f(w)
    B, C, M = feats.shape
    N = idx.shape[1]
    out = np.empty((B, C, N), np.float32)
    lib().orc_three_weighted_sum(_pf(feats), _pi(idx), _pf(w), B, C, M, N, _pf(out))
    return out


def three_weighted_sum_grad(grad_out, idx, w, M):
    """Adjoint of three_weighted_sum w.r.t. feats.  grad_out [B,C,N] -> [B,C,M]."""
    grad_out, idx, w = _f(grad_out), _i(idx), _f(w)
    B, C, N = grad_out.shape
    out = np.empty((B, C, M), np.float32)
    lib().orc_three_weighted_sum_grad(_pf(grad_out), _pi(idx), _pf(w), B, C, N, int(M), _pf(out))
    return out


def group_points(points, idx):
    """select_point_subset, modules/geometry_utils.py:26-44.
    points [B,C,N]; idx [B,S] or [B,S,K] -> [B,C,S] or [B,C,S,K]."""
    points, idx = _f(points), _i(idx)
    B, C, N = points.shape
    squeeze = idx.ndim == 2
    idx3 = idx[:, :, None] if squeeze else idx
    idx3 = np.ascontiguousarray(idx3)
    S, K = idx3.shape[1:]
    out = np.empty((B, C, S, K), np.float32)
    lib().orc_group_points(_pf(points), _pi(idx3), B, C, N, S, K, _pf(out))
    return out[..., 0] if squeeze else out


def group_points_grad(grad_out, idx, N):
    """Adjoint of group_points.  grad_out [B,C,S(,K)] -> [B,C,N]."""
    grad_out, idx = _f(grad_out), _i(idx)
    if idx.ndim == 2:
        idx = idx[:, :, None]
        grad_out = grad_out[..., None]
    idx = np.ascontiguousarray(idx)
    grad_out = np.ascontiguousarray(grad_out)
    B, C, S, K = grad_out.shape
    out = np.empty((B, C, N), np.float32)
    lib().orc_group_points_grad(_pf(grad_out), _pi(idx), B, C, int(N), S, K, _pf(out))
    return out


def num_threads():
    return int(lib().orc_num_threads())


# ---------------------------------------------------------------- CUDA route (PARITY UNPINNED)
# Restatements of the reference's compiled `fast=True` kernels; they cannot be checked against
# a CUDA build here (see the header of that section in cpfn_oracle.c).
def ball_query_cuda(radius, num_samples, xyz, new_xyz):
    """cuda_ops/src/ball_query_gpu.cu:9-44.  xyz [B,N,3], new_xyz [B,S,3] -> [B,S,K] int64."""
    xyz, new_xyz = _f(xyz), _f(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    out = np.empty((B, S, num_samples), np.int64)
    lib().orc_ball_query_direct(_pf(xyz), _pf(new_xyz), B, N, S, ctypes.c_float(radius), int(num_samples), _pi(out))
    return out


def three_nn_cuda(unknown, known, sqrt=True):
    """cuda_ops/src/interpolate_gpu.cu:9-59 (+ the sqrt of modules/geometry_utils.py:184)."""
    unknown, known = _f(unknown), _f(known)
    B, N, _ = unknown.shape
    M = known.shape[1]
    d = np.empty((B, N, 3), np.float32)
    i = np.empty((B, N, 3), np.int64)
    lib().orc_three_nn_direct(_pf(unknown), _pf(known), B, N, M, 1 if sqrt else 0, _pf(d), _pi(i))
    return d, i


def farthest_point_sample_cuda(xyz, num_point):
    """cuda_ops/src/sampling_gpu.cu:63-159: start 0, near-origin points skipped."""
    xyz = _f(xyz)
    B, N, _ = xyz.shape
    out = np.empty((B, num_point), np.int64)
    lib().orc_fps_cuda(_pf(xyz), B, N, int(num_point), _pi(out))
    return out
