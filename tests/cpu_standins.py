"""TEST INFRASTRUCTURE: CPU stand-ins for the HIP data movers of cpfn_amd.ops, built on the oracle (oracle/geometry.py)
and stock torch indexing, so that the HOST logic of the product — the real PointNet2 module tree in fp32 mode, the
trainer, the flat gradient bucket and the data-parallel exchange — can be exercised in multi-process CPU tests (gloo).
The product itself has no CPU path and never imports this file; `installed()` monkeypatches cpfn_amd.ops for the
duration of a test only."""
import contextlib

import numpy as np
import torch

from oracle import geometry as og


def _i32(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.int32)


def fps(xyz, num_samples, start=None, skip_near_origin=False):
    assert not skip_near_origin
    s = np.zeros(xyz.shape[0], np.int64) if start is None else start.cpu().numpy().astype(np.int64)
    return _i32(og.farthest_point_sample(xyz.numpy(), int(num_samples), s))


def ball_query(new_xyz, xyz, radius, nsample, cuda_route=False):
    assert not cuda_route
    return _i32(og.ball_query(radius, int(nsample), xyz.numpy(), new_xyz.numpy()))


def three_nn(unknown, known, cuda_route=False, sqrt=False):
    assert not cuda_route
    d, i = og.three_nn(unknown.numpy(), known.numpy())
    return torch.from_numpy(d), _i32(i)


def three_weights(dist):
    return torch.from_numpy(og.three_weights(dist.numpy()))


def gather_rows(rows, idx):
    B, N, C = rows.shape
    flat = idx.reshape(B, -1).long()
    out = torch.gather(rows, 1, flat.unsqueeze(2).expand(B, flat.shape[1], C))
    return out.reshape(tuple(idx.shape) + (C,))


def fps_centres(xyz, num_samples, start=None, skip_near_origin=False):
    idx = fps(xyz, num_samples, start, skip_near_origin)
    return idx, gather_rows(xyz, idx)


def ball_query_rel(new_xyz, xyz, radius, nsample, cuda_route=False):
    idx = ball_query(new_xyz, xyz, radius, nsample, cuda_route)
    return idx, group_xyz_centered(xyz, new_xyz, idx)


def three_nn_weights(unknown, known, cuda_route=False, sqrt=False):
    d, i = three_nn(unknown, known, cuda_route, sqrt)
    return d, i, three_weights(d)


def scatter_add_rows(grad_out, idx, N):
    B, C = grad_out.shape[0], grad_out.shape[-1]
    flat = idx.reshape(B, -1).long()
    out = torch.zeros(B, int(N), C, dtype=torch.float32)
    return out.scatter_add_(1, flat.unsqueeze(2).expand(B, flat.shape[1], C), grad_out.reshape(B, -1, C))


def group_xyz_centered(xyz, new_xyz, idx):
    return gather_rows(xyz, idx) - new_xyz.unsqueeze(2)


def interp_rows_fwd(feats, idx, w):
    return (gather_rows(feats, idx) * w.unsqueeze(3)).sum(2)


def interp_rows_bwd(grad_out, idx, w, M):
    B, N, C = grad_out.shape
    contrib = (grad_out.unsqueeze(2) * w.unsqueeze(3)).reshape(B, N * 3, C)
    return scatter_add_rows(contrib, idx.reshape(B, N * 3), M)


def csr_build(idx, M):
    B = idx.shape[0]
    return torch.zeros(B, M + 1, dtype=torch.int32), torch.zeros(B, idx[0].numel(), dtype=torch.int32)


_NAMES = ("fps", "fps_centres", "ball_query", "ball_query_rel", "three_nn", "three_nn_weights", "three_weights", "gather_rows", "scatter_add_rows", "group_xyz_centered",
          "interp_rows_fwd", "interp_rows_bwd", "csr_build")


@contextlib.contextmanager
def installed():
    from cpfn_amd import ops
    saved = {n: getattr(ops, n) for n in _NAMES}
    for n in _NAMES:
        setattr(ops, n, globals()[n])
    try:
        yield
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
