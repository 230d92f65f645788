"""The reference's CPU geometry route restated op for op in torch (NOT the C restatement of oracle/geometry.py): what
`fast=False` executes in the reference — a 512-iteration FPS loop of tensor ops, the full [S, N] distance matrix with a sort for
the ball query, a full sort for the 3-NN.

TEST INFRASTRUCTURE ONLY — see ``oracle/__init__.py``.  It exists for ONE purpose: `bench.py`'s `cpu_baseline.reference_like`,
a CPU baseline whose geometry costs what the reference's own CPU path costs (the C geometry of the default
oracle makes the port ~2x faster than the reference's torch path on the same step).  Same call signatures as oracle/geometry.py
(numpy in, numpy out); `tests/test_oracle_golden.py` holds its indices to the C oracle's.
"""
import numpy as np
import torch


def pairwise_squared_distance(src, dst):
    """modules/geometry_utils.py:4-23.  src [B,S,3], dst [B,N,3] (points-major here) -> [B,S,N]: -2 src.dst + |src|^2 + |dst|^2,
    in that order."""
    d = -2.0 * torch.matmul(src, dst.transpose(1, 2))                      # :20
    d = d + torch.sum(src ** 2, dim=2).unsqueeze(2)                        # :21
    return d + torch.sum(dst ** 2, dim=2).unsqueeze(1)                     # :22


def farthest_point_sample(xyz, num_point, start):
    """modules/geometry_utils.py:88-101 (the start indices are the caller's: the reference draws them with torch.randint, :92)."""
    p = torch.from_numpy(np.ascontiguousarray(xyz, dtype=np.float32))
    B, N, _ = p.shape
    out = torch.zeros(B, num_point, dtype=torch.long)
    dist = torch.full((B, N), 1e10)
    far = torch.as_tensor(np.asarray(start), dtype=torch.long).clone()
    rows = torch.arange(B)
    for i in range(num_point):                                             # :94
        out[:, i] = far                                                    # :95
        c = p[rows, far].unsqueeze(1)                                      # :96
        d = torch.sum((p - c) ** 2, dim=2)                                 # :97
        m = d < dist                                                       # :98
        dist[m] = d[m]                                                     # :99
        far = torch.max(dist, dim=1)[1]                                    # :100
    return out.numpy()


def ball_query(radius, num_samples, xyz, new_xyz):
    """modules/geometry_utils.py:151-161: indices whose squared distance exceeds r**2 are replaced by N, the rest sorted ascending,
    the first K kept, short rows padded with their first entry."""
    p = torch.from_numpy(np.ascontiguousarray(xyz, dtype=np.float32))
    q = torch.from_numpy(np.ascontiguousarray(new_xyz, dtype=np.float32))
    B, N, _ = p.shape
    S = q.shape[1]
    idx = torch.arange(N, dtype=torch.long).view(1, 1, N).repeat(B, S, 1)  # :153
    d = pairwise_squared_distance(q, p)                                    # :155
    idx[d > radius ** 2] = N                                               # :156
    idx = idx.sort(dim=2)[0][:, :, :num_samples]                           # :157
    first = idx[:, :, 0:1].repeat(1, 1, num_samples)                       # :158
    mask = idx == N                                                        # :159
    idx[mask] = first[mask]                                                # :160
    return idx.numpy()


def three_nn(unknown, known):
    """modules/geometry_utils.py:212-215: full ascending sort of the squared distances, first three."""
    u = torch.from_numpy(np.ascontiguousarray(unknown, dtype=np.float32))
    k = torch.from_numpy(np.ascontiguousarray(known, dtype=np.float32))
    d, i = pairwise_squared_distance(u, k).sort(dim=2)                     # :213
    return d[:, :, :3].contiguous().numpy(), i[:, :, :3].contiguous().numpy()
