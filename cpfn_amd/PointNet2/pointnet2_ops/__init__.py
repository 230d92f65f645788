"""Drop-in for the reference's `PointNet2.pointnet2_ops` package: `cuda_ops` here is the
ctypes shim over libcpfn_hip.so (the reference builds a pybind11 CUDA extension of that
name, PointNet2/pointnet2_ops/setup.py:20-28)."""
from ... import cuda_ops  # noqa: F401
