"""ctypes binding of libcpfn_hip.so (the C ABI declared in include/cpfn_hip.h).

There is deliberately NO fallback: if the shared object is missing or a symbol does
not resolve, importing/using the product raises — a silent CPU or eager-PyTorch
path would void every parity and performance claim.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libcpfn_hip.so")

_vp, _i, _f, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/cpfn_hip.h
SIGNATURES = {
    "cpfn_abi_version": [],
    "cpfn_build_info": [],
    "cpfn_fps": [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp],
    "cpfn_ball_query": [_vp, _vp, _i, _i, _i, _f, _i, _vp, _vp],
    "cpfn_three_nn": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "cpfn_pairwise_sqdist": [_vp, _vp, _i, _i, _i, _vp, _vp],
    "cpfn_three_weights": [_vp, _i64, _vp, _vp],
    "cpfn_three_interp_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "cpfn_three_interp_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "cpfn_group_fwd": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "cpfn_group_bwd": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "cpfn_gather_rows": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "cpfn_scatter_add_rows_f32": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "cpfn_group_xyz_centered": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "cpfn_interp_rows_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "cpfn_interp_rows_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "cpfn_fit_num_chunks": [_i, _i],
    "cpfn_fit_moments_fwd": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "cpfn_fit_moments_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "cpfn_cone_pass_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "cpfn_cone_pass_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "cpfn_eigh3": [_vp, _i64, _vp, _vp, _vp],
}
_RESTYPES = {"cpfn_build_info": ctypes.c_char_p}

_lib = None


class CpfnHipError(RuntimeError):
    pass


def lib():
    """Load the shared object once and attach every prototype.  Raises if it is
    missing (build it with `python -m cpfn_amd.build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise CpfnHipError(
                "%s not found: the HIP extension is not built. Run `python -m cpfn_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % SO_PATH)
        h = ctypes.CDLL(SO_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError if the .so is stale: loud by design
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, ctypes.c_int)
        if h.cpfn_abi_version() != 1:
            raise CpfnHipError("libcpfn_hip.so ABI version mismatch")
        _lib = h
    return _lib


def check(status, what):
    if status != 0:
        raise CpfnHipError("%s failed with status %d%s" % (
            what, status, " (invalid argument)" if status == -22 else " (hipError_t)"))
