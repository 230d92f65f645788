"""ctypes binding of libcpfn_hip.so (the C ABI declared in include/cpfn_hip.h).

There is deliberately NO fallback: if the shared object is missing or a symbol does
not resolve, importing/using the product raises — a silent CPU or eager-PyTorch
path would void every parity and performance claim.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libcpfn_hip.so")
ABI_VERSION = 3          # = CPFN_ABI_VERSION of include/cpfn_hip.h; bumped whenever an exported signature changes

_vp, _i, _f, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64
_ll = ctypes.c_longlong

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/cpfn_hip.h
SIGNATURES = {
    "cpfn_abi_version": [],
    "cpfn_build_info": [],
    "cpfn_fps": [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp],
    "cpfn_fps_centres": [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp],
    "cpfn_fps_max_resident": [],
    "cpfn_fps_faults": [],
    "cpfn_fps_debug_drop": [_i],
    "cpfn_fps_profile": [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp],
    "cpfn_ball_query": [_vp, _vp, _i, _i, _i, _f, _i, _vp, _vp],
    "cpfn_three_nn": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "cpfn_three_nn_weights": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "cpfn_ball_query_direct": [_vp, _vp, _i, _i, _i, _f, _i, _vp, _vp],
    "cpfn_set_background_geometry": [_i],
    "cpfn_pack_xyzn": [_vp, _i, _i, _vp, _vp], "cpfn_ball_query_packed": [_vp, _vp, _i, _i, _i, _f, _i, _vp, _vp],
    "cpfn_ball_query_packed_rel": [_vp, _vp, _i, _i, _i, _f, _i, _vp, _vp, _vp],
    "cpfn_three_nn_direct": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
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
    "cpfn_interp_rows_bf16": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "cpfn_concat_interp_bf16": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "cpfn_colsum_rows_bf16": [_vp, _i, _i, _i, _i, _vp, _vp],
    "cpfn_colsum_rows_pass1_bf16": [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "cpfn_scatter_rows_bf16": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "cpfn_group_concat_bf16": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "cpfn_multi_copy": [_vp, _i, _vp],
    "cpfn_multi_copy_blocks": [_vp, _i],
    "cpfn_multi_copy_checked": [_vp, _i, _vp, _i, _vp],
    "cpfn_multi_cast": [_vp, _i, _vp],
    "cpfn_concat_pos_feats_bf16": [_vp, _vp, _ll, _i, _i, _vp, _vp],
    "cpfn_count_labels": [_vp, _i, _i, _vp, _vp],
    "cpfn_csr_build": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "cpfn_csr_build_ws": [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp],
    "cpfn_csr_gather_sum_bf16": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "cpfn_csr_gather_sum_add_bf16": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp],
    "cpfn_fit_num_chunks": [_i, _i],
    "cpfn_fit_moments_fwd": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "cpfn_fit_moments_fwd_match": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "cpfn_fit_moments_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "cpfn_cone_pass_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "cpfn_cone_pass_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp],
    "cpfn_fit_algebra_fwd": [_vp, _i64, _vp, _vp, _vp],
    "cpfn_fit_algebra_bwd": [_vp, _vp, _vp, _i64, _vp, _vp, _vp],
    "cpfn_fit_moments_algebra_fwd": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cpfn_fit_params_bwd_cone": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "cpfn_fit_params_bwd_algebra": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "cpfn_fit_pack_fwd": [_vp, _vp, _vp, _i64, _vp, _vp],
    "cpfn_fit_pack_fwd_partials": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "cpfn_fit_pack_bwd": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp],
    "cpfn_nonfinite_flag": [_vp, _ll, _vp, _vp, _vp],
    "cpfn_adam_flat": [_vp, _vp, _vp, _vp, _ll, _vp, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "cpfn_adam_flat_xw": [_vp, _vp, _vp, _vp, _ll, _vp, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp],
    "cpfn_adam_flat_sticky": [_vp, _vp, _vp, _vp, _ll, _vp, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp],
    "cpfn_nonfinite_blocks": [_ll],
    "cpfn_nonfinite_partial": [_vp, _ll, _vp, _vp],
    "cpfn_hungarian_match": [_vp, _vp, _i, _i, _vp, _vp],
    "cpfn_p_coverage": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp],
    "cpfn_ce2_blocks": [_ll],
    "cpfn_ce2": [_vp, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp],
    "cpfn_metrics_workspace": [_i, _i, _i, _i],
    "cpfn_metrics_points": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cpfn_metrics_tail": [_vp] * 10 + [_i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp],
    "cpfn_loss_tail": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cpfn_eigh3": [_vp, _i64, _vp, _vp, _vp],
    "cpfn_similarity_soft_workspace": [_i, _i, _i, _i, _i],
    "cpfn_similarity_soft": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "cpfn_label_pool": [_vp, _vp, _ll, _i, _i, _vp, _vp, _vp],
    "cpfn_mlp_gemm_blocks": [_ll, _i],
    "cpfn_mlp_gemm": [_vp, _i, _vp, _vp, _i, _ll, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "cpfn_seam_words": [_i, _i],
    "cpfn_mlp_gemm_seam_ok": [_ll, _i, _i],
    "cpfn_mlp_gemm_seam": [_vp, _vp, _ll, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cpfn_mlp_gemm_xyz_seam": [_vp, _vp, _vp, _vp, _ll, _i, _i, _vp, _vp, _vp],
    "cpfn_mlp_gemm_pool_ok": [_ll, _i, _i, _i],
    "cpfn_mlp_gemm_pool": [_vp, _vp, _ll, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp],
    "cpfn_bn_pool_finish": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cpfn_smallk_fwd_seam": [_vp, _i, _vp, _i, _vp, _ll, _i, _vp, _vp, _vp],
    "cpfn_mlp_gemm_can_fuse_bwd_stats": [_ll, _i, _i],
    "cpfn_mlp_gemm_xyz_ok": [_ll, _i, _i],
    "cpfn_mlp_gemm_xyz": [_vp, _i, _vp, _vp, _vp, _ll, _i, _i, _vp, _i, _vp, _vp],
    "cpfn_flag_wait": [_vp, ctypes.c_uint, ctypes.c_uint64, _vp, _vp, _vp], "cpfn_flag_set": [_vp, ctypes.c_uint, _vp],
    "cpfn_flag_set_payload": [_vp, ctypes.c_uint, _vp, _vp, _i, _vp],
    "cpfn_mlp_gemm_set_probe": [_vp, _i, _i],
    "cpfn_wall_clock_khz": [_i],
    "cpfn_stamp": [_vp, _vp],
    "cpfn_bn_finalize": [_vp, _i, _i, _f, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cpfn_bn_eval_affine": [_vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _vp],
    "cpfn_bn_relu_apply": [_vp, _vp, _vp, _ll, _i, _vp, _vp, ctypes.c_uint64, _f, _vp, _vp],
    "cpfn_bn_relu_maxpool": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "cpfn_bn_bwd_blocks": [_ll],
    "cpfn_bn_relu_bwd": [_vp, _vp, _vp, _vp, _ll, _i, _vp, _vp, _vp, _f, _vp],
    "cpfn_bn_relu_bwd_join": [_vp, _i, _vp, _i, _vp, _vp, _vp, _ll, _i, _vp, _vp, _vp],
    "cpfn_bn_bwd_finalize": [_vp, _i, _i, _f, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp],
    "cpfn_bn_bwd_finalize_ride": [_vp, _i, _i, _f, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "cpfn_bn_bwd_apply": [_vp, _vp, _vp, _vp, _vp, _ll, _i, _vp, _vp, _f, _vp],
    "cpfn_bn_pool_bwd_apply": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "cpfn_mlp_wgrad_splits": [_ll, _i, _i],
    "cpfn_multi_split_reduce": [_vp, _i, _vp],
    "cpfn_multi_split_reduce_checked": [_vp, _i, _vp, _vp],
    "cpfn_bn_bwd_finalize_checked": [_vp, _i, _i, _f, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "cpfn_bn_bwd_finalize_ride_checked": [_vp, _i, _i, _f, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "cpfn_mlp_wgrad": [_vp, _i, _vp, _i, _vp, _ll, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "cpfn_mlp_bwd_fused_ok": [_ll, _i, _i],
    "cpfn_mlp_wgrad_apply_ok": [_ll, _i, _i],
    "cpfn_mlp_bwd_small_ok": [_ll, _i, _i],
    "cpfn_mlp_bwd_small_blocks": [_ll],
    "cpfn_mlp_bwd_small": [_vp, _i, _vp, _i, _vp, _ll, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "cpfn_mlp_dgrad_small_ok": [_ll, _i, _i],
    "cpfn_mlp_dgrad_small": [_vp, _vp, _ll, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "cpfn_mlp_bwd_fused": [_vp, _i, _vp, _i, _vp, _ll, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                           _vp, _f, _vp, _vp, _i, _vp, _vp, _vp],
    "cpfn_mlp_gemm_xyz_gather": [_vp, _vp, _i, _i, _vp, _vp, _vp, _ll, _i, _i, _vp, _vp, _vp, _vp],
    "cpfn_mlp_bwd_fused_xt_gather": [_vp, _vp, _vp, _i, _i, _vp, _ll, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cpfn_mlp_bwd_fused_xw": [_vp, _vp, _vp, _ll, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cpfn_colsum_f32": [_vp, _ll, _i, _vp, _vp, _vp, _vp],
    "cpfn_smallk_fwd": [_vp, _i, _vp, _ll, _i, _vp, _vp, _vp],
    "cpfn_smallk_fwd_cast": [_vp, _i, _vp, _i, _vp, _ll, _i, _vp, _vp, _vp],
    "cpfn_smallk_wgrad": [_vp, _vp, _i, _ll, _i, _vp, _vp, _vp],
    "cpfn_smallk_wgrad_apply": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, _i, _vp, _vp, _vp],
    "cpfn_smallk_wgrad_apply_xyz": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, _i, _vp, _vp, _vp],
    "cpfn_head_post_chunks": [_i],
    "cpfn_head_post_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cpfn_head_post_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "cpfn_seg_stats_chunks": [_i, _i],
    "cpfn_seg_stats_fwd": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "cpfn_seg_stats_bwd": [_vp, _vp, _i, _i, _i, _vp, _vp],
    "cpfn_residue_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "cpfn_residue_bwd": [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp],
}
_RESTYPES = {"cpfn_build_info": ctypes.c_char_p, "cpfn_similarity_soft_workspace": ctypes.c_longlong,
             "cpfn_metrics_workspace": ctypes.c_longlong}

_lib = None
_raw = None

# ---- optional per-entry-point device timing (bench.py's roofline leg) -------------------
# When a symbol is listed here, every call is bracketed by two events recorded on torch's
# current stream (the stream the kernels are launched on); nothing synchronises until
# `timed_report()` is called after the timed region.
_timed = {}


_bytes = {}
_census = None      # {symbol: [launches, algorithmic bytes]} while a byte census is running


def time_symbols(names):
    """Enable event timing for the given C-ABI entry points (empty list = off)."""
    _timed.clear()
    _bytes.clear()
    for n in names:
        _timed[n] = []
        _bytes[n] = 0


def add_bytes(name, nbytes):
    """Callers that know a launch's ALGORITHMIC traffic (every operand read once, every result written once) report
    it here; counted only while timing of that symbol or a byte census is on."""
    if name in _bytes:
        _bytes[name] += int(nbytes)
    if _census is not None:
        ent = _census.setdefault(name, [0, 0])
        ent[0] += 1
        ent[1] += int(nbytes)


def timed_bytes(name):
    return _bytes.get(name, 0)


def byte_census(on):
    """Start (True) / stop (False) counting the algorithmic bytes of every instrumented entry point.
    Stop returns {symbol: (launches, bytes)}."""
    global _census
    if on:
        _census = {}
        return None
    out, _census = _census, None
    return {k: tuple(v) for k, v in (out or {}).items()}


def timed_report():
    """-> {symbol: (calls, total_ms)}; synchronises on the recorded events."""
    out = {}
    for n, pairs in _timed.items():
        total = 0.0
        for a, b in pairs:
            b.synchronize()
            total += a.elapsed_time(b)
        out[n] = (len(pairs), total)
    return out


class _Proxy:
    def __init__(self, h):
        self._h = h

    def __getattr__(self, name):
        fn = getattr(self._h, name)

        def call(*args, _fn=fn, _name=name):
            rec = _timed.get(_name)
            if rec is None:
                return _fn(*args)
            import torch
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            r = _fn(*args)
            b.record()
            rec.append((a, b))
            return r

        setattr(self, name, call)
        return call


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
        # The kernels are launched on torch's streams with torch-allocated pointers, so the
        # library must bind to the SAME HIP runtime instance torch uses: load torch (and its
        # bundled libamdhip64) first; our NEEDED libamdhip64.so.7 then resolves to it.
        import torch  # noqa: F401
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(bundled):
            ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)
        h = ctypes.CDLL(SO_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError if the .so is stale: loud by design
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, ctypes.c_int)
        if h.cpfn_abi_version() != ABI_VERSION:
            raise CpfnHipError("libcpfn_hip.so ABI version %d, this package binds version %d: rebuild it "
                               "(python -m cpfn_amd.build)" % (h.cpfn_abi_version(), ABI_VERSION))
        _lib = _Proxy(h)
    return _lib


def check(status, what):
    if status != 0:
        raise CpfnHipError("%s failed with status %d%s" % (
            what, status, " (invalid argument)" if status == -22 else " (hipError_t)"))
