"""Patch merging on the MI355X path: the two tensor functions of the reference's Utils/merging_utils.py that
evaluation_localSPFN.py:101-110 runs on the device, same names, arguments and results.

  similarity_soft(spfn_labels, predicted_labels, point_indices)      (merging_utils.py:6-15)
  get_point_final(point2primitive_prediction, output_labels_heuristic)  (merging_utils.py:56-60)

Both are hand-written HIP (csrc/merging.hip) behind the C ABI (cpfn_similarity_soft, cpfn_label_pool); there is
no CPU or framework fallback — CPU tensors raise, like the reference's native ops ("CPU not supported").
The greedy label solver between the two calls (heuristic_merging / run_heuristic_solver) is numba-jitted host
code in the reference and stays on the host; it is not part of this package.
"""
import torch

from .. import lib as _l
from ..ops import _ptr, _stream


def _need_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError("%s: CPU not supported (cpfn_amd runs on the HIP path only)" % name)


def similarity_soft(spfn_labels, predicted_labels, point_indices):
    """spfn_labels [N, Lo] (any dtype; the reference passes a LongTensor of one-hot rows), predicted_labels
    [nb, npp, Lp] float, point_indices [nb, npp] long (each patch lists a point at most once) ->
    [nb*Lp + Lo, nb*Lp + Lo] fp32: M^T M of the point-to-primitive matrix, which is never materialised."""
    _need_cuda(predicted_labels, "similarity_soft")
    N, Lo = spfn_labels.shape
    nb, npp, Lp = predicted_labels.shape
    dev = predicted_labels.device
    spfn = spfn_labels.to(device=dev, dtype=torch.float32).contiguous()
    pred = predicted_labels.to(torch.float32).contiguous()
    pidx = point_indices.to(device=dev, dtype=torch.int64).contiguous()
    if tuple(pidx.shape) != (nb, npp):
        raise RuntimeError("similarity_soft: point_indices must be [nb, npp]")
    h = _l.lib()
    nbytes = h.cpfn_similarity_soft_workspace(N, nb, npp, Lp, Lo)
    if nbytes < 0:
        raise RuntimeError("similarity_soft: invalid sizes")
    C = nb * Lp + Lo
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = torch.empty(C, C, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _l.check(h.cpfn_similarity_soft(_ptr(spfn), _ptr(pred), _ptr(pidx), N, nb, npp, Lp, Lo, _ptr(ws), _ptr(out),
                                        _stream()), "cpfn_similarity_soft")
    return out


def get_point_final(point2primitive_prediction, output_labels_heuristic):
    """point2primitive_prediction [N, C] float, output_labels_heuristic [C] integer labels ->
    [N, max(label)+1]: columns pooled by label, each divided by (columns with that label + 1e-10)."""
    _need_cuda(point2primitive_prediction, "get_point_final")
    M = point2primitive_prediction.to(torch.float32).contiguous()
    dev = M.device
    lab = output_labels_heuristic.to(device=dev, dtype=torch.int64)
    N, C = M.shape
    if lab.numel() != C:
        raise RuntimeError("get_point_final: one label per column expected")
    G = int(lab.max()) + 1                                   # (:57 builds torch.eye(max+1): the same host read)
    from ..ops import check_fps_faults
    check_fps_faults("get_point_final")
    ws = torch.empty(C + 2 * G + 2, dtype=torch.int32, device=dev)
    out = torch.empty(N, G, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _l.check(_l.lib().cpfn_label_pool(_ptr(M), _ptr(lab.contiguous()), N, C, G, _ptr(ws), _ptr(out), _stream()),
                 "cpfn_label_pool")
    return out


# ---- the rest of the reference module (the greedy host solver) --------------------------------------------
# `evaluation_localSPFN.py` also calls `merging_utils.run_heuristic_solver` (numba-jitted host code).  When this
# module stands in for `Utils.merging_utils` (cpfn_amd.dropin), any name it does not define is looked up in the
# reference's own file, loaded from wherever `Utils/merging_utils.py` lies on sys.path — unchanged host code,
# needing whatever it needs (numba).
_reference_module = None


def _load_reference_module():
    global _reference_module
    if _reference_module is None:
        import importlib.util
        import os
        import sys
        for root in sys.path:
            cand = os.path.join(root or ".", "Utils", "merging_utils.py")
            if os.path.isfile(cand) and os.path.abspath(cand) != os.path.abspath(__file__):
                spec = importlib.util.spec_from_file_location("_cpfn_reference_merging_utils", cand)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                _reference_module = mod
                break
        else:
            raise ImportError("the reference's Utils/merging_utils.py is not on sys.path")
    return _reference_module


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    try:
        return getattr(_load_reference_module(), name)
    except ImportError as e:
        raise AttributeError("cpfn_amd.Utils.merging_utils has no %r and the reference module could not be loaded: %s"
                             % (name, e))
