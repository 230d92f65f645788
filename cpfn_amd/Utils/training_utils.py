"""`Utils.training_utils` behind `cpfn_amd.dropin.install(fast_epoch=True)`: `spfn_train_val_epoch` and
`patch_selection_train_val_epoch` are the replayed-step epoch loops of cpfn_amd/epoch.py (same signatures and returns as
Utils/training_utils.py:84-176 and :33-82); EVERY other name — `get_batch_norm_decay`, `update_momentum`,
`get_learning_rate` — is the reference's own, looked up in the user's checkout (`Utils/training_utils.py` on sys.path), unchanged.
"""
from ..epoch import patch_selection_train_val_epoch, spfn_train_val_epoch  # noqa: F401

_reference_module = None


def _load_reference_module():
    global _reference_module
    if _reference_module is None:
        import importlib.util
        import os
        import sys
        for root in sys.path:
            cand = os.path.join(root or ".", "Utils", "training_utils.py")
            if os.path.isfile(cand) and os.path.abspath(cand) != os.path.abspath(__file__):
                spec = importlib.util.spec_from_file_location("_cpfn_reference_training_utils", cand)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                _reference_module = mod
                break
        else:
            raise ImportError("the reference's Utils/training_utils.py is not on sys.path")
    return _reference_module


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    try:
        return getattr(_load_reference_module(), name)
    except ImportError as e:
        raise AttributeError("cpfn_amd.Utils.training_utils has no %r and the reference module could not be loaded: %s"
                             % (name, e))
