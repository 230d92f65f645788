"""Make the reference's import paths resolve to this package, so that the reference's
`training_SPFN.py` / `Utils/training_utils.py` run on the MI355X path unchanged:

    import cpfn_amd.dropin; cpfn_amd.dropin.install()
    from PointNet2 import pn2_network            # -> cpfn_amd.PointNet2.pn2_network
    from SPFN import fitter_factory, losses_implementation

Call it before the reference's modules are imported (e.g. first line of training_SPFN.py, or a
`sitecustomize`).  Nothing is copied: the names are aliased in `sys.modules`.

What must be on `sys.path`: this repository's root (for `cpfn_amd`) and the root of the user's reference checkout
(it is `sys.path[0]` when one of the reference's scripts runs from its directory).  The checkout keeps providing
everything this package does not replace — `Dataset`, `Utils.*`, `Configs`, and the HOST-SIDE helpers inside the
replaced `SPFN` package: `SPFN.primitives`, `fitter_factory.create_primitive_from_dict`,
`*_fitter.extract_parameter_data_as_dict` / `extract_predicted_parameters_as_json` / `create_primitive_from_dict`
(callers: Utils/dataset_utils.py:79, :112-120, SPFN/metric_implementation.py:593-599) resolve to the reference's own
files through `cpfn_amd/SPFN/_reference.py`.
"""
import importlib
import sys

_ALIASES = {
    "PointNet2": "cpfn_amd.PointNet2",
    "PointNet2.pn2_network": "cpfn_amd.PointNet2.pn2_network",
    "PointNet2.pointnet2_ops": "cpfn_amd.PointNet2.pointnet2_ops",
    "PointNet2.pointnet2_ops.cuda_ops": "cpfn_amd.cuda_ops",
    "PointNet2.pointnet2_ops.modules": "cpfn_amd.PointNet2.pointnet2_ops.modules",
    "PointNet2.pointnet2_ops.modules.geometry_utils": "cpfn_amd.PointNet2.pointnet2_ops.modules.geometry_utils",
    "PointNet2.pointnet2_ops.modules.pointset_abstraction":
        "cpfn_amd.PointNet2.pointnet2_ops.modules.pointset_abstraction",
    "PointNet2.pointnet2_ops.modules.pointset_feature_propagation":
        "cpfn_amd.PointNet2.pointnet2_ops.modules.pointset_feature_propagation",
    "SPFN": "cpfn_amd.SPFN",
    "SPFN.fitter_factory": "cpfn_amd.SPFN.fitter_factory",
    "SPFN.losses_implementation": "cpfn_amd.SPFN.losses_implementation",
    "SPFN.plane_fitter": "cpfn_amd.SPFN.plane_fitter",
    "SPFN.sphere_fitter": "cpfn_amd.SPFN.sphere_fitter",
    "SPFN.cylinder_fitter": "cpfn_amd.SPFN.cylinder_fitter",
    "SPFN.cone_fitter": "cpfn_amd.SPFN.cone_fitter",
    "SPFN.metric_implementation": "cpfn_amd.SPFN.metric_implementation",
    "SPFN.geometry_utils": "cpfn_amd.SPFN.geometry_utils",
    "SPFN.differentiable_tls": "cpfn_amd.SPFN.differentiable_tls",
    "Utils.merging_utils": "cpfn_amd.Utils.merging_utils",
}


# the caller harness (training_SPFN.py:14 `from Utils import training_utils`), aliased unless install(fast_epoch=False)
_EPOCH_ALIASES = {"Utils.training_utils": "cpfn_amd.Utils.training_utils"}


def alias_names():
    """Every module name install() may bind (tests snapshot / restore these)."""
    return sorted(list(_ALIASES) + list(_EPOCH_ALIASES))


def install(compute_dtype=None, fast_epoch=True):
    """Alias the reference's module names.  `compute_dtype=torch.bfloat16` additionally makes
    every PointNet2 built afterwards use the fused bf16 MFMA stacks by default.
    `fast_epoch` (default since round 5: the unchanged script should get the replayed step, not 0.13 x of it):
    `Utils.training_utils.spfn_train_val_epoch` / `patch_selection_train_val_epoch` (the loops training_SPFN.py:105-108 and
    training_PatchSelection.py:79-86 call) resolve to the replayed-step epoch loops (cpfn_amd/epoch.py: one graph replay per
    batch, pinned look-ahead input staging, deferred logging, same signature / return / prints) — WHEN the caller's optimizer
    is the plain `torch.optim.Adam` those scripts build; with any other optimizer the same call runs the reference's own loop
    (decided per call, with a warning).  Every other name of `Utils.training_utils` stays the reference's own.
    `fast_epoch=False` leaves `Utils.training_utils` alone: the reference's loop on the eager modules."""
    for ref_name, ours in _ALIASES.items():
        sys.modules[ref_name] = importlib.import_module(ours)
    if fast_epoch:
        for ref_name, ours in _EPOCH_ALIASES.items():
            sys.modules[ref_name] = importlib.import_module(ours)
    from .SPFN import _reference
    _reference.attach()            # SPFN.primitives & co. of a checkout that is already on sys.path; lazy otherwise
    if compute_dtype is not None:
        from .PointNet2 import pn2_network
        orig = pn2_network.PointNet2.__init__

        def patched(self, *a, **k):
            orig(self, *a, **k)
            self.set_compute_dtype(compute_dtype)
        pn2_network.PointNet2.__init__ = patched
    return sorted(list(_ALIASES) + (list(_EPOCH_ALIASES) if fast_epoch else []))
