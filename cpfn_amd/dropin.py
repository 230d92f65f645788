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


def install(compute_dtype=None, fast_epoch=False):
    """Alias the reference's module names.  `compute_dtype=torch.bfloat16` additionally makes
    every PointNet2 built afterwards use the fused bf16 MFMA stacks by default.
    `fast_epoch=True`: `Utils.training_utils.spfn_train_val_epoch` (the loop training_SPFN.py:105-108 calls) resolves to
    the replayed-step epoch loop (cpfn_amd/epoch.py: one graph replay per batch, pinned look-ahead input staging, deferred
    logging, same signature / return / prints); every other name of `Utils.training_utils` stays the reference's own."""
    for ref_name, ours in _ALIASES.items():
        sys.modules[ref_name] = importlib.import_module(ours)
    if fast_epoch:
        sys.modules["Utils.training_utils"] = importlib.import_module("cpfn_amd.Utils.training_utils")
    from .SPFN import _reference
    _reference.attach()            # SPFN.primitives & co. of a checkout that is already on sys.path; lazy otherwise
    if compute_dtype is not None:
        from .PointNet2 import pn2_network
        orig = pn2_network.PointNet2.__init__

        def patched(self, *a, **k):
            orig(self, *a, **k)
            self.set_compute_dtype(compute_dtype)
        pn2_network.PointNet2.__init__ = patched
    return sorted(_ALIASES)
