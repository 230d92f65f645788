"""CPU: the drop-in aliasing and the reference-shaped API surface (no compute)."""
import inspect
import sys


def test_install_aliases_reference_import_paths():
    import cpfn_amd.dropin as d
    saved = {k: sys.modules.get(k) for k in d.alias_names()}
    try:
        d.install()
        from PointNet2 import pn2_network
        from PointNet2.pointnet2_ops import cuda_ops
        from PointNet2.pointnet2_ops.modules import geometry_utils
        from SPFN import fitter_factory, losses_implementation, metric_implementation, plane_fitter
        assert pn2_network.__name__ == "cpfn_amd.PointNet2.pn2_network"
        # the nine bound functions of cuda_ops/src/bindings.cpp:6-19
        for name in ("gather_points", "gather_points_grad", "farthest_point_sampling", "three_nn",
                     "three_weighted_sum", "three_weighted_sum_grad", "ball_query", "group_points",
                     "group_points_grad"):
            assert callable(getattr(cuda_ops, name)), name
        for name in ("pairwise_squared_distance", "select_point_subset", "farthest_point_sample", "ball_query",
                     "three_nn", "three_weighted_sum"):
            assert callable(getattr(geometry_utils, name)), name
        assert "fast" in inspect.signature(geometry_utils.ball_query).parameters
        assert list(inspect.signature(losses_implementation.compute_all_losses).parameters)[:9] == [
            "P", "W", "I_gt", "X", "X_gt", "T", "T_gt", "gt_parameters", "points_per_instance"]
        assert list(inspect.signature(metric_implementation.compute_all_metrics).parameters) == [
            "P", "X", "X_gt", "W", "I_gt", "T", "T_gt", "points_per_instance", "gt_parameters", "list_epsilon", "classes"]
        for fn in ("hungarian_matching", "hard_W_encoding", "get_instance_type", "get_residual_loss", "compute_segmentation_iou",
                   "compute_type_accuracy", "compute_normal_difference", "compute_axis_difference",
                   "compute_meanstd_Sk_residual", "compute_Sk_coverage", "compute_P_coverage"):
            assert callable(getattr(metric_implementation, fn)), fn
        for fn in ("compute_parameters", "compute_residue_single", "compute_parameter_loss"):
            assert callable(getattr(plane_fitter, fn))
        fitter_factory.register_primitives(["sphere", "plane", "cylinder", "cone"])
        assert fitter_factory.primitive_name_to_id("cylinder") == 2
        assert fitter_factory.get_n_registered_primitives() == 4
        # the caller harness: bound by default (the unchanged script must get the replayed step) ...
        assert sys.modules["Utils.training_utils"].__name__ == "cpfn_amd.Utils.training_utils"
        assert sys.modules["Utils.training_utils"].spfn_train_val_epoch.__module__ == "cpfn_amd.epoch"
        # ... and left alone on request
        del sys.modules["Utils.training_utils"]
        assert "Utils.training_utils" not in d.install(fast_epoch=False) and "Utils.training_utils" not in sys.modules
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_state_dict_keys_match_reference_layout():
    from cpfn_amd import synthetic
    from cpfn_amd.PointNet2 import pn2_network
    for sizes in ([3, 4, 28], [3, 4, 21], [2]):
        m = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=sizes)
        shapes = synthetic.pointnet2_state_shapes(sizes)
        sd = m.state_dict()
        assert list(sd.keys()) == list(shapes.keys())
        assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in sd)
    assert sum(p.numel() for p in pn2_network.PointNet2(3, 3, [3, 4, 28]).parameters()) == 1406307


def test_merging_utils_alias_keeps_the_host_solver_of_the_reference(tmp_path):
    """`Utils.merging_utils` resolves to the HIP-backed module; names it does not define (the greedy host solver)
    come from the reference's own file found on sys.path."""
    import cpfn_amd.dropin as d
    (tmp_path / "Utils").mkdir()
    (tmp_path / "Utils" / "__init__.py").write_text("")
    (tmp_path / "Utils" / "merging_utils.py").write_text(
        "def run_heuristic_solver(*a, **k):\n    return 'host solver of the reference'\n"
        "def similarity_soft(*a):\n    raise AssertionError('must not be reached')\n")
    saved = {k: sys.modules.get(k) for k in d.alias_names() + ["Utils"]}
    sys.path.insert(0, str(tmp_path))
    try:
        d.install()
        import cpfn_amd.Utils.merging_utils as mine
        mine._reference_module = None
        from Utils import merging_utils
        assert merging_utils is mine
        assert merging_utils.similarity_soft.__module__ == "cpfn_amd.Utils.merging_utils"
        assert merging_utils.get_point_final.__module__ == "cpfn_amd.Utils.merging_utils"
        assert merging_utils.run_heuristic_solver() == "host solver of the reference"
        assert list(inspect.signature(merging_utils.similarity_soft).parameters) == ["spfn_labels", "predicted_labels", "point_indices"]
        assert list(inspect.signature(merging_utils.get_point_final).parameters) == ["point2primitive_prediction", "output_labels_heuristic"]
    finally:
        sys.path.remove(str(tmp_path))
        mine._reference_module = None
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


# ---- the reference's host-side SPFN helpers behind the alias (second boundary) -------------------------
def _restore_modules(saved_keys_snapshot):
    import cpfn_amd.SPFN._reference as r
    r.reset()
    for k in [k for k in sys.modules if k == "SPFN" or k.startswith("SPFN.") or k == "Utils" or k.startswith("Utils.")
              or k == "Dataset" or k.startswith("Dataset.")]:
        del sys.modules[k]
    for k, v in saved_keys_snapshot.items():
        if v is not None:
            sys.modules[k] = v


def _snapshot():
    import cpfn_amd.dropin as d
    return {k: sys.modules.get(k) for k in d.alias_names() + ["Utils", "Dataset", "SPFN.primitives"]}


FAKE_PRIMITIVES = "class Plane:\n    def __init__(self, n, c):\n        self.n, self.c = n, c\n"
FAKE_PLANE_FITTER = (
    "from SPFN.primitives import Plane\n"
    "from SPFN.geometry_utils import host_only_helper\n"          # a name only the fake reference's sibling has
    "def compute_parameters(*a):\n    raise AssertionError('the device path must win')\n"
    "def create_primitive_from_dict(d):\n    return Plane(d['axis'], host_only_helper(d['c']))\n"
    "def extract_parameter_data_as_dict(primitives, n):\n"
    "    return {'plane_n_gt': [p.n for p in primitives if isinstance(p, Plane)]}\n")
FAKE_FACTORY = (
    "from SPFN import plane_fitter\n"
    "def register_primitives(*a):\n    raise AssertionError('the device path must win')\n"
    "def create_primitive_from_dict(d):\n    return plane_fitter.create_primitive_from_dict(d)\n")


def test_host_helpers_of_a_fake_reference_tree_resolve_behind_the_alias(tmp_path):
    """With `SPFN` aliased to cpfn_amd.SPFN, the names the device path does not define come from the checkout on
    sys.path: `SPFN.primitives`, `fitter_factory.create_primitive_from_dict`, `plane_fitter.extract_*` — loaded with
    the checkout's own siblings visible to them — while everything the device path defines stays ours."""
    import cpfn_amd.dropin as d
    (tmp_path / "SPFN").mkdir()
    (tmp_path / "SPFN" / "primitives.py").write_text(FAKE_PRIMITIVES)
    (tmp_path / "SPFN" / "plane_fitter.py").write_text(FAKE_PLANE_FITTER)
    (tmp_path / "SPFN" / "fitter_factory.py").write_text(FAKE_FACTORY)
    (tmp_path / "SPFN" / "geometry_utils.py").write_text("def host_only_helper(c):\n    return c + 1\n")
    saved = _snapshot()
    _restore_modules({})
    sys.path.insert(0, str(tmp_path))
    try:
        d.install()
        from SPFN import fitter_factory, plane_fitter, primitives
        from SPFN.primitives import Plane
        import SPFN.primitives as by_import
        assert primitives.__file__ == str(tmp_path / "SPFN" / "primitives.py")
        assert by_import is primitives and Plane is primitives.Plane
        assert fitter_factory.__name__ == "cpfn_amd.SPFN.fitter_factory"
        assert plane_fitter.compute_parameters.__module__ == "cpfn_amd.SPFN.plane_fitter"
        fitter_factory.register_primitives(["sphere", "plane", "cylinder", "cone"])          # ours: does not raise
        prim = fitter_factory.create_primitive_from_dict({"axis": (0, 0, 1), "c": 2})
        assert isinstance(prim, Plane) and prim.c == 3
        assert plane_fitter.extract_parameter_data_as_dict([prim, object()], 4) == {"plane_n_gt": [(0, 0, 1)]}
        # the alias is back in place after the private load, and the device modules were not replaced
        assert sys.modules["SPFN"].__name__ == "cpfn_amd.SPFN"
        assert sys.modules["SPFN.geometry_utils"].__name__ == "cpfn_amd.SPFN.geometry_utils"
        import pytest
        with pytest.raises(AttributeError, match="neither"):
            plane_fitter.extract_no_such_name                      # a pass-through NAME neither side defines
        with pytest.raises(AttributeError, match="only host-side helpers"):
            plane_fitter.no_such_name                              # anything else is not passed through at all
        from SPFN import geometry_utils
        with pytest.raises(AttributeError, match="only host-side helpers"):
            geometry_utils.host_only_helper                        # ... even if the checkout's file defines it
    finally:
        sys.path.remove(str(tmp_path))
        _restore_modules(saved)


def test_missing_checkout_is_reported_not_shadowed():
    """No checkout on sys.path: the lookup says what is missing (round 2 raised NotImplementedError / ImportError)."""
    import cpfn_amd.SPFN._reference as r
    import pytest
    saved = _snapshot()
    path = list(sys.path)
    _restore_modules({})
    sys.path[:] = [p for p in sys.path if not __import__("os").path.isfile(__import__("os").path.join(p or ".", "SPFN", "primitives.py"))]
    try:
        from cpfn_amd.SPFN import fitter_factory
        with pytest.raises(AttributeError, match="reference checkout"):
            fitter_factory.create_primitive_from_dict
        assert getattr(fitter_factory, "create_primitive_from_dict", None) is None
    finally:
        sys.path[:] = path
        _restore_modules(saved)
        r.reset()


REFERENCE = "/root/reference"


class _FakeH5Item:
    def __init__(self, value=None, attrs=None, children=None):
        self.value, self.attrs, self.children = value, attrs or {}, children or {}

    def __getitem__(self, key):
        return self.value if key == () else self.children[key]

    def keys(self):
        return self.children.keys()


def _fake_shape_file(n_points=64):
    """What Utils/dataset_utils.py:34-80 reads from one HDF5 shape: point arrays plus one `<name>_soup_<i>` group per GT
    primitive with its `meta` attribute (a dict repr, :77)."""
    import numpy as np
    rng = np.random.default_rng(0)
    metas = [
        dict(type="plane", location_x=0.1, location_y=0.2, location_z=0.3, axis_x=0.0, axis_y=0.0, axis_z=1.0),
        dict(type="sphere", location_x=0.1, location_y=0.2, location_z=0.3, radius=0.5),
        dict(type="cylinder", location_x=0.1, location_y=0.2, location_z=0.3, axis_x=0.0, axis_y=1.0, axis_z=0.0, radius=0.3),
        dict(type="cone", apex_x=0.0, apex_y=0.0, apex_z=0.0, axis_x=1.0, axis_y=0.0, axis_z=0.0, angle=0.4, semi_angle=0.4,
             location_x=0.0, location_y=0.0, location_z=0.0, radius=0.1),
    ]
    children = {
        "noisy_points": _FakeH5Item(rng.standard_normal((n_points, 3)).astype("float32")),
        "gt_points": _FakeH5Item(rng.standard_normal((n_points, 3)).astype("float32")),
        "gt_normals": _FakeH5Item(rng.standard_normal((n_points, 3)).astype("float32")),
        "gt_labels": _FakeH5Item(rng.integers(0, len(metas), n_points)),
    }
    for i, m in enumerate(metas):
        children["shape_soup_%d" % i] = _FakeH5Item(attrs={"meta": repr(m)}, children={
            "gt_points": _FakeH5Item(rng.standard_normal((16, 3)).astype("float32"))})
    return _FakeH5Item(children=children), metas


def test_the_reference_s_own_data_path_runs_behind_the_alias():
    """Build container only (needs /root/reference; skipped on the GPU box).  Installs the drop-in and drives the REAL
    `Utils.dataset_utils.create_unit_data_from_hdf5_spfn` (the function `training_SPFN.py`'s data loader calls per
    sample, Dataset/dataloaders.py) on a fake HDF5 shape: `fitter_factory.create_primitive_from_dict` (:79), the four
    `extract_parameter_data_as_dict` (:112-120), and the JSON export of the evaluation scripts
    (`metric_implementation.creates_json`, SPFN/metric_implementation.py:590-603)."""
    import os
    import types

    import numpy as np
    import pytest
    if not os.path.isdir(os.path.join(REFERENCE, "SPFN")):
        pytest.skip("the reference checkout is not on this machine")
    import cpfn_amd.dropin as d
    saved = _snapshot()
    saved_h5 = sys.modules.get("h5py")
    _restore_modules({})
    sys.path.insert(0, REFERENCE)
    if saved_h5 is None:
        sys.modules["h5py"] = types.ModuleType("h5py")                 # imported at module level only (:4)
    try:
        d.install()
        from SPFN import fitter_factory, metric_implementation, primitives
        from Utils import dataset_utils, training_utils                 # noqa: F401  (the reference's own files)
        assert dataset_utils.__file__.startswith(REFERENCE)
        assert dataset_utils.fitter_factory is fitter_factory and fitter_factory.__name__ == "cpfn_amd.SPFN.fitter_factory"
        assert training_utils.losses_implementation.__name__ == "cpfn_amd.SPFN.losses_implementation"
        assert primitives.__file__ == os.path.join(REFERENCE, "SPFN", "primitives.py")
        fitter_factory.register_primitives(["sphere", "plane", "cylinder", "cone"])
        f, metas = _fake_shape_file()
        np.random.seed(0)
        out = dataset_utils.create_unit_data_from_hdf5_spfn(f, n_max_instances=6, noisy=True, n_points=64)
        assert out is not None
        assert out["T_gt"].tolist() == [1, 0, 2, 3, 0, 0]              # ids in the registry's order
        assert out["P_gt"].shape == (6, 16, 3) and out["P"].shape == (64, 3)
        assert np.allclose(out["plane_n_gt"][0], [0, 0, 1]) and not out["plane_n_gt"][1:].any()
        assert np.allclose(out["cylinder_axis_gt"][2], [0, 1, 0])
        assert np.allclose(out["cone_axis_gt"][3], [1, 0, 0])
        # round trip per type through the helper modules themselves
        from SPFN import cone_fitter, cylinder_fitter, plane_fitter, sphere_fitter
        for mod, meta, cls in zip((plane_fitter, sphere_fitter, cylinder_fitter, cone_fitter), metas,
                                  (primitives.Plane, primitives.Sphere, primitives.Cylinder, primitives.Cone)):
            prim = mod.create_primitive_from_dict(meta)
            assert type(prim) is cls
            assert type(fitter_factory.create_primitive_from_dict(meta)) is cls
            assert isinstance(mod.extract_parameter_data_as_dict([prim], 2), dict)
        # JSON export (evaluation scripts): one primitive of every type id 0..3 of metric_implementation.creates_json
        import torch
        params = {"plane_normal": torch.tensor([[[0.0, 0.0, 1.0]] * 4]), "plane_center": torch.full((1, 4), 0.3),
                  "sphere_center": torch.zeros(1, 4, 3), "sphere_radius_squared": torch.full((1, 4), 0.25),
                  "cylinder_center": torch.zeros(1, 4, 3), "cylinder_radius_squared": torch.full((1, 4), 0.09),
                  "cylinder_axis": torch.tensor([[[0.0, 1.0, 0.0]] * 4]),
                  "cone_apex": torch.zeros(1, 4, 3), "cone_axis": torch.tensor([[[1.0, 0.0, 0.0]] * 4]),
                  "cone_half_angle": torch.full((1, 4), 0.4)}
        js = metric_implementation.creates_json([0, 1, 2, 3], params)
        assert [j["type"] for j in js] == ["plane", "sphere", "cylinder", "cone"]
        assert abs(js[1]["radius"] - 0.5) < 1e-6 and js[1]["label"] == 1
        assert sphere_fitter.extract_predicted_parameters_as_json(np.zeros(3), np.float32(0.25), 7)["label"] == 7
    finally:
        sys.path.remove(REFERENCE)
        if saved_h5 is None:
            sys.modules.pop("h5py", None)
        _restore_modules(saved)


def test_every_reference_function_is_defined_here_or_passed_through_by_name():
    """walk every top-level `def` / `class` of the reference's SPFN/*.py.  Each is either DEFINED by the same-named
    module of this package (a `def` of its own, not a fall-through) or is on the pass-through allow-list (host-side GT parsing /
    JSON export, TensorFlow twins).  A compute function of the reference that silently ran the reference's own op-by-op file would
    fail here.  Build container only: needs the reference checkout."""
    import ast
    import importlib
    import os
    import pytest
    from cpfn_amd.SPFN import _reference as r
    ref_dir = os.path.join(REFERENCE, "SPFN")
    if not os.path.isdir(ref_dir):
        pytest.skip("no reference checkout (GPU box)")
    ours_dir = os.path.dirname(r.__file__)
    checked = 0
    for f in sorted(os.listdir(ref_dir)):
        name = f[:-3]
        if not f.endswith(".py") or name == "__init__" or not r.overridden(name):
            continue                                            # (`primitives`: not overridden, the reference's file IS the module)
        tree = ast.parse(open(os.path.join(ref_dir, f)).read())
        ref_defs = [n.name for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef))]
        own = ast.parse(open(os.path.join(ours_dir, f)).read())
        own_defs = {n.name for n in own.body if isinstance(n, (ast.FunctionDef, ast.ClassDef))}
        mod = importlib.import_module("cpfn_amd.SPFN." + name)
        for d in ref_defs:
            checked += 1
            if r.passes_through(d):
                continue
            assert d in own_defs, "SPFN/%s.py:%s is neither defined in cpfn_amd/SPFN/%s nor on the pass-through list" % (name, d, f)
            assert getattr(mod, d).__module__.startswith("cpfn_amd."), (name, d)
    assert checked > 100
    # the allow-list is narrow: none of the reference's compute functions match it
    for d in ("guarded_matrix_solve_ls", "compute_parameter_loss", "acos_safe", "compute_parameters", "weighted_sphere_fitting",
              "solve_weighted_tls", "compute_all_losses", "compute_all_metrics", "hungarian_matching"):
        assert not r.passes_through(d), d
