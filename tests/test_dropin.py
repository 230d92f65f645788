"""CPU: the drop-in aliasing and the reference-shaped API surface (no compute)."""
import inspect
import sys


def test_install_aliases_reference_import_paths():
    import cpfn_amd.dropin as d
    saved = {k: sys.modules.get(k) for k in d._ALIASES}
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
    saved = {k: sys.modules.get(k) for k in list(d._ALIASES) + ["Utils"]}
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
