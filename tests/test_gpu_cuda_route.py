"""GPU: the opt-in CUDA route (`cuda_ops.CUDA_ROUTE`, CPFN_CUDA_ROUTE=1 + `fast=True`) — what the reference's compiled
ops return (FPS from index 0 skipping near-origin points, direct-distance ball query / 3-NN, SQRT 3-NN distances in
the interpolation weights; sampling_gpu.cu:63-159, ball_query_gpu.cu:9-44, interpolate_gpu.cu:9-59,
modules/geometry_utils.py:184) — against its scalar restatement in oracle/ (parity UNPINNED: no CUDA build of the
reference can run here), and the check that the two routes differ exactly where the reference's two routes differ."""
import numpy as np
import pytest
import torch

from cpfn_amd import synthetic
from oracle import geometry as og

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    return t if dtype is None else t.to(dtype)


@pytest.fixture
def cuda_route():
    from cpfn_amd import cuda_ops
    prev, cuda_ops.CUDA_ROUTE = cuda_ops.CUDA_ROUTE, True
    yield
    cuda_ops.CUDA_ROUTE = prev


def _cloud():
    P = synthetic.primitive_cloud(2, 4096, n_prims=7, seed=5)["P"].numpy()
    P[:, 100:140] *= 0.02                   # a cluster of near-origin points (|p|² <= 1e-3): the CUDA FPS skips them
    P[:, 900:905] = P[:, 300:305]           # exact duplicates -> distance ties
    return np.ascontiguousarray(P)


def test_ops_match_cuda_route_restatement():
    from cpfn_amd import cuda_ops, ops
    P = _cloud()
    xyz = T(P)
    f = cuda_ops.farthest_point_sampling(xyz, 512, cuda_compat=True)
    want = og.farthest_point_sample_cuda(P, 512)
    assert np.array_equal(f.cpu().numpy(), want.astype(np.int32))
    assert (want[:, 0] == 0).all()
    near = (P ** 2).sum(-1) <= 1e-3
    assert near.sum() >= 40 and not near[0, want[0, 1:]].any()
    ctr = ops.gather_rows(xyz, f)
    for r, K in ((0.2, 64), (0.4, 16), (0.01, 8)):
        got = cuda_ops.ball_query(ctr, xyz, r, K, cuda_compat=True).cpu().numpy()
        assert np.array_equal(got, og.ball_query_cuda(r, K, P, ctr.cpu().numpy()).astype(np.int32)), (r, K)
    d2, i = cuda_ops.three_nn(xyz, ctr, cuda_compat=True)
    wd2, wi = og.three_nn_cuda(P, ctr.cpu().numpy(), sqrt=False)
    assert np.array_equal(i.cpu().numpy(), wi.astype(np.int32))
    assert np.array_equal(d2.cpu().numpy().view(np.uint32), wd2.view(np.uint32))
    d, _ = ops.three_nn(xyz, ctr, cuda_route=True, sqrt=True)
    wd, _ = og.three_nn_cuda(P, ctr.cpu().numpy(), sqrt=True)
    assert np.array_equal(d.cpu().numpy().view(np.uint32), wd.view(np.uint32))


def test_fast_flag_selects_the_route(cuda_route):
    """With the switch on, `fast=True` gives the CUDA route and `fast=False` the CPU route — the reference's own two
    results; the interpolation weights are 1/(d + 1e-8) vs 1/(d² + 1e-8)."""
    from cpfn_amd import ops
    from cpfn_amd.PointNet2.pointnet2_ops.modules import geometry_utils as gu
    P = _cloud()
    pos = T(P).transpose(1, 2).contiguous()                       # [B,3,N]
    torch.manual_seed(3)
    start = torch.randint(0, P.shape[1], (2,))
    torch.manual_seed(3)
    f_cpu = gu.farthest_point_sample(pos, 128, fast=False)
    f_cuda = gu.farthest_point_sample(pos, 128, fast=True)
    assert np.array_equal(f_cpu.cpu().numpy(), og.farthest_point_sample(P, 128, start.numpy()))
    assert np.array_equal(f_cuda.cpu().numpy(), og.farthest_point_sample_cuda(P, 128))
    ctr = gu.select_point_subset(pos, f_cuda)                      # [B,3,128]
    cn = np.ascontiguousarray(ctr.transpose(1, 2).cpu().numpy())
    d_cpu, i_cpu = gu.three_nn(ctr, pos, fast=False)
    d_cuda, i_cuda = gu.three_nn(ctr, pos, fast=True)
    od, oi = og.three_nn(P, cn)
    assert np.array_equal(d_cpu.cpu().numpy().view(np.uint32), od.view(np.uint32)) and np.array_equal(i_cpu.cpu().numpy(), oi)
    od, oi = og.three_nn_cuda(P, cn, sqrt=True)
    assert np.array_equal(d_cuda.cpu().numpy().view(np.uint32), od.view(np.uint32)) and np.array_equal(i_cuda.cpu().numpy(), oi)
    # the weights the feature-propagation level derives from them (pointset_feature_propagation.py:40-42)
    w_cpu, w_cuda = ops.three_weights(d_cpu.contiguous()), ops.three_weights(d_cuda.contiguous())
    r = 1.0 / (od.astype(np.float64) + 1e-8)
    np.testing.assert_allclose(w_cuda.cpu().numpy(), r / r.sum(-1, keepdims=True), rtol=1e-5)
    far = od.min(-1) > 1e-3                                        # away from coincident points the two weightings differ
    assert np.abs(w_cpu.cpu().numpy() - w_cuda.cpu().numpy())[far].max() > 1e-2
    b_cpu = gu.ball_query(0.2, 32, pos, ctr, fast=False).cpu().numpy()
    b_cuda = gu.ball_query(0.2, 32, pos, ctr, fast=True).cpu().numpy()
    assert np.array_equal(b_cpu, og.ball_query(0.2, 32, P, cn)) and np.array_equal(b_cuda, og.ball_query_cuda(0.2, 32, P, cn))


def test_network_routes(cuda_route):
    """The whole network: switch on + fast=False is bit-identical to the default (switch off); fast=True differs
    through the interpolation weights, and its geometry is the CUDA-route restatement's."""
    from cpfn_amd import cuda_ops
    from cpfn_amd.PointNet2 import pn2_network
    m = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 28])
    m.load_state_dict(synthetic.synthetic_state_dict(synthetic.pointnet2_state_shapes(), seed=0), strict=True)
    m.dropout_p = 0.0
    m = m.to(dev()).train()
    P = _cloud()
    x = T(P)
    starts = (torch.tensor([5, 9]), torch.tensor([1, 2]))
    with torch.no_grad():
        slow = m(x, fast=False, fps_start=starts)
        fast = m(x, fast=True, fps_start=starts)
        fps_fast = m.aux_sa1["fps_idx"].cpu().numpy()
        nn_w_fast = m.aux_sfp3["nn_w"].cpu().numpy()
        cuda_ops.CUDA_ROUTE = False
        default = m(x, fps_start=starts)
        cuda_ops.CUDA_ROUTE = True
    for a, b in zip(slow[:3], default[:3]):
        assert torch.equal(a, b)
    assert float((fast[2] - slow[2]).abs().max()) > 1e-3
    want = og.farthest_point_sample_cuda(P, 512)
    assert np.array_equal(fps_fast, want.astype(np.int32))
    l1 = np.take_along_axis(P, want[:, :, None], axis=1)
    d, _ = og.three_nn_cuda(P, l1, sqrt=True)
    r = 1.0 / (d.astype(np.float64) + 1e-8)
    np.testing.assert_allclose(nn_w_fast, r / r.sum(-1, keepdims=True), rtol=1e-5)
