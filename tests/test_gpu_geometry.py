"""GPU parity: HIP kernels (through the C ABI) vs the oracle and the golden fixtures.
Bit-exact for every integer output and for the fp32 squared distances."""
import numpy as np
import pytest
import torch

from oracle import geometry as og

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    return t if dtype is None else t.to(dtype)


def _i64(a):
    return a.astype(np.int64)


def test_golden_8192(golden):
    from cpfn_amd import cuda_ops, ops
    g = golden("geometry_8192.npz")
    xyz = T(g["xyz"])
    idx1 = cuda_ops.farthest_point_sampling(xyz, 512, start_idx=T(g["fps1_start"]))
    assert idx1.dtype == torch.int32
    assert np.array_equal(idx1.cpu().numpy(), g["fps1_idx"].astype(np.int32))
    l1 = ops.gather_rows(xyz, idx1)
    idx2 = cuda_ops.farthest_point_sampling(l1, 128, start_idx=T(g["fps2_start"]))
    assert np.array_equal(idx2.cpu().numpy(), g["fps2_idx"].astype(np.int32))
    l2 = ops.gather_rows(l1, idx2)
    b1 = cuda_ops.ball_query(l1, xyz, 0.2, 64)
    assert np.array_equal(b1.cpu().numpy(), g["ball1_idx"].astype(np.int32))
    b2 = cuda_ops.ball_query(l2, l1, 0.4, 64)
    assert np.array_equal(b2.cpu().numpy(), g["ball2_idx"].astype(np.int32))
    d, i = cuda_ops.three_nn(xyz, l1)
    assert np.array_equal(i.cpu().numpy(), g["nn3_idx"].astype(np.int32))
    assert np.array_equal(d.cpu().numpy().view(np.uint32), g["nn3_dist"].view(np.uint32))
    d, i = cuda_ops.three_nn(l1, l2)
    assert np.array_equal(i.cpu().numpy(), g["nn2_idx"].astype(np.int32))
    assert np.array_equal(d.cpu().numpy().view(np.uint32), g["nn2_dist"].view(np.uint32))


def test_golden_ragged(golden):
    from cpfn_amd import cuda_ops, ops
    g = golden("geometry_ragged.npz")
    xyz = T(g["xyz"])
    idx = cuda_ops.farthest_point_sampling(xyz, 37, start_idx=T(g["fps_start"]))
    assert np.array_equal(idx.cpu().numpy(), g["fps_idx"].astype(np.int32))
    ctr = ops.gather_rows(xyz, idx)
    for r, K in [(0.3, 16), (0.2, 5), (0.7, 128), (0.05, 8)]:
        got = cuda_ops.ball_query(ctr, xyz, r, K).cpu().numpy()
        assert np.array_equal(got, g["ball_r%g_k%d" % (r, K)].astype(np.int32)), (r, K)
    d, i = cuda_ops.three_nn(xyz, ctr)
    assert np.array_equal(d.cpu().numpy().view(np.uint32), g["nn_dist"].view(np.uint32))
    od, oi = og.three_nn(g["xyz"], ctr.cpu().numpy())      # ties: compare with the oracle's rule
    assert np.array_equal(i.cpu().numpy(), oi.astype(np.int32))
    w = ops.three_weights(d)
    assert np.array_equal(w.cpu().numpy().view(np.uint32), og.three_weights(od).view(np.uint32))
    # channel-major drop-in ops
    nn_idx = T(g["nn_idx"], torch.int32)
    out = cuda_ops.three_weighted_sum(T(g["feats"]), nn_idx, T(g["w"]))
    np.testing.assert_allclose(out.cpu().numpy(), g["interp"], rtol=1e-6, atol=1e-6)
    gf = cuda_ops.three_weighted_sum_grad(T(g["interp_gout"]), nn_idx, T(g["w"]), 37)
    np.testing.assert_allclose(gf.cpu().numpy(), g["interp_gfeats"], rtol=1e-4, atol=1e-4)
    bidx = T(g["ball_r0.3_k16"], torch.int32)
    grouped = cuda_ops.group_points(T(g["pts"]), bidx)
    assert np.array_equal(grouped.cpu().numpy(), g["grouped"])
    gp = cuda_ops.group_points_grad(T(g["grouped_gout"]), bidx, 1000)
    np.testing.assert_allclose(gp.cpu().numpy(), g["grouped_gpts"], rtol=1e-5, atol=1e-5)
    # gather_points = K==1 grouping
    gi = T(g["fps_idx"], torch.int32)
    gpts = cuda_ops.gather_points(T(g["pts"]), gi)
    assert np.array_equal(gpts.cpu().numpy(), og.group_points(g["pts"], _i64(g["fps_idx"])))
    gg = cuda_ops.gather_points_grad(gpts, gi, 1000)
    np.testing.assert_allclose(gg.cpu().numpy(), og.group_points_grad(gpts.cpu().numpy(), _i64(g["fps_idx"]), 1000),
                               rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("B,N,S", [(1, 64, 64), (3, 65, 7), (2, 513, 100), (2, 2049, 33), (1, 8192, 512), (3, 5000, 77),
                                   (2, 10000, 50), (1, 40000, 64)])
def test_fps_sizes_vs_oracle(B, N, S):
    from cpfn_amd import cuda_ops
    rng = np.random.default_rng(N + S)
    xyz = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    start = rng.integers(0, N, B)
    got = cuda_ops.farthest_point_sampling(T(xyz), S, start_idx=T(start)).cpu().numpy()
    want = og.farthest_point_sample(xyz, S, start).astype(np.int32)
    assert np.array_equal(got, want)
    from cpfn_amd import ops
    with ops.background_geometry():       # the instantiations used beside a training step: 4 waves x 32 points per lane at N > 2048,
        #                                   and no packed fp32 at any size (csrc/sampling.hip, fps_update)
        assert np.array_equal(cuda_ops.farthest_point_sampling(T(xyz), S, start_idx=T(start)).cpu().numpy(), want)


@pytest.mark.parametrize("B,N,S,dup", [(1, 8193, 64, False), (3, 20000, 200, True), (2, 131072, 512, False), (16, 131072, 32, False),
                                         (1, 150001, 100, True), (1, 300000, 64, False), (1, 524288, 40, False),
                                         (20, 131072, 16, False), (1, 600000, 20, False)])
def test_fps_large_clouds_multi_workgroup(B, N, S, dup):
    """N > 8192: several workgroups per cloud exchanging one 8-byte key per sample (fps_shared_kernel, 8 / 16 / 32 points
    per lane), and its fall-backs to the one-workgroup streaming kernel (more workgroups than the device can hold at once, N > 524288) —
    all bit-identical to the oracle, duplicated points (exact ties: lowest index wins) included; also the CUDA-route
    flag (near-origin points skipped) on the multi-workgroup path."""
    from cpfn_amd import cuda_ops, ops
    faults0 = ops.fps_faults()               # (absolute count since the library was loaded: other tests raise it on purpose)
    rng = np.random.default_rng(N + S)
    xyz = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    if dup:
        xyz[:, N // 2:N // 2 + 500] = xyz[:, 100:600]
    start = rng.integers(0, N, B)
    got = cuda_ops.farthest_point_sampling(T(xyz), S, start_idx=T(start), cuda_compat=False).cpu().numpy()
    assert got.min() >= 0
    assert np.array_equal(got, og.farthest_point_sample(xyz, S, start).astype(np.int32))
    from cpfn_amd import ops
    torch.cuda.synchronize()
    assert ops.fps_faults() == faults0      # no workgroup gave up on a sibling (co-residency derived from occupancy), no lost update
    if B == 3:
        xyz[:, 50:90] *= 0.01
        got = cuda_ops.farthest_point_sampling(T(xyz), S, cuda_compat=True).cpu().numpy()
        assert np.array_equal(got, og.farthest_point_sample_cuda(xyz, S).astype(np.int32))


def test_fps_ties_and_default_start():
    from cpfn_amd import cuda_ops
    # a lattice has many exactly equal distances: the lowest index must win each tie
    ax = np.linspace(-1, 1, 9, dtype=np.float32)
    xyz = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(1, -1, 3)
    got = cuda_ops.farthest_point_sampling(T(xyz), 100).cpu().numpy()
    assert np.array_equal(got, og.farthest_point_sample(xyz, 100, [0]).astype(np.int32))


@pytest.mark.parametrize("B,N,S,K,r", [(2, 100, 10, 4, 0.5), (1, 8192, 512, 64, 0.2), (3, 777, 99, 32, 0.3),
                                       (2, 4096, 64, 128, 1.5), (1, 300, 300, 1, 0.1),
                                       # sixteen queries per workgroup, the cloud through LDS tiles of 2048 points: ragged
                                       # last tile, balls that never fill (every tile scanned), empty balls (pad = N)
                                       (3, 5000, 48, 64, 0.2), (2, 2049, 16, 64, 0.02), (16, 512, 128, 64, 0.4),
                                       (2, 8192, 32, 64, 0.001), (1, 2048, 16, 7, 2.0)])
def test_ball_query_vs_oracle(B, N, S, K, r):
    from cpfn_amd import cuda_ops
    rng = np.random.default_rng(N + K)
    xyz = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    q = xyz[:, rng.permutation(N)[:S]]
    got = cuda_ops.ball_query(T(q), T(xyz), r, K).cpu().numpy()
    want = og.ball_query(r, K, xyz, q).astype(np.int32)
    assert np.array_equal(got, want)
    from cpfn_amd import ops
    with ops.background_geometry():           # the wave-per-query kernel (what runs beside a training step)
        assert np.array_equal(cuda_ops.ball_query(T(q), T(xyz), r, K).cpu().numpy(), want)


def test_ball_query_no_neighbour_pads_with_N():
    from cpfn_amd import cuda_ops
    xyz = np.zeros((1, 10, 3), np.float32)
    q = np.full((1, 1, 3), 5.0, np.float32)
    assert np.array_equal(cuda_ops.ball_query(T(q), T(xyz), 0.2, 4).cpu().numpy(), np.full((1, 1, 4), 10))


@pytest.mark.parametrize("B,N,M", [(2, 1000, 3), (1, 8192, 512), (2, 333, 1500), (1, 5, 2), (3, 70, 64), (2, 513, 1027), (1, 100, 65)])
def test_three_nn_vs_oracle(B, N, M):
    from cpfn_amd import cuda_ops
    rng = np.random.default_rng(N + M)
    u = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    k = rng.uniform(-1, 1, (B, M, 3)).astype(np.float32)
    if M >= 3 and N >= M:
        u[:, :M] = k                      # coincident points -> slightly negative squared distances
    d, i = cuda_ops.three_nn(T(u), T(k))
    od, oi = og.three_nn(u, k)
    assert np.array_equal(i.cpu().numpy(), oi.astype(np.int32))
    assert np.array_equal(d.cpu().numpy().view(np.uint32), od.view(np.uint32))
    from cpfn_amd import ops
    with ops.background_geometry():           # the lane-per-query kernel (what runs beside a training step)
        d, i = cuda_ops.three_nn(T(u), T(k))
    assert np.array_equal(i.cpu().numpy(), oi.astype(np.int32))
    assert np.array_equal(d.cpu().numpy().view(np.uint32), od.view(np.uint32))


def test_background_geometry_switch_nests_and_restores():
    from cpfn_amd import lib as _l, ops
    h = _l.lib()
    assert h.cpfn_set_background_geometry(0) == 0
    with ops.background_geometry():
        with ops.background_geometry():
            assert h.cpfn_set_background_geometry(1) == 1
        assert h.cpfn_set_background_geometry(1) == 1          # the inner context restored "on"
    assert h.cpfn_set_background_geometry(0) == 0              # the outer one restored "off"


def test_three_nn_ties_keep_the_lower_index():
    """Known points on a lattice, queries at cell centres, edge midpoints and lattice points: many exactly equal distances.
    The four-lanes-per-query kernel merges its lanes' triples in (distance, index) order — the sequential scan's result."""
    from cpfn_amd import cuda_ops
    ax = np.linspace(-1, 1, 9, dtype=np.float32)
    k = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(1, -1, 3)          # M = 729
    rng = np.random.default_rng(7)
    k = k[:, rng.permutation(k.shape[1])]                                               # ties in arbitrary index order
    cell = (k[0, rng.integers(0, k.shape[1], 400)] + np.float32(0.125)).astype(np.float32)
    edge = (k[0, rng.integers(0, k.shape[1], 400)] + np.array([0.125, 0, 0], np.float32)).astype(np.float32)
    u = np.concatenate([cell, edge, k[0, :224]], 0)[None]
    od, oi = og.three_nn(u, k)
    from cpfn_amd import ops
    import contextlib
    for ctx in (contextlib.nullcontext(), ops.background_geometry()):
        with ctx:
            d, i = cuda_ops.three_nn(T(u), T(k))
        assert np.array_equal(i.cpu().numpy(), oi.astype(np.int32))
        assert np.array_equal(d.cpu().numpy().view(np.uint32), od.view(np.uint32))
    # identical known points (every distance ties): indices 0, 1, 2
    same = np.zeros((1, 100, 3), np.float32)
    d, i = cuda_ops.three_nn(T(u), T(same))
    assert np.array_equal(i.cpu().numpy(), np.broadcast_to(np.arange(3, dtype=np.int32), (1, u.shape[1], 3)))


def test_points_major_ops_vs_oracle():
    from cpfn_amd import ops
    rng = np.random.default_rng(3)
    B, N, S, K, C = 2, 500, 40, 16, 20
    xyz = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    feats = rng.normal(size=(B, N, C)).astype(np.float32)
    idx = rng.integers(0, N, (B, S, K))
    new_xyz = xyz[:, :S]
    # row gather == channel-major grouping transposed
    got = ops.gather_rows(T(feats), T(idx, torch.int32)).cpu().numpy()
    want = og.group_points(feats.transpose(0, 2, 1), idx).transpose(0, 2, 3, 1)
    assert np.array_equal(got, want)
    got16 = ops.gather_rows(T(feats).to(torch.bfloat16), T(idx, torch.int32))
    assert torch.equal(got16.cpu(), torch.from_numpy(want).to(torch.bfloat16))
    c = ops.group_xyz_centered(T(xyz), T(new_xyz), T(idx, torch.int32)).cpu().numpy()
    wantc = og.group_points(xyz.transpose(0, 2, 1), idx).transpose(0, 2, 3, 1) - new_xyz[:, :, None, :]
    assert np.array_equal(c, wantc)
    go = rng.normal(size=(B, S, K, C)).astype(np.float32)
    gs = ops.scatter_add_rows(T(go), T(idx, torch.int32), N).cpu().numpy()
    wants = og.group_points_grad(go.transpose(0, 3, 1, 2), idx, N).transpose(0, 2, 1)
    np.testing.assert_allclose(gs, wants, rtol=1e-5, atol=1e-5)
    nn = rng.integers(0, S, (B, N, 3))
    w = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    f2 = rng.normal(size=(B, S, C)).astype(np.float32)
    o = ops.interp_rows_fwd(T(f2), T(nn, torch.int32), T(w)).cpu().numpy()
    wo = og.three_weighted_sum(f2.transpose(0, 2, 1), nn, w).transpose(0, 2, 1)
    assert np.array_equal(o.view(np.uint32), np.ascontiguousarray(wo).view(np.uint32))
    gb = ops.interp_rows_bwd(T(o), T(nn, torch.int32), T(w), S).cpu().numpy()
    wb = og.three_weighted_sum_grad(np.ascontiguousarray(o.transpose(0, 2, 1)), nn, w, S).transpose(0, 2, 1)
    np.testing.assert_allclose(gb, wb, rtol=1e-4, atol=1e-4)


def test_bf16_row_movers_vs_oracle():
    """bf16 interpolation, its LDS-privatised scatter adjoint, and the fused grouped-input rows."""
    from cpfn_amd import autograd_ops
    rng = np.random.default_rng(11)
    B, M, N, C = 2, 300, 2500, 64
    feats = torch.from_numpy(rng.normal(size=(B, M, C)).astype(np.float32)).to(torch.bfloat16)
    nn = rng.integers(0, M, (B, N, 3))
    w = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    f_dev = feats.to(dev()).requires_grad_(True)
    out = autograd_ops.interp_rows(f_dev, T(nn, torch.int32), T(w))
    assert out.dtype == torch.bfloat16
    f32 = feats.float().numpy()
    want = og.three_weighted_sum(f32.transpose(0, 2, 1), nn, w).transpose(0, 2, 1)
    np.testing.assert_allclose(out.detach().float().cpu().numpy(), want, rtol=1e-2, atol=2e-2)
    g = torch.from_numpy(rng.normal(size=(B, N, C)).astype(np.float32)).to(torch.bfloat16)
    out.backward(g.to(dev()))
    wantg = og.three_weighted_sum_grad(np.ascontiguousarray(g.float().numpy().transpose(0, 2, 1)), nn, w, M).transpose(0, 2, 1)
    got = f_dev.grad.float().cpu().numpy()
    assert np.abs(got - wantg).max() <= 2e-2 * np.abs(wantg).max()
    # grouped input rows: [gathered feats | rel xyz | zero pad], and the gather adjoint
    S, K = 40, 16
    idx = rng.integers(0, M, (B, S, K))
    rel = rng.normal(size=(B, S, K, 3)).astype(np.float32)
    f2 = feats.to(dev()).requires_grad_(True)
    x = autograd_ops.GroupConcat.apply(f2, T(rel), T(idx, torch.int32), 128)
    assert x.shape == (B * S * K, 128) and x.dtype == torch.bfloat16
    xs = x.detach().float().cpu().numpy().reshape(B, S, K, 128)
    gathered = og.group_points(f32.transpose(0, 2, 1), idx).transpose(0, 2, 3, 1)
    assert np.array_equal(xs[..., :C], gathered)
    np.testing.assert_allclose(xs[..., C:C + 3], rel, rtol=1e-2, atol=1e-3)
    assert not xs[..., C + 3:].any()
    gx = torch.from_numpy(rng.normal(size=(B * S * K, 128)).astype(np.float32)).to(torch.bfloat16)
    x.backward(gx.to(dev()))
    wantgg = og.group_points_grad(np.ascontiguousarray(gx.float().numpy().reshape(B, S, K, 128)[..., :C].transpose(0, 3, 1, 2)), idx, M).transpose(0, 2, 1)
    gotg = f2.grad.float().cpu().numpy()
    assert np.abs(gotg - wantgg).max() <= 2e-2 * np.abs(wantgg).max()


@pytest.mark.parametrize("route", ["ordered", "ordered_4_waves", "ordered_16_waves", "radix", "sorted_lists"])
@pytest.mark.parametrize("kind", ["ball_padded", "one_target", "two_alternating", "runs_of_64", "uniform", "beyond_lds"])
def test_inverse_index_lists_are_ascending(kind, route, monkeypatch):
    """cpfn_csr_build(_ws) = a stable sort of the entries by target, whatever the list lengths and whichever of the three builds
    runs (ops.CSR_THREADS / CSR_RADIX): rows padded like a ball query's (a popular target collects hundreds of entries), every
    entry on ONE target, two targets alternating lane by lane and runs of 64 equal targets (the worst cases of the ordered build's
    same-counter LDS atomics), uniform targets, and more entries than any LDS slab holds.  The ordered build must not have needed
    its fall-back sort."""
    from cpfn_amd import ops
    monkeypatch.setattr(ops, "CSR_THREADS", {"ordered": -8, "ordered_4_waves": -1, "ordered_16_waves": -16, "radix": 1024, "sorted_lists": 0}[route])
    monkeypatch.setattr(ops, "CSR_RADIX", route != "sorted_lists")
    fallbacks0 = ops.csr_fallbacks()
    rng = np.random.default_rng(21)
    B, M = 3, 512
    if kind == "ball_padded":
        R, K = 128, 64
        idx = np.empty((B, R, K), np.int64)
        for b in range(B):
            for r in range(R):
                n = int(rng.integers(1, K + 1))
                hit = np.sort(rng.choice(M, n, replace=False))
                idx[b, r, :n] = hit
                idx[b, r, n:] = hit[0]
            idx[b, :40, 5:] = idx[b, :40, :1]            # forty rows padded with the same few targets
    elif kind == "one_target":
        idx = np.full((B, 2048, 4), 9, np.int64)
        idx[1, ::3] = 300
    elif kind == "two_alternating":
        idx = np.tile((np.arange(8192) % 2 * 300).reshape(1, 8192, 1), (B, 1, 1))
    elif kind == "runs_of_64":
        idx = np.repeat(rng.integers(0, M, (B, 128)), 64, axis=1).reshape(B, 8192, 1)
    elif kind == "uniform":
        idx = rng.integers(0, M, (B, 8192, 3))
    else:
        idx = rng.integers(0, M, (B, 40000, 1))
        idx[0, :5000] = 17
    off, ent = ops.csr_build(T(idx, torch.int32), M)
    off, ent = off.cpu().numpy(), ent.cpu().numpy()
    for b in range(B):
        flat = idx[b].reshape(-1)
        assert np.array_equal(ent[b], np.argsort(flat, kind="stable")), kind
        assert np.array_equal(off[b], np.concatenate([[0], np.cumsum(np.bincount(flat, minlength=M))]))
    assert ops.csr_fallbacks() == fallbacks0


def test_inverse_index_adjoints_vs_oracle():
    """cpfn_csr_build gives, per target, the ascending list of referencing entries; the atomic-free adjoints
    through it match the oracle's scatter-adds and are bitwise reproducible."""
    from cpfn_amd import autograd_ops, ops
    rng = np.random.default_rng(12)
    B, M, N, C = 2, 300, 2500, 64
    nn = rng.integers(0, M, (B, N, 3))
    nn[0, :200] = 7                                     # one very long list
    nn[1][nn[1] == 5] = 6                               # and an empty one
    off, ent = ops.csr_build(T(nn, torch.int32), M)
    off, ent = off.cpu().numpy(), ent.cpu().numpy()
    for b in range(B):
        flat = nn[b].reshape(-1)
        order = np.argsort(flat, kind="stable")
        assert np.array_equal(ent[b], order)
        assert np.array_equal(off[b], np.concatenate([[0], np.cumsum(np.bincount(flat, minlength=M))]))
    feats = torch.from_numpy(rng.normal(size=(B, M, C)).astype(np.float32)).to(torch.bfloat16)
    w = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    g = torch.from_numpy(rng.normal(size=(B, N, C)).astype(np.float32)).to(torch.bfloat16)
    grads = []
    for _ in range(2):
        f_dev = feats.to(dev()).requires_grad_(True)
        idx_d = T(nn, torch.int32)
        out = autograd_ops.interp_rows(f_dev, idx_d, T(w), ops.csr_build(idx_d, M))
        out.backward(g.to(dev()))
        grads.append(f_dev.grad.float().cpu().numpy())
    assert np.array_equal(grads[0], grads[1])
    wantg = og.three_weighted_sum_grad(np.ascontiguousarray(g.float().numpy().transpose(0, 2, 1)), nn, w, M).transpose(0, 2, 1)
    assert np.abs(grads[0] - wantg).max() <= 1e-2 * np.abs(wantg).max()
    S, K = 40, 16
    idx = rng.integers(0, M, (B, S, K))
    rel = rng.normal(size=(B, S, K, 3)).astype(np.float32)
    f2 = feats.to(dev()).requires_grad_(True)
    idx_d = T(idx, torch.int32)
    inv = ops.csr_build(idx_d, M)
    x = autograd_ops.GroupConcat.apply(f2, T(rel), idx_d, 128, inv[0], inv[1])
    gx = torch.from_numpy(rng.normal(size=(B * S * K, 128)).astype(np.float32)).to(torch.bfloat16)
    x.backward(gx.to(dev()))
    wantgg = og.group_points_grad(np.ascontiguousarray(gx.float().numpy().reshape(B, S, K, 128)[..., :C].transpose(0, 3, 1, 2)), idx, M).transpose(0, 2, 1)
    gotg = f2.grad.float().cpu().numpy()
    assert np.abs(gotg - wantgg).max() <= 1e-2 * np.abs(wantgg).max()


def test_pairwise_squared_distance_matches_reference(golden):
    """§8 a6: the MATERIALISED distance matrix (modules/geometry_utils.py:4-23) bit for bit against the imported
    reference's own output (fixture `pdist`: 37 FPS centres x the first 257 points), through the C ABI and through
    the drop-in `pairwise_squared_distance` ([B,3,N] channel-major arguments)."""
    from cpfn_amd import cuda_ops, ops
    from cpfn_amd.PointNet2.pointnet2_ops.modules import geometry_utils as gu
    g = golden("geometry_ragged.npz")
    xyz = T(g["xyz"])
    ctr = ops.gather_rows(xyz, T(g["fps_idx"], torch.int32))                 # [3,37,3]
    sub = xyz[:, :257].contiguous()
    got = ops.pairwise_sqdist(ctr, sub)
    assert got.shape == (3, 37, 257) and got.dtype == torch.float32
    assert np.array_equal(got.cpu().numpy().view(np.uint32), g["pdist"].view(np.uint32))
    got2 = gu.pairwise_squared_distance(ctr.transpose(1, 2), sub.transpose(1, 2))
    assert np.array_equal(got2.cpu().numpy().view(np.uint32), g["pdist"].view(np.uint32))
    # and against the oracle on a full 8192 x 512 matrix of another cloud
    g8 = golden("geometry_8192.npz")
    x8 = T(g8["xyz"][:1])
    l1 = ops.gather_rows(x8, T(g8["fps1_idx"][:1], torch.int32))
    full = ops.pairwise_sqdist(l1, x8).cpu().numpy()
    want = og.pairwise_squared_distance(l1.cpu().numpy(), g8["xyz"][:1])
    assert np.array_equal(full.view(np.uint32), want.view(np.uint32))


def test_random_shapes_sweep_vs_oracle():
    """60 random configurations (ragged sizes from 1 point up, duplicated points = exact distance ties, radii from
    "nothing in the ball" to "everything", M < 3 known points for the 3-NN, K larger than the cloud): FPS, ball query,
    3-NN indices and squared distances, interpolation weights — all bit-identical to the oracle."""
    from cpfn_amd import cuda_ops, ops
    rng = np.random.default_rng(2024)
    for trial in range(60):
        B = int(rng.integers(1, 4))
        N = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 127, 300, 1000, 2049, 5000]))
        S = int(rng.integers(1, min(N, 600) + 1))
        K = int(rng.choice([1, 3, 16, 64, 100]))
        r = float(rng.choice([0.01, 0.1, 0.2, 0.4, 0.9, 3.0]))
        xyz = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
        if N > 10 and trial % 3 == 0:
            xyz[:, N // 2:N // 2 + N // 10] = xyz[:, :N // 10]                     # exact duplicates
        if trial % 7 == 0:
            xyz = np.round(xyz * 4) / 4                                           # lattice: many equal distances
        start = rng.integers(0, N, B)
        tag = (trial, B, N, S, K, r)
        sel = cuda_ops.farthest_point_sampling(T(xyz), S, start_idx=T(start))
        want = og.farthest_point_sample(xyz, S, start)
        assert np.array_equal(sel.cpu().numpy(), want.astype(np.int32)), ("fps", tag)
        ctr = np.take_along_axis(xyz, want[:, :, None], axis=1)
        got = cuda_ops.ball_query(T(ctr), T(xyz), r, K).cpu().numpy()
        assert np.array_equal(got, og.ball_query(r, K, xyz, ctr).astype(np.int32)), ("ball", tag)
        d, i = cuda_ops.three_nn(T(xyz), T(ctr))
        od, oi = og.three_nn(xyz, ctr)
        if S >= 3:
            assert np.array_equal(i.cpu().numpy(), oi.astype(np.int32)), ("3nn idx", tag)
            assert np.array_equal(d.cpu().numpy().view(np.uint32), od.view(np.uint32)), ("3nn dist", tag)
            assert np.array_equal(ops.three_weights(d).cpu().numpy().view(np.uint32), og.three_weights(od).view(np.uint32)), ("w", tag)
        else:       # fewer than 3 known points: the first S slots are defined (the reference's sort would fail here)
            assert np.array_equal(i.cpu().numpy()[..., :S], oi.astype(np.int32)[..., :S]), ("3nn idx", tag)


@pytest.mark.parametrize("B,N,S", [(3, 300, 64), (2, 2048, 512), (2, 8192, 512), (1, 5000, 37)])
@pytest.mark.parametrize("background", [False, True])
def test_fps_centres_equals_sampling_then_gather(B, N, S, background):
    """cpfn_fps_centres: the sampling kernels also write the centres they hold anyway — bit for bit what cpfn_fps followed by
    a gather of xyz gives (pointset_abstraction.py:49-50), in both kernel shapes (critical path / beside a training step)."""
    from cpfn_amd import ops
    g = torch.Generator().manual_seed(N + S)
    xyz = torch.randn(B, N, 3, generator=g).cuda()
    start = torch.randint(0, N, (B,), generator=g).to(torch.int32).cuda()
    import contextlib
    with (ops.background_geometry() if background else contextlib.nullcontext()):
        idx0 = ops.fps(xyz, S, start)
        idx1, ctr = ops.fps_centres(xyz, S, start)
        idx2, ctr2 = ops.fps_centres(xyz, S, None, skip_near_origin=True)
        idx3 = ops.fps(xyz, S, None, skip_near_origin=True)
    assert torch.equal(idx0, idx1) and torch.equal(idx2, idx3)
    assert torch.equal(ctr, ops.gather_rows(xyz, idx0)) and torch.equal(ctr2, ops.gather_rows(xyz, idx3))


def test_fps_centres_on_large_clouds_falls_back_to_two_launches():
    from cpfn_amd import ops
    g = torch.Generator().manual_seed(1)
    xyz = torch.randn(1, 20000, 3, generator=g).cuda()
    start = torch.tensor([7], dtype=torch.int32).cuda()
    idx, ctr = ops.fps_centres(xyz, 64, start)
    assert torch.equal(idx, ops.fps(xyz, 64, start)) and torch.equal(ctr, ops.gather_rows(xyz, idx))


@pytest.mark.parametrize("B,N,M", [(2, 8192, 512), (3, 512, 128), (2, 1000, 37), (1, 131, 3)])
@pytest.mark.parametrize("background", [False, True])
@pytest.mark.parametrize("cuda_route", [False, True])
def test_three_nn_weights_equals_three_nn_then_three_weights(B, N, M, background, cuda_route):
    """cpfn_three_nn_weights: distances, indices AND the normalised inverse-distance weights (pointset_feature_propagation.py:38-42)
    from one launch, bit for bit what the two launches give — lane-per-query and four-lanes-per-query kernels, both routes."""
    from cpfn_amd import ops
    g = torch.Generator().manual_seed(N * 7 + M)
    u, k = torch.randn(B, N, 3, generator=g).cuda(), torch.randn(B, M, 3, generator=g).cuda()
    import contextlib
    with (ops.background_geometry() if background else contextlib.nullcontext()):
        d0, i0 = ops.three_nn(u, k, cuda_route=cuda_route, sqrt=cuda_route)
        w0 = ops.three_weights(d0)
        d1, i1, w1 = ops.three_nn_weights(u, k, cuda_route=cuda_route, sqrt=cuda_route)
    assert torch.equal(d0, d1) and torch.equal(i0, i1) and torch.equal(w0, w1)


@pytest.mark.parametrize("B,N,S,K,r", [(2, 8192, 512, 64, 0.2), (3, 512, 128, 64, 0.4), (2, 777, 99, 32, 0.3), (1, 600, 40, 16, 0.01)])
@pytest.mark.parametrize("background", [False, True])
def test_ball_query_rel_equals_ball_query_then_centred_grouping(B, N, S, K, r, background):
    """ball_query_rel: neighbours AND their centred coordinates (pointset_abstraction.py:59-63) — beside a training step from ONE
    launch, bit for bit what ball_query + group_xyz_centered give;
    padded slots and empty balls (radius 0.01) included."""
    from cpfn_amd import ops
    import contextlib
    g = torch.Generator().manual_seed(N + K)
    xyz = (torch.rand(B, N, 3, generator=g) * 2 - 1).cuda()
    start = torch.randint(0, N, (B,), generator=g).to(torch.int32).cuda()
    with (ops.background_geometry() if background else contextlib.nullcontext()):
        sel, ctr = ops.fps_centres(xyz, S, start)
        idx0 = ops.ball_query(ctr, xyz, r, K)
        rel0 = ops.group_xyz_centered(xyz, ctr, idx0)
        idx1, rel1 = ops.ball_query_rel(ctr, xyz, r, K)
    assert torch.equal(idx0, idx1) and torch.equal(rel0, rel1)
