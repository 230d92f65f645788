"""GPU: patch merging (cpfn_amd/Utils/merging_utils.py -> csrc/merging.hip through the C ABI) against the
reference's fixture, the oracle (oracle/merging.py) and — at config-5 size — a dense fp32 evaluation of the
reference's formula on the device plus size-independent properties."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def test_similarity_and_pooling_match_reference_fixture(golden):
    from cpfn_amd.Utils import merging_utils as mu
    g = golden("merging_small.npz")
    sim = mu.similarity_soft(torch.from_numpy(g["spfn_labels"]).to(dev()), torch.from_numpy(g["predicted_labels"]).to(dev()),
                             torch.from_numpy(g["point_indices"]).to(dev()))
    np.testing.assert_allclose(sim.cpu().numpy(), g["similarity"], rtol=2e-5, atol=1e-5)
    fin = mu.get_point_final(torch.from_numpy(g["point2primitive"]).to(dev()), torch.from_numpy(g["merged_labels"]).to(dev()))
    np.testing.assert_allclose(fin.cpu().numpy(), g["point_final"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("N,nb,npp,Lp,Lo", [(50, 1, 1, 1, 1), (777, 3, 100, 32, 32), (5000, 7, 1031, 21, 28),
                                             (64, 0, 0, 5, 3), (4000, 2, 4000, 3, 2)])
def test_similarity_vs_oracle_ragged(N, nb, npp, Lp, Lo):
    """ragged sizes (rows not a multiple of 32, label counts up to the 32-column limit, no patches at all, a patch
    covering every point) against the float64 oracle"""
    from cpfn_amd.Utils import merging_utils as mu
    from oracle import merging as om
    rng = np.random.default_rng(N + nb)
    pidx = np.stack([rng.permutation(N)[:npp] for _ in range(nb)]).astype(np.int64) if nb else np.zeros((0, npp), np.int64)
    pred = rng.random((nb, npp, Lp)).astype(np.float32)
    spfn = rng.random((N, Lo)).astype(np.float32)
    want = om.similarity_soft(spfn, pred, pidx)
    got = mu.similarity_soft(torch.from_numpy(spfn).to(dev()), torch.from_numpy(pred).to(dev()).reshape(nb, npp, Lp),
                             torch.from_numpy(pidx).to(dev()))
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=3e-5, atol=1e-5)
    # bitwise reproducible: fixed summation order, no atomics
    again = mu.similarity_soft(torch.from_numpy(spfn).to(dev()), torch.from_numpy(pred).to(dev()).reshape(nb, npp, Lp),
                               torch.from_numpy(pidx).to(dev()))
    assert torch.equal(got, again)


def test_config5_size_against_dense_formula_and_properties():
    """131072 points, 32 patches of 8192 points, 21 local + 28 global labels: the reference's formula evaluated
    densely on the device (fp32 scatter + matmul), and properties that do not need it."""
    from cpfn_amd.Utils import merging_utils as mu
    N, nb, npp, Lp, Lo = 131072, 32, 8192, 21, 28
    g = torch.Generator().manual_seed(5)
    # patches: contiguous-ish neighbourhoods (a random window of 3*npp points, npp of them) -> realistic overlaps
    pidx = torch.empty(nb, npp, dtype=torch.int64)
    for b in range(nb):
        start = int(torch.randint(0, N - 3 * npp, (1,), generator=g))
        pidx[b] = start + torch.randperm(3 * npp, generator=g)[:npp]
    pred = torch.softmax(torch.randn(nb, npp, Lp, generator=g) * 2, dim=2).to(dev())
    lab = torch.randint(0, Lo, (N,), generator=g)
    spfn = torch.eye(Lo, dtype=torch.int64)[lab].to(dev())
    pidx = pidx.to(dev())
    sim = mu.similarity_soft(spfn, pred, pidx)
    C = nb * Lp + Lo
    M = torch.zeros(N, C, device=dev())
    for b in range(nb):
        M[pidx[b], b * Lp:(b + 1) * Lp] += pred[b]
    M[:, nb * Lp:] = spfn.float()
    dense = (M.double().t() @ M.double()).float()
    scale = float(dense.abs().max())
    assert float((sim - dense).abs().max()) <= 2e-5 * scale
    # properties: symmetry up to summation order; diagonal blocks of the global labels count the points per label;
    # every patch row sums to the patch's total soft mass against the global columns where labels exist
    assert float((sim - sim.t()).abs().max()) <= 2e-5 * scale
    counts = torch.bincount(lab, minlength=Lo).float().to(dev())
    assert torch.equal(torch.diagonal(sim[nb * Lp:, nb * Lp:]), counts)
    mass = pred.sum(1).reshape(-1)                                    # [nb*Lp]: every point carries one global label
    assert float((sim[:nb * Lp, nb * Lp:].sum(1) - mass).abs().max()) <= 1e-4 * float(mass.max())
    # get_point_final on the caller's matrix with a random merged labelling
    labels = torch.randint(0, 60, (C,), generator=g)
    labels[:60] = torch.arange(60)
    fin = mu.get_point_final(M, labels.to(dev()))
    onehot = torch.eye(60, device=dev())[labels.to(dev())]
    want = M @ (onehot / (onehot.sum(0, keepdim=True) + 1e-10))
    assert float((fin - want).abs().max()) <= 1e-5 * float(want.abs().max())


def test_cpu_tensors_are_refused():
    from cpfn_amd.Utils import merging_utils as mu
    with pytest.raises(RuntimeError, match="CPU not supported"):
        mu.similarity_soft(torch.zeros(4, 2), torch.zeros(1, 2, 2), torch.zeros(1, 2, dtype=torch.long))
    with pytest.raises(RuntimeError, match="CPU not supported"):
        mu.get_point_final(torch.zeros(4, 2), torch.zeros(2, dtype=torch.long))
