"""Golden vectors for the patch-merging tensor functions (run in the BUILD container only, where
/root/reference exists):

    python tests/golden/make_golden_merging.py     ->  tests/golden/merging_small.npz

`Utils/merging_utils.py` imports numba (absent here) at module level for its greedy host solver, so the module
cannot be imported as a whole.  The two functions pinned here, `similarity_soft` and `get_point_final`, are pure
torch: this script compiles exactly those two function definitions out of the reference file's syntax tree and
runs them — the reference's code, unmodified, executed from where it lies; nothing of it is copied or stubbed.
"""
import ast
import os
import sys

import numpy as np
import torch

REF = "/root/reference/Utils/merging_utils.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_functions(names):
    tree = ast.parse(open(REF).read(), REF)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names and not n.decorator_list]
    assert sorted(n.name for n in keep) == sorted(names)
    ns = {"torch": torch, "np": np}
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF, "exec"), ns)
    return [ns[n] for n in names]


def main():
    similarity_soft, get_point_final = reference_functions(["similarity_soft", "get_point_final"])
    rng = np.random.default_rng(11)
    N, nb, npp, Lp, Lo = 3000, 5, 700, 6, 9
    # patches = random point subsets (unique inside a patch, overlapping between patches)
    pidx = np.stack([rng.permutation(N)[:npp] for _ in range(nb)]).astype(np.int64)
    logits = rng.normal(size=(nb, npp, Lp)).astype(np.float32) * 2.0
    pred = torch.softmax(torch.from_numpy(logits), dim=2)
    lab = rng.integers(0, Lo, N)
    spfn = np.eye(Lo, dtype=np.int64)[lab]                      # evaluation_localSPFN.py:79: a LongTensor of one-hot rows
    spfn[rng.random(N) < 0.1] = 0                               # some points carry no global label
    sim = similarity_soft(torch.from_numpy(spfn), pred, torch.from_numpy(pidx))
    C = nb * Lp + Lo
    assert tuple(sim.shape) == (C, C)
    # get_point_final on the caller's matrix (evaluation_localSPFN.py:103-110) with some merged labelling
    M = torch.zeros(N, C)
    for b in range(nb):
        M[torch.from_numpy(pidx[b]), b * Lp:(b + 1) * Lp] = pred[b]
    M[:, nb * Lp:] = torch.from_numpy(spfn).float()
    flag = M[:, :nb * Lp].sum(1) > 0
    M[flag, nb * Lp:] = 0
    labels = torch.from_numpy(rng.integers(0, 14, C))
    labels[:14] = torch.arange(14)                              # every label present
    final = get_point_final(M, labels)
    np.savez_compressed(os.path.join(HERE, "merging_small.npz"), spfn_labels=spfn, predicted_labels=pred.numpy(),
                        point_indices=pidx, similarity=sim.numpy(), point2primitive=M.numpy(),
                        merged_labels=labels.numpy(), point_final=final.numpy())
    print("similarity", sim.shape, float(sim.abs().max()), "final", final.shape)


if __name__ == "__main__":
    sys.exit(main())
