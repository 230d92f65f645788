"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) — CPU restatement of the assignment solver behind the
reference's `hungarian_matching` (SPFN/losses_implementation.py:10-30 calls
`scipy.optimize.linear_sum_assignment(-cost)` per cloud).

SciPy is a third-party dependency of the reference (not vendored in /root/reference; the image has SciPy
1.15.3).  Its solver (scipy/optimize/rectangular_lsap/rectangular_lsap.cpp) is the shortest-augmenting-path
algorithm of D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE TAES 52(4), 2016, with
two published tie-breaking choices that decide WHICH optimal assignment comes out and therefore matter for
parity: the list of unvisited columns is filled in reverse order, and among columns of equal reduced cost an
unassigned one (the last such in list order) is preferred.  This file restates that algorithm in plain Python
(nr <= nc, the only case the reference produces: n_gt rows, K columns); it is pinned against SciPy itself by
tests/test_oracle_golden.py::test_lsap_matches_scipy on random, tie-heavy and constant matrices, and is the
checker for the HIP kernel cpfn_hungarian_match.
"""
import numpy as np


def linear_sum_assignment_min(cost):
    """cost [nr, nc] float64 with nr <= nc -> col4row [nr]: the column assigned to every row (minimisation)."""
    cost = np.asarray(cost, dtype=np.float64)
    nr, nc = cost.shape
    assert nr <= nc
    u = np.zeros(nr)
    v = np.zeros(nc)
    spc = np.empty(nc)
    path = np.full(nc, -1, dtype=np.int64)
    col4row = np.full(nr, -1, dtype=np.int64)
    row4col = np.full(nc, -1, dtype=np.int64)
    for cur in range(nr):
        # ---- shortest augmenting path from row `cur`
        min_val = 0.0
        remaining = [nc - it - 1 for it in range(nc)]        # reverse order: identity for a constant matrix
        SR = np.zeros(nr, dtype=bool)
        SC = np.zeros(nc, dtype=bool)
        spc[:] = np.inf
        sink, i = -1, cur
        while sink == -1:
            index, lowest = -1, np.inf
            SR[i] = True
            for it, j in enumerate(remaining):
                r = ((min_val + cost[i, j]) - u[i]) - v[j]
                if r < spc[j]:
                    path[j] = i
                    spc[j] = r
                # equal reduced costs: prefer a column that is a new sink (the last such in list order)
                if spc[j] < lowest or (spc[j] == lowest and row4col[j] == -1):
                    lowest = spc[j]
                    index = it
            min_val = lowest
            if min_val == np.inf:
                raise ValueError("cost matrix is infeasible")
            j = remaining[index]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            SC[j] = True
            remaining[index] = remaining[-1]
            remaining.pop()
        # ---- dual variables
        u[cur] += min_val
        for i in range(nr):
            if SR[i] and i != cur:
                u[i] += min_val - spc[col4row[i]]
        for j in range(nc):
            if SC[j]:
                v[j] -= min_val - spc[j]
        # ---- augment along the path
        j = sink
        while True:
            i = path[j]
            row4col[j] = i
            col4row[i], j = j, col4row[i]
            if i == cur:
                break
    return col4row


def hungarian_from_stats(S, n_gt):
    """S [B, K+2, K] float32 (rows < K: D = W_gt^T W_pred, row K: column sums of W_pred, row K+1: label counts),
    n_gt [B] -> match [B, K] int64, the restatement of hungarian_matching (losses_implementation.py:19-29):
    cost = D / clamp(cnt + col - D, 1e-10) in fp32, maximised over the first n_gt rows."""
    S = np.asarray(S, dtype=np.float32)
    B, K2, K = S.shape
    match = np.zeros((B, K), dtype=np.int64)
    for b in range(B):
        n = int(min(max(int(n_gt[b]), 0), K))
        D, col, cnt = S[b, :K], S[b, K], S[b, K + 1]
        den = (cnt[:, None] + col[None, :]) - D
        cost = D / np.maximum(den, np.float32(1e-10))
        if n:
            match[b, :n] = linear_sum_assignment_min(-cost[:n].astype(np.float64))
    return match
