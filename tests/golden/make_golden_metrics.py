"""Evaluation-metric fixture: SPFN/metric_implementation.compute_all_metrics of the
reference, imported here on CPU with the harness shims of make_golden.py, on a synthetic batch whose
predictions are a noisy copy of the ground truth.

    cd tests/golden && python make_golden_metrics.py      # needs /root/reference; writes metrics_2x2048.npz
                                                          # and metrics_padded_2x1024.npz

The second file holds the two PADDING branches of compute_all_metrics (reference lines 487-492, 505-508): more GT slots than
prediction columns (W / T padded) and more prediction columns than GT slots (T_gt, the GT axes and points_per_instance
padded; the reference hard-codes 512 points per instance there).
"""
import os
import sys

import numpy as np
import torch

from make_golden import import_reference, save

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CLASSES = ["plane", "sphere", "cylinder", "cone"]


def make_inputs(N=2048, K=28, Kgt=28, n_prims=6, seed=77):
    from cpfn_amd import synthetic
    B = 2
    batch = synthetic.training_batch(B, N=N, n_max_instances=Kgt, n_prims=n_prims, n_inst_points=512, seed=seed)
    g = torch.Generator().manual_seed(5)
    I = batch["I_gt"]
    logits = torch.randn(B, N, K, generator=g)
    lab = I.clamp(min=0)
    logits.scatter_add_(2, lab.unsqueeze(2), torch.full((B, N, 1), 3.0))
    perm = torch.stack([torch.randperm(K, generator=g) for _ in range(B)])          # predictions come in a shuffled order
    W = torch.softmax(torch.gather(logits, 2, perm.unsqueeze(1).expand(B, N, K)), dim=2)
    X = torch.nn.functional.normalize(batch["X_gt"] + 0.2 * torch.randn(B, N, 3, generator=g), dim=2)
    T_gt = batch["T_gt"]
    T = torch.randn(B, N, 4, generator=g)
    T.scatter_add_(2, torch.gather(T_gt, 1, lab).unsqueeze(2), torch.full((B, N, 1), 1.5))
    gt = {"plane_normal": batch["plane_n_gt"], "cylinder_axis": batch["cylinder_axis_gt"], "cone_axis": batch["cone_axis_gt"]}
    return dict(P=batch["P"], X=X, X_gt=batch["X_gt"], W=W, I_gt=I, T=T, T_gt=T_gt,
                points_per_instance=batch["points_per_instance"], **{"gt_" + k: v for k, v in gt.items()})


def run_reference(d):
    from SPFN import metric_implementation as mi
    gt = {k[3:]: v.clone() for k, v in d.items() if k.startswith("gt_")}
    with torch.no_grad():
        out = mi.compute_all_metrics(d["P"], d["X"], d["X_gt"], d["W"], d["I_gt"], d["T"], d["T_gt"],
                                     d["points_per_instance"], gt, list_epsilon=[0.01, 0.02], classes=CLASSES)
        mIoU, type_acc, normal_diff, axis_diff, mean_res, std_res, Sk, Pc, Wh, params, Tinst = out
        match, mask = mi.hungarian_matching(Wh, d["I_gt"])
    arrays = {k: v.numpy() for k, v in d.items()}
    arrays.update(mIoU=mIoU.numpy(), type_accuracy=type_acc.numpy(), normal_difference=normal_diff.numpy(),
                  axis_difference=axis_diff.numpy(), mean_residual=mean_res.numpy(), std_residual=std_res.numpy(),
                  Sk_coverage=np.stack([s.numpy() for s in Sk]), P_coverage=np.stack([p.numpy() for p in Pc]),
                  T_instance=Tinst.numpy(), matching=match.numpy(), mask=mask.numpy(), epsilons=np.array([0.01, 0.02]))
    arrays["W_hard"] = Wh.numpy()
    arrays.update({"param_" + k: v.numpy() for k, v in params.items()})
    for k in ("mIoU", "type_accuracy", "normal_difference", "axis_difference", "mean_residual", "std_residual",
              "Sk_coverage", "P_coverage"):
        print(k, arrays[k])
    return arrays


def main():
    import_reference()
    a = run_reference(make_inputs())
    for k in [k for k in a if k == "W_hard" or k.startswith("param_")]:       # (the first fixture keeps its round-2 contents)
        del a[k]
    save("metrics_2x2048.npz", **a)
    padded = {}
    # K = 12 predictions against 16 GT slots (6 of them used); K = 20 predictions against 12 GT slots (7 used)
    for tag, kw in (("few_", dict(N=1024, K=12, Kgt=16, n_prims=6, seed=78)), ("many_", dict(N=1024, K=20, Kgt=12, n_prims=7, seed=79))):
        a = run_reference(make_inputs(**kw))
        padded.update({tag + k: v for k, v in a.items()})
    save("metrics_padded_2x1024.npz", **padded)


if __name__ == "__main__":
    main()
