"""One SPFN training step, reproducing the sequence of the reference's
`spfn_train_val_epoch` (Utils/training_utils.py:84-176): BN-momentum / LR staircase
schedules, forward, normalise + softmax of the heads, all five losses (incl. the fused
fitters), backward, non-finite-gradient guard, Adam — plus what the reference lacks:
data-parallel training, one process per GPU, with ONE flat fp32 gradient bucket
all-reduced over RCCL (xGMI) per step.
"""
import numpy as np
import torch
import torch.distributed as dist

from .SPFN import losses_implementation

GLOBAL_SPFN_CLASSES = ['sphere', 'plane', 'cylinder', 'cone']   # Configs/config_globalSPFN.yml:13-17


def get_batch_norm_decay(global_step, batch_size, bn_decay_step, staircase=True):
    """max(0.5 * 0.5^floor(step*bs/decay_step), 0.01)   (training_utils.py:9-17)."""
    p = global_step * batch_size / bn_decay_step
    if staircase:
        p = int(np.floor(p))
    return max(0.5 * (0.5 ** p), 1 - 0.99)


def update_momentum(module, bn_momentum):
    """Every sub-module whose qualified name contains 'bn'   (training_utils.py:19-22)."""
    for name, sub in module.named_modules():
        if 'bn' in name:
            sub.momentum = bn_momentum


def get_learning_rate(init_learning_rate, global_step, batch_size, decay_step, decay_rate, staircase=True):
    """init * rate^floor(step*bs/decay_step)   (training_utils.py:25-30)."""
    p = global_step * batch_size / decay_step
    if staircase:
        p = int(np.floor(p))
    return init_learning_rate * (decay_rate ** p)


class FlatGradBucket:
    """All gradients of a module as views into one contiguous fp32 buffer, so the
    data-parallel exchange is a single all-reduce with no packing copies.  5.6 MB for
    GlobalSPFN: latency-bound on xGMI, hence one bucket rather than DDP's 25 MB chunks."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=torch.float32, device=ref.device)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())

    def finite(self):
        """One fused reduction instead of the reference's per-parameter isinf/isnan scan
        (training_utils.py:151-156: two host syncs per tensor)."""
        return torch.isfinite(self.flat).all()


def broadcast_parameters(module, src=0):
    """Identical replicas at start: rank-0 state to everyone (parameters and BN buffers)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)


class SPFNTrainer:
    """Holds the optimizer state and the schedule bookkeeping of the reference's epoch loop."""

    def __init__(self, module, batch_size=16, init_learning_rate=1e-3, decay_step=200000, decay_rate=0.7,
                 bn_decay_step=200000, multipliers=None, classes=None, fused_adam=None):
        self.module = module
        self.batch_size = batch_size
        self.init_learning_rate, self.decay_step, self.decay_rate = init_learning_rate, decay_step, decay_rate
        self.bn_decay_step = bn_decay_step
        self.mult = dict(miou=1.0, normal=1.0, type=1.0, parameter=1.0, residue=1.0, total=1.0)
        if multipliers:
            self.mult.update(multipliers)
        self.classes = list(classes) if classes is not None else list(GLOBAL_SPFN_CLASSES)
        self.bucket = FlatGradBucket(module)
        on_gpu = self.bucket.flat.is_cuda
        self.optimizer = torch.optim.Adam(module.parameters(), lr=init_learning_rate,
                                          fused=on_gpu if fused_adam is None else fused_adam)
        self.global_step = 0
        self._bn_momentum = get_batch_norm_decay(0, batch_size, bn_decay_step)
        self._lr = get_learning_rate(init_learning_rate, 0, batch_size, decay_step, decay_rate)
        update_momentum(module, self._bn_momentum)
        self.skipped_steps = 0
        self.fused_losses = True      # HIP loss kernels when the model exposes its packed fp32 heads

    def _schedules(self):
        m = get_batch_norm_decay(self.global_step, self.batch_size, self.bn_decay_step)
        if m != self._bn_momentum:
            update_momentum(self.module, m)
            self._bn_momentum = m
        lr = get_learning_rate(self.init_learning_rate, self.global_step, self.batch_size, self.decay_step,
                               self.decay_rate)
        if lr != self._lr:
            for group in self.optimizer.param_groups:
                group['lr'] = lr
            self._lr = lr

    def losses(self, batch, fps_start=None):
        """Forward + losses (training_utils.py:140-146).  Returns the reference's 6 scalars."""
        P = batch["P"]
        X, T, W, _, _ = self.module(P, fps_start=fps_start)
        packed = getattr(self.module, "heads_packed", None)
        if self.fused_losses and packed is not None and len(self.classes) == 4 and T.shape[2] == 4:
            from .SPFN import fused_losses
            return fused_losses.fused_losses(P, packed, batch, self.mult, self.classes)
        X = torch.nn.functional.normalize(X, p=2, dim=2, eps=1e-12)
        W = torch.softmax(W, dim=2)
        gt = {'plane_normal': batch["plane_n_gt"], 'cylinder_axis': batch["cylinder_axis_gt"],
              'cone_axis': batch["cone_axis_gt"]}
        m = self.mult
        out = losses_implementation.compute_all_losses(
            P, W, batch["I_gt"], X, batch["X_gt"], T, batch["T_gt"], gt, batch["points_per_instance"],
            m["normal"], m["type"], m["miou"], m["residue"], m["parameter"], m["total"], False,
            mode_seg='mIoU', classes=self.classes)
        return out[:6]

    def step(self, batch, fps_start=None):
        """One optimisation step; returns the 6 loss tensors (still on the device — the
        reference's six `.item()` syncs per step, training_utils.py:169-174, are left to the caller)."""
        self.module.train()
        self.bucket.zero()
        self._schedules()
        out = self.losses(batch, fps_start)
        out[0].backward()
        self.bucket.all_reduce_mean()
        if bool(self.bucket.finite()):                 # single host sync (reference: 148)
            self.optimizer.step()
        else:
            self.skipped_steps += 1
        self.global_step += 1
        return out
