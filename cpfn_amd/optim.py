"""Adam on flat buffers (cpfn_adam_flat): the optimizer of the training step as one streaming kernel.

`FlatAdam` is a torch.optim.Optimizer with torch.optim.Adam's arithmetic and hyper-parameters (the reference
builds `optim.Adam(spfn_module.parameters(), lr=...)`, training_SPFN.py; torch/optim/adam.py).  It re-points
every parameter at a view of ONE contiguous fp32 buffer and reads the gradients from the trainer's
FlatGradBucket, so a step is a single launch over 1.4 M elements instead of a multi-tensor launch over ~100
tensors.  Capturable: learning rate, step count and `found_inf` (set it before `step()` to skip on non-finite
gradients) are device scalars.  GPU only — on CPU the trainer keeps torch.optim.Adam."""
import torch

from . import lib as _l
from .ops import _ptr, _stream


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = list(bucket.params)
        dev = bucket.flat.device
        if not bucket.flat.is_cuda:
            raise RuntimeError("FlatAdam runs on the GPU only (CPU not supported)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.bucket = bucket
        n = bucket.flat.numel()
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in params:
                if p.dtype != torch.float32:
                    raise RuntimeError("FlatAdam expects fp32 master parameters")
                k = p.numel()
                view = self.flat_p[off:off + k].view_as(p)
                view.copy_(p.data)
                p.data = view                      # same nn.Parameter object, storage inside the flat buffer
                off += k
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step_count = torch.zeros((), dtype=torch.float32, device=dev)
        self.lr_dev = torch.zeros((), dtype=torch.float32, device=dev)
        self._coef = torch.zeros(3, dtype=torch.float32, device=dev)
        self.beta_pows = torch.ones(2, dtype=torch.float64, device=dev)      # beta1^step, beta2^step
        self._lr_host = None
        self.found_inf = None
        self._nf_ws = None
        g = self.param_groups[0]
        g["lr"] = self.lr_dev                      # the trainer's staircase does `group['lr'].fill_(lr)`
        self.lr_dev.fill_(float(lr))

    @torch.no_grad()
    def step(self, closure=None, check_gradients=False, skipped=None, nf_flags=None, combine=None):
        """check_gradients: scan the flat gradient for NaN / inf in the same launch sequence and skip the step if any
        is found (the reference's per-parameter isinf/isnan scan, Utils/training_utils.py:151-156) — the scan's final
        reduction and the optional `skipped` device counter (+1 per skipped step) live in the 1-wave prepare kernel.
        nf_flags = (int32 flags tensor, count[, n_sticky]): per-workgroup flags of a scan that already happened (the checked packing
        copy of FlatGradBucket.collect); the step is skipped if any of them is set.  The first n_sticky words are OR-ed into by
        the launches that wrote the gradients (fused_mlp.GradSink) and are cleared by this step's prepare kernel.
        combine = (S [7 C], coef [3, C], C, out [C, 3] — a slice of the flat gradient): the pass's last gradient, finished (and
        checked) by the prepare kernel (fused_mlp.GradSink.combine, cpfn_adam_flat_xw)."""
        g = self.param_groups[0]
        lr = g["lr"]
        if not isinstance(lr, torch.Tensor):       # someone assigned a float: mirror it into the device scalar
            if lr != self._lr_host:
                self.lr_dev.fill_(float(lr))
                self._lr_host = lr
            lr = self.lr_dev
        b1, b2 = g["betas"]
        h = _l.lib()
        grads = self.bucket.flat
        with torch.cuda.device(self.flat_p.device):
            nf_ws, nf_count, n_sticky = None, 0, 0
            if nf_flags is not None:
                nf_ws, nf_count = nf_flags[0], nf_flags[1]
                n_sticky = nf_flags[2] if len(nf_flags) > 2 else 0       # words OR-ed into by the gradients' writers: cleared here
            elif check_gradients:
                if self._nf_ws is None:
                    self._nf_ws = torch.empty(256, dtype=torch.int32, device=self.flat_p.device)
                nf_ws, nf_count = self._nf_ws, h.cpfn_nonfinite_blocks(grads.numel())
                _l.check(h.cpfn_nonfinite_partial(_ptr(grads), grads.numel(), _ptr(nf_ws), _stream()), "cpfn_nonfinite_partial")
            xS, xc, xC, xo = combine if combine is not None else (None, None, 0, None)
            _l.check(h.cpfn_adam_flat_xw(_ptr(self.flat_p), _ptr(grads), _ptr(self.exp_avg), _ptr(self.exp_avg_sq),
                                         self.flat_p.numel(), _ptr(lr), float(b1), float(b2), float(g["eps"]),
                                         float(g["weight_decay"]), _ptr(self.step_count), _ptr(self.beta_pows),
                                         _ptr(self.found_inf), _ptr(self._coef), _ptr(nf_ws), nf_count, n_sticky, _ptr(skipped),
                                         _ptr(xS), _ptr(xc), int(xC), _ptr(xo), _stream()), "cpfn_adam_flat")
        _l.add_bytes("cpfn_adam_flat", 28 * self.flat_p.numel() + (4 * grads.numel() if check_gradients and nf_flags is None else 0))

    def state_dict(self):
        return {"flat": {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self.step_count,
                         "beta_pows": self.beta_pows},
                "hyper": {k: (float(v) if isinstance(v, torch.Tensor) else v) for k, v in self.param_groups[0].items()
                          if k != "params"}}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["flat"]["exp_avg"])
        self.exp_avg_sq.copy_(sd["flat"]["exp_avg_sq"])
        self.step_count.copy_(sd["flat"]["step"])
        self.beta_pows.copy_(sd["flat"]["beta_pows"])
        for k, v in sd.get("hyper", {}).items():
            if k == "lr":
                self.lr_dev.fill_(float(v))
            else:
                self.param_groups[0][k] = v
