"""The reference's epoch loop, `spfn_train_val_epoch` (Utils/training_utils.py:84-176), on the replayed step.

Same arguments, same `(global_step, total_loss_)` return, same prints and the same sequence of `visualiser.log_loss` /
`visualiser.update` calls — so `training_SPFN.py:94-115` runs unchanged on it (`cpfn_amd.dropin.install(fast_epoch=True)`
makes `Utils.training_utils.spfn_train_val_epoch` resolve here; every other name of that module stays the reference's).
What differs is WHEN the host touches the device:

  reference, per batch                                   here
  ------------------------------------------------------ ------------------------------------------------------------
  8 blocking `.to(device)` of the batch (:122-132)       batch i+1 goes pinned -> device on a copy stream while step i-1
                                                         runs (three staging slots), and is announced to step i so that
                                                         its FPS / ball query / 3-NN run beside it (SPFNTrainer.step)
  forward, losses, backward, Adam as ~2000 launches      one hipGraph replay (bf16 model), or the trainer's eager step
  148 isinf/isnan syncs (:151-156)                       finite flag + skip on the device
  7 `.item()` (:147, :169-174) + 6 log_loss per batch    six loss scalars per batch into a device ring; ONE device->host
                                                         copy every 100th batch (where the reference prints anyway) and
                                                         at the end of the epoch, then the deferred log_loss / update
                                                         calls in their original order and the same `total_loss_` sum

`patch_selection_train_val_epoch` (Utils/training_utils.py:33-82, the loop of training_PatchSelection.py:79-86) is the same
machinery on `PatchSelectionTrainer`: two tensors per batch (points, per-point labels), one loss (cross-entropy of the two-class
heat map), one `visualiser.log_loss` per batch, the one-line print of :70 — and the reference's quirk that the module is put in
training mode whatever `network_mode` says (:50: the validation pass runs on batch statistics and moves the running ones).

`network_mode='val'`: forward + losses under `no_grad` in evaluation mode (running statistics; dropout stays on as in
pn2_network.py:63), the next batch's geometry prefetched on the side stream; nothing is back-propagated, `global_step` is
returned unchanged.
"""
import warnings

import torch

from . import training as _tr

_F, _L = torch.float32, torch.int64
# position in the data loader's tuple -> (key of the trainer's batch dict, dtype the reference casts to)   (:122-138)
_FIELDS = ((0, "P", _F), (1, "X_gt", _F), (2, "points_per_instance", _F), (3, "I_gt", _L), (4, "T_gt", _L),
           (5, "plane_n_gt", _F), (6, "cylinder_axis_gt", _F), (7, "cone_axis_gt", _F))
_LOCAL_FIELDS = ((8, "glob_features", _F), (9, "loc_features", _F))
_LOG_NAMES = ("loss", "normal_loss", "type_loss", "miou_loss", "residue_loss", "parameter_loss")
_PS_FIELDS = ((0, "P", _F), (1, "labels", _L))                 # patch_selection_train_val_epoch, :57-60
LOG_EVERY = 100                # the reference prints the six losses at every 100th batch (:87, :167-174)


class _Staging:
    """Host batches -> device, one batch ahead of the step that consumes them.  Three slots: the slot batch i+1 is copied
    into held batch i-2, whose last reader is step i-2 — complete (host-side wait on its event: the host never runs more
    than two steps ahead) long before step i-1, which the copy overlaps, has finished.  No device-side dependency of the
    copy stream on the trainer's stream; the trainer's stream waits for the copy's event, which has long fired."""
    SLOTS = 3

    def __init__(self, device):
        self.device = device
        self.stream = torch.cuda.Stream(device=device)
        self.buf = [dict() for _ in range(self.SLOTS)]           # slot -> {shape signature: (pinned dict, device dict)}
        self.copied = [None] * self.SLOTS                         # event: the slot's host -> device copies are done
        self.read = [None] * self.SLOTS                           # event: the last step that reads the slot is done
        self.keep = [None] * self.SLOTS                           # loader tensors a copy still reads from
        self.turn = 0

    def put(self, fields):
        """fields: list of (key, CPU or device tensor, dtype).  Returns (slot, {key: device tensor})."""
        s = self.turn
        self.turn = (s + 1) % self.SLOTS
        if self.read[s] is not None:
            self.read[s].synchronize()
        if self.copied[s] is not None:
            self.copied[s].synchronize()
        sig = tuple((k, tuple(t.shape)) for k, t, _ in fields)
        if sig not in self.buf[s]:
            if len(self.buf[s]) >= 2:                             # (full batches + the epoch's ragged last one)
                self.buf[s].clear()
            self.buf[s][sig] = ({}, {k: torch.empty(t.shape, dtype=dt, device=self.device) for k, t, dt in fields})
        pinned, dev = self.buf[s][sig]
        keep = []
        with torch.cuda.stream(self.stream):
            for k, t, dt in fields:
                if t.is_cuda:                                      # (a loader that already yields device tensors)
                    self.stream.wait_stream(torch.cuda.current_stream(t.device))
                    dev[k].copy_(t, non_blocking=True)
                    keep.append(t)
                    continue
                if not (t.is_pinned() and t.dtype == dt and t.is_contiguous()):
                    # pageable (or to be cast like `.type(torch.FloatTensor)`): through this slot's pinned twin
                    if k not in pinned:
                        pinned[k] = torch.empty(t.shape, dtype=dt).pin_memory()
                    pinned[k].copy_(t)
                    t = pinned[k]
                else:
                    keep.append(t)                                 # DataLoader(pin_memory=True): copied from where it lies
                dev[k].copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.copied[s], self.keep[s] = ev, keep
        return s, dev

    def consumed(self, slot, stream):
        ev = torch.cuda.Event()
        ev.record(stream)
        self.read[slot] = ev


class EpochRunner:
    """What persists between two calls of `spfn_train_val_epoch` for one network: the trainer (optimizer state, captured
    graphs), the staging buffers and the loss ring.  Kept on the module as `_cpfn_epoch_runner`."""

    def __init__(self, spfn_module, optimizer, conf, device, kind="spfn"):
        g = optimizer.param_groups[0]
        lr0 = g["lr"]
        self.kind = kind
        self.device = torch.device(device)
        on_gpu = self.device.type == "cuda"
        bf16 = getattr(spfn_module, "compute_dtype", torch.float32) == torch.bfloat16
        common = dict(batch_size=conf.get_batch_size(), init_learning_rate=conf.get_init_learning_rate(),
                      decay_step=conf.get_decay_step(), decay_rate=conf.get_decay_rate(), bn_decay_step=conf.get_bn_decay_step(),
                      use_graphs=on_gpu and bf16)
        if kind == "patch_selection":
            self.trainer = _tr.PatchSelectionTrainer(spfn_module, **common)
            self.n_losses = 1
        else:
            mult = dict(miou=conf.get_miou_loss_multiplier(), normal=conf.get_normal_loss_multiplier(),
                        type=conf.get_type_loss_multiplier(), parameter=conf.get_parameter_loss_multiplier(),
                        residue=conf.get_residue_loss_multiplier(), total=conf.get_total_loss_multiplier())
            self.trainer = _tr.SPFNTrainer(spfn_module, multipliers=mult, classes=conf.get_list_of_primitives(), **common)
            self.n_losses = 6
        self.optimizer = optimizer
        self._adopt_optimizer(optimizer, float(lr0))
        self.staging = _Staging(self.device) if on_gpu else None
        self.ring = torch.zeros(LOG_EVERY + 1, 6, dtype=torch.float32, device=self.device)
        self.ring_host = torch.zeros(LOG_EVERY + 1, 6, dtype=torch.float32)
        if on_gpu:
            self.ring_host = self.ring_host.pin_memory()

    def _adopt_optimizer(self, optimizer, lr0):
        """The caller's `torch.optim.Adam` (training_SPFN.py:90) keeps describing the training: its hyper-parameters are
        the flat optimizer's, moments it already holds are taken over, and afterwards its per-parameter state ARE views of
        the flat moment buffers, so `optimizer.state_dict()` saves what was trained."""
        from .optim import FlatAdam
        flat = self.trainer.optimizer
        if not isinstance(flat, FlatAdam):
            # CPU (tests) / fused_adam off: the trainer would step its own torch Adam — use the caller's instead
            self.trainer.optimizer = optimizer
            return
        g = optimizer.param_groups[0]
        fg = flat.param_groups[0]
        fg["betas"], fg["eps"], fg["weight_decay"] = tuple(g["betas"]), g["eps"], g["weight_decay"]
        flat.lr_dev.fill_(lr0)
        off = 0
        with torch.no_grad():
            for p in self.trainer.bucket.params:
                n = p.numel()
                st = optimizer.state.get(p)
                if st and "exp_avg" in st:
                    flat.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                    flat.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                    step = float(st["step"])
                    flat.step_count.fill_(step)
                    flat.beta_pows.copy_(torch.tensor([fg["betas"][0] ** step, fg["betas"][1] ** step], dtype=torch.float64))
                optimizer.state[p] = {"step": flat.step_count, "exp_avg": flat.exp_avg[off:off + n].view_as(p),
                                      "exp_avg_sq": flat.exp_avg_sq[off:off + n].view_as(p)}
                off += n

    # ---- one epoch ----------------------------------------------------------------------------------------------
    def _fields(self, data, local):
        if self.kind == "patch_selection":
            return [(k, data[i], dt) for i, k, dt in _PS_FIELDS]
        f = [(k, data[i], dt) for i, k, dt in _FIELDS]
        m = self.trainer.module
        if local and (getattr(m, "use_glob_features", False) or getattr(m, "use_loc_features", False)):
            f += [(k, data[i], dt) for i, k, dt in _LOCAL_FIELDS]       # (:136-137; ignored by a network built without them)
        return f

    def _stage(self, data, local):
        if data is None:
            return None
        f = self._fields(data, local)
        if self.staging is None:
            return None, {k: t.to(dt) for k, t, dt in f}
        return self.staging.put(f)

    def _loss_vector(self, out):
        """The step's loss scalars as one [6] tensor (unused entries stay zero).  The fused loss tail writes the six side by
        side in one buffer: then this is a view (no launch); otherwise (op-by-op losses, one-loss objectives) they are stacked."""
        o0 = out[0]
        if len(out) >= 6:
            try:
                base = o0.untyped_storage().data_ptr()
                if all(o.dtype == torch.float32 and o.dim() == 0 and o.untyped_storage().data_ptr() == base and
                       o.storage_offset() == o0.storage_offset() + i for i, o in enumerate(out[:6])):
                    return torch.as_strided(o0.detach(), (6,), (1,))
            except RuntimeError:
                pass
        v = [o.detach().reshape(()).float() for o in out[:6]]
        return torch.stack(v + [torch.zeros_like(v[0])] * (6 - len(v)))

    def run(self, dataloader, epoch, global_step, visualiser, args, network_mode):
        tr, mod = self.trainer, self.trainer.module
        train = network_mode == 'train'
        local = getattr(args, "network", "GlobalSPFN") == 'LocalSPFN'
        B_full = tr.batch_size
        # the staircases start from the values at the epoch's first step, like the reference's `old_*` (:99-100): a value is
        # only written to the modules / the optimizer when it CHANGES inside an epoch
        tr.global_step = int(global_step)
        tr._bn_momentum = _tr.get_batch_norm_decay(global_step, B_full, tr.bn_decay_step)
        tr._lr = _tr.get_learning_rate(tr.init_learning_rate, global_step, B_full, tr.decay_step, tr.decay_rate)
        ps = self.kind == "patch_selection"
        # (patch_selection_train_val_epoch puts the module in training mode whatever the mode says, :45-50)
        mod.train() if (train or ps) else mod.eval()
        total, pending, sizes = 0.0, 0, []
        on_gpu = self.staging is not None

        def flush(last_print=False):
            nonlocal total, pending
            if pending == 0:
                return None
            self.ring_host[:pending].copy_(self.ring[:pending], non_blocking=on_gpu)
            if on_gpu:
                torch.cuda.current_stream(self.device).synchronize()
            if on_gpu:
                from . import ops as _ops
                _ops.check_fps_faults("this point of the epoch")
            rows = self.ring_host[:pending].tolist()
            for b, row in zip(sizes, rows):
                total += b * row[0]                                                   # (:147; :69)
                for v, name in zip(row[:self.n_losses], _LOG_NAMES):
                    visualiser.log_loss(v, '%s_%s' % (network_mode, name))          # (:176-181; :80)
                visualiser.update()
            pending = 0
            del sizes[:]
            return rows[-1]

        stream_ctx = torch.cuda.stream(tr.stream(self.device)) if on_gpu and tr.use_graphs else _Null()
        with stream_ctx:
            it = iter(dataloader)
            cur = self._stage(next(it, None), local)
            batch_id = 0
            while cur is not None:
                nxt = self._stage(next(it, None), local)
                if batch_id % LOG_EVERY == 0 and not ps:
                    print('[%s][Epoch %d - Iteration %d]' % (network_mode, epoch, batch_id))
                slot, batch = cur
                if on_gpu:
                    me = torch.cuda.current_stream(self.device)
                    me.wait_event(self.staging.copied[slot])
                    if nxt is not None:
                        me.wait_event(self.staging.copied[nxt[0]])
                B_cur = batch["P"].shape[0]
                if train:
                    lr_before = tr._lr
                    # the epoch's ragged last batch runs as eager launches; the batch in front of it does not announce it
                    # (the replayed graphs' static buffers have the full batch's shapes)
                    ragged = tr.use_graphs and (B_cur != B_full or (tr._graph is not None and any(
                        k in tr._graph["batch"] and tr._graph["batch"][k].shape != v.shape for k, v in batch.items())))
                    announce = nxt is not None and not ragged and nxt[1]["P"].shape == batch["P"].shape
                    out = tr.step(batch, next_batch=nxt[1] if announce else None, force_eager=ragged)
                    if tr._lr != lr_before:                                 # the caller's optimizer shows the staircase (:119-121)
                        for group in self.optimizer.param_groups:
                            if not isinstance(group['lr'], torch.Tensor):
                                group['lr'] = tr._lr
                else:
                    out = tr.eval_losses(batch, next_batch=nxt[1] if nxt is not None else None)
                self.ring[pending].copy_(self._loss_vector(out), non_blocking=True)
                pending += 1
                sizes.append(B_cur)
                if on_gpu:
                    self.staging.consumed(slot, torch.cuda.current_stream(self.device))
                if batch_id % LOG_EVERY == 0 or pending >= LOG_EVERY:
                    row = flush()
                    if batch_id % LOG_EVERY == 0 and ps:
                        print('[%s][Epoch %d - Iteration %d] Loss: %f' % (network_mode, epoch, batch_id, row[0]))      # (:70)
                    elif batch_id % LOG_EVERY == 0:
                        for label, v in zip(('Loss Value: ', 'Normal Loss', 'Type Loss', 'mIoU Loss', 'Residue Loss',
                                             'Parameter Loss'), row):
                            print(label, v)
                cur = nxt
                batch_id += 1
            flush()
        return tr.global_step if train else global_step, total


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _epoch(kind, ref_name, dataloader, module, epoch, optimizer, global_step, visualiser, args, conf, device, network_mode):
    assert network_mode in ['train', 'val']
    runner = module.__dict__.get("_cpfn_epoch_runner")
    if runner is None or runner.optimizer is not optimizer or runner.kind != kind:
        if not isinstance(optimizer, torch.optim.Adam) or len(optimizer.param_groups) != 1 or \
                optimizer.param_groups[0].get("amsgrad") or optimizer.param_groups[0].get("maximize"):
            from .Utils import training_utils as _tu
            warnings.warn("cpfn_amd: the fast epoch loop takes a plain torch.optim.Adam with one parameter group; running "
                          "the reference's own loop with this optimizer")
            return getattr(_tu._load_reference_module(), ref_name)(dataloader, module, epoch, optimizer, global_step, visualiser,
                                                                  args, conf, device, network_mode)
        runner = EpochRunner(module, optimizer, conf, device, kind)
        module.__dict__["_cpfn_epoch_runner"] = runner
    if network_mode == 'val':
        with torch.no_grad():
            return runner.run(dataloader, epoch, global_step, visualiser, args, network_mode)
    return runner.run(dataloader, epoch, global_step, visualiser, args, network_mode)


def spfn_train_val_epoch(dataloader, spfn_module, epoch, optimizer, global_step, visualiser, args, conf, device,
                         network_mode='train'):
    """Drop-in for Utils/training_utils.py:84-176 (see the module docstring).  `optimizer` must be the
    `torch.optim.Adam` over `spfn_module.parameters()` that training_SPFN.py:90 builds; anything else runs the
    reference's own loop."""
    return _epoch("spfn", "spfn_train_val_epoch", dataloader, spfn_module, epoch, optimizer, global_step, visualiser, args, conf,
                  device, network_mode)


def patch_selection_train_val_epoch(dataloader, patchselec_module, epoch, optimizer, global_step, visualiser, args, conf, device,
                                    network_mode='train'):
    """Drop-in for Utils/training_utils.py:33-82 (training_PatchSelection.py:79-86): same arguments, return, print and
    visualiser calls, on the replayed step of `PatchSelectionTrainer`."""
    return _epoch("patch_selection", "patch_selection_train_val_epoch", dataloader, patchselec_module, epoch, optimizer,
                  global_step, visualiser, args, conf, device, network_mode)
