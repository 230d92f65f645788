"""One SPFN training step, reproducing the sequence of the reference's
`spfn_train_val_epoch` (Utils/training_utils.py:84-176): BN-momentum / LR staircase
schedules, forward, normalise + softmax of the heads, all five losses (incl. the fused
fitters), backward, non-finite-gradient guard, Adam — plus what the reference lacks:
data-parallel training, one process per GPU, with ONE flat fp32 gradient bucket
all-reduced over RCCL (xGMI) per step.
"""
import os

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


# The next batch's geometry runs concurrently with the step as a SECOND linear graph on a side stream (a forked branch
# inside the step's graph switches the whole replay to a slower dispatch mode on this stack: DESIGN.md §4), and the two
# graphs order themselves through two device flags polled / set by one-lane kernels (cpfn_flag_wait / cpfn_flag_set) —
# the closed cross-queue event cycle of two events per step costs ~100 us of idle GPU per step.
#
# CPFN_FLAG_TIMEOUT_S: how long a flag waiter polls before it gives up (then: a sticky pinned error word the host checks
# at every step AND a sticky device word that is the optimizer's skip flag, so that nothing computed after a broken
# hand-over reaches the weights even though the host may have queued several steps by then).  The waiters only wait long
# when the OTHER stream is stalled from outside — with the gradient all-reduce inside the step's graph that is any peer
# stall (a rank-0 checkpoint or evaluation, a data-loader hiccup) — so the default is far above any collective stall
# for data-parallel runs and short for one GPU (where a timeout can only mean a host-side error between two launches).
# CPFN_DP_COLLECTIVE: how the flat gradient bucket is averaged over the ranks.  "all_reduce" (default): ONE collective with
# in-collective averaging.  "rs_ag": reduce-scatter + all-gather on the same bucket (SURVEY §8e argues that on 7
# point-to-point xGMI links a direct reduce-scatter / all-gather pair can beat a ring all-reduce of this size; RCCL picks its
# own algorithm for either, so this is an A/B switch for a SCALE run, reported by bench.py as config.collective).

class _CopyDesc(__import__("ctypes").Structure):
    _fields_ = [("src", __import__("ctypes").c_void_p), ("dst", __import__("ctypes").c_void_p),
                ("bytes", __import__("ctypes").c_longlong)]


_copy_desc_cache = {}

DP_COLLECTIVE = os.environ.get("CPFN_DP_COLLECTIVE", "all_reduce")
_SIDE_GRAPH_FIRST = os.environ.get("CPFN_SIDE_GRAPH_FIRST", "0") == "1"
_BUCKET_PAD = 3360            # elements; 4 x lcm(1..8): every world size up to 8 (and 10, 12, 14, 15, 16 ...) gets 16-byte-aligned shards


def _flag_timeout_ticks(world):
    s = os.environ.get("CPFN_FLAG_TIMEOUT_S")
    seconds = float(s) if s else (10.0 if world == 1 else 1800.0)
    return int(seconds * 100e6)            # ticks of the 100 MHz device wall clock


class FlatGradBucket:
    """All gradients of a module in one contiguous fp32 buffer, so the data-parallel exchange is a
    single all-reduce.  5.6 MB for GlobalSPFN: latency-bound on xGMI, hence one bucket rather than
    DDP's 25 MB chunks.

    Gradients are produced by autograd into fresh tensors (`.grad = None` before the backward pass, so
    no per-parameter accumulate kernels run), gathered into the flat buffer by ONE multi-tensor copy,
    and `.grad` is then re-pointed at views of the flat buffer for the all-reduce and the optimizer."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        # parameters whose gradients are produced as ONE packed block (PointNet2.packed_parameter_groups: the fc2 heads' weights,
        # then their biases) sit adjacent, in that order, at the END of the bucket: the launch that reduces the packed gradient then
        # writes the bucket directly (fused_mlp.GradSink); behind them nothing needs 16-byte alignment any more
        groups = module.packed_parameter_groups() if hasattr(module, "packed_parameter_groups") else []
        moved = [p for g in groups for p in g if p.requires_grad]
        if moved and len({id(p) for p in moved}) == len(moved) and all(any(p is q for q in self.params) for p in moved):
            ids = {id(p) for p in moved}
            self.params = [p for p in self.params if id(p) not in ids] + moved
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        # (padded so that the bucket splits into equal shards for the reduce-scatter / all-gather layout; `flat` is the
        #  unpadded view everybody else uses, the padding stays zero)
        #  ... except its first element, the FAULT SLOT: under data parallelism a rank's sticky "a cross-stream flag wait timed
        #  out" word is packed there by the gradient-packing launch and rides on the same collective, so that every rank skips
        #  the optimizer step when ANY rank faulted — a per-rank skip would let the replicas diverge)
        self.padded = torch.zeros((n + 1 + _BUCKET_PAD - 1) // _BUCKET_PAD * _BUCKET_PAD, dtype=torch.float32, device=ref.device)
        self.flat = self.padded[:n]
        self.fault_slot = self.padded[n:n + 1]
        self._exchanged = self.padded[:n + 1]          # what the all-reduce layout moves: gradients + fault slot
        self._shard = None
        self.collective = DP_COLLECTIVE if DP_COLLECTIVE in ("all_reduce", "rs_ag") else "all_reduce"
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self._has_grad = set()            # indices of the parameters whose flat slice holds a gradient of an earlier step
        # the gradient sink's flags (allocated HERE, never inside a capture — a torch.zeros there is a fill node of every replay):
        # word 0 = the writers' sticky flag, words 1.. = the per-workgroup flags of a checked copy of whatever was not written in place
        self._sink_flags = torch.zeros(4096, dtype=torch.int32, device=ref.device) if ref.is_cuda else None
        self.zero()

    def zero(self):
        """Call before every backward pass."""
        for p in self.params:
            p.grad = None

    def sink(self, check, combine_ok=False):
        """fused_mlp.grad_sink over this bucket for the backward pass of a step (GPU, bf16 fused path): the launches that produce
        parameter gradients write them into the bucket's views and OR a NaN / inf into `sink_flag` (check=True).
        combine_ok: the caller's optimizer step takes GradSink.combine (FlatAdam.step(combine=...))."""
        from . import fused_mlp
        flag = self._sink_flags[0:1] if (self.flat.is_cuda and check) else None
        return fused_mlp.grad_sink(self.params, self.views, self.flat, flag, combine_ok)

    def collect(self, check=False, fault=None, sink=None):
        """Call after the backward pass: pack the fresh gradients into the flat buffer.  Parameters that
        received no gradient (conv biases in front of a training-mode BatchNorm) keep a zero slice and
        `.grad = None`, which the optimizer skips — identical to a zero update.  A parameter that HAD a gradient in
        an earlier step and has none now (a loss multiplier switched off, an unused head) gets its slice cleared,
        so neither the flat optimizer nor the all-reduce ever sees a stale gradient.
        check=True (GPU): the packing copy also scans what it copies for NaN / inf (cpfn_multi_copy_checked) and the
        per-workgroup flags are returned as (flags int32 tensor, count) for FlatAdam.step(nf_flags=...); None when
        nothing had to be copied or the scan could not ride along.
        sink: the fused_mlp.GradSink that was armed during the backward pass (self.sink(check)): gradients it covered are
        already in place and checked by the launches that wrote them (its sticky flag word = flags[0]); anything else is
        copied — and scanned — as before, its flags behind that word.  Returns (flags, count, n_sticky = 1) then.
        fault (data parallel): this rank's 0-dim fp32 fault word; the same launch copies it into the bucket's fault slot."""
        src, dst, who, in_place = [], [], [], set()
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            if p.grad is None:
                if i in self._has_grad:
                    v.zero_()
                    self._has_grad.discard(i)
                continue
            self._has_grad.add(i)
            if p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
                who.append(p)
            else:
                in_place.add(id(p))
        flags = None
        if fault is not None:
            src.append(fault.reshape(1))
            dst.append(self.fault_slot)
        if sink is not None and sink.flag is not None and check and self.flat.is_cuda:
            # every gradient that is in place must have been written by a checked launch of THIS pass
            if in_place <= sink.covered:
                count = 1
                if src:
                    if not all(t.dtype == torch.float32 for t in src):
                        sink = None
                    else:
                        got = SPFNTrainer._copy_all(dst, src, flags=self._sink_flags[1:])
                        if got is None:
                            sink = None                     # (the scan could not ride on the copy: full scan by the optimizer)
                        else:
                            count += got[1]
                    for p, v in zip(who, dst):
                        p.grad = v
                return (self._sink_flags, count, 1) if sink is not None else None
        if src:
            if self.flat.is_cuda:
                if check:
                    if getattr(self, "_copy_flags", None) is None:
                        self._copy_flags = torch.empty(4096, dtype=torch.int32, device=self.flat.device)
                    flags = SPFNTrainer._copy_all(dst, src, flags=self._copy_flags)
                else:
                    SPFNTrainer._copy_all(dst, src)            # one launch (cpfn_multi_copy)
            else:
                torch._foreach_copy_(dst, src)
            for p, v in zip(who, dst):
                p.grad = v
        return flags

    def all_reduce_mean(self):
        """The step's one exchange: mean of the flat gradient over the ranks.  RCCL (backend "nccl"): ONE collective
        with the averaging done inside it (ncclAvg) — no scaling launch, capturable in the step's graph.  Other
        backends (gloo on CPU): sum + divide."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            world = dist.get_world_size()
            avg = self.flat.is_cuda and dist.get_backend() == "nccl"
            if self.collective == "rs_ag" and self.padded.numel() % world == 0:
                # reduce-scatter into this rank's shard, all-gather the averaged shards back into the bucket
                if self._shard is None:
                    self._shard = torch.empty(self.padded.numel() // world, dtype=torch.float32, device=self.padded.device)
                dist.reduce_scatter_tensor(self._shard, self.padded, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM)
                if not avg:
                    self._shard.div_(world)
                dist.all_gather_into_tensor(self.padded, self._shard)
            elif avg:
                dist.all_reduce(self._exchanged, op=dist.ReduceOp.AVG)
            else:
                dist.all_reduce(self._exchanged, op=dist.ReduceOp.SUM)
                self._exchanged.div_(world)

    def finite(self):
        """One fused reduction instead of the reference's per-parameter isinf/isnan scan
        (training_utils.py:151-156: two host syncs per tensor)."""
        return torch.isfinite(self.flat).all()

    def nonfinite_flag(self):
        """0-dim fp32 tensor: 1.0 if any gradient is NaN / inf, else 0.0 — one streaming kernel on the GPU
        (cpfn_nonfinite_flag), capturable; the optimizer skips its step when it is set."""
        if not self.flat.is_cuda:
            return (~torch.isfinite(self.flat).all()).float().reshape(())
        from . import lib as _l
        if getattr(self, "_nf_ws", None) is None:
            self._nf_ws = torch.empty(256, dtype=torch.int32, device=self.flat.device)
        flag = torch.empty((), dtype=torch.float32, device=self.flat.device)
        with torch.cuda.device(self.flat.device):
            _l.check(_l.lib().cpfn_nonfinite_flag(self.flat.data_ptr(), self.flat.numel(), self._nf_ws.data_ptr(), flag.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream), "cpfn_nonfinite_flag")
        return flag


class capture_guard:
    """No Python garbage collection inside a stream capture.  A dead reference cycle that owns device-side objects — a model
    with the trainer / epoch runner / auto-replay state hanging off it holds hipGraphs, pinned buffers, events — is destroyed
    whenever the cyclic collector happens to run; if that is in the middle of a capture, the destructors' HIP calls abort the
    process (seen as "Fatal Python error: Aborted ... Garbage-collecting" in a capture that followed the epoch tests).  torch < 2.10
    ran gc.collect() at the start of every `torch.cuda.graph`; 2.10 does not (`torch.compiler.config.force_cudagraph_gc`).  So:
    collect BEFORE the capture, keep the collector off DURING it."""

    def __enter__(self):
        import gc
        gc.collect()
        self._was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self._was:
            gc.enable()
        return False


def _quiesce_collectives(dev):
    """Before a stream capture in a multi-rank process: let the process group's watchdog thread retire every finished
    collective.  It polls the completion events of outstanding work from ITS thread (hipEventQuery, every 100 ms); a
    query that lands while this thread is capturing was seen to fail on this stack even in thread-local capture mode
    (ProcessGroupNCCL::WorkNCCL::finishedGPUExecutionInternal -> process abort, about 1 run in 5 when an eager all-reduce
    had been issued just before the capture).  After a device synchronisation and 0.25 s the watchdog's list is empty,
    and nothing is added to it while capturing (captured collectives are not tracked by the watchdog)."""
    import time
    torch.cuda.synchronize(dev)
    time.sleep(0.25)


def broadcast_parameters(module, src=0):
    """Identical replicas at start: rank-0 state to everyone (parameters and BN buffers)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)


class SPFNTrainer:
    """Holds the optimizer state and the schedule bookkeeping of the reference's epoch loop."""

    def __init__(self, module, batch_size=16, init_learning_rate=1e-3, decay_step=200000, decay_rate=0.7,
                 bn_decay_step=200000, multipliers=None, classes=None, fused_adam=None, use_graphs=False,
                 require_graphs=False):
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
        self.use_graphs = bool(use_graphs) and on_gpu
        self.require_graphs = bool(require_graphs)    # a failed capture raises instead of degrading to eager launches
        # GPU: Adam as one kernel over flat buffers (optim.FlatAdam; lr / step count / skip flag on the device, so
        # the LR staircase needs no re-capture).  `fused_adam=False` keeps torch.optim.Adam (also used on CPU).
        if on_gpu and fused_adam is not False:
            from .optim import FlatAdam
            self.optimizer = FlatAdam(self.bucket, lr=init_learning_rate)
        elif self.use_graphs:     # torch's optimizer inside a capture: fused + capturable, lr as a device tensor
            self.optimizer = torch.optim.Adam(module.parameters(), lr=torch.tensor(float(init_learning_rate), device=self.bucket.flat.device),
                                              fused=True, capturable=True)
        else:
            self.optimizer = torch.optim.Adam(module.parameters(), lr=init_learning_rate)
        self._graph, self._graph_warm = None, 0
        self._exchange_in_graph = os.environ.get("CPFN_EXCHANGE_IN_GRAPH", "1") != "0"
        self._gstream, self._in_gstream, self._gside = None, False, None
        self.global_step = 0
        # Like the reference (training_utils.py:100, 111-114) the momentum is only WRITTEN when the staircase value
        # changes: until the first change (step 12500 at batch 16) the BatchNorm layers keep their constructor
        # momentum (nn default 0.1), although the schedule's value is 0.5.
        self._bn_momentum = get_batch_norm_decay(0, batch_size, bn_decay_step)
        self._lr = get_learning_rate(init_learning_rate, 0, batch_size, decay_step, decay_rate)
        self._host_skipped = 0            # steps skipped on the host path (one sync per step, like the reference)
        self._skipped_dev = None          # device-side counter of the capturable / graph paths (survives re-captures)
        self.fused_losses = True      # HIP loss kernels when the model exposes its packed fp32 heads
        module.return_point_features = False    # the step consumes the heads only (no [B,128,N] fp32 conversion)
        self._side, self._prefetched = None, None
        self._comm_stamps = None          # two device wall-clock readings around the gradient exchange (world > 1)

    @property
    def skipped_steps(self):
        """Steps whose optimizer update was skipped because a gradient was NaN / inf (training_utils.py:151-158).
        Reading it synchronises with the device when the capturable / graph path keeps the count there."""
        dev = 0 if self._skipped_dev is None else int(round(float(self._skipped_dev)))
        return self._host_skipped + dev

    def _skip_counter(self, device):
        if self._skipped_dev is None:
            self._skipped_dev = torch.zeros((), dtype=torch.float32, device=device)
        return self._skipped_dev

    # ---- the rank's FAULT WORD (data parallel) --------------------------------------------------------------------------
    # One sticky 0-dim fp32 word per rank: non-zero = "what this rank computed may be wrong" (a cross-stream flag wait of the
    # replayed step timed out; `raise_fault()` for anything the caller finds — an FPS fault count, a failed loader).  The
    # gradient-packing launch copies it into the flat bucket's fault slot, it rides on the step's ONE collective (any rank's 1
    # makes the reduced slot non-zero everywhere) and the reduced value is the optimizer's skip flag: every replica skips the
    # same step, so the weights stay identical.  Then the raising rank errors (taking the process group down where torch can)
    # and its peers error too instead of waiting in the next collective.
    def fault_word(self, device):
        if getattr(self, "_fault", None) is None or self._fault.device != torch.device(device):
            self._fault = torch.zeros((), dtype=torch.float32, device=device)
            self._fault_raised = None
        return self._fault

    def raise_fault(self, reason="fault raised by the caller"):
        """Mark this rank's next step as not to be applied — on EVERY rank.  The step that carries the word still runs (its
        collective must match the peers'), skips the optimizer everywhere, and then raises."""
        dev = self.bucket.flat.device
        self.fault_word(dev).fill_(1.0)
        self._fault_raised = str(reason)

    @staticmethod
    def _abort_group():
        abort = getattr(dist.distributed_c10d, "_abort_process_group", None)
        try:
            if abort is not None:
                abort()
        except Exception:
            pass

    def _raise_reduced_fault(self, abort_as_peer=False):
        """The reduced fault slot was non-zero: this rank's own word (the process group goes down where torch can) or a peer's.
        abort_as_peer: a peer also takes the group down first — the replayed path, where this rank may already have queued later
        steps whose in-graph collective would wait for the partner that has stopped."""
        if getattr(self, "_fault_raised", None) is not None:
            self._abort_group()
            raise RuntimeError("cpfn_amd: this rank raised its fault word (%s); every rank skipped the optimizer step" % self._fault_raised)
        if abort_as_peer:
            self._abort_group()
        raise RuntimeError("cpfn_amd: a PEER rank raised its fault word; every rank skipped the optimizer step (this rank's "
                           "replica is intact: restart from it)")

    def _after_exchange_fault_check(self):
        """Host side of the fault word on the paths that read the device anyway (eager launches): after the exchange the
        bucket's fault slot is the ranks' mean."""
        if float(self.bucket.fault_slot) == 0.0:
            return
        self._raise_reduced_fault()

    def _schedules(self):
        m = get_batch_norm_decay(self.global_step, self.batch_size, self.bn_decay_step)
        if m != self._bn_momentum:
            update_momentum(self.module, m)
            self._bn_momentum = m
            self._graph = None            # the momentum is a kernel argument baked into the capture
        lr = get_learning_rate(self.init_learning_rate, self.global_step, self.batch_size, self.decay_step,
                               self.decay_rate)
        if lr != self._lr:
            for group in self.optimizer.param_groups:
                if isinstance(group['lr'], torch.Tensor):
                    group['lr'].fill_(lr)
                else:
                    group['lr'] = lr
            self._lr = lr

    # ---- geometry prefetch: FPS / ball query / 3-NN depend on coordinates only -----------------
    @staticmethod
    def _batch_key(P):
        """Identity of a batch's coordinates for the geometry hand-off: storage address AND the tensor's in-place
        version counter, so a caller that refills an announced buffer in place gets fresh geometry (recomputed)
        instead of the indices of the old contents.  `next_batch` must hold its final contents when announced."""
        return (P.data_ptr(), P._version)

    def prefetch(self, batch, fps_start=None):
        """Compute the next batch's index tensors on a side stream (overlaps with whatever the
        main stream is doing, typically the current step's backward pass: FPS alone is 0.7 ms of
        a 16-workgroup latency chain)."""
        P = batch["P"]
        if not (P.is_cuda and hasattr(self.module, "compute_geometry")):
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=P.device)
        main = torch.cuda.current_stream(P.device)
        self._side.wait_stream(main)
        from . import ops as _ops
        with torch.cuda.stream(self._side), _ops.background_geometry():
            geom = self.module.compute_geometry(P, fps_start)
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._prefetched = (self._batch_key(P), geom, ev)

    def _take_prefetched(self, P):
        pf, self._prefetched = self._prefetched, None
        if pf is None or pf[0] != self._batch_key(P):
            return None
        _, geom, ev = pf
        main = torch.cuda.current_stream(P.device)
        main.wait_event(ev)

        def mark(o):                    # allocated on the side stream, consumed on the main one
            if isinstance(o, torch.Tensor):
                o.record_stream(main)
            elif isinstance(o, dict):
                for v in o.values():
                    mark(v)
            elif isinstance(o, (list, tuple)):
                for v in o:
                    mark(v)
        mark(geom)
        return geom

    def _checked_optimizer_step(self, skipped, nf_flags=None, fault=None, combine=None):
        """Finite check of the flat gradient + optimizer step (skipped on the device when a NaN / inf is found) +
        `skipped` counter.  FlatAdam does all of it in its own launches — two when the scan already rode on the
        packing copy (nf_flags from FlatGradBucket.collect(check=True)); other optimizers get the flag tensor.
        fault: 0-dim fp32 device word that also skips the step when non-zero (the flag waiters' sticky time-out word)."""
        from .optim import FlatAdam
        if isinstance(self.optimizer, FlatAdam):
            self.optimizer.found_inf = fault
            self.optimizer.step(check_gradients=nf_flags is None, skipped=skipped, nf_flags=nf_flags, combine=combine)
        else:
            assert combine is None
            found = self.bucket.nonfinite_flag()
            self.optimizer.found_inf = found if fault is None else torch.maximum(found, fault)
            self.optimizer.step()
            skipped += self.optimizer.found_inf

    @staticmethod
    def _feature_kwargs(batch):
        """LocalSPFN's extra inputs (training_utils.py:136-137), for a network built with use_glob_features / use_loc_features."""
        return {k: batch[k] for k in ("glob_features", "loc_features") if k in batch}

    def losses(self, batch, fps_start=None, geometry=False):
        """Forward + losses (training_utils.py:140-146).  Returns the reference's 6 scalars.
        geometry: the batch's index tensors if the caller already holds them (None = compute them in the forward pass);
        by default whatever `prefetch` left for this batch."""
        P = batch["P"]
        if geometry is False:
            geometry = self._take_prefetched(P) if fps_start is None else None
        kw = {"geometry": geometry} if geometry is not None else {}
        kw.update(self._feature_kwargs(batch))
        X, T, W, _, _ = self.module(P, fps_start=fps_start, **kw)
        packed = getattr(self.module, "heads_packed", None)
        if self.fused_losses and packed is not None and len(self.classes) == 4 and T.shape[2] == 4:
            from .SPFN import fused_losses
            return fused_losses.fused_losses(P, packed, batch, self.mult, self.classes,
                                             handover=getattr(self.module, "handover", None))
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

    def eval_losses(self, batch, next_batch=None):
        """The validation pass of the reference's epoch loop (`network_mode='val'`, training_utils.py:104-105, 140-147 under
        the caller's `torch.no_grad()`): forward + the six losses in whatever mode the module is in, nothing back-propagated,
        no optimizer.  `next_batch`: its FPS / ball query / 3-NN run on the side stream beside this batch's forward pass."""
        st = self._graph
        if (self.use_graphs and st is not None and st.get("single") and self._gstream is not None and batch["P"].is_cuda
                and torch.cuda.current_stream(batch["P"].device) == self._gstream and self._prefetched is None
                and set(batch) >= {k for k in st["batch"] if k != "gt_axes"}
                and all(batch[k].shape == st["batch"][k].shape for k in st["batch"] if k != "gt_axes")
                and (next_batch is None or next_batch["P"].shape == st["P_next"].shape)):
            # the replayed form: the step's static inputs, geometry hand-over and side graph, with the validation twin of the
            # step's graph in place of the step (same FPS-seed draws as the eager form: one pair per announced batch)
            return self._graph_step(batch, next_batch, val=True)
        with torch.no_grad():
            geom = self._take_prefetched(batch["P"])
            if next_batch is not None:
                self.prefetch(next_batch)
            out = self.losses(batch, geometry=geom)
        return tuple(o.detach() for o in out)

    def _single_graph(self, K):
        """Can the step be ONE graph?  (SPFN: the assignment on the device, at most 32 instance columns.)"""
        from .SPFN import fused_losses as fl
        return not fl.HOST_ASSIGNMENT and K <= 32

    def _graph_losses(self, sb, st):
        """The loss section inside the captured step, after the network's forward pass on the static batch `sb`: heads
        post-processing + segmented sums -> assignment + fits -> matched losses (everything capturable, nothing on the host).
        Returns the tuple whose first entry is back-propagated.  Other objectives on the same network (PatchSelectionTrainer)
        override this and `losses`."""
        from .SPFN import fused_losses as fl
        Xn, W, nl, tl, S = fl.pre_match(self.module.heads_packed, sb, getattr(self.module, "handover", None))
        n_gt = fl.count_gt(sb["I_gt"])
        params, st["match"] = fl.fit_params_and_match(sb["P"], W, Xn, self.mult, S, n_gt)
        with fl.unit_loss_gradient():
            return fl.post_match(sb["P"], Xn, W, nl, tl, S, st["match"], sb, self.mult, self.classes, n_gt, params)

    # ---- hipGraph replay of the step -----------------------------------------------------------
    # At 16 clouds per GPU the step is ~150 launches and host-bound when launched one by one.  With the
    # assignment solved on the device (cpfn_hungarian_match) nothing in the step needs the host, so it is
    # captured ONCE:
    #   G  = network forward + heads post-processing + segmented sums + assignment + fits + matched losses
    #        + full backward + gradient packing (+ finite scan) + [all-reduce] + Adam            (trainer's stream)
    #   GS = FPS / ball query / 3-NN / inverse indices of the NEXT batch into the geometry buffers B (side stream)
    #   G0 = the geometry pass alone (only replayed when the next batch was not announced)
    # and per step: geomA <- geomB + new inputs (one multi-tensor copy), flag kernels, replay G || GS.
    # The geometry of a batch depends on its coordinates only, so computing it one step ahead hides the 0.6 ms FPS
    # latency chain (16 workgroups) behind the forward and backward passes.
    # CPFN_HOST_ASSIGNMENT=1 solves the assignment with SciPy on the host like the reference; the step is then
    # split at that round trip: G1 (forward .. cost matrices -> pinned host memory, event), G1b (the fits, which
    # do not depend on the assignment, run while the host works), G2 (the rest, with the geometry as a forked branch).
    @staticmethod
    def _flatten_geom(g):
        out = []
        for lvl in ("sa1", "sa2"):
            out += [g[lvl]["fps_idx"], g[lvl]["new_xyz"]]
            for nbr, rel in g[lvl]["scales"]:
                out += [nbr, rel]
            if "inv" in g[lvl]:
                out += list(g[lvl]["inv"])
        for lvl in ("sfp2", "sfp3"):
            out += [g[lvl]["nn_idx"], g[lvl]["nn_w"]]
            if "inv" in g[lvl]:
                out += list(g[lvl]["inv"])
        return out

    @staticmethod
    def _copy_all(dst, src, flags=None):
        """One launch for a whole set of device-to-device copies (cpfn_multi_copy); tensors that are not contiguous
        (or oddly aligned) go through torch.  flags (int32 device tensor): fp32 copies with the finite scan riding
        along; returns (flags, count) when every tensor went through the checked launch, else None."""
        from . import lib as _l
        _D = _CopyDesc
        # (the same tensors are copied step after step: the descriptor array of a set of (source, destination, strides) is
        #  built once — the checks and the ctypes construction were ~60 us of host time per step)
        key = tuple((d.data_ptr(), t.data_ptr(), d.numel(), d.stride(), t.stride(), d.dtype, t.dtype) for d, t in zip(dst, src))
        ent = _copy_desc_cache.get(key)
        if ent is not None and flags is None:
            arr, n_fast, nbytes = ent
            _l.add_bytes("cpfn_multi_copy", 2 * nbytes)
            with torch.cuda.device(dst[0].device):
                _l.check(_l.lib().cpfn_multi_copy(arr, n_fast, torch.cuda.current_stream().cuda_stream), "cpfn_multi_copy")
            return None
        fast, slow = [], 0
        for d, t in zip(dst, src):
            if (d.is_cuda and t.is_cuda and d.is_contiguous() and t.is_contiguous() and d.dtype == t.dtype
                    and d.numel() == t.numel() and ((d.data_ptr() | t.data_ptr()) % 16 == 0 or
                                                    ((d.data_ptr() | t.data_ptr()) % 4 == 0 and d.element_size() % 4 == 0))):
                fast.append(_D(t.data_ptr(), d.data_ptr(), d.numel() * d.element_size()))
            else:
                d.copy_(t, non_blocking=True)
                slow += 1
        if not fast:
            return None
        arr = (_D * len(fast))(*fast)
        h = _l.lib()
        if slow == 0 and flags is None:
            if len(_copy_desc_cache) > 64:
                _copy_desc_cache.clear()
            _copy_desc_cache[key] = (arr, len(fast), sum(f.bytes for f in fast))
        _l.add_bytes("cpfn_multi_copy", 2 * sum(f.bytes for f in fast))
        with torch.cuda.device(dst[0].device):
            stream = torch.cuda.current_stream().cuda_stream
            if flags is not None and slow == 0 and all(t.dtype == torch.float32 for t in src):
                count = h.cpfn_multi_copy_blocks(arr, len(fast))
                if 0 < count <= flags.numel():
                    _l.check(h.cpfn_multi_copy_checked(arr, len(fast), flags.data_ptr(), flags.numel(), stream),
                             "cpfn_multi_copy_checked")
                    return flags, count
            _l.check(h.cpfn_multi_copy(arr, len(fast), stream), "cpfn_multi_copy")
        return None

    @staticmethod
    def _like_geom(g, tensors):
        it = iter(tensors)
        out = {}
        for lvl in ("sa1", "sa2"):
            d = {"fps_idx": next(it), "new_xyz": next(it), "scales": []}
            for _ in g[lvl]["scales"]:
                d["scales"].append((next(it), next(it)))
            if "inv" in g[lvl]:
                d["inv"] = (next(it), next(it))
            out[lvl] = d
        for lvl in ("sfp2", "sfp3"):
            d = {"nn_idx": next(it), "nn_w": next(it)}
            if "inv" in g[lvl]:
                d["inv"] = (next(it), next(it))
            out[lvl] = d
        return out

    @staticmethod
    def _probe_exchange_capture(dev):
        """Can this stack capture the gradient exchange?  Only RCCL's collectives can be stream-captured (gloo
        synchronises the stream: attempting it invalidates the capture and leaves the stream unusable), and even there
        it is tried on a THROW-AWAY stream first: a tiny averaging all-reduce is captured, replayed and its result
        checked, and the ranks agree on the verdict — so a stack that cannot do it never touches the step's own capture
        stream, and every rank takes the same layout."""
        if dist.get_backend() != "nccl":
            return False
        ok = 1.0
        try:
            _quiesce_collectives(dev)
            rank = dist.get_rank()
            probe = torch.full((1024,), float(rank), dtype=torch.float32, device=dev)
            ref = probe.clone()
            dist.all_reduce(ref, op=dist.ReduceOp.AVG)                    # eager: the expected result (and the communicator
            #                                                               is set up outside any capture)
            _quiesce_collectives(dev)
            stream = torch.cuda.Stream(device=dev)
            stream.wait_stream(torch.cuda.current_stream(dev))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(stream), capture_guard():
                with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                    dist.all_reduce(probe, op=dist.ReduceOp.AVG)
                probe.fill_(float(rank))
                g.replay()
            stream.synchronize()
            if not torch.equal(probe, ref):
                ok = 0.0
        except Exception:
            ok = 0.0
        flag = torch.tensor([ok], device=dev)
        try:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        except Exception:
            return False
        return float(flag) >= 1.0

    def _capture(self, batch, exchange_in_graph=True):
        with capture_guard():
            return self._capture_impl(batch, exchange_in_graph)

    def _capture_impl(self, batch, exchange_in_graph=True):
        """exchange_in_graph (data parallel only): capture the RCCL all-reduce and the optimizer inside the step's
        graph; False = the graph ends after the gradient packing and the exchange + optimizer follow as eager
        launches (what step() falls back to if the collective cannot be captured on this stack)."""
        from .SPFN import fused_losses as fl
        # The parameters' AccumulateGrad nodes were created by the eager warm-up steps on the default stream and are
        # reused while capturing on the capture stream: autograd warns about that on every backward pass although the
        # capture is self-contained (every gradient is produced and consumed on the capture stream).
        if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        dev = batch["P"].device
        B, N, _ = batch["P"].shape
        K = batch["T_gt"].shape[1] if "T_gt" in batch else 0
        st = {"batch": {k: v.clone() for k, v in batch.items()},
              "P_next": batch["P"].clone(),
              "start_dev": torch.zeros(2, B, dtype=torch.int32, device=dev),
              "start_host": [torch.zeros(2, B, dtype=torch.int32).pin_memory() for _ in range(2)],
              "start_done": [torch.cuda.Event(), torch.cuda.Event()], "start_turn": 0,
              "match": torch.zeros(B, K, dtype=torch.long, device=dev),
              "skipped": self._skip_counter(dev)}
        st["start1"], st["start2"] = st["start_dev"][0], st["start_dev"][1]
        st["unit"] = torch.ones((), dtype=torch.float32, device=dev)      # d total / d total, allocated outside the graph
        sb = st["batch"]
        # the three GT axis tensors live stacked ([3,B,K,3], what the residue kernel reads): the per-key static
        # buffers are views of it, so staging a batch fills the stacked tensor without a torch.stack per step
        axes = ("plane_n_gt", "cylinder_axis_gt", "cone_axis_gt")
        if all(k in sb for k in axes) and len({(tuple(sb[k].shape), sb[k].dtype) for k in axes}) == 1 and \
                sb[axes[0]].dtype == torch.float32:
            stacked = torch.stack([sb[k] for k in axes], 0).contiguous()
            for i, k in enumerate(axes):
                sb[k] = stacked[i]
            sb["gt_axes"] = stacked
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if world > 1 and batch["P"].is_cuda:
            if self._comm_stamps is None:                  # (allocated outside any capture)
                self._comm_stamps = torch.zeros(2, dtype=torch.int64, device=dev)
            _quiesce_collectives(dev)
        starts = (st["start1"], st["start2"])
        # static geometry buffers (shapes from one eager pass)
        g_example = self.module.compute_geometry(sb["P"], starts)
        geomB = [t.clone() for t in self._flatten_geom(g_example)]
        geomA = [t.clone() for t in geomB]
        st["geomA"] = self._like_geom(g_example, geomA)
        st["geomA_flat"] = geomA
        st["geomB"] = geomB
        if self._gside is None:
            self._gside = torch.cuda.Stream(device=dev)

        def geometry_into_B(P, beside=True):
            # beside: the pass runs on the side stream next to a step -> the narrower kernels (csrc/neighbors.hip)
            import contextlib
            from . import ops as _ops
            with (_ops.background_geometry() if beside else contextlib.nullcontext()):
                fresh = self._flatten_geom(self.module.compute_geometry(P, starts))
            self._copy_all(geomB, fresh)

        # capture on the same side stream the eager warm-up steps ran on, so that the parameters'
        # AccumulateGrad nodes do not belong to the default stream (which cannot take part in a capture)
        # capture_error_mode="thread_local": other threads (e.g. the RCCL watchdog of a data-parallel job)
        # may touch the HIP API while this thread captures
        g0 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g0, stream=self._gstream, capture_error_mode="thread_local"):
            geometry_into_B(sb["P"], beside=False)
        st["world"], st["g0"] = world, g0
        st["single"] = self._single_graph(K)
        if st["single"]:
            # Device-side assignment (cpfn_hungarian_match): the step has no host round trip and is ONE linear graph; the
            # geometry of the NEXT batch is a second linear graph, replayed on the side stream while the step's graph runs
            # (own memory pool: the two replay concurrently).
            g = torch.cuda.CUDAGraph()
            stamps = st["stamps"] = (torch.zeros(5, dtype=torch.int64, device=dev)
                                     if os.environ.get("CPFN_STEP_STAMPS") == "1" else None)

            def stamp(i):       # debugging: device wall-clock readings inside the replayed step
                if stamps is not None:
                    from . import lib as _l
                    _l.check(_l.lib().cpfn_stamp(stamps[i:].data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "cpfn_stamp")

            from . import ops as _ops
            gs = torch.cuda.CUDAGraph()
            self._gside.wait_stream(self._gstream)
            # the tensors the captured pass allocates have fixed addresses (the graph's private pool): they ARE the B set —
            # no 12 MB copy at the end of every side replay.  G0 (the serial pass of an un-announced batch) is
            # re-captured so that it fills them.
            with torch.cuda.graph(gs, stream=self._gside, capture_error_mode="thread_local"):
                with _ops.background_geometry():
                    fresh_side = self._flatten_geom(self.module.compute_geometry(st["P_next"], starts))
                stamp(1)
            self._gstream.wait_stream(self._gside)
            assert len(fresh_side) == len(geomB) and all(a.shape == b.shape and a.dtype == b.dtype and a.is_contiguous()
                                                         for a, b in zip(fresh_side, geomB))
            geomB[:] = fresh_side                       # (st["geomB"] and geometry_into_B see the same list)
            g0b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g0b, pool=g0.pool(), stream=self._gstream, capture_error_mode="thread_local"):
                geometry_into_B(sb["P"], beside=False)
            st["g0_first"], st["g0"] = g0, g0b          # (the first one owns the pool the step's graph shares)
            st["gs"] = gs
            st["side_pending"] = False
            # The two streams' flags: [0] = step graphs whose geometry hand-off is done ("geomB consumed, next inputs
            # written"), [1] = side graphs finished.  A waiter that gives up (_flag_timeout_ticks) raises flag_err (pinned,
            # polled by the host at every step) and flag_fault (device): the latter is the optimizer's skip flag inside the
            # graph, so a step computed from a broken hand-over can never update the weights.
            st["flags"] = torch.zeros(4, dtype=torch.int32, device=dev)
            st["flag_err"] = torch.zeros(4, dtype=torch.int32).pin_memory()
            st["flag_fault"] = self.fault_word(dev)            # (the rank's one fault word: raise_fault() sets the same tensor)
            st["flag_timeout"] = _flag_timeout_ticks(world)
            st["fault_host"] = torch.zeros(1, dtype=torch.float32).pin_memory()      # the reduced fault slot of the last step (world > 1)
            st["n_main"], st["n_side"] = 0, 0
            with torch.cuda.graph(g, pool=g0.pool(), stream=self._gstream, capture_error_mode="thread_local"):
                if stamps is not None:
                    stamps[4:5].copy_(stamps[3:4])                  # when the PREVIOUS replay ended
                stamp(0)
                self.bucket.zero()
                self.module(sb["P"], geometry=st["geomA"], **self._feature_kwargs(sb))
                out = self._graph_losses(sb, st)
                # (the launches that produce parameter gradients write them into the flat bucket, finite check included:
                #  FlatGradBucket.sink — no packing copy on one GPU, a one-word copy of the fault slot under data parallelism)
                from .optim import FlatAdam as _FA
                with self.bucket.sink(check=world == 1, combine_ok=world == 1 and isinstance(self.optimizer, _FA)) as gsink:
                    out[0].backward(st["unit"])          # (no ones_like fill inside the graph)
                nf = self.bucket.collect(check=world == 1, fault=st["flag_fault"] if world > 1 else None, sink=gsink)
                if world == 1:
                    self._checked_optimizer_step(st["skipped"], nf, fault=st["flag_fault"],
                                                 combine=None if gsink is None else gsink.combine)
                elif exchange_in_graph:
                    # data parallel: the gradient exchange (RCCL, over the 5.6 MB flat bucket) and the optimizer are nodes
                    # of the SAME graph: still one replay per step.  Two one-lane stamp kernels bracket it (device wall
                    # clock): `comm_us()` of the last replayed step, so that a scaling run can tell exchange time from
                    # everything else (2 x ~3 us on a 1.9 ms chain).
                    self._exchange_with_stamps(dev)
                    # (the fault word every rank packed into the bucket's fault slot came back averaged: non-zero on EVERY rank
                    #  when any rank's flag wait timed out — all replicas skip together)
                    self._checked_optimizer_step(st["skipped"], fault=self.bucket.fault_slot.reshape(()))
                    # ... and lands in a pinned host word every rank polls at the head of its next steps (one 4-byte copy node)
                    st["fault_host"].copy_(self.bucket.fault_slot, non_blocking=True)
                st["out"] = tuple(o.detach() for o in out)
                stamp(2)
                stamp(3)
            st["g"] = g
            st["exchange_in_graph"] = world > 1 and exchange_in_graph
            st["geom_ready_for"] = None
            return st
        # ---- CPFN_HOST_ASSIGNMENT=1: SciPy on the host like the reference; graphs split at that round trip
        st["cost_host"] = torch.empty(B, K * K + 1, dtype=torch.float32).pin_memory()   # (not inside a capture)
        st["match_host"] = torch.zeros(B, K, dtype=torch.long).pin_memory()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, pool=g0.pool(), stream=self._gstream, capture_error_mode="thread_local"):
            self._copy_all(geomA, geomB)
            self.bucket.zero()
            self.module(sb["P"], geometry=st["geomA"], **self._feature_kwargs(sb))
            st["pre"] = fl.pre_match(self.module.heads_packed, sb, getattr(self.module, "handover", None))
            st["n_gt"] = fl.count_gt(sb["I_gt"])
            st["cost_pack"] = fl.hungarian_cost_pack(st["pre"][4].detach(), sb["I_gt"], st["n_gt"])   # device part, in-graph
            st["cost_host"].copy_(st["cost_pack"], non_blocking=True)          # D2H node at the end of G1
        # G1b: the fits do not depend on the assignment: they run while the host solves it
        g1b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1b, pool=g0.pool(), stream=self._gstream, capture_error_mode="thread_local"):
            Xn, W, nl, tl, S = st["pre"]
            st["params"] = fl.fit_params(sb["P"], W, Xn, self.mult)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, pool=g0.pool(), stream=self._gstream, capture_error_mode="thread_local"):
            self._gside.wait_stream(self._gstream)                  # fork: next batch's geometry
            with torch.cuda.stream(self._gside):
                geometry_into_B(st["P_next"])
            Xn, W, nl, tl, S = st["pre"]
            with fl.unit_loss_gradient():
                out = fl.post_match(sb["P"], Xn, W, nl, tl, S, st["match"], sb, self.mult, self.classes, st["n_gt"],
                                    st["params"])
            out[0].backward(st["unit"])
            nf = self.bucket.collect(check=world == 1)
            if world == 1:
                self._checked_optimizer_step(st["skipped"], nf)
            st["out"] = tuple(o.detach() for o in out)
            self._gstream.wait_stream(self._gside)                  # join
        st["g1"], st["g1b"], st["g2"] = g1, g1b, g2
        st["cost_ready"] = torch.cuda.Event()
        st["geom_ready_for"] = None
        return st

    def _draw_starts(self, st, B, N, as_payload=False):
        # the same two CPU-generator draws the eager path makes (geometry_utils.py:92), in the same order.
        # They go through PINNED staging buffers that live as long as the graphs, so the asynchronous copy never
        # reads from a temporary pageable tensor the host may already have reused.  Two buffers take turns and
        # each is rewritten only after the copy that last read it has executed (its event): without a host
        # sync in the step the host may run ahead of the device.
        k = st["start_turn"]
        st["start_turn"] = 1 - k
        host, done = st["start_host"][k], st["start_done"][k]
        done.synchronize()
        host[0].copy_(torch.randint(0, N, (B,), dtype=torch.long))
        host[1].copy_(torch.randint(0, self.module.sa1.num_points, (B,), dtype=torch.long))
        if as_payload:          # the caller sends them with its flag kernel (cpfn_flag_set_payload): no host-to-device copy
            return host
        st["start_dev"].copy_(host, non_blocking=True)
        done.record()
        return None

    def _val_graph(self, st):
        """The validation twin of the step's graph: forward + loss section under no_grad in the module's CURRENT mode (evaluation
        mode for `spfn_train_val_epoch(..., 'val')`; training mode for the PatchSelection loop's quirk), on the same static
        inputs and geometry set A, sharing the step graph's memory pool.  Captured on first use, one per mode."""
        key = bool(self.module.training)
        ent = st.setdefault("val", {}).get(key)
        if ent is None:
            dev = st["batch"]["P"].device
            if st["world"] > 1:
                _quiesce_collectives(dev)
            gv = torch.cuda.CUDAGraph()
            sb = st["batch"]
            with torch.no_grad(), capture_guard():
                with torch.cuda.graph(gv, pool=st.get("g0_first", st["g0"]).pool(), stream=self._gstream, capture_error_mode="thread_local"):
                    self.module(sb["P"], geometry=st["geomA"], **self._feature_kwargs(sb))
                    out = self._graph_losses(sb, {})
            ent = st["val"][key] = (gv, tuple(o.detach() for o in out))
        return ent

    def _graph_step(self, batch, next_batch=None, val=False):
        from .SPFN import fused_losses as fl
        st = self._graph
        single = st["single"]
        main_graph, main_out = (st.get("g"), None) if not val else self._val_graph(st)
        if single and int(st["flag_err"][0]) != 0:
            if st["world"] > 1:
                # the peers would sit in the next step's in-graph collective until the RCCL time-out: take the group down
                abort = getattr(dist.distributed_c10d, "_abort_process_group", None)
                try:
                    if abort is not None:
                        abort()
                except Exception:
                    pass
            raise RuntimeError("cpfn_amd: a cross-stream flag wait of the replayed step timed out after %.0f s (the other "
                               "stream's graph was never launched, or it stalled for longer than CPFN_FLAG_TIMEOUT_S); the "
                               "optimizer has skipped every step since, the losses of the last steps are invalid"
                               % (st["flag_timeout"] / 100e6))
        if single and st["world"] > 1 and float(st["fault_host"][0]) != 0.0:
            # the REDUCED fault slot of an earlier replayed step (copied to this pinned word by the step itself, right behind the
            # exchange): some rank raised its fault word, every replica skipped that step — and every rank stops here, the peers
            # too, instead of replaying into a collective whose partner is gone
            self._raise_reduced_fault(abort_as_peer=True)
        if (self.global_step & 63) == 0:
            # the sampling kernels' fault count (a pinned host word: no synchronisation): the several-workgroups time-out and the
            # tripwire — a sample whose own min-distance was not zeroed, i.e. a lost update beside this very step (VERDICT r4 #1a)
            from . import ops as _ops
            try:
                _ops.check_fps_faults("training step %d" % self.global_step)
            except RuntimeError as e:
                if not (single and st["world"] > 1):
                    raise
                # data parallel: leaving the loop here would strand the peers in this step's collective — the fault rides on the
                # rank's fault word instead: the step runs, EVERY rank skips its update, then this rank (and its peers) raise
                self.raise_fault("sampling fault: %s" % e)
        if single and st["side_pending"]:
            # the side graph of the previous step (reads P_next / the FPS seeds, writes geomB) must be done before
            # this step overwrites its inputs and reads its result
            self._flag_wait(st, 1, st["n_side"], torch.cuda.current_stream(batch["P"].device))
            st["side_pending"] = False
        # inputs into the static buffers: ONE multi-tensor copy (the batch tensors that are not already the
        # static ones + the next batch's coordinates for the geometry graph)
        dst, src = [], []
        for k, v in batch.items():
            if v.data_ptr() != st["batch"][k].data_ptr():
                dst.append(st["batch"][k])
                src.append(v)
        if next_batch is not None and next_batch["P"].data_ptr() != st["P_next"].data_ptr():
            dst.append(st["P_next"])
            src.append(next_batch["P"])
        B, N, _ = batch["P"].shape
        # (geomA <- geomB is an eager launch in front of the step's graph; when the geometry was announced one step
        #  ahead — the normal case — it rides on the input copy: one launch per step, not two)
        merged = single and self._prefetched is None and st["geom_ready_for"] == self._batch_key(batch["P"])
        if merged:
            dst, src = dst + st["geomA_flat"], src + st["geomB"]
        if dst:
            self._copy_all(dst, src)
        if self._prefetched is not None:                       # geometry prefetched by an eager (warm-up) step
            geom = self._take_prefetched(batch["P"])
            if geom is not None:
                self._copy_all(st["geomB"], self._flatten_geom(geom))
                st["geom_ready_for"] = self._batch_key(batch["P"])
        if st["geom_ready_for"] != self._batch_key(batch["P"]):      # not announced one step ahead: do it now
            self._draw_starts(st, B, N)
            st["g0"].replay()
        # inputs of the geometry graph: the NEXT batch's FPS seeds (its coordinates were copied above)
        # (without an announced next batch no geometry graph runs and no FPS seeds are drawn, so the CPU generator is
        #  consumed exactly as in eager mode)
        announce = next_batch is not None
        if single:
            st["geom_ready_for"] = self._batch_key(next_batch["P"]) if announce else None
            # geomA <- geomB, then the side stream may overwrite geomB with the next batch's geometry while this
            # stream replays the step
            if not merged:
                self._copy_all(st["geomA_flat"], st["geomB"])
            if announce:
                from . import lib as _l
                h = _l.lib()
                cur = torch.cuda.current_stream(batch["P"].device)
                flags = st["flags"]
                # the 2 x B seeds travel in the arguments of the flag kernel that releases the geometry graph (they were a
                # 128-byte host-to-device copy: a blit kernel of its own between two replays)
                as_payload = 2 * B <= 64
                payload = self._draw_starts(st, B, N, as_payload=as_payload)
                with torch.cuda.device(cur.device):
                    if as_payload:
                        _l.check(h.cpfn_flag_set_payload(flags[0:].data_ptr(), st["n_main"] + 1, st["start_dev"].data_ptr(),
                                                         payload.data_ptr(), payload.numel(), cur.cuda_stream), "cpfn_flag_set_payload")
                    else:
                        _l.check(h.cpfn_flag_set(flags[0:].data_ptr(), st["n_main"] + 1, cur.cuda_stream), "cpfn_flag_set")
                    # the step's own graph is submitted BEFORE the side stream's launches: from an idle GPU (the first step after
                    # a synchronisation) the device then waits for one graph launch less (~0.1 ms of host time).
                    # CPFN_SIDE_GRAPH_FIRST=1 (profiling only): under rocprofv3 a graph launch costs the host > 1 ms, the side
                    # graph submitted second then trails the step by most of its length, lands on the last backward launches
                    # (csr_build holds 80 KB of LDS per CU) and on the NEXT step's waiter — per-kernel times of a trace taken that
                    # way describe the profiler, not the step (profiles/README.md, round 4).
                    if not _SIDE_GRAPH_FIRST:
                        main_graph.replay()                    # the whole step: no host synchronisation
                    self._flag_wait(st, 0, st["n_main"] + 1, self._gside)
                    with torch.cuda.stream(self._gside):
                        st["gs"].replay()
                    _l.check(h.cpfn_flag_set(flags[1:].data_ptr(), st["n_side"] + 1, self._gside.cuda_stream), "cpfn_flag_set")
                    if _SIDE_GRAPH_FIRST:
                        main_graph.replay()
                st["n_side"] += 1
                st["side_pending"] = True
            else:
                main_graph.replay()                            # the whole step: no host synchronisation
            st["n_main"] += 1
            if val:                                            # (no backward pass, no exchange, no optimizer, no step count)
                return main_out
            if st["world"] > 1 and not st["exchange_in_graph"]:
                self._exchange_with_stamps(batch["P"].device)
                self._checked_optimizer_step(st["skipped"], fault=self.bucket.fault_slot.reshape(()))
                st["fault_host"].copy_(self.bucket.fault_slot, non_blocking=True)
            self.global_step += 1
            if st["world"] > 1 and getattr(self, "_fault_raised", None) is not None:
                # raise_fault(): the replay just issued carries the word through the collective (every rank skips); now stop.
                # (The peers see the same reduced slot in their pinned word at the head of one of their next steps.)
                torch.cuda.current_stream(batch["P"].device).synchronize()
                self._after_exchange_fault_check()
            return st["out"]
        st["g1"].replay()
        st["cost_ready"].record()                              # the cost matrices are in pinned host memory after this
        st["g1b"].replay()                                     # fits: GPU work for the time of the host round trip
        if announce:
            self._draw_starts(st, B, N)
        st["geom_ready_for"] = self._batch_key(next_batch["P"]) if announce else None
        st["cost_ready"].synchronize()                         # the step's one host sync (waits for G1, not G1b)
        fl.hungarian_host(st["cost_host"], st["match"].shape[1], out=st["match_host"])
        st["match"].copy_(st["match_host"], non_blocking=True)
        st["g2"].replay()
        if st["world"] > 1:
            self.bucket.all_reduce_mean()
            self._checked_optimizer_step(st["skipped"])
        self.global_step += 1
        return st["out"]

    def _exchange_with_stamps(self, dev):
        """The step's gradient exchange between two device wall-clock stamps (cpfn_stamp: one-lane kernels, capturable)."""
        from . import lib as _l
        if self._comm_stamps is None:
            self._comm_stamps = torch.zeros(2, dtype=torch.int64, device=dev)
        h = _l.lib()
        with torch.cuda.device(dev):
            _l.check(h.cpfn_stamp(self._comm_stamps[0:].data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "cpfn_stamp")
            self.bucket.all_reduce_mean()
            _l.check(h.cpfn_stamp(self._comm_stamps[1:].data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "cpfn_stamp")

    def comm_us(self):
        """Microseconds between the two stamps around the LAST step's gradient exchange (from the end of the gradient
        packing to the end of the collective on this rank: wire time plus the wait for the slowest peer); None on one GPU.
        Synchronises with the device."""
        if self._comm_stamps is None:
            return None
        a, b = (int(v) for v in self._comm_stamps.cpu())
        from . import lib as _l
        khz = float(_l.lib().cpfn_wall_clock_khz(self._comm_stamps.device.index or 0))
        return (b - a) / khz * 1e3 if khz > 0 and b >= a else None

    @staticmethod
    def _flag_wait(st, which, value, stream):
        from . import lib as _l
        with torch.cuda.device(stream.device):
            _l.check(_l.lib().cpfn_flag_wait(st["flags"][which:].data_ptr(), int(value) & 0xffffffff, st["flag_timeout"],
                                             st["flag_err"].data_ptr(), st["flag_fault"].data_ptr(), stream.cuda_stream),
                     "cpfn_flag_wait")

    def stream(self, device):
        """The stream the replayed steps run on.  A training loop that runs under it (`with torch.cuda.stream(...)`)
        saves the two cross-stream dependencies per step that calling step() from another stream costs (~40 us of idle
        GPU between consecutive replays, measured with CPFN_STEP_STAMPS)."""
        if self._gstream is None:
            self._gstream = torch.cuda.Stream(device=device)
        return self._gstream

    def _all_training(self):
        # (walks the module tree every step — ~60 attribute reads, 10 us — instead of caching the list: a submodule added or
        #  swapped after the first step must be seen)
        for m in self.module.modules():
            if not m.training:
                return False
        return True

    def step(self, batch, fps_start=None, next_batch=None, force_eager=False):
        """One optimisation step; returns the 6 loss tensors (still on the device — the
        reference's six `.item()` syncs per step, training_utils.py:169-174, are left to the caller).
        `next_batch`: if given, its geometry is prefetched on a side stream during this step's backward."""
        # (nn.Module.train() walks all ~60 submodules through __setattr__: ~0.1 ms of host time per call, twice per step with
        #  the re-entry on the trainer's stream — exposed whenever the GPU is idle, e.g. on the first step after a sync.  The
        #  walk only happens when some module is NOT in training mode.)
        if not self._all_training():
            self.module.train()
        self._schedules()
        if self.use_graphs and "T_gt" in batch and batch["T_gt"].shape[1] > 32:
            # the fused loss kernels (and with them the captured step) take at most 32 instance columns; wider label
            # sets run eagerly on the op-by-op losses (HIP fitters, stock reductions)
            if self.require_graphs:
                raise RuntimeError("graph replay supports at most 32 instance columns (got %d)" % batch["T_gt"].shape[1])
            force_eager = True
        if self.use_graphs and not force_eager and fps_start is None and batch["P"].is_cuda and not self._in_gstream:
            if self._gstream is None:
                self._gstream = torch.cuda.Stream(device=batch["P"].device)
            cur = torch.cuda.current_stream(batch["P"].device)
            same = cur == self._gstream          # the caller already runs on the trainer's stream (see stream()): no hops
            if not same:
                self._gstream.wait_stream(cur)
            self._in_gstream = True
            try:
                with torch.cuda.stream(self._gstream):
                    out = self.step(batch, None, next_batch)
            finally:
                self._in_gstream = False
            if not same:
                cur.wait_stream(self._gstream)
            return out
        if self.use_graphs and not force_eager and fps_start is None and batch["P"].is_cuda:
            if self._graph is None and self._graph_warm >= 2:
                try:
                    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
                    if world > 1 and self._exchange_in_graph and not self._probe_exchange_capture(batch["P"].device):
                        self._exchange_in_graph = False
                    if world == 1 or not self._exchange_in_graph:
                        self._graph = self._capture(batch, exchange_in_graph=self._exchange_in_graph)
                    else:
                        # Data parallel: try the step WITH the collective inside the graph; all ranks must end up with
                        # the same layout (a rank replaying the all-reduce while another issues it eagerly is still a
                        # matching collective, but a rank whose capture failed must not leave the others waiting), so
                        # the outcome is agreed on with one tiny all-reduce before anybody replays.
                        graph, err = None, None
                        try:
                            graph = self._capture(batch, exchange_in_graph=True)
                        except Exception as e:
                            err = e
                            torch.cuda.synchronize()
                        ok = torch.tensor([0.0 if graph is None else 1.0], device=batch["P"].device)
                        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                        if float(ok) < 1.0:
                            # the collective could not be captured on (some rank of) this stack: keep the replayed step,
                            # with the exchange + optimizer as eager launches after it (still no host synchronisation)
                            import warnings
                            warnings.warn("RCCL all-reduce not capturable on every rank (%s); exchange stays outside the graph"
                                          % (("%s: %s" % (type(err).__name__, err)) if err is not None else "another rank failed"))
                            self._exchange_in_graph = False
                            graph = None
                            torch.cuda.synchronize()
                            graph = self._capture(batch, exchange_in_graph=False)
                        self._graph = graph
                except Exception as e:          # capture is an optimisation: fall back to eager launches
                    if self.require_graphs:     # ... unless the caller asked for the replayed step (bench.py does)
                        raise
                    import warnings
                    warnings.warn("hipGraph capture failed (%s: %s); running eagerly" % (type(e).__name__, e))
                    self.use_graphs = False
                    torch.cuda.synchronize()
            if self._graph is not None:
                return self._graph_step(batch, next_batch)
            self._graph_warm += 1
        self.bucket.zero()
        out = self.losses(batch, fps_start)
        if next_batch is not None:
            self.prefetch(next_batch)
        out[0].backward()
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.bucket.collect(fault=self.fault_word(self.bucket.flat.device) if world > 1 else None)
        self.bucket.all_reduce_mean()
        reduced_fault = self.bucket.fault_slot.reshape(()) if world > 1 else None
        if self.use_graphs:                            # capturable optimizer: skip decided (and counted) on the device
            self._checked_optimizer_step(self._skip_counter(self.bucket.flat.device), fault=reduced_fault)
        elif bool(self.bucket.finite()) and (reduced_fault is None or float(reduced_fault) == 0.0):   # host sync (reference: 148)
            self.optimizer.step()
        else:
            self._host_skipped += 1
        self.global_step += 1
        if world > 1 and not self.use_graphs:
            self._after_exchange_fault_check()
        return tuple(o.detach() for o in out)     # do not keep the autograd graph alive across steps




class PatchSelectionTrainer(SPFNTrainer):
    """The same step machinery for the OTHER objective trained on this network: PatchSelection
    (`PointNet2(output_sizes=[2])`, training_PatchSelection.py:55) — the two-class heat map with a cross-entropy against
    per-point labels (`patch_selection_train_val_epoch`, Utils/training_utils.py:62-75).  batch = {"P": [B,N,3] fp32,
    "labels": [B,N] int64}; `step` / `eval_losses` return a 1-tuple (the loss).  The loss itself is two stock kernels
    (log-soft-max + NLL over [B*N, 2]) inside the replayed graph; forward, backward, geometry look-ahead, gradient bucket and
    Adam are the SPFN trainer's.  Unlike the reference's loop (which steps unconditionally, :72-74) a step with non-finite
    gradients is skipped on the device."""

    def _single_graph(self, K):
        return True

    def _cross_entropy(self, batch):
        heat = self._heat
        B, N, _ = heat.shape
        packed = getattr(self.module, "heads_packed", None)
        if heat.is_cuda and packed is not None and packed.shape[2] == 2:
            # the fused pass (cpfn_ce2): loss, its gradient and the heads' padded gradient rows / column sums in one launch,
            # instead of ~8 stock kernels around a [B*N, 2] tensor (torch's mean reduction alone is one workgroup)
            from .SPFN import fused_losses as fl
            with fl.unit_loss_gradient():
                return fl.HeatCrossEntropy.apply(packed, batch["labels"], getattr(self.module, "handover", None))
        return torch.nn.functional.cross_entropy(heat.contiguous().view(B * N, 2), batch["labels"].view(B * N))    # (:66-68)

    def losses(self, batch, fps_start=None, geometry=False):
        P = batch["P"]
        if geometry is False:
            geometry = self._take_prefetched(P) if fps_start is None else None
        kw = {"geometry": geometry} if geometry is not None else {}
        self._heat = self.module(P, fps_start=fps_start, **kw)[0]
        return (self._cross_entropy(batch),)

    def _graph_losses(self, sb, st):
        self._heat = self.module.heads_packed if self.module.heads_packed is not None else None
        if self._heat is None or self._heat.shape[2] != 2:
            raise RuntimeError("PatchSelectionTrainer: a PointNet2(output_sizes=[2]) in the bf16 compute mode is expected")
        return (self._cross_entropy(sb),)


def __getattr__(name):
    # the reference's epoch loop on this trainer (same signature as Utils/training_utils.py:84-176) lives in cpfn_amd/epoch.py,
    # which imports this module: resolved on first use
    if name in ("spfn_train_val_epoch", "patch_selection_train_val_epoch", "EpochRunner"):
        from . import epoch
        return getattr(epoch, name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
