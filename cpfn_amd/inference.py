"""Evaluation forward of a PointNet2 (GlobalSPFN / LocalSPFN / PatchSelection) as ONE replayed hipGraph.

The reference evaluates with batch 1 on full-resolution clouds (evaluation_globalSPFN.py:62-64) and with 32 patches per
cloud (evaluation_localSPFN.py:95); at those sizes an eager forward is ~100 kernel launches whose host-side launch cost
(40-60 us each) exceeds the kernels' run time: 6.4 ms for one 131072-point cloud of which ~2 ms are kernels.
`GraphedForward` captures the forward pass once per input shape and replays it; FPS start indices are still drawn from
the CPU generator per call like the reference's CPU route (modules/geometry_utils.py:92) and staged through pinned
memory, so `torch.manual_seed(s)` selects the same points with and without the graph.
"""
import torch


class GraphedForward:
    """model: a cpfn_amd PointNet2 in eval mode on a HIP device.  `__call__(x [B,N,C], glob_features=None,
    loc_features=None, fps_start=None)` -> the model's output list (tensors owned by the graph: copy what must
    outlive the next call — or construct with clone_outputs=True).

    max_shapes: graphs kept (least recently used out first).  A captured graph reads the parameters and BatchNorm buffers
    where they lay at capture time; the storage addresses are part of a graph's key, so a model whose tensors moved
    (`.to()`, an optimizer that re-points them) is captured again instead of replaying on stale memory.

    `PointNet2.forward` uses one of these by itself for evaluation-mode forwards under `torch.no_grad()` (max_shapes=4,
    clone_outputs=True; `model.auto_graph = False` opts out), so that the reference's evaluation scripts
    (evaluation_globalSPFN.py:85, evaluation_localSPFN.py:95) get the replayed forward unedited."""

    def __init__(self, model, max_shapes=None, clone_outputs=False, weak=False):
        # weak: the instance the model keeps for itself (PointNet2.forward's auto replay) must not keep the model alive
        import weakref
        self._model_ref = weakref.ref(model) if weak else (lambda m=model: m)
        self._graphs = {}
        self.max_shapes = max_shapes
        self.clone_outputs = clone_outputs

    @property
    def model(self):
        return self._model_ref()

    def _storage_key(self):
        m = self.model
        return hash(tuple(t.data_ptr() for t in m.parameters()) + tuple(t.data_ptr() for t in m.buffers()))

    def _capture(self, x, glob, loc):
        m = self.model
        dev = x.device
        B, N, _ = x.shape
        st = {"x": x.clone(), "glob": None if glob is None else glob.clone(), "loc": None if loc is None else loc.clone(),
              "start_dev": torch.zeros(2, B, dtype=torch.int32, device=dev),
              "start_host": [torch.zeros(2, B, dtype=torch.int32).pin_memory() for _ in range(2)],
              "done": [torch.cuda.Event(), torch.cuda.Event()], "turn": 0}
        starts = (st["start_dev"][0], st["start_dev"][1])
        stream = torch.cuda.Stream(device=dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        from .training import capture_guard
        with torch.cuda.stream(stream), torch.no_grad():
            m(st["x"], glob_features=st["glob"], loc_features=st["loc"], fps_start=starts)     # warm-up: lazily created state
            g = torch.cuda.CUDAGraph()
            with capture_guard(), torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                st["out"] = m(st["x"], glob_features=st["glob"], loc_features=st["loc"], fps_start=starts)
        torch.cuda.current_stream(dev).wait_stream(stream)
        st["g"], st["stream"] = g, stream
        st["aux"] = {k: getattr(m, k) for k in ("aux_sa1", "aux_sa2", "aux_sfp3", "heads_packed") if hasattr(m, k)}
        return st

    @torch.no_grad()
    def __call__(self, x, glob_features=None, loc_features=None, fps_start=None):
        m = self.model
        if m.training:
            raise RuntimeError("GraphedForward replays an evaluation-mode forward: call model.eval() first")
        if not x.is_cuda:
            raise RuntimeError("GraphedForward: CPU not supported")
        from . import cuda_ops as _co, ops as _ops
        _ops.check_fps_faults("this evaluation forward")       # (a pinned host word: no synchronisation)
        key = (tuple(x.shape), x.dtype, None if glob_features is None else tuple(glob_features.shape),
               None if loc_features is None else tuple(loc_features.shape), getattr(m, "compute_dtype", torch.float32),
               float(m.dropout_p), bool(_co.CUDA_ROUTE), self._storage_key())
        st = self._graphs.pop(key, None)
        if st is None:
            m.__dict__["_graph_busy"] = True         # (the warm-up and the captured forward call the model itself: no nesting)
            try:
                st = self._capture(x, glob_features, loc_features)
            finally:
                m.__dict__["_graph_busy"] = False
            if self.max_shapes and len(self._graphs) >= self.max_shapes:
                self._graphs.pop(next(iter(self._graphs)))          # least recently used
        self._graphs[key] = st                                       # (re-inserted: most recently used last)
        B, N, _ = x.shape
        k = st["turn"]
        st["turn"] = 1 - k
        host, done = st["start_host"][k], st["done"][k]
        done.synchronize()                      # the copy that last read this staging buffer has executed
        if fps_start is None:                   # the same two CPU-generator draws, in the same order, as the eager path
            host[0].copy_(torch.randint(0, N, (B,), dtype=torch.long))
            host[1].copy_(torch.randint(0, m.sa1.num_points, (B,), dtype=torch.long))
        else:
            host[0].copy_(fps_start[0].to("cpu", torch.int32))
            host[1].copy_(fps_start[1].to("cpu", torch.int32))
        cur = torch.cuda.current_stream(x.device)
        st["stream"].wait_stream(cur)
        with torch.cuda.stream(st["stream"]):
            st["start_dev"].copy_(host, non_blocking=True)
            done.record()
            st["x"].copy_(x, non_blocking=True)
            if glob_features is not None:
                st["glob"].copy_(glob_features, non_blocking=True)
            if loc_features is not None:
                st["loc"].copy_(loc_features, non_blocking=True)
            st["g"].replay()
        cur.wait_stream(st["stream"])
        # the module's aux_* attributes (index tensors of the forward pass) are the graph's too: put this graph's back
        for k, v in st["aux"].items():
            setattr(m, k, v)
        if self.clone_outputs:
            return self._clones(st["out"])
        return st["out"]

    @staticmethod
    def _clones(out):
        """Copies of the graph's outputs that keep their view structure: outputs that share a storage (the three heads are
        slices of one packed [rows, 35] tensor) are copied ONCE and re-sliced."""
        copies, res = {}, []
        for o in out:
            stg = o.untyped_storage()
            key = stg.data_ptr()
            if key not in copies:
                whole = torch.empty(0, dtype=o.dtype, device=o.device).set_(stg, 0, (stg.nbytes() // o.element_size(),), (1,))
                copies[key] = (whole.clone(), o.dtype)
            c, dt = copies[key]
            res.append(torch.as_strided(c, o.size(), o.stride(), o.storage_offset()) if dt == o.dtype else o.clone())
        return type(out)(res)
