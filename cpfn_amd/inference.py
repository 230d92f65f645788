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
    outlive the next call)."""

    def __init__(self, model):
        self.model = model
        self._graphs = {}

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
        with torch.cuda.stream(stream), torch.no_grad():
            m(st["x"], glob_features=st["glob"], loc_features=st["loc"], fps_start=starts)     # warm-up: lazily created state
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                st["out"] = m(st["x"], glob_features=st["glob"], loc_features=st["loc"], fps_start=starts)
        torch.cuda.current_stream(dev).wait_stream(stream)
        st["g"], st["stream"] = g, stream
        return st

    @torch.no_grad()
    def __call__(self, x, glob_features=None, loc_features=None, fps_start=None):
        m = self.model
        if m.training:
            raise RuntimeError("GraphedForward replays an evaluation-mode forward: call model.eval() first")
        if not x.is_cuda:
            raise RuntimeError("GraphedForward: CPU not supported")
        key = (tuple(x.shape), x.dtype, None if glob_features is None else tuple(glob_features.shape),
               None if loc_features is None else tuple(loc_features.shape), getattr(m, "compute_dtype", torch.float32),
               float(m.dropout_p))
        st = self._graphs.get(key)
        if st is None:
            st = self._graphs[key] = self._capture(x, glob_features, loc_features)
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
        return st["out"]
