"""Floor of back-to-back hipGraph replays on this stack: a graph of n tiny kernels (one stream, or with one forked
branch), replayed 300 times -> microseconds per replay.   python tools/dbg/graph_gap.py"""
import time
import torch
dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev)


def run(n, fork):
    s = torch.cuda.Stream()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        for _ in range(3):
            x.add_(1)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            if fork:
                side.wait_stream(s)
                with torch.cuda.stream(side):
                    y = x * 2
            for _ in range(n):
                x.add_(1)
            if fork:
                s.wait_stream(side)
    torch.cuda.synchronize()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 300 * 1e6


for n in (1, 10, 50, 150):
    print("n=%3d kernels: %.1f us/replay   with one forked branch: %.1f us/replay" % (n, run(n, False), run(n, True)))
