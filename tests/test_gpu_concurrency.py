"""Kernels of the step run concurrently (the next batch's geometry on a side stream / a forked graph branch
next to the backward pass), so they must be immune to what their neighbours on the same CU do.

Regression test for a hardware interaction found in round 1: `ds_read_b96` (what the compiler emits for three
floats of a float4 in LDS) returned wrong data now and then while a workgroup of another kernel on the same
CU was writing LDS heavily.  Furthest-point sampling next to the weight-gradient kernel picked a spurious point
in 195 of 200 runs; with b32 / b128 reads: 0 of 200.  (cpfn_amd/csrc/common.h, cpfn_lds_read4; the build rejects
96-bit DS instructions.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup():
    from cpfn_amd import lib as _l, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    P1 = torch.rand(16, 8192, 3, device=dev)
    P2 = torch.rand(16, 512, 3, device=dev)
    start = torch.randint(0, 512, (16,), device=dev, dtype=torch.int32)
    Y = torch.randn(131072, 128, device=dev).to(torch.bfloat16)
    h = _l.lib()
    splits = h.cpfn_mlp_wgrad_splits(131072, 128, 128)
    ws = torch.empty(splits * 128 * 128, device=dev)
    dW = torch.empty(128, 128, device=dev)

    def wgrad(n):
        for _ in range(n):
            rc = h.cpfn_mlp_wgrad(Y.data_ptr(), 128, Y.data_ptr(), 128, None, 131072, 128, 128, None, None, ws.data_ptr(), dW.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream)
            assert rc == 0
    return ops, P1, P2, start, wgrad


def test_fps_is_immune_to_a_neighbouring_lds_heavy_kernel():
    ops, P1, P2, start, wgrad = _setup()
    ref1, ref2 = ops.fps(P1, 512, start), ops.fps(P2, 128, start)
    side = torch.cuda.Stream()
    bad = 0
    for _ in range(100):
        side.wait_stream(torch.cuda.current_stream())
        wgrad(3)
        with torch.cuda.stream(side):
            s2 = ops.fps(P2, 128, start)
            s1 = ops.fps(P1, 512, start)
        wgrad(30)
        torch.cuda.synchronize()
        bad += int(not (torch.equal(s1, ref1) and torch.equal(s2, ref2)))
    assert bad == 0, "%d of 100 overlapped FPS runs differ from the quiet run" % bad


def test_fps_on_a_forked_graph_branch():
    ops, P1, P2, start, wgrad = _setup()
    ref2 = ops.fps(P2, 128, start)
    gs, side = torch.cuda.Stream(), torch.cuda.Stream()
    out = {}
    with torch.cuda.stream(gs):
        wgrad(2)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=gs, capture_error_mode="thread_local"):
            side.wait_stream(gs)
            with torch.cuda.stream(side):
                out["s2"] = ops.fps(P2, 128, start)
            wgrad(20)
            gs.wait_stream(side)
        bad = 0
        for _ in range(100):
            g.replay()
            torch.cuda.synchronize()
            bad += int(not torch.equal(out["s2"], ref2))
    assert bad == 0, "%d of 100 graph replays produced different FPS indices" % bad


def test_cross_stream_flags_order_two_streams_and_time_out():
    """cpfn_flag_wait / cpfn_flag_set (what orders the step's two replayed graphs instead of events): a consumer stream
    that waits for flag >= k sees everything the producer stream wrote before it set the flag to k, over many rounds and
    in both directions; a waiter whose setter never comes gives up after its timeout and reports it instead of hanging."""
    from cpfn_amd import lib as _l
    dev = torch.device("cuda:0")
    h = _l.lib()
    flags = torch.zeros(4, dtype=torch.int32, device=dev)
    err = torch.zeros(4, dtype=torch.int32).pin_memory()
    a, b = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    data = torch.zeros(1 << 20, dtype=torch.float32, device=dev)
    seen = torch.zeros(64, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    TMO = 200_000_000          # 2 s of the 100 MHz clock
    for k in range(1, 65):
        with torch.cuda.stream(a):                     # producer: waits until round k - 1 was consumed, then writes round k
            _l.check(h.cpfn_flag_wait(flags[1:].data_ptr(), k - 1, TMO, err.data_ptr(), None, a.cuda_stream), "cpfn_flag_wait")
            data.fill_(float(k))
            _l.check(h.cpfn_flag_set(flags[0:].data_ptr(), k, a.cuda_stream), "cpfn_flag_set")
        with torch.cuda.stream(b):                     # consumer
            _l.check(h.cpfn_flag_wait(flags[0:].data_ptr(), k, TMO, err.data_ptr(), None, b.cuda_stream), "cpfn_flag_wait")
            seen[k - 1:k].copy_(data.min().reshape(1))
            _l.check(h.cpfn_flag_set(flags[1:].data_ptr(), k, b.cuda_stream), "cpfn_flag_set")
    torch.cuda.synchronize()
    assert int(err[0]) == 0
    assert torch.equal(seen.cpu(), torch.arange(1, 65, dtype=torch.float32)), seen
    assert flags[:2].tolist() == [64, 64]
    # a setter with a payload (the trainer's FPS seeds): the ints are in place when the waiter's stream goes on
    import ctypes
    dst = torch.zeros(64, dtype=torch.int32, device=dev)
    got = torch.zeros(8, 64, dtype=torch.int32, device=dev)
    for k in range(65, 73):
        vals = (ctypes.c_int * 40)(*[k * 1000 + i for i in range(40)])
        with torch.cuda.stream(a):
            _l.check(h.cpfn_flag_wait(flags[1:].data_ptr(), k - 1, TMO, err.data_ptr(), None, a.cuda_stream), "cpfn_flag_wait")
            _l.check(h.cpfn_flag_set_payload(flags[0:].data_ptr(), k, dst.data_ptr(), vals, 40, a.cuda_stream), "cpfn_flag_set_payload")
        with torch.cuda.stream(b):
            _l.check(h.cpfn_flag_wait(flags[0:].data_ptr(), k, TMO, err.data_ptr(), None, b.cuda_stream), "cpfn_flag_wait")
            got[k - 65].copy_(dst)
            _l.check(h.cpfn_flag_set(flags[1:].data_ptr(), k, b.cuda_stream), "cpfn_flag_set")
    torch.cuda.synchronize()
    want = torch.tensor([[k * 1000 + i if i < 40 else 0 for i in range(64)] for k in range(65, 73)], dtype=torch.int32)
    assert torch.equal(got.cpu(), want) and int(err[0]) == 0
    assert h.cpfn_flag_set_payload(flags.data_ptr(), 1, dst.data_ptr(), vals, 65, None) != 0       # more than 64 ints
    # nobody sets flag 2: the waiter returns after ~10 ms and raises the error word
    fault = torch.zeros((), dtype=torch.float32, device=dev)
    _l.check(h.cpfn_flag_wait(flags[2:].data_ptr(), 1, 1_000_000, err.data_ptr(), fault.data_ptr(),
                              torch.cuda.current_stream().cuda_stream), "cpfn_flag_wait")
    torch.cuda.synchronize()
    assert int(err[0]) == 1 and float(fault) == 1.0          # host word for the host, device word for the optimizer
    assert h.cpfn_flag_wait(None, 1, 1, None, None, None) != 0 and h.cpfn_flag_set(None, 1, None) != 0
