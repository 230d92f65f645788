"""What would the step cost without the eager launches between consecutive replays?  Replays the trainer's captured
graphs directly (same kernels; the inputs are simply not refreshed): (a) the trainer's own step(), (b) step graph + side
graph with the two events but NO copy launches in between, (c) the step graph alone back to back.
    python tools/dbg/twin_potential.py [steps=300]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cpfn_amd import synthetic, training, lib
from cpfn_amd.PointNet2 import pn2_network
from cpfn_amd.SPFN import fitter_factory
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    fitter_factory.register_primitives(training.GLOBAL_SPFN_CLASSES)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[3, 4, 28]).to(dev)
model.set_compute_dtype(torch.bfloat16)
tr = training.SPFNTrainer(model, batch_size=16, use_graphs=True, require_graphs=True)
batch = {k: v.to(dev) for k, v in synthetic.training_batch(16, 8192, 28, seed=1000).items()}
torch.cuda.set_stream(tr.stream(dev))
for _ in range(8):
    tr.step(batch, next_batch=batch)
torch.cuda.synchronize()
st = tr._graph


def timed(fn, label):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("%-60s %.4f ms/step" % (label, 1e3 * dt))


timed(lambda: tr.step(batch, next_batch=batch), "(a) trainer.step")
cur = torch.cuda.current_stream(dev)
side = tr._gside


def both():
    cur.wait_event(st["b_written"])
    st["b_read"].record(cur)
    with torch.cuda.stream(side):
        side.wait_event(st["b_read"])
        st["gs"].replay()
        st["b_written"].record(side)
    st["g"].replay()


timed(both, "(b) step graph + side graph, no copy launches in between")
torch.cuda.synchronize()
timed(lambda: st["g"].replay(), "(c) step graph alone, back to back")


def events_only():
    cur.wait_event(st["b_written"])
    st["b_read"].record(cur)
    with torch.cuda.stream(side):
        side.wait_event(st["b_read"])
        st["b_written"].record(side)
    st["g"].replay()


torch.cuda.synchronize()
timed(events_only, "(d) the two events, side graph NOT replayed")


def no_main_wait():
    st["b_read"].record(cur)
    with torch.cuda.stream(side):
        side.wait_event(st["b_read"])
        st["gs"].replay()
        st["b_written"].record(side)
    st["g"].replay()


torch.cuda.synchronize()
timed(no_main_wait, "(e) as (b) without the main stream's wait on the side graph")


def unsync():
    with torch.cuda.stream(side):
        st["gs"].replay()
    st["g"].replay()


torch.cuda.synchronize()
timed(unsync, "(f) both graphs, no events at all")
torch.cuda.synchronize()
timed(lambda: st["gs"].replay(), "(g) side graph alone on the main stream")


def rec_only():
    st["b_read"].record(cur)
    st["g"].replay()


def wait_only():
    cur.wait_event(st["b_written"])
    st["g"].replay()


stamp_buf = torch.zeros(4, dtype=torch.int64, device=dev)


def tiny_kernel_between():
    lib.lib().cpfn_stamp(stamp_buf.data_ptr(), cur.cuda_stream)
    st["g"].replay()


def side_events_only():
    with torch.cuda.stream(side):
        side.wait_event(st["b_read"])
        st["b_written"].record(side)
    st["g"].replay()


for fn, label in () and ((rec_only, "(h) one event record on the main stream per step, nothing else"),
                  (wait_only, "(i) one wait on an already completed event per step, nothing else"),
                  (tiny_kernel_between, "(j) one tiny eager kernel between the replays"),
                  (side_events_only, "(k) wait + record on the SIDE stream only")):
    torch.cuda.synchronize()
    timed(fn, label)


# ---- device flags polled / set by tiny eager kernels on each stream
flags = torch.zeros(4, dtype=torch.int32, device=dev)
hh = lib.lib()
f_consumed, f_geom, f_err = flags[0:].data_ptr(), flags[1:].data_ptr(), flags[2:].data_ptr()
nn = [0]
TMO = 50_000_000      # 0.5 s


def both_flags():
    nn[0] += 1
    k = nn[0]
    hh.cpfn_flag_wait(f_geom, k - 1, TMO, f_err, cur.cuda_stream)
    hh.cpfn_flag_set(f_consumed, k, cur.cuda_stream)
    hh.cpfn_flag_wait(f_consumed, k, TMO, f_err, side.cuda_stream)
    with torch.cuda.stream(side):
        st["gs"].replay()
    hh.cpfn_flag_set(f_geom, k, side.cuda_stream)
    st["g"].replay()


def flags_only():
    nn[0] += 1
    k = nn[0]
    hh.cpfn_flag_wait(f_geom, k - 1, TMO, f_err, cur.cuda_stream)
    hh.cpfn_flag_set(f_consumed, k, cur.cuda_stream)
    hh.cpfn_flag_wait(f_consumed, k, TMO, f_err, side.cuda_stream)
    hh.cpfn_flag_set(f_geom, k, side.cuda_stream)
    st["g"].replay()


# ---- which part of the geometry costs the step what?  Side graphs with ONE part of the work each
from cpfn_amd import ops
Pn = st["P_next"]
s1, s2 = st["start1"], st["start2"]
gA = st["geomA"]
xyz1 = gA["sa1"]["new_xyz"].clone()
xyz2 = gA["sa2"]["new_xyz"].clone()
idx_sa2 = gA["sa2"]["scales"][0][0].clone()
nn3, nn2 = gA["sfp3"]["nn_idx"].clone(), gA["sfp2"]["nn_idx"].clone()
parts = {
    "FPS (8192 -> 512, 512 -> 128)": lambda: (ops.fps(Pn, 512, s1), ops.fps(xyz1, 128, s2)),
    "ball query x2 + 3-NN x2": lambda: (ops.ball_query(xyz1, Pn, 0.2, 64), ops.ball_query(xyz2, xyz1, 0.4, 64),
                                         ops.three_nn(Pn, xyz1), ops.three_nn(xyz1, xyz2)),
    "inverse-index build x3": lambda: (ops.csr_build(idx_sa2, 512), ops.csr_build(nn3, 512), ops.csr_build(nn2, 128)),
}
_bg = ops.background_geometry()
_bg.__enter__()                  # the shapes the trainer's geometry pass uses beside a step
for name, fn in parts.items():
    with torch.cuda.stream(side):
        fn()
    torch.cuda.synchronize()
    gpart = torch.cuda.CUDAGraph()
    side.wait_stream(cur)
    with torch.cuda.graph(gpart, stream=side, capture_error_mode="thread_local"):
        keep = fn()
    cur.wait_stream(side)

    def part_step():
        nn[0] += 1
        k = nn[0]
        hh.cpfn_flag_wait(f_geom, k - 1, TMO, f_err, cur.cuda_stream)
        hh.cpfn_flag_set(f_consumed, k, cur.cuda_stream)
        hh.cpfn_flag_wait(f_consumed, k, TMO, f_err, side.cuda_stream)
        with torch.cuda.stream(side):
            gpart.replay()
        hh.cpfn_flag_set(f_geom, k, side.cuda_stream)
        st["g"].replay()
    torch.cuda.synchronize()
    timed(part_step, "(q) side graph = %s" % name)
    torch.cuda.synchronize()
    timed(lambda: gpart.replay(), "    that side graph alone")
_bg.__exit__(None, None, None)
# ---- phase of the side graph relative to the step: released X us after the step's start (spin kernel on the side stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(1_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 1_000_000 / (1e3 * e0.elapsed_time(e1))
for delay_us in ():
    def delayed():
        nn[0] += 1
        k = nn[0]
        hh.cpfn_flag_wait(f_geom, k - 1, TMO, f_err, cur.cuda_stream)
        hh.cpfn_flag_set(f_consumed, k, cur.cuda_stream)
        hh.cpfn_flag_wait(f_consumed, k, TMO, f_err, side.cuda_stream)
        with torch.cuda.stream(side):
            if delay_us:
                torch.cuda._sleep(int(delay_us * cyc_per_us))
            st["gs"].replay()
        hh.cpfn_flag_set(f_geom, k, side.cuda_stream)
        st["g"].replay()
    torch.cuda.synchronize()
    timed(delayed, "(p) flags, side graph released %d us after the step's start" % delay_us)
torch.cuda.synchronize()
timed(both_flags, "(n) both graphs ordered by device flags (eager 1-lane kernels)")
torch.cuda.synchronize()
timed(flags_only, "(o) the four flag kernels, side graph NOT replayed")
torch.cuda.synchronize()
print("flags (consumed, geometry, error):", flags[:3].tolist(), "steps issued", nn[0])
sys.exit(0)
# ---- stream memory operations (hipStreamWaitValue32 / hipStreamWriteValue32) instead of events
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
consumed, geom = ctypes.c_void_p(), ctypes.c_void_p()
hip.hipExtMallocWithFlags.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
rc = hip.hipExtMallocWithFlags(ctypes.byref(consumed), 8, 0x2) or hip.hipExtMallocWithFlags(ctypes.byref(geom), 8, 0x2)
print("hipExtMallocWithFlags(signal memory):", rc)
if rc == 0:
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    hip.hipMemset(consumed, 0, 8)
    hip.hipMemset(geom, 0, 8)
    torch.cuda.synchronize()
    hip.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
    hip.hipStreamWriteValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
    n = [0]
    errs = set()

    def both_values():
        n[0] += 1
        k = n[0]
        errs.add(hip.hipStreamWaitValue32(cur.cuda_stream, geom, k - 1, 0, 0xffffffff))      # geometry of this step ready
        errs.add(hip.hipStreamWriteValue32(cur.cuda_stream, consumed, k, 0))                  # ... and consumed
        errs.add(hip.hipStreamWaitValue32(side.cuda_stream, consumed, k, 0, 0xffffffff))
        with torch.cuda.stream(side):
            st["gs"].replay()
        errs.add(hip.hipStreamWriteValue32(side.cuda_stream, geom, k, 0))
        st["g"].replay()

    def values_only():
        n[0] += 1
        k = n[0]
        errs.add(hip.hipStreamWaitValue32(cur.cuda_stream, geom, k - 1, 0, 0xffffffff))
        errs.add(hip.hipStreamWriteValue32(cur.cuda_stream, consumed, k, 0))
        errs.add(hip.hipStreamWaitValue32(side.cuda_stream, consumed, k, 0, 0xffffffff))
        errs.add(hip.hipStreamWriteValue32(side.cuda_stream, geom, k, 0))
        st["g"].replay()

    torch.cuda.synchronize()
    timed(both_values, "(l) both graphs ordered by stream wait / write values")
    torch.cuda.synchronize()
    timed(values_only, "(m) the four value operations, side graph NOT replayed")
    print("status codes seen:", errs)
sys.exit(0)
# ---- the same choreography with device-scope events (cpfn_event_*)
import ctypes
h = lib.lib()
for flags, name in ((0x2 | 0x40000000, "DisableTiming|ReleaseToDevice"), (0x2 | 0x20000000, "DisableTiming|DisableSystemFence"),
                    (0x2, "DisableTiming (torch's)")):
    evs = []
    for _ in range(2):
        e = ctypes.c_void_p()
        rc = h.cpfn_event_create(ctypes.byref(e), flags)
        if rc:
            print("event flags %s: create failed" % name)
            break
        evs.append(e)
    if len(evs) < 2:
        continue
    b_read, b_written = evs
    h.cpfn_event_record(b_written, side.cuda_stream)

    def both_dev():
        h.cpfn_stream_wait_event(cur.cuda_stream, b_written)
        h.cpfn_event_record(b_read, cur.cuda_stream)
        h.cpfn_stream_wait_event(side.cuda_stream, b_read)
        with torch.cuda.stream(side):
            st["gs"].replay()
        h.cpfn_event_record(b_written, side.cuda_stream)
        st["g"].replay()

    def ev_only_dev():
        h.cpfn_stream_wait_event(cur.cuda_stream, b_written)
        h.cpfn_event_record(b_read, cur.cuda_stream)
        h.cpfn_stream_wait_event(side.cuda_stream, b_read)
        h.cpfn_event_record(b_written, side.cuda_stream)
        st["g"].replay()

    torch.cuda.synchronize()
    timed(both_dev, "(b') both graphs, events %s" % name)
    torch.cuda.synchronize()
    timed(ev_only_dev, "(d') events only, %s" % name)
