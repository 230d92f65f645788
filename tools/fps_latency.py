"""FPS latency model ("report it against its LDS/DPP step-latency model, not HBM").

Times cpfn_fps on the shapes the step and the evaluation use, and runs the stamped diagnostic twin (cpfn_fps_profile) of the
resident kernel: shader-clock cycles per sample and phase as wave 0 sees them.  Writes the markdown table behind
profiles/r03_fps_latency.md.   usage: python tools/fps_latency.py [out.md]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                              # noqa: E402
import torch                                                    # noqa: E402
from cpfn_amd import lib as _l, ops, synthetic                  # noqa: E402

dev = torch.device("cuda:0")
h = _l.lib()
PHASES = ("sample broadcast read (3 ds_read_b32)", "distance update + lane max (VALU)", "wave max (6 DPP steps + readlane)",
          "index of the max (ballots)", "LDS slot write + workgroup barrier", "slot read + max over waves (DPP)")
STAMP = 40          # cycles one stamp costs itself (two stamps back to back; cdna_hip_programming.md §7)


def timed(fn, reps=30):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


def main():
    lines = []
    P = synthetic.uniform_cloud(16, 8192, seed=1).to(dev)
    start = torch.randint(0, 8192, (16,), dtype=torch.int32, device=dev)
    ref = ops.fps(P, 512, start)
    rows = []
    rows.append(("16 x 8192 -> 512, critical path (8 waves x 16 points / lane)", timed(lambda: ops.fps(P, 512, start)), 512))
    with ops.background_geometry():
        rows.append(("16 x 8192 -> 512, beside a step (4 waves x 32 points / lane)", timed(lambda: ops.fps(P, 512, start)), 512))
    P2 = synthetic.uniform_cloud(16, 512, seed=2).to(dev)
    s2 = torch.randint(0, 512, (16,), dtype=torch.int32, device=dev)
    rows.append(("16 x 512 -> 128 (1 wave x 8)", timed(lambda: ops.fps(P2, 128, s2)), 128))
    P3 = synthetic.uniform_cloud(1, 131072, seed=3).to(dev)
    s3 = torch.randint(0, 131072, (1,), dtype=torch.int32, device=dev)
    rows.append(("1 x 131072 -> 512 (16 workgroups x 4 waves x 32, keys exchanged through memory)", timed(lambda: ops.fps(P3, 512, s3)), 512))
    P4 = synthetic.uniform_cloud(32, 8192, seed=4).to(dev)
    s4 = torch.randint(0, 8192, (32,), dtype=torch.int32, device=dev)
    rows.append(("32 x 8192 -> 512, critical path", timed(lambda: ops.fps(P4, 512, s4)), 512))
    lines.append("| call | us | us / sample |\n|---|---|---|")
    for name, us, S in rows:
        lines.append("| %s | %.1f | %.3f |" % (name, us, us / S))
    # ---- stamped twin
    lines.append("")
    lines.append("| phase (cycles per sample, wave 0, stamp cost of %d subtracted) | 8 waves x 16 | 4 waves x 32 |\n|---|---|---|" % STAMP)
    cols, totals, clocks = [], [], []
    for variant in (1, 2):
        out = torch.empty(16, 512, dtype=torch.int32, device=dev)
        prof = torch.zeros(16, 6, dtype=torch.int64, device=dev)

        def run():
            _l.check(h.cpfn_fps_profile(P.data_ptr(), 16, 8192, 512, start.data_ptr(), variant, out.data_ptr(), prof.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream), "cpfn_fps_profile")
        us = timed(run)
        assert torch.equal(out, ref), "the stamped kernel must select the same points"
        pr = prof.cpu().numpy().astype(np.float64) / 512.0 - STAMP
        cols.append(np.median(pr, axis=0))
        totals.append(us)
        clocks.append((prof.cpu().numpy().sum(1).astype(np.float64).mean()) / us)     # cycles per us = MHz (stamps included)
    for i, ph in enumerate(PHASES):
        lines.append("| %s | %.0f | %.0f |" % (ph, cols[0][i], cols[1][i]))
    lines.append("| sum | %.0f | %.0f |" % (cols[0].sum(), cols[1].sum()))
    lines.append("| stamped kernel, us per sample (shader clock from cycles / time: MHz) | %.3f (%.0f) | %.3f (%.0f) |"
                 % (totals[0] / 512, clocks[0], totals[1] / 512, clocks[1]))
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
