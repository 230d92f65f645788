"""Does the packed-fp32 sampling kernel lose an update beside a weight-gradient workgroup ON THIS BOX?  Works on the PRODUCT library
(no debugging symbols): the packed instantiation loops on a side stream, the 64 x 64-tile `cpfn_mlp_wgrad` (8192 rows, 128 -> 128:
116 registers, 9 KB of LDS — it fits beside a sampling workgroup) on the main stream; a launch counts as bad when its indices differ
from the quiet run's.  Shapes: 4 x 2048 points (fps_resident_kernel<256, 8>, packed outside background_geometry) and 16 x 8192
(<512, 16> packed; with CPFN_FPS_BESIDE_MODE=2 and `beside` the 4-wave x 32-point shape without its LDS claim).
    python tools/dbg/pk_repro.py [seconds per case] [cases: small,large,beside2,scalar]
Prints one line per case and the device's identity (the pool's boxes may differ)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
from cpfn_amd import lib as _l, ops          # noqa: E402


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    cases = (sys.argv[2] if len(sys.argv) > 2 else "small,large,beside2,scalar").split(",")
    dev = torch.device("cuda:0")
    h = _l.lib()
    p = torch.cuda.get_device_properties(0)
    uid = ""
    try:
        import glob
        for f in sorted(glob.glob("/sys/class/drm/card*/device/unique_id")):
            uid += open(f).read().strip() + " "
    except Exception:
        pass
    print("device: %s, %d CUs, unique ids: %s" % (p.name, p.multi_processor_count, uid or "?"), flush=True)
    P_, N_ = 8192, 128
    gk = torch.Generator().manual_seed(7)
    Gy = torch.randn(P_, N_, generator=gk).to(dev).to(torch.bfloat16)
    A = torch.randn(P_, N_, generator=gk).to(dev).to(torch.bfloat16)
    splits = h.cpfn_mlp_wgrad_splits(P_, N_, N_)
    ws = torch.empty(splits * N_ * N_, device=dev)

    def wgrad():
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(60):
            _l.check(h.cpfn_mlp_wgrad(Gy.data_ptr(), N_, A.data_ptr(), N_, None, P_, N_, N_, None, None, ws.data_ptr(), None, s), "wgrad")

    side = torch.cuda.Stream()
    for case in cases:
        B, N = (4, 2048) if case in ("small", "scalar") else (16, 8192)
        g = torch.Generator().manual_seed(1)
        xyz = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(dev)
        start = torch.randint(0, N, (B,), generator=g).to(torch.int32).to(dev)
        beside = case in ("beside2", "scalar", "beside1")
        if case == "beside2":
            os.environ["CPFN_FPS_BESIDE_MODE"] = "2"          # (read once per process by the launcher: run this case in its own process)

        def fps():
            if beside:
                with ops.background_geometry():
                    return ops.fps(xyz, 512, start)
            return ops.fps(xyz, 512, start)
        ref = fps().clone()
        torch.cuda.synchronize()
        t0, launches, bad = time.time(), 0, torch.zeros((), dtype=torch.int32, device=dev)
        while time.time() - t0 < secs:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(16):
                    bad += (fps() != ref).any().int()
                    launches += 1
            wgrad()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
        print("%-8s (%2d x %4d points%s): %6d sampling launches beside mlp_wgrad, %5d with different indices; sampling faults word %d"
              % (case, B, N, ", background_geometry" if beside else "", launches, int(bad), ops.fps_faults()), flush=True)


if __name__ == "__main__":
    main()
