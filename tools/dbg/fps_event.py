"""Offline replay of a sampling divergence saved by tools/dbg/step_repro.py (gpurun_out/fps_event_*.pt): furthest-point sampling
of the saved cloud on the CPU with the kernel's arithmetic (fp32, ((dx*dx + dy*dy) + dz*dz), lowest index on ties); at the first
sample where the two recorded sequences part it prints which one the replay agrees with and how the other's pick ranks."""
import sys

import numpy as np
import torch


def main():
    for path in sys.argv[1:]:
        ev = torch.load(path)
        P, ref, bad = ev["P"].numpy().astype(np.float32), ev["idx_ref"].numpy(), ev["idx_bad"].numpy()
        print(path, "clouds equal:", bool((ev["P"] == ev["P_bad"]).all()))
        for b in range(P.shape[0]):
            if (ref[b] == bad[b]).all():
                continue
            i0 = int(np.nonzero(ref[b] != bad[b])[0][0])
            p = P[b]
            md = np.full(p.shape[0], 1e10, np.float32)
            far = int(ref[b, 0])
            for i in range(i0):
                assert far == ref[b, i], ("replay leaves the common prefix", b, i, far, ref[b, i])
                d = p - p[far]
                d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + (d[:, 2] * d[:, 2]).astype(np.float32)
                md = np.minimum(md, d2.astype(np.float32))
                far = int(np.argmax(md))
            order = np.argsort(-md, kind="stable")
            r, w = int(ref[b, i0]), int(bad[b, i0])
            print("  cloud %d parts at sample %d: replay picks %d; first run %d (rank %d, md %.6g, lane %d wave %d slot %d); "
                  "this run %d (rank %d, md %.6g, lane %d wave %d slot %d); top md %s" %
                  (b, i0, far, r, int(np.nonzero(order == r)[0][0]), md[r], r % 64, (r % 256) // 64, r // 256,
                   w, int(np.nonzero(order == w)[0][0]), md[w], w % 64, (w % 256) // 64, w // 256, md[order[:4]].tolist()))
            # was the other pick a point that had ALREADY been taken (md == 0), or the maximum of an earlier sample?
            prev = list(ref[b, :i0])
            for tag, x in (("first run", r), ("this run", w)):
                if x in prev:
                    print("    %s's pick was already sampled at %d" % (tag, prev.index(x)))


if __name__ == "__main__":
    main()


def detail(path):
    """For the wave that owns the wrong pick: is its min-distance the maximum over a SUBSET of the wave's lanes (a lost reduction
    step), with the min-distances after or before the sample's update?"""
    ev = torch.load(path)
    P, ref, bad = ev["P"].numpy().astype(np.float32), ev["idx_ref"].numpy(), ev["idx_bad"].numpy()
    for b in range(P.shape[0]):
        if (ref[b] == bad[b]).all():
            continue
        i0 = int(np.nonzero(ref[b] != bad[b])[0][0])
        p = P[b]
        md = np.full(p.shape[0], 1e10, np.float32)
        hist = []
        for i in range(i0):
            far = int(ref[b, i])
            hist.append(md.copy())
            d = p - p[far]
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + (d[:, 2] * d[:, 2]).astype(np.float32)
            md = np.minimum(md, d2.astype(np.float32))
        w = int(bad[b, i0])
        t = w % 256
        wave, lane = t // 64, t % 64
        for tag, m in (("after the update", md), ("before the update", hist[-1]), ("two updates back", hist[-2])):
            grid = m.reshape(8, 256)[:, wave * 64:(wave + 1) * 64]          # [slot][lane] of that wave
            lm = grid.max(0)
            v = m[w]
            sets = {"lane": [lane], "quad": list(range(lane & ~3, (lane & ~3) + 4)), "half row": list(range(lane & ~7, (lane & ~7) + 8)),
                    "row": list(range(lane & ~15, (lane & ~15) + 16)), "rows 2-3": list(range(32, 64)), "wave": list(range(64))}
            print("  %s: md[pick] %.6g; lane max %.6g;" % (tag, v, lm[lane]),
                  ", ".join("%s %.6g%s" % (k, lm[s].max(), "*" if lm[s].max() == v else "") for k, s in sets.items()))


if __name__ == "__main__" and len(sys.argv) > 1:
    for path in sys.argv[1:]:
        print(path)
        detail(path)
