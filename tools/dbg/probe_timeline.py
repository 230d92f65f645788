"""Timeline of the GEMM-family, weight-gradient and one-pass backward launches inside the LAST replayed step of a
bench.py run, from the in-kernel probe (device wall clock, 100 MHz): `bench.py --probe-dump f.json` -> this script.
Shows per launch: offset from the first anchor, duration, the time since the previous anchor ENDED (= everything that
ran in between on the critical path) and how ragged the workgroups were.
    python tools/dbg/probe_timeline.py a.json [b.json]      (two files: side by side, matching launches by order)"""
import json, sys
KIND = {1: "stream", 2: "generic", 3: "smallp", 4: "wgrad", 5: "onepass", 6: "small-bwd"}


def load(p):
    f = json.load(open(p))
    d, st = f["launches"], f.get("stamps")
    out, prev_end = [], 0.0 if st else d[0]["start"]
    for r in d:
        out.append((KIND.get(r["kind"], "?"), r["nwg"], r["start"] / 100.0, (r["end"] - r["start"]) / 100.0,
                    (r["start"] - prev_end) / 100.0, r["wg_mean"] / 100.0, (r["last_start"] - r["start"]) / 100.0))
        prev_end = r["end"]
    return out, st


files = [load(p) for p in sys.argv[1:]]
for p, (rows, st) in zip(sys.argv[1:], files):
    print("%s: %d launches (mean of 40 replays), sum of launch durations %.1f us" % (p, len(rows), sum(r[3] for r in rows)))
    if st:
        print("  stamps: geometry branch ends %.1f, main chain ends %.1f, joined %.1f; idle before the step's first node %.1f"
              % tuple(st[k] / 100.0 for k in ("geometry_end", "main_end", "joined", "gap_before")))
    for k, nwg, off, dur, since, wgm, ls in rows:
        print("  %8.1f  %-8s nwg %5d  dur %6.1f  since-prev-end %7.1f  wg-mean %6.1f  last-wg-start +%5.1f" % (off, k, nwg, dur, since, wgm, ls))
