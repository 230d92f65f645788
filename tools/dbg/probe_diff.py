"""Two probe dumps (bench.py --probe-dump) side by side, launches matched by order: duration and the time since the
previous probed launch ended, and their differences.   python tools/dbg/probe_diff.py a.json b.json"""
import json, sys
KIND = {1: "stream", 2: "generic", 3: "smallp", 4: "wgrad", 5: "onepass", 6: "small-bwd"}
a, b = (json.load(open(p))["launches"] for p in sys.argv[1:3])
assert len(a) == len(b), (len(a), len(b))
pa = pb = None
td = tg = 0.0
for x, y in zip(a, b):
    da, db = (x["end"] - x["start"]) / 100.0, (y["end"] - y["start"]) / 100.0
    ga = 0.0 if pa is None else (x["start"] - pa) / 100.0
    gb = 0.0 if pb is None else (y["start"] - pb) / 100.0
    pa, pb = x["end"], y["end"]
    td += da - db
    tg += ga - gb
    print("%-8s nwg %5d   dur %6.1f %6.1f  (%+5.1f)    since-prev-end %7.1f %7.1f  (%+6.1f)" % (KIND.get(x["kind"], "?"), x["nwg"], da, db, da - db, ga, gb, ga - gb))
print("sum of duration differences %+.1f us, of in-between differences %+.1f us" % (td, tg))
