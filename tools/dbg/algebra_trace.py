"""Reads the kernel trace of tools/dbg/algebra_time.py: per-launch durations of the algebra kernels, warm (first 65) vs cold (last 60)."""
import csv, glob, sys, statistics as st
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for key in ("fit_algebra_fwd_kernel", "fit_algebra_bwd_kernel"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000 for r in rows if key in r["Kernel_Name"]]
    print(key, "launches", len(d), "warm median %.1f us (min %.1f)" % (st.median(d[5:65]), min(d[5:65])), "cold median %.1f us (min %.1f, max %.1f)" % (st.median(d[65:]), min(d[65:]), max(d[65:])))
