"""Per-kernel time inside the REPLAYED steps of a rocprofv3 kernel trace of bench.py (tools/profile_bench.sh):
the eager warm-up / capture / roofline steps in the same trace are left out.
    python tools/replay_breakdown.py gpurun_out/prof_<tag>/<tag>_kernel_trace.csv [--torch]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
nm = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
idx = [i for i, r in enumerate(rows) if 'adam_flat_kernel' in r['Kernel_Name']]
agg = collections.defaultdict(lambda: [0, 0.0])
steps = range(7, min(14, len(idx) - 1))
for k in steps:
    for r in rows[idx[k] + 1:idx[k + 1] + 1]:
        agg[nm(r)][0] += 1
        agg[nm(r)][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
n = len(steps)
tot = sum(v[1] for v in agg.values()) / n
print("%d replayed steps: %.0f kernels, %.1f us of kernel time per step" % (n, sum(v[0] for v in agg.values()) / n, tot))
only_torch = "--torch" in sys.argv
tt = 0.0
for name, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    is_t = name.startswith('at::') or 'rocclr' in name
    tt += v[1] / n if is_t else 0
    if only_torch and not is_t:
        continue
    print("%-104s %5.1f/step %8.1f us/step" % (name[:104], v[0] / n, v[1] / n))
print("framework kernels: %.1f us per step" % tt)
