"""Timeline of ONE replayed step from a rocprofv3 kernel trace of bench.py: every kernel in start order with its
offset from the step's first kernel, its duration, the queue it ran on and the idle gap since the previous kernel
ended on ANY queue (a gap with nothing running is pure dependency / launch latency).
    python tools/replay_timeline.py <kernel_trace.csv> [step_index=9] [--gaps-only US]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
nm = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
idx = [i for i, r in enumerate(rows) if 'adam_flat_kernel' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 9
step = rows[idx[k] + 1:idx[k + 1] + 1]
t0 = int(step[0]['Start_Timestamp'])
qs = {}
busy_end = t0
idle = 0.0
span = (int(step[-1]['End_Timestamp']) - t0) / 1e3
print("step %d: %d kernels, span %.1f us" % (k, len(step), span))
thr = None
if "--gaps-only" in sys.argv:
    thr = float(sys.argv[sys.argv.index("--gaps-only") + 1])
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    q = qs.setdefault(r.get('Queue_Id', '?'), len(qs))
    gap = (s - busy_end) / 1e3
    if gap > 0:
        idle += gap
    if thr is None or gap > thr:
        print("%9.1f  q%d  %7.1f us  gap %6.1f  %s  grid %s wg %s" % ((s - t0) / 1e3, q, (e - s) / 1e3, gap, nm(r)[:70],
              r.get('Grid_Size', r.get('Grid_Size_X', '?')), r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?'))))
    busy_end = max(busy_end, e)
print("idle (no kernel running on any queue): %.1f us of %.1f" % (idle, span))
