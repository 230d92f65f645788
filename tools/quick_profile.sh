#!/bin/bash
# One gpurun call while iterating on a kernel: GPU test subset (optional), the bench (ms / step), and the per-kernel
# breakdown of the replayed steps under rocprofv3.     bash tools/quick_profile.sh [tag] ["pytest args"]
tag=${1:-q}
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd "$repo"
mkdir -p gpurun_out/$tag
if [ -n "$2" ]; then timeout 900 python3 -m pytest $2 -x -q 2>&1 | tail -3; fi
python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-routes --no-traffic --no-rocprof 2> gpurun_out/$tag/bench.err | tail -1 > gpurun_out/$tag/bench.json
python3 -c "
import json; d=json.load(open('gpurun_out/$tag/bench.json')); r=d['roofline']
print('ms/step %.4f  clouds/s %.1f  step_frac %.3f' % (d['ms_per_step'], d['value'], r['step_frac']))
for k,v in r['families'].items(): print('  %-20s %2d launches %7.1f us  %.3f of peak' % (k, v['launches'], v['us_per_step'], v['frac']))
"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/stats -o s -- python3 bench.py --no-cpu-baseline --no-routes --no-traffic --no-rocprof --steps 10 --warmup 5 > gpurun_out/$tag/stats.log 2>&1
python3 tools/replay_breakdown.py $(find gpurun_out/$tag/stats -name '*kernel_trace.csv' | head -1) > gpurun_out/$tag/breakdown.txt 2> gpurun_out/$tag/breakdown.err
head -${3:-60} gpurun_out/$tag/breakdown.txt
find gpurun_out/$tag -name '*kernel_trace.csv' -size +8M -delete
find gpurun_out/$tag -name '*.db' -delete
