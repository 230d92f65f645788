#!/bin/bash
# rocprofv3 kernel statistics of the default bench (run on the GPU box from the repo root):
#   bash tools/profile_bench.sh <tag>   ->  gpurun_out/prof_<tag>/<tag>_kernel_stats.csv
tag=${1:-run}
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd "$repo"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-routes --no-traffic --no-rocprof > gpurun_out/prof_$tag.log 2>&1
