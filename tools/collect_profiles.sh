#!/bin/bash
# Everything profiles/ is refreshed from, in one gpurun call (run on the GPU box from the repo root):
#   bash tools/collect_profiles.sh   ->  gpurun_out/collect/{stats,fetch,write}/..., bench_n1.json, bench_n1_eager.json
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd "$repo"
out=gpurun_out/collect
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o f -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-graphs > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o w -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-graphs > $out/write.log 2>&1
python3 tools/pmc_traffic.py $out/fetch $out/write > $out/mlp_gemm_traffic.json 2> $out/traffic.err
cp $out/mlp_gemm_traffic.json profiles/r01_mlp_gemm_traffic.json     # bench.py reads the traffic figure from here
python3 bench.py --steps 300 --warmup 30 2> $out/bench.err | tail -1 > $out/bench_n1.json
python3 bench.py --steps 50 --warmup 10 --no-graphs --no-cpu-baseline 2> $out/bench_eager.err | tail -1 > $out/bench_n1_eager.json
rm -f $out/fetch/*kernel_trace.csv $out/write/*kernel_trace.csv
ls -la $out $out/stats | head -30
