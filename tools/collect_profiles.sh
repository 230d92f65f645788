#!/bin/bash
# Everything profiles/ is refreshed from, in one gpurun call (run on the GPU box from the repo root):
#   bash tools/collect_profiles.sh [tag]  ->  gpurun_out/collect/... and profiles/<tag>_*
# Counter passes are separate runs with --kernel-trace only (never combined with sys/hip/hsa tracing), eager launches.
tag=${1:-r06}
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd "$repo"
mkdir -p gpurun_out/collect
B="bench.py --no-cpu-baseline --no-routes --no-traffic --no-rocprof"
# (CPFN_SIDE_GRAPH_FIRST=1: under the profiler a graph launch costs the host > 1 ms; with the step's graph submitted first the side
#  graph trails it by most of a step and the trace describes the profiler — cpfn_amd/training.py, profiles/README.md)
export CPFN_SIDE_GRAPH_FIRST=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/collect/stats -o s -- python3 $B --steps 10 --warmup 5 > gpurun_out/collect/stats.log 2>&1
unset CPFN_SIDE_GRAPH_FIRST
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/collect/fetch -o f -- python3 $B --steps 3 --warmup 3 --no-graphs > gpurun_out/collect/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/collect/write -o w -- python3 $B --steps 3 --warmup 3 --no-graphs > gpurun_out/collect/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/collect/mfma -o m -- python3 $B --steps 3 --warmup 3 --no-graphs > gpurun_out/collect/mfma.log 2>&1
python3 $B --steps 5 --warmup 5 --census-out gpurun_out/collect/census.json > gpurun_out/collect/census.log 2>&1
python3 tools/rooflines.py --trace gpurun_out/collect/stats --fetch gpurun_out/collect/fetch --write gpurun_out/collect/write \
    --mfma gpurun_out/collect/mfma --census gpurun_out/collect/census.json \
    --traffic-out profiles/${tag}_family_traffic.json > profiles/${tag}_rooflines.json 2> gpurun_out/collect/rooflines.err
python3 tools/replay_breakdown.py $(find gpurun_out/collect/stats -name '*kernel_trace.csv' | head -1) > profiles/${tag}_replayed_step_breakdown.txt 2> gpurun_out/collect/breakdown.err
cp $(find gpurun_out/collect/stats -name '*kernel_stats.csv' | head -1) profiles/${tag}_graph_kernel_stats.csv
python3 bench.py --steps 300 --warmup 30 2> gpurun_out/collect/bench.err | tail -1 > profiles/${tag}_bench_n1.json
python3 bench.py --steps 50 --warmup 10 --no-graphs --no-cpu-baseline --no-routes --no-rocprof 2> gpurun_out/collect/bench_eager.err | tail -1 > profiles/${tag}_bench_n1_eager.json
python3 bench.py --steps 100 --warmup 10 --workload local --no-routes --no-rocprof 2> gpurun_out/collect/bench_local.err | tail -1 > profiles/${tag}_bench_local_n1.json
python3 tools/fps_latency.py gpurun_out/collect/fps_latency_table.md > /dev/null 2> gpurun_out/collect/fps_latency.err
python3 tools/cascade_probe.py > profiles/${tag}_cascade_probe.txt 2> gpurun_out/collect/cascade.err
cp gpurun_out/collect/fps_latency_table.md profiles/${tag}_fps_latency_table.md 2>/dev/null
(CPFN_CSR_RADIX=0 python3 tools/dbg/csr_time.py; CPFN_CSR_THREADS=1024 python3 tools/dbg/csr_time.py; CPFN_CSR_THREADS=256 python3 tools/dbg/csr_time.py; CPFN_CSR_THREADS=-1 python3 tools/dbg/csr_time.py; python3 tools/dbg/csr_time.py) > profiles/${tag}_csr_time.txt 2> gpurun_out/collect/csr.err
# what the full-size parity tests ACHIEVED on this build: (A) fp32 mode, (B) every fused stack teacher-forced at bench
# size, (C) the replayed bf16 step against the fp32 oracle, and the fitters' worst per-instance errors at full size
python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fitters.py -m gpu -s -q 2>&1 | grep -E "^\[|per-instance|accepted|passed|failed" > profiles/${tag}_fullsize_parity.txt
# bench.py reads roofline.traffic from profiles/<tag>_family_traffic.json: a file older than the library it describes is a lie
if [ ! -s profiles/${tag}_family_traffic.json ] || [ profiles/${tag}_family_traffic.json -ot cpfn_amd/libcpfn_hip.so ]; then
  echo "collect_profiles: profiles/${tag}_family_traffic.json is missing or older than cpfn_amd/libcpfn_hip.so" >&2
  stale=1
fi
# the traces are large: only the summaries travel back (gpurun merges <= 64 MiB)
find gpurun_out/collect -name '*kernel_trace.csv' -size +8M -delete
find gpurun_out/collect -name '*.db' -delete
find gpurun_out/collect -name '*counter_collection.csv' -size +8M -delete
tail -n 3 gpurun_out/collect/rooflines.err; tail -n 3 gpurun_out/collect/breakdown.err; ls -la profiles | tail -12; head -c 1200 profiles/${tag}_rooflines.json
# (gpurun only merges gpurun_out/ back: the judged copies travel through it, then `cp gpurun_out/profiles_<tag>/* profiles/`)
mkdir -p gpurun_out/profiles_${tag} && cp profiles/${tag}_* gpurun_out/collect/fps_latency_table.md gpurun_out/profiles_${tag}/ 2>/dev/null
[ -z "$stale" ] || exit 1
