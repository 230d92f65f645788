# rocprofv3 kernel trace of bench.py -> per-kernel breakdown + timeline of one replayed step (gpurun_out/tl_*)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tl -o s -- python3 bench.py --no-cpu-baseline --no-routes --no-traffic --steps 20 --warmup 5 > gpurun_out/tl.log 2>&1
python3 tools/replay_timeline.py $(find gpurun_out/tl -name '*kernel_trace.csv' | head -1) 12 > gpurun_out/tl_timeline.txt 2>&1
python3 tools/replay_breakdown.py $(find gpurun_out/tl -name '*kernel_trace.csv' | head -1) > gpurun_out/tl_breakdown.txt 2>&1
find gpurun_out/tl -name '*kernel_trace.csv' -delete
find gpurun_out/tl -name '*.db' -delete
