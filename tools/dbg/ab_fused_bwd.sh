cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for f in 0 1; do
CPFN_FUSED_BWD=$f rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab$f -o s -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/ab$f.log 2>&1
python3 tools/replay_timeline.py $(find gpurun_out/ab$f -name '*kernel_trace.csv' | head -1) 12 > gpurun_out/ab${f}_timeline.txt 2>&1
done
find gpurun_out/ab0 gpurun_out/ab1 -name '*kernel_trace.csv' -delete
find gpurun_out/ab0 gpurun_out/ab1 -name '*.db' -delete
