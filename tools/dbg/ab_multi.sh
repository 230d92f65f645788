# interleaved comparison of several built source trees on ONE box:  bash tools/dbg/ab_multi.sh ".r2ref . .r2mid . .r2ref .r2mid"
for d in $1; do
  (cd $d && python3 bench.py --steps ${STEPS:-500} --warmup 20 --no-cpu-baseline --no-routes --no-traffic $(grep -q no-rocprof bench.py && echo --no-rocprof) 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$d', round(d['ms_per_step'],4), round(d['value'],1))")
done
