# same-box A/B of environment settings on this tree:  bash tools/dbg/ab_env2.sh "VAR=1" "VAR=2" ...   (each setting run three times, interleaved)
for rep in 1 2 3; do
  for e in "$@"; do
    env $e python3 bench.py --steps ${STEPS:-400} --warmup 20 --no-cpu-baseline --no-routes --no-traffic --no-rocprof 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$e', round(d['ms_per_step'],4), round(d['value'],1))"
  done
done
