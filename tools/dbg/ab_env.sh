# interleaved A/B of an environment flag on bench.py:  bash tools/dbg/ab_env.sh CPFN_FUSED_BWD "0 1 1 0 0 1 1 0"
var=$1; seq=$2
for f in $seq; do
  env $var=$f python3 bench.py --steps ${STEPS:-300} --warmup 30 --no-cpu-baseline --no-routes --no-traffic --no-rocprof 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$var=$f', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"
done
