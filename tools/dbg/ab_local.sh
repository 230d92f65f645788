# ab_multi.sh for the LocalSPFN workload (32 patches x 8192 points):  bash tools/dbg/ab_local.sh ".r2ref . . .r2ref"
for d in $1; do
  (cd $d && python3 bench.py --workload local --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-routes --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$d', round(d['ms_per_step'],4), round(d['value'],1))")
done
