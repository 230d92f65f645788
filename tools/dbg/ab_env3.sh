# one box, interleaved: each argument "label dir [ENV=.. ENV=..]" -> ms per step (bench.py without its side measurements),
# skipped optimizer steps, last loss
#   bash tools/dbg/ab_env3.sh "a .r6c" "b ." "c . CPFN_OUTPUT_JOIN=0" ...   (REPS rounds, default 2: forward then reversed order)
run() {
  l=$1; d=$2; shift 2
  (cd $d && env "$@" python3 bench.py --steps ${STEPS:-300} --warmup 20 --no-cpu-baseline --no-routes --no-traffic --no-rocprof 2>/dev/null |
   L="$l $d $*" python3 -c 'import sys, json, os
d = json.loads(sys.stdin.readlines()[-1])
print(os.environ["L"], round(d["ms_per_step"], 4), d.get("skipped_steps"), d.get("loss_last"))')
}
for r in $(seq ${REPS:-2}); do
  for spec in "$@"; do run $spec; done
  for ((i=$#; i>=1; i--)); do run ${!i}; done
done
