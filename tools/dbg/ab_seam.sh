# same-box A/B of the BatchNorm seams: a = reference tree (.r6ref), b = this tree, c = this tree without the arena fill (timing only)
for v in ${1:-a b c c b a a b c}; do
  d=.; e=""
  if [ $v = a ]; then d=.r6ref; fi
  if [ $v = c ]; then e="CPFN_SEAM_NOFILL=1"; fi
  (cd $d && env $e python3 bench.py --steps ${STEPS:-400} --warmup 20 --no-cpu-baseline --no-routes --no-traffic --no-rocprof 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', round(d['ms_per_step'],4), round(d['value'],1))")
done
