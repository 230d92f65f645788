# interleaved A/B of two source trees on ONE box (the boxes of the pool differ by +-1.5 %):
#   bash tools/dbg/ab_trees.sh <other-tree> "a b b a a b"      (a = the other tree, b = this one; both built)
other=$1; seq=${2:-"a b b a a b"}
for v in $seq; do
  if [ $v = a ]; then d=$other; else d=.; fi
  (cd $d && python3 bench.py --steps ${STEPS:-500} --warmup 20 --no-cpu-baseline --no-routes --no-traffic $(grep -q no-rocprof bench.py && echo --no-rocprof) 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', '$d', round(d['ms_per_step'],4), round(d['value'],1))")
done
