# interleaved A/B of two built trees on the DRIVER's command (python bench.py --gpus 1 --steps 20 --warmup 5): what a change
# does to the fill of the pipeline from an idle GPU, which a 300-step run does not see.   bash tools/dbg/short_ab.sh   (a = .r3ref)
for v in a b b a a b b a; do
  if [ $v = a ]; then d=.r3ref; else d=.; fi
  (cd $d && python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-routes --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', round(d['ms_per_step'],4))")
done
