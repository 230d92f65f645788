#!/bin/bash
# One box's contribution to the packed-fp32 hunt (round 5): does the PRODUCT's packed sampling kernel lose an update beside
# mlp_wgrad on THIS box?  (tools/dbg/pk_repro.py: ~25 s.)  Only if it does, the scratch trees of tools/dbg/fps_ablate.py that
# travelled with the snapshot (.tdiag .tabl .tpairs .tstaged) run their experiments on the same box, because the next call
# may get another one.  Appends to gpurun_out/r05/pk_boxes.txt.
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/pk_boxes.txt
echo "==== $(date -u +%FT%TZ) $(hostname)" >> $OUT
(rocm-smi --showclocks --showpower --showdriverversion --showfwinfo 2>/dev/null | grep -iE "sclk|power|driver|SMC|MEC |PM4" | head -12) >> $OUT
python tools/dbg/pk_repro.py ${1:-6} small,large >> $OUT 2>&1
python tools/dbg/pk_repro.py ${1:-6} beside2 2>&1 | grep -v "^device\|amdgpu.ids" >> $OUT
if grep -A6 "==== " $OUT | tail -6 | grep -E " +[1-9][0-9]* with different" > /dev/null; then
  echo "THIS BOX REPRODUCES: running the ablations" >> $OUT
  for v in diag; do [ -d .t$v ] && (cd .t$v && timeout 150 python tools/dbg/pk_aggressor.py 30 diag) >> $OUT 2>&1; done
  [ -d .tabl ] && (cd .tabl && timeout 250 python tools/dbg/pk_aggressor.py 8 ablate) >> $OUT 2>&1
  for v in pairs staged; do [ -d .t$v ] && (cd .t$v && timeout 120 python tools/dbg/pk_aggressor.py 20 ablate0) >> $OUT 2>&1; done
fi
tail -n 24 $OUT
