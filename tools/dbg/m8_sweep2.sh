#!/bin/bash
# band-order jk_m8: cost of a tile / of a column group in the split (host-side switches)
export PYTHONPATH=$PWD
for tg in $@; do
  t=${tg%%:*}; g=${tg##*:}
  r=$(NBX_M8_TC=$t NBX_M8_GC=$g timeout -k 10 240 python tools/dbg/m8_time.py 148 40 2>&1 | tail -1)
  echo "tc $t gc $g: $r"
done
NBX_JK_M8=0 timeout -k 10 240 python tools/dbg/m8_time.py 148 40 2>&1 | tail -1
