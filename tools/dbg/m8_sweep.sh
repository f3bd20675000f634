#!/bin/bash
# jk_m8's split of the tile sequence under different cost models (host-side switches: no rebuild)
export PYTHONPATH=$PWD
mkdir -p gpurun_out
for ft in $@; do
  f=${ft%%:*}; t=${ft##*:}
  NBX_M8_TC=$t NBX_JK_M8=1 timeout -k 10 240 python tools/dbg/m8_time.py 148 40 > gpurun_out/m8_sweep_${f}_$t.log 2>&1 || { echo "run $ft failed"; tail -5 gpurun_out/m8_sweep_${f}_$t.log; exit 1; }
  echo "floor $f tc $t: $(grep -c 'reproducible True' gpurun_out/m8_sweep_${f}_$t.log) $(tail -2 gpurun_out/m8_sweep_${f}_$t.log | tr '\n' ' ')"
done
timeout -k 10 240 python tools/dbg/m8_time.py 148 40 2>&1 | tail -1
