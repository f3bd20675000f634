#!/bin/bash
# m8_sweepv.sh VARIANT tc...: variant library under different tile costs
export PYTHONPATH=$PWD
v=$1; shift
export NBX_LIB=$PWD/build/variants/libnbx_$v.so
for t in $@; do
  NBX_M8_TC=$t NBX_JK_M8=1 timeout -k 10 240 python tools/dbg/m8_time.py 148 40 > gpurun_out/m8_sweepv.log 2>&1 || { echo failed; tail -5 gpurun_out/m8_sweepv.log; exit 1; }
  echo "$v tc $t: $(tail -1 gpurun_out/m8_sweepv.log)"
done
