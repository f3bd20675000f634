#!/bin/bash
# correctness + timing of jk_m8 (and variants of it under build/variants) against jk_m4, each in its own process
export PYTHONPATH=$PWD
mkdir -p gpurun_out
for v in "" $@; do
  if [ -n "$v" ]; then export NBX_LIB=$PWD/build/variants/libnbx_$v.so; fi
  NBX_JK_M8=1 timeout -k 10 240 python tools/dbg/m8_time.py 148 50 > gpurun_out/m8_time_$v.log 2>&1 || { echo "m8 run $v failed"; tail -20 gpurun_out/m8_time_$v.log; exit 1; }
  echo "== variant '$v'"; tail -3 gpurun_out/m8_time_$v.log
done
unset NBX_LIB
timeout -k 10 240 python tools/dbg/m8_time.py 148 50 > gpurun_out/m4_time.log 2>&1
echo "== m4"; tail -2 gpurun_out/m4_time.log
