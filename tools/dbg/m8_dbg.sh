#!/bin/bash
export PYTHONPATH=$PWD
mkdir -p gpurun_out
export NBX_LIB=$PWD/build/variants/libnbx_$1.so
NBX_JK_M8=1 timeout -k 10 240 python tools/dbg/m8_time.py 148 30 > gpurun_out/m8_dbg_$1.log 2>&1
grep -c m8dbg gpurun_out/m8_dbg_$1.log; tail -2 gpurun_out/m8_dbg_$1.log
