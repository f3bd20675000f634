#!/bin/bash
# same-box A/B: the library in the tree, variants of it, jk_m4 -- three rounds each, interleaved
export PYTHONPATH=$PWD
mkdir -p gpurun_out
for round in 1 2 3; do
  for v in "" $@ m4; do
    unset NBX_LIB; export NBX_JK_M8=1
    if [ "$v" = "m4" ]; then export NBX_JK_M8=0; elif [ -n "$v" ]; then export NBX_LIB=$PWD/build/variants/libnbx_$v.so; fi
    r=$(timeout -k 10 240 python tools/dbg/m8_time.py ${M8N:-148} 40 2>&1 | tail -1)
    echo "round $round variant '$v': $r"
  done
done
