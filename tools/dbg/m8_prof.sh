#!/bin/bash
export PYTHONPATH=$PWD
mkdir -p gpurun_out/m8prof
cd /tmp && export TMPDIR=/tmp
NBX_JK_M8=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/m8prof -o m8 -- python3 $GRAFT_REPO_ROOT/tools/dbg/m8_time.py 148 20 > $GRAFT_REPO_ROOT/gpurun_out/m8prof/run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/m8prof -name "*kernel_stats.csv" | head -1)
head -12 $f | cut -c1-160
