#!/bin/bash
# An EXPERIMENTAL libnbx.so whose J/K kernel is the ROUND-3 jk_m4 translation unit with its ablation switches
# (tools/variants/jk_m4_r03_ablations.hip: NBX_M4_NO_WALK, NBX_M4_NO_STAGE, NBX_M4_ROWS_*, NBX_M4_CLOCKS, ... -- several of
# them give wrong results on purpose), compiled with extra flags.  The production csrc/jk_m4.hip carries none of them.
#   tools/build_m4_variant.sh NAME "-DNBX_M4_NO_WALK"   ->  scratch/libnbx_m4_NAME.so     (run with NBX_LIB=... NBX_JK_M4=1)
set -e
cd "$(dirname "$0")/../nbed_amd/csrc"
make EXPERIMENTAL=1 TARGET=../../scratch/libnbx_exp_base.so -j8 > /dev/null
mkdir -p ../../build/variants ../../scratch
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function -DNBX_EXPERIMENTAL $2 -I. -c ../../tools/variants/jk_m4_r03_ablations.hip -o ../../build/variants/jk_m4_$1.o
OBJS=$(ls ../../build/nbx_experimental/*.o | grep -v "/jk_m4.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $OBJS ../../build/variants/jk_m4_$1.o -o ../../scratch/libnbx_m4_$1.so
echo built scratch/libnbx_m4_$1.so
