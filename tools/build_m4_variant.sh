#!/bin/bash
# An EXPERIMENTAL libnbx.so whose jk_m4.hip is compiled with extra flags (NBX_M4_NO_WALK, NBX_M4_NO_STAGE, ...):
#   tools/build_m4_variant.sh NAME "-DNBX_M4_NO_WALK"   ->  scratch/libnbx_m4_NAME.so     (run with NBX_LIB=... NBX_JK_M4=1)
set -e
cd "$(dirname "$0")/../nbed_amd/csrc"
make EXPERIMENTAL=1 TARGET=../../scratch/libnbx_exp_base.so -j8 > /dev/null
mkdir -p ../../build/variants ../../scratch
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function -DNBX_EXPERIMENTAL $2 -c jk_m4.hip -o ../../build/variants/jk_m4_$1.o
OBJS=$(ls ../../build/nbx_experimental/*.o | grep -v "/jk_m4.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $OBJS ../../build/variants/jk_m4_$1.o -o ../../scratch/libnbx_m4_$1.so
echo built scratch/libnbx_m4_$1.so
