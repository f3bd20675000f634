#!/bin/bash
# An alternative libnbx.so whose gemm.hip is compiled with extra flags (A/B measurements through NBX_LIB):
#   tools/build_gemm_variant.sh NAME "-DNBX_TN_DBG=5"   ->  scratch/libnbx_NAME.so   (bits: see gemm.hip)
set -e
cd "$(dirname "$0")/../nbed_amd/csrc"
mkdir -p ../../build/variants ../../scratch
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function $2 -c gemm.hip -o ../../build/variants/gemm_$1.o
OBJS=$(ls ../../build/nbx/*.o | grep -v "/gemm.o" | grep -v "jk_p8\|jk_s8\|jk_s4d")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $OBJS ../../build/variants/gemm_$1.o -o ../../scratch/libnbx_$1.so
echo built scratch/libnbx_$1.so
