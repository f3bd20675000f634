#!/bin/bash
# tools/build_m8_variant.sh NAME [-DNBX_M8_...=..]: libnbx with jk_m8.hip (M8_SRC=tools/variants/jk_m8_r04_instrumented.hip: the copy
# with the ablation bits NBX_M8_ABL and the time stamps) compiled under the given switches -> build/variants/libnbx_NAME.so
# (A/B measurements: NBX_LIB=build/variants/libnbx_NAME.so python ...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants
make -C nbed_amd/csrc -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Inbed_amd/csrc -Wno-unused-function -Wno-inline-asm "$@" -c ${M8_SRC:-nbed_amd/csrc/jk_m8.hip} -o build/variants/jk_m8_$name.o 2> /dev/null
objs=$(ls build/nbx/*.o | grep -v "/jk_m8.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $objs build/variants/jk_m8_$name.o -o build/variants/libnbx_$name.so
echo build/variants/libnbx_$name.so
