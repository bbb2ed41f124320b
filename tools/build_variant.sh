#!/bin/bash
# tools/build_variant.sh <name> <file.hip|file.cpp> <extra hipcc flags...> -- build/libexp_<name>.so: the library with ONE kernel
# file recompiled with extra flags (experiment knobs are -D macros), the other objects taken from build/obj
set -e
NAME=$1; FILE=$2; shift 2
cd "$(dirname "$0")/../ryg_rans_amd/csrc"
make -s >/dev/null
O=../../build/obj
if [ -n "$MEASURE" ]; then make -s measure >/dev/null; O=../../build/obj_measure; set -- -DRANS_AMD_MEASURE "$@"; fi
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-const-variable --offload-arch=gfx950 "$@" -c $FILE -o $O/variant_$NAME.o
OBJS=""
for f in decode_wave decode_dual decode_groups encode_groups encode_wave encode_adaptive lanes container_kernels dispatch api model container; do
  if [ "$f.hip" = "$FILE" ] || [ "$f.cpp" = "$FILE" ]; then OBJS="$OBJS $O/variant_$NAME.o"; else OBJS="$OBJS $O/$f.o"; fi
done
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o ../../build/libexp_$NAME.so $OBJS -Wl,-rpath,/opt/rocm/lib
echo built build/libexp_$NAME.so
