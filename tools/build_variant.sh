#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags]  -> build/variants/libx265_amd_<name>.so
set -e
name=$1; shift
mkdir -p build/variants/obj_$name
for f in x265_amd/csrc/*.hip; do
  extra=""; [ "$(basename $f)" = tu_chain.hip ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $extra "$@" -c $f -o build/variants/obj_$name/$(basename $f .hip).o &
done
for f in x265_amd/csrc/host/*.cpp; do
  g++ -O2 -std=c++17 -fPIC -pthread $HOSTFLAGS -c $f -o build/variants/obj_$name/$(basename $f .cpp).o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC build/variants/obj_$name/*.o -o build/variants/libx265_amd_$name.so -pthread
echo built build/variants/libx265_amd_$name.so
