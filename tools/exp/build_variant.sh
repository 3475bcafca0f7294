#!/bin/bash
# Perf lab: builds a variant of the engine in which ONE source file is compiled with extra -D flags.
# usage: tools/exp/build_variant.sh <source.hip> <name> [-DFLAG ...]   -> tools/exp/libsdr_amd_<name>.so
# Run a tool against it with SDR_AMD_LIB=tools/exp/libsdr_amd_<name>.so.
set -e
cd "$(dirname "$0")/../../sdr_amd/csrc"
make -s -j8
src=$1; name=$2; shift 2
base=$(basename "$src" .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "$@" \
   -c -o /tmp/var_$name.o "$src"
objs=$(ls _obj/*.o | grep -v "_obj/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/libsdr_amd_$name.so $objs /tmp/var_$name.o -L/opt/rocm/lib -lrocprofiler-sdk-roctx
echo built tools/exp/libsdr_amd_$name.so
