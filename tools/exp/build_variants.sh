#!/bin/bash
# Perf lab: builds libsdr_amd variants with different bit-slice tuning constants.
# usage: tools/exp/build_variants.sh "256 3" "512 4" ...   -> tools/exp/libsdr_amd_<block>_<waves>.so
set -e
cd "$(dirname "$0")/../../sdr_amd/csrc"
make -s -j8
for cfg in "$@"; do
  set -- $cfg
  out=../../tools/exp/libsdr_amd_$1_$2.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
     -DSDRB_BS_BLOCK=$1 -DSDRB_BS_MIN_WAVES=$2 $3 -c -o /tmp/bs_var.o size_bitslice.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $(ls _obj/*.o | grep -v size_bitslice) /tmp/bs_var.o -L/opt/rocm/lib -lrocprofiler-sdk-roctx
  echo built $out
done
