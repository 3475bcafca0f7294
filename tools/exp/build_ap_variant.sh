#!/bin/bash
# Perf lab: builds a libsdr_amd variant whose all-pairs kernel is the lab copy (tools/exp/allpairs_mfma_lab.hip: the product kernel plus its
# timing-experiment switches SDRB_AP_EXP_*, SDRB_AP_FORCE_SPLITS) compiled with extra -D flags.
# usage: tools/exp/build_ap_variant.sh <name> [-DFLAG ...]   -> tools/exp/libsdr_amd_ap_<name>.so
set -e
cd "$(dirname "$0")/../../sdr_amd/csrc"
make -s -j8
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "$@" \
   -I. -c -o /tmp/ap_var.o ../../tools/exp/allpairs_mfma_lab.hip
objs=$(ls _obj/*.o | grep -v allpairs_mfma)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/libsdr_amd_ap_$name.so $objs /tmp/ap_var.o -L/opt/rocm/lib -lrocprofiler-sdk-roctx
echo built tools/exp/libsdr_amd_ap_$name.so
