#!/bin/bash
# Perf lab: product objects + tools/exp/lane_queue_lab.hip -> tools/exp/libsdr_amd_queue.so
# usage: tools/exp/build_queue_lab.sh [-DLQ_CHUNK=16 ...]; run a tool against it with SDR_AMD_LIB=tools/exp/libsdr_amd_queue.so
set -e
cd "$(dirname "$0")/../../sdr_amd/csrc"
make -s -j8
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I. -I../../include "$@" \
   -c -o /tmp/lane_queue_lab.o ../../tools/exp/lane_queue_lab.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/libsdr_amd_queue.so _obj/*.o /tmp/lane_queue_lab.o -L/opt/rocm/lib -lrocprofiler-sdk-roctx
echo built tools/exp/libsdr_amd_queue.so
