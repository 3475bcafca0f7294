#!/bin/bash
# Perf lab: UnitData size ops at 4096 x 1 Mi through each library variant named on the command line ("base" = the product library).
for v in "$@"; do
  if [ "$v" = base ]; then lib=""; else lib="tools/exp/libsdr_amd_$v.so"; fi
  echo "== $v"
  for op in ands xors rms; do
    SDR_AMD_LIB=$lib timeout -k 10 100 python tools/exp/run_unit_size.py 4096 $op default 3 2>&1 | grep " ms" | tail -n 1 || exit 1
  done
done
