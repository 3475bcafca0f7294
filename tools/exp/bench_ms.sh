#!/bin/bash
# Perf lab: headline bench through each library variant named on the command line ("base" = the product library).
for v in "$@"; do
  if [ "$v" = base ]; then lib=""; else lib="tools/exp/libsdr_amd_$v.so"; fi
  SDR_AMD_LIB=$lib timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --steps 30 --warmup 5 > /tmp/bench_$v.json || exit 1
  python - "$v" /tmp/bench_$v.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
print(sys.argv[1], "ms/step %.4f kernel %.4f frac %.4f parity %s" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["parity_checked"]), flush=True)
PY
done
