#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + separate PMC passes for the all-pairs top-k bench.
# usage: tools/profile_allpairs.sh <tag> [bench args]     -> gpurun_out/prof_ap_<tag>/...
set -u
tag=${1:-r01}; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_ap_$tag
mkdir -p $O
B="python3 $R/tools/bench_allpairs.py --check-rows 0 $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B --steps 3 > $O/kt.log 2>&1 || { echo kt failed; exit 1; }
for pass in "GRBM_GUI_ACTIVE" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | tr ' ' '+' | cut -c1-60)
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $O/pmc_$name -- $B --steps 1 > $O/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -3 $O/pmc_$name.log; exit 1; }
done
echo profile done
