#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace (with the entry points' named ranges) + separate PMC passes for the headline bench.
# usage: tools/profile_bench.sh <tag>      -> gpurun_out/prof_<tag>/...
set -u
tag=${1:-r01}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_$tag
mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-secondary"
timeout -k 10 300 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $O/kt -- $B --steps 20 --warmup 3 > $O/kt.log 2>&1 || { echo kt failed; exit 1; }
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" \
            "SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  name=$(echo $pass | tr ' ' '+')
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $O/pmc_$name -- $B --steps 3 --warmup 1 --no-verify > $O/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -3 $O/pmc_$name.log; exit 1; }
done
# calibration of WRITE_SIZE on a known byte count with the same store shape (dword per lane)
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cal_WRITE_SIZE -- $R/tools/exp/_bin/lab_store cal > $O/cal_write.log 2>&1 || { echo cal failed; exit 1; }
echo profile done
