#!/bin/bash
# Perf lab: SQ counter passes over one payload-dependent size op (run_unit_size.py).  usage: pmc_unit_size.sh <tag> [args of run_unit_size.py]
set -u
tag=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_unit_$tag
mkdir -p $O
P="python3 $R/tools/exp/run_unit_size.py $*"
n=0
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" \
            "SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
            "GRBM_GUI_ACTIVE"; do
  n=$((n+1))
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $O/p$n -- $P > $O/p$n.log 2>&1 || { echo "pass $n failed"; tail -n 3 $O/p$n.log; exit 1; }
done
echo pmc done
