#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + separate PMC passes for the element ops of configs 3 and 4.
# usage: tools/profile_ops.sh <tag>      -> gpurun_out/prof_ops_<tag>/<case>/...
set -u
tag=${1:-r02}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_ops_$tag
mkdir -p $O
for spec in "c3 1 ande" "c3 16 ande" "c3 1 ore" "c3 16 ore" "c4 64 ande" "c3u 1 ande"; do
  set -- $spec
  name=$1_q$2_$3
  P="python3 $R/tools/exp/run_one_op.py $1 $2 $3"
  mkdir -p $O/$name
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name/kt -- $P > $O/$name/kt.log 2>&1 || { echo "$name kt failed"; tail -n 5 $O/$name/kt.log; exit 1; }
  for pass in "FETCH_SIZE" "WRITE_SIZE" \
              "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY"; do
    pn=$(echo $pass | tr ' ' '+' | cut -c1-24)
    timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $O/$name/pmc_$pn -- $P > $O/$name/pmc_$pn.log 2>&1 || { echo "$name pmc $pn failed"; tail -n 3 $O/$name/pmc_$pn.log; exit 1; }
  done
  echo "$name done"
done
echo ops profile done
