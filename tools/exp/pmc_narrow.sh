#!/bin/bash
# Lab: counters of the narrow-count kernels (run on the GPU box): tools/exp/pmc_narrow.sh <op> <u8|u16|u32>
set -e
op=$1; dt=$2
out=$PWD/gpurun_out/pmc_narrow_${op}_${dt}
mkdir -p $out
repo=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $repo/tools/exp/run_narrow.py $op $dt > $out/kt.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY --output-format csv -d $out/sq -o sq -- python3 $repo/tools/exp/run_narrow.py $op $dt > $out/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum --output-format csv -d $out/tcc -o tcc -- python3 $repo/tools/exp/run_narrow.py $op $dt > $out/tcc.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/ws -o ws -- python3 $repo/tools/exp/run_narrow.py $op $dt > $out/ws.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_WAIT_ANY --output-format csv -d $out/sq2 -o sq2 -- python3 $repo/tools/exp/run_narrow.py $op $dt > $out/sq2.log 2>&1
echo done
