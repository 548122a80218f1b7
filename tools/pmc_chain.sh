#!/bin/bash
# SQ counters + kernel trace of the chain stages on tools/chain_lab.py (scratch data).  usage: pmc_chain.sh <tag> [cfg] [ENV=.. ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-a}; CFG=${2:-c3}; shift; shift
O=$R/gpurun_out/pmcc_$TAG; mkdir -p $O
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python $R/tools/chain_lab.py $CFG 2 > $O/kt.log 2>&1
env "$@" rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p1 -o p -- python $R/tools/chain_lab.py $CFG 1 > $O/p1.log 2>&1
env "$@" rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM --output-format csv -d $O/p2 -o p -- python $R/tools/chain_lab.py $CFG 1 > $O/p2.log 2>&1
python $R/tools/pmc_chain_sum.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
