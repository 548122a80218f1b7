#!/bin/bash
# SQ counters of the Legendre kernels (incl. the FP64-MFMA analysis) on a batched bench command; separate --pmc passes with kernel-trace only.
# usage: pmc_mm.sh <tag> [bench args...]     default: --config c4 --no-cpu --steps 1 --warmup 1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-a}; shift; O=$R/gpurun_out/pmcmm_$TAG; mkdir -p $O
ARGS=${@:---config c4 --no-cpu --steps 1 --warmup 1}
CMD="python $R/bench.py $ARGS"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- $CMD > $O/kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p1 -o p -- $CMD > $O/p1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p2 -o p -- $CMD > $O/p2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p3 -o p -- $CMD > $O/p3.log 2>&1
python $R/tools/pmc_leg_sum.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
