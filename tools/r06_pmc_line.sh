#!/bin/bash
# SQ counters of the single-kernel line engines (theta_line_kernel, ring_line_kernel) and, beside them, of the chain stages they replace: tools/chain_lab.py c4
# (8 maps per launch) under separate --pmc passes with kernel-trace only, once with the engines and once with PXS_THETA_LINE=0 PXS_RING_LINE=0.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_pmc_line; mkdir -p $O
for v in 1 0; do
	CMD="env PXS_THETA_LINE=$v PXS_RING_LINE=$v python $R/tools/chain_lab.py c4 2"
	timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt$v -o p -- $CMD > $O/kt$v.log 2>&1
	timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p1_$v -o p -- $CMD > $O/p1_$v.log 2>&1
	timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p2_$v -o p -- $CMD > $O/p2_$v.log 2>&1
	timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT --output-format csv -d $O/p3_$v -o p -- $CMD > $O/p3_$v.log 2>&1
done
python $R/tools/r06_pmc_line_sum.py $O > $O/summary.txt 2>&1; cat $O/summary.txt
