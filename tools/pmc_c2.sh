cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PXS_K_SYN0=8 PXS_K_ANA0=8 PXS_K_SYNS=3 PXS_K_ANAS=3
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc1 -o pmc1 -- python $R/bench.py --config c2 --no-cpu --steps 1 --warmup 0 > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc2 -o pmc2 -- python $R/bench.py --config c2 --no-cpu --steps 1 --warmup 0 > $R/gpurun_out/pmc2.log 2>&1
ls $R/gpurun_out/pmc1 $R/gpurun_out/pmc2
