# SQ counters of the Legendre kernels at the bench configuration (two passes: 8 SQ counters each)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; CFG=${1:-c3}
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmcl1 -o p -- python $R/bench.py --config $CFG --no-cpu --steps 1 --warmup 0 > $R/gpurun_out/pmcl1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/pmcl2 -o p -- python $R/bench.py --config $CFG --no-cpu --steps 1 --warmup 0 > $R/gpurun_out/pmcl2.log 2>&1
ls $R/gpurun_out/pmcl1 $R/gpurun_out/pmcl2
