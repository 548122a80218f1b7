# LDS / issue counters of the FFT kernel on tools/fft_bench.py (optionally with env knobs given as arguments)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-a}; shift
env "$@" rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT --output-format csv -d $R/gpurun_out/pmcf_$TAG -o p -- python $R/tools/fft_bench.py > $R/gpurun_out/pmcf_$TAG.log 2>&1
env "$@" rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $R/gpurun_out/pmcg_$TAG -o p -- python $R/tools/fft_bench.py > $R/gpurun_out/pmcg_$TAG.log 2>&1
