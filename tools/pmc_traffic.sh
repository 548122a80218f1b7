# HBM traffic of the Legendre kernels (separate --pmc passes; no trace domains besides kernel-trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; CFG=${1:-c3}
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$CFG -o f -- python $R/bench.py --config $CFG --no-cpu --steps 1 --warmup 0 > $R/gpurun_out/pmc_fetch_$CFG.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$CFG -o w -- python $R/bench.py --config $CFG --no-cpu --steps 1 --warmup 0 > $R/gpurun_out/pmc_write_$CFG.log 2>&1
ls $R/gpurun_out/pmc_fetch_$CFG $R/gpurun_out/pmc_write_$CFG
