# On the GPU box: GPU tests, the bench lines of the round, rocprofv3 kernel stats and the FETCH/WRITE counter passes of the same
# bench command (separate --pmc passes, kernel-trace only).  Everything guarded by timeouts; results under gpurun_out/<tag>/.
TAG=${1:-r02}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; tail -1 $O/bench_c3.err
timeout 600 python bench.py --config c2 --no-cpu > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.err
timeout 900 python bench.py --config c4 --no-cpu > $O/bench_c4.json 2> $O/bench_c4.err; tail -1 $O/bench_c4.err
timeout 300 python bench.py --config ref --no-cpu --steps 40 > $O/bench_ref.json 2> $O/bench_ref.err; tail -1 $O/bench_ref.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o c3 -- python $R/bench.py --no-cpu --steps 2 > $O/prof_c3.log 2>&1
f=$(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/c3_kernel_stats.csv && head -16 "$f" | cut -c1-150
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fft -o fft -- python $R/tools/fft_bench2.py > $O/prof_fft.log 2>&1
f=$(find /tmp/prof_fft -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/fft_kernel_stats.csv
grep GB/s $O/prof_fft.log | tail -12
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python $R/bench.py --no-cpu --steps 1 --warmup 0 > $O/pmc_fetch.log 2>&1
f=$(find /tmp/pmc_f -name "*counter_collection.csv" | head -1); [ -n "$f" ] && mkdir -p $R/gpurun_out/pmc_fetch_c3 && cp "$f" $R/gpurun_out/pmc_fetch_c3/f_counter_collection.csv
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python $R/bench.py --no-cpu --steps 1 --warmup 0 > $O/pmc_write.log 2>&1
f=$(find /tmp/pmc_w -name "*counter_collection.csv" | head -1); [ -n "$f" ] && mkdir -p $R/gpurun_out/pmc_write_c3 && cp "$f" $R/gpurun_out/pmc_write_c3/w_counter_collection.csv
ls -la $R/gpurun_out/pmc_fetch_c3 $R/gpurun_out/pmc_write_c3 2>&1 | tail -4
