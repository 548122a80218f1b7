# On the GPU box: GPU tests, the bench lines of the round, rocprofv3 kernel stats and the FETCH/WRITE counter passes of the same
# bench command (separate --pmc passes, kernel-trace only).  Everything guarded by timeouts; results under gpurun_out/<tag>/.
TAG=${1:-r05}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
PXS_REQUIRE_FULL=1 timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; tail -1 $O/bench_c3.err
timeout 600 python bench.py --config c2 --no-cpu --steps 10 --warmup 2 > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.err
timeout 900 python bench.py --config c4 --no-cpu > $O/bench_c4.json 2> $O/bench_c4.err; tail -1 $O/bench_c4.err
timeout 900 python bench.py --config c5 --no-cpu --steps 1 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err; tail -1 $O/bench_c5.err
timeout 300 python bench.py --config ref --no-cpu --steps 40 > $O/bench_ref.json 2> $O/bench_ref.err; tail -1 $O/bench_ref.err
PXS_BENCH_FORCE_PG=1 timeout 600 python bench.py --config c4 --no-cpu > $O/bench_c4_forcepg.json 2> $O/bench_c4_forcepg.err; tail -1 $O/bench_c4_forcepg.err
# the N-rank launch path end to end on this one-GPU box: two ranks share the device, the gather goes through gloo (times mean nothing)
PXS_BENCH_BACKEND=gloo PXS_BENCH_NBATCH=8 timeout 600 python bench.py --gpus 2 --config c4 --no-cpu --steps 1 --warmup 1 > $O/rehearsal_c4_2ranks_gloo.json 2> $O/rehearsal_c4.err; tail -2 $O/rehearsal_c4.err
PXS_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --config c2 --no-cpu --no-legs --steps 2 --warmup 1 > $O/rehearsal_c2_2ranks_gloo.json 2> $O/rehearsal_c2.err; tail -1 $O/rehearsal_c2.err
timeout 300 python tools/adj_bench.py > $O/adj_bench.log 2>&1; tail -6 $O/adj_bench.log
timeout 600 python tools/host_bench.py c3 3 > $O/host_bench.log 2>&1; tail -1 $O/host_bench.log
timeout 300 python tools/band_bench.py > $O/band_bench.log 2>&1; tail -4 $O/band_bench.log
for cfg in c3 c4; do python tools/chain_lab.py $cfg 3 >> $O/chain_lab.jsonl 2>> $O/lab.err; done; cat $O/chain_lab.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o c3 -- env PXS_BENCH_NO_WEIGHTS=1 python $R/bench.py --no-cpu --no-legs --steps 2 > $O/prof_c3.log 2>&1
f=$(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/c3_kernel_stats.csv && head -16 "$f" | cut -c1-150
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -o c4 -- python $R/bench.py --no-cpu --config c4 --steps 1 > $O/prof_c4.log 2>&1
f=$(find /tmp/prof_c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/c4_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -o c5 -- env PXS_BENCH_NREAL=10 python $R/bench.py --no-cpu --config c5 --steps 1 > $O/prof_c5.log 2>&1
f=$(find /tmp/prof_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/c5_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- env PXS_BENCH_NO_WEIGHTS=1 python $R/bench.py --no-cpu --no-legs --steps 1 --warmup 0 > $O/pmc_fetch.log 2>&1
f=$(find /tmp/pmc_f -name "*counter_collection.csv" | head -1); [ -n "$f" ] && mkdir -p $R/gpurun_out/pmc_fetch_c3 && cp "$f" $R/gpurun_out/pmc_fetch_c3/f_counter_collection.csv
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- env PXS_BENCH_NO_WEIGHTS=1 python $R/bench.py --no-cpu --no-legs --steps 1 --warmup 0 > $O/pmc_write.log 2>&1
f=$(find /tmp/pmc_w -name "*counter_collection.csv" | head -1); [ -n "$f" ] && mkdir -p $R/gpurun_out/pmc_write_c3 && cp "$f" $R/gpurun_out/pmc_write_c3/w_counter_collection.csv
cd $R && python tools/pmc_traffic_sum.py c3 $TAG > $O/traffic_c3.txt 2>&1; cp profiles/${TAG}_traffic_c3.json $O/ 2>/dev/null; tail -5 $O/traffic_c3.txt | cut -c1-160
