# rocprofv3 kernel stats + FETCH/WRITE counter passes of the C3 bench command only (the tail of tools/profile_round.sh)
TAG=${1:-r05c}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o c3 -- env PXS_BENCH_NO_WEIGHTS=1 python $R/bench.py --no-cpu --no-legs --steps 2 > $O/prof_c3.log 2>&1
f=$(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/c3_kernel_stats.csv && head -12 "$f" | cut -c1-150
tail -c 600 $O/prof_c3.log | grep -o '"stage_ms_per_step": {[^}]*}'
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- env PXS_BENCH_NO_WEIGHTS=1 python $R/bench.py --no-cpu --no-legs --steps 1 --warmup 0 > $O/pmc_fetch.log 2>&1
f=$(find /tmp/pmc_f -name "*counter_collection.csv" | head -1); [ -n "$f" ] && mkdir -p $R/gpurun_out/pmc_fetch_c3 && cp "$f" $R/gpurun_out/pmc_fetch_c3/f_counter_collection.csv
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- env PXS_BENCH_NO_WEIGHTS=1 python $R/bench.py --no-cpu --no-legs --steps 1 --warmup 0 > $O/pmc_write.log 2>&1
f=$(find /tmp/pmc_w -name "*counter_collection.csv" | head -1); [ -n "$f" ] && mkdir -p $R/gpurun_out/pmc_write_c3 && cp "$f" $R/gpurun_out/pmc_write_c3/w_counter_collection.csv
cd $R && python tools/pmc_traffic_sum.py c3 $TAG > $O/traffic_c3.txt 2>&1; cp profiles/${TAG}_traffic_c3.json $O/ 2>/dev/null; tail -3 $O/traffic_c3.txt | cut -c1-160
rm -rf $R/gpurun_out/pmc_fetch_c3 $R/gpurun_out/pmc_write_c3
