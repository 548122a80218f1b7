# round 6, last GPU pass: counter traffic of the C4 bench command (separate FETCH_SIZE / WRITE_SIZE passes, kernel-trace only) -> profiles/<tag>_traffic_c4.json,
# then what the driver runs (tools/gpu_driver_like.sh): GPU tests, smoke, bench.py --gpus 1 --steps 20 --warmup 5
TAG=${1:-r06}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}_final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
	d=/tmp/pmc_c4_$c; rm -rf $d
	timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- python $R/bench.py --config c4 --no-cpu --steps 1 --warmup 0 > $O/pmc_c4_$c.log 2>&1
	f=$(find $d -name "*counter_collection.csv" | head -1)
	k=$([ $c = FETCH_SIZE ] && echo fetch || echo write); mkdir -p $R/gpurun_out/pmc_${k}_c4
	[ -n "$f" ] && cp "$f" $R/gpurun_out/pmc_${k}_c4/${k:0:1}_counter_collection.csv
done
cd $R && python tools/pmc_traffic_sum.py c4 $TAG > $O/traffic_c4.txt 2>&1; cp profiles/${TAG}_traffic_c4.json $O/ 2>/dev/null; cut -c1-170 $O/traffic_c4.txt | tail -14
bash tools/gpu_driver_like.sh ${TAG}_final
