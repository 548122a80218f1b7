# round 6: HBM traffic of the single-kernel engines (theta_line_kernel, ring_line_kernel) at C4 from the FETCH_SIZE / WRITE_SIZE counters
# (separate --pmc passes, kernel trace only), next to the stage chains they replace, on tools/chain_lab.py (8 maps 5400x10800, lmax 4000).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r06_line_traffic}; mkdir -p $O
for v in 1 0; do
	for c in FETCH_SIZE WRITE_SIZE; do
		rm -rf /tmp/pmc_$c$v
		env PXS_THETA_LINE=$v PXS_RING_LINE=$v timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c$v -o p -- python $R/tools/chain_lab.py c4 1 > $O/pmc_$c$v.log 2>&1
		f=$(find /tmp/pmc_$c$v -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${c}_line$v.csv
	done
done
rm -rf /tmp/kt_line; env timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_line -o p -- python $R/tools/chain_lab.py c4 3 > $O/kt.log 2>&1
f=$(find /tmp/kt_line -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/c4_chain_lab_kernel_stats.csv
cd $R && python tools/r06_line_traffic_sum.py $O | tee $O/summary.txt
