# round 6: analysis ring FFT (map -> leg), single kernel with 32-byte pieces merged in the XCD's L2 against the two-stage chain:
# time (tools/chain_lab.py, 8 maps 5400x10800) and the WRITE_SIZE / FETCH_SIZE counters of both
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r06_ring_ana}; mkdir -p $O
cd $R; python -m pytest tests/test_theta_line.py -x -q -m gpu -k "ring_line_ana" 2>&1 | tail -1
for v in 1 0; do PXS_RING_LINE_ANA=$v python tools/chain_lab.py c4 5 2>&1 | tail -1 | cut -c1-200 | sed "s/^/ring_line_ana=$v /"; done | tee $O/chain_lab.txt
cd /tmp
for v in 1 0; do for c in FETCH_SIZE WRITE_SIZE; do
	rm -rf /tmp/pa_$c$v; env PXS_RING_LINE_ANA=$v timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pa_$c$v -o p -- python $R/tools/chain_lab.py c4 1 > /dev/null 2>&1
	f=$(find /tmp/pa_$c$v -name "*counter_collection.csv" | head -1)
	python3 - "$f" $c $v <<'PY'
import csv, sys
f, c, v = sys.argv[1:4]
for row in csv.DictReader(open(f)):
	if row["Counter_Name"] == c and ("ring_line_ana" in row["Kernel_Name"] or "StRingA" in row["Kernel_Name"]):
		k = "ring_line_ana" if "ring_line_ana" in row["Kernel_Name"] else ("StRingA1" if "StRingA1" in row["Kernel_Name"] else "StRingA2")
		print("ring_line_ana=%s %-10s %-14s %9.1f MB %8.1f us" % (v, c, k, float(row["Counter_Value"])*1024/1e6, (int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))/1e3))
PY
done; done | tee $O/counters.txt
