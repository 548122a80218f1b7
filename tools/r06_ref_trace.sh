# kernel trace of the reference's benchmark shape (900x1800, lmax 750): how much of a round trip is kernel time, how much the gaps between ~40 launches
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_ref_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/reftrace -o ref -- python $R/bench.py --config ref --no-cpu --steps 40 > $O/run.log 2>&1
f=$(find /tmp/reftrace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp "$f" $O/ref_kernel_trace.csv
cd $R && python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/r06_ref_trace/ref_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pxs = [r for r in rows if "pxs::" in r["Kernel_Name"]]
# the last 40 round trips: take the last 60 % of the pxs launches
n = len(pxs); tail = pxs[int(n*0.5):]
t0, t1 = int(tail[0]["Start_Timestamp"]), int(tail[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in tail)
gaps = [int(b["Start_Timestamp"])-int(a["End_Timestamp"]) for a, b in zip(tail[:-1], tail[1:])]
print("launches in window %d, window %.3f ms, kernel time %.3f ms (%.1f %%), mean kernel %.2f us, median gap %.2f us, mean gap %.2f us" % (len(tail), (t1-t0)/1e6, busy/1e6, 100.0*busy/(t1-t0), busy/len(tail)/1e3, sorted(gaps)[len(gaps)//2]/1e3, sum(gaps)/len(gaps)/1e3))
cnt = collections.Counter(); dur = collections.Counter()
for r in tail:
	k = r["Kernel_Name"].split("(")[0][:70]; cnt[k] += 1; dur[k] += int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for k, v in dur.most_common(30): print("  %-72s n=%4d  mean %.2f us" % (k, cnt[k], v/cnt[k]/1e3))
PY
