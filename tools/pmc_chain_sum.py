"""Summarises tools/pmc_chain.sh: per kernel name the average duration and the SQ counters (sum over launches / launches)."""
import sys, os, glob, csv, collections, re
O = sys.argv[1]
def short(n):
	m = re.search(r"chain_kernel<pxs::(\w+(?:<\d+>)?)", n)
	return m.group(1) if m else n[:40]
dur = collections.defaultdict(list)
for f in glob.glob(O+"/kt/**/*kernel_trace.csv", recursive=True):
	for r in csv.DictReader(open(f)):
		dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
cnt = collections.defaultdict(lambda: collections.defaultdict(float)); nl = collections.defaultdict(lambda: collections.defaultdict(int))
for p in ("p1", "p2"):
	for f in glob.glob(O+"/"+p+"/**/*counter_collection.csv", recursive=True):
		for r in csv.DictReader(open(f)):
			k = short(r["Kernel_Name"]); c = r["Counter_Name"]
			cnt[k][c] += float(r["Counter_Value"]); nl[k][c] += 1
print("kernel                launches  avg_ms  total_ms")
for k in sorted(dur, key=lambda k: -sum(dur[k])):
	print("%-22s %6d %8.3f %9.2f" % (k, len(dur[k]), sum(dur[k])/len(dur[k]), sum(dur[k])))
for k in sorted(cnt, key=lambda k: -sum(dur.get(k, [0]))):
	c = {n: cnt[k][n]/max(nl[k][n], 1) for n in cnt[k]}
	g = lambda n: c.get(n, float("nan"))
	print("\n%s (per launch)" % k)
	print("  " + "  ".join("%s=%.4g" % (n, v) for n, v in sorted(c.items())))
	# derived: fractions of wave-cycles / busy cycles
	wc = g("SQ_WAVE_CYCLES"); bc = g("SQ_BUSY_CYCLES")
	print("  valu_active/busy=%.3f  inst_any_active/busy=%.3f  wait_any/wave_cycles=%.3f  wait_inst_any/wave_cycles=%.3f  lds_active/busy=%.3f  lds_conflict/lds_active=%.3f  valu_per_wave=%.1f lds_per_wave=%.1f" % (
		g("SQ_ACTIVE_INST_VALU")/bc, g("SQ_ACTIVE_INST_ANY")/bc, g("SQ_WAIT_ANY")/wc, g("SQ_WAIT_INST_ANY")/wc, g("SQ_ACTIVE_INST_LDS")/bc, g("SQ_LDS_BANK_CONFLICT")/max(g("SQ_ACTIVE_INST_LDS"), 1), g("SQ_INSTS_VALU")/g("SQ_WAVES"), g("SQ_INSTS_LDS")/g("SQ_WAVES")))
