"""Summarises tools/pmc_leg.sh: per Legendre kernel the average duration and the SQ counters per launch, with the ratios that say
where the VALU issue slots go (SQ_* cycle counters are in quad-cycles: x4 = shader cycles, summed over the waves)."""
import sys, glob, csv, collections, re
O = sys.argv[1]
def short(n):
	m = re.search(r"pxs::(leg_\w+<[\d, ]+>|alm_\w+|reduce_partials)", n)
	return m.group(1) if m else None
dur = collections.defaultdict(list)
for f in glob.glob(O+"/kt/**/*kernel_trace.csv", recursive=True):
	for r in csv.DictReader(open(f)):
		k = short(r["Kernel_Name"])
		if k: dur[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
cnt = collections.defaultdict(lambda: collections.defaultdict(float)); nl = collections.defaultdict(lambda: collections.defaultdict(int))
for p in ("p1", "p2", "p3"):
	for f in glob.glob(O+"/"+p+"/**/*counter_collection.csv", recursive=True):
		for r in csv.DictReader(open(f)):
			k = short(r["Kernel_Name"])
			if not k: continue
			c = r["Counter_Name"]; cnt[k][c] += float(r["Counter_Value"]); nl[k][c] += 1
print("kernel                launches  avg_ms  total_ms")
for k in sorted(dur, key=lambda k: -sum(dur[k])): print("%-22s %6d %8.3f %9.2f" % (k, len(dur[k]), sum(dur[k])/len(dur[k]), sum(dur[k])))
for k in sorted(cnt, key=lambda k: -sum(dur.get(k, [0]))):
	if not k.startswith("leg_"): continue
	c = {n: cnt[k][n]/max(nl[k][n], 1) for n in cnt[k]}
	g = lambda n: c.get(n, float("nan"))
	print("\n%s (per launch)" % k)
	print("  " + "  ".join("%s=%.4g" % (n, v) for n, v in sorted(c.items())))
	wc = g("SQ_WAVE_CYCLES"); bc = g("SQ_BUSY_CYCLES")
	print("  valu_insts_per_wave=%.0f  salu_per_wave=%.0f smem_per_wave=%.0f lds_per_wave=%.1f vmem_rd_per_wave=%.1f" % (g("SQ_INSTS_VALU")/g("SQ_WAVES"), g("SQ_INSTS_SALU")/g("SQ_WAVES"), g("SQ_INSTS_SMEM")/g("SQ_WAVES"), g("SQ_INSTS_LDS")/g("SQ_WAVES"), g("SQ_INSTS_VMEM_RD")/g("SQ_WAVES")))
	print("  of the wave cycles: valu_active=%.3f  any_active=%.3f  wait_inst_any (issue stalls)=%.3f  wait_any (waitcnt)=%.3f;  valu quad-cycles per valu inst=%.2f" % (
		g("SQ_ACTIVE_INST_VALU")/wc, g("SQ_ACTIVE_INST_ANY")/wc, g("SQ_WAIT_INST_ANY")/wc, g("SQ_WAIT_ANY")/wc, g("SQ_ACTIVE_INST_VALU")/g("SQ_INSTS_VALU")))
