"""Summary of tools/r06_pmc_line.sh: per kernel of the FFT family the average duration and the SQ counters per launch (SQ_* cycle counters are in quad-cycles, summed
over the waves), engines on (1) and off (0)."""
import sys, glob, csv, collections, re
O = sys.argv[1]
def short(n):
	m = re.search(r"pxs::(theta_line_kernel|ring_line_kernel|transpose_mul_outcol)", n)
	if m:
		mid = re.search(r"RfSeq<16, 15, 15, 3>, pxs::RfSeq<(7, 9|)", n)
		return m.group(1) + (" (to_cc)" if mid and mid.group(1) else (" (from_cc_adjoint)" if m.group(1) == "theta_line_kernel" else ""))
	m = re.search(r"chain_kernel<pxs::(\w+(?:<\d>)?)", n)
	return "chain " + m.group(1) if m else None
for v in ("1", "0"):
	dur = collections.defaultdict(list)
	for f in glob.glob(O+"/kt%s/**/*kernel_trace.csv" % v, recursive=True):
		for r in csv.DictReader(open(f)):
			k = short(r["Kernel_Name"])
			if k: dur[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
	cnt = collections.defaultdict(lambda: collections.defaultdict(float)); nl = collections.defaultdict(lambda: collections.defaultdict(int))
	for p in ("p1", "p2", "p3"):
		for f in glob.glob(O+"/%s_%s/**/*counter_collection.csv" % (p, v), recursive=True):
			for r in csv.DictReader(open(f)):
				k = short(r["Kernel_Name"])
				if not k: continue
				c = r["Counter_Name"]; cnt[k][c] += float(r["Counter_Value"]); nl[k][c] += 1
	print("==== engines %s (tools/chain_lab.py c4: 8 maps per launch)" % ("ON" if v == "1" else "OFF: the stage chains"))
	for k in sorted(dur, key=lambda k: -sum(dur[k])):
		c = {n: cnt[k][n]/max(nl[k][n], 1) for n in cnt[k]}
		g = lambda n: c.get(n, float("nan"))
		wc = g("SQ_WAVE_CYCLES")
		print("%-36s n=%3d avg %.3f ms | per wave: VALU %.0f LDS %.0f SALU %.0f VMEM rd %.0f wr %.0f | of the wave cycles: valu_active %.3f lds_active %.3f issue stalls %.3f waitcnt %.3f lds_wait %.3f | LDS bank conflict cycles / LDS active %.3f" % (
			k, len(dur[k]), sum(dur[k])/len(dur[k]), g("SQ_INSTS_VALU")/g("SQ_WAVES"), g("SQ_INSTS_LDS")/g("SQ_WAVES"), g("SQ_INSTS_SALU")/g("SQ_WAVES"), g("SQ_INSTS_VMEM_RD")/g("SQ_WAVES"), g("SQ_INSTS_VMEM_WR")/g("SQ_WAVES"),
			g("SQ_ACTIVE_INST_VALU")/wc, g("SQ_ACTIVE_INST_LDS")/wc, g("SQ_WAIT_INST_ANY")/wc, g("SQ_WAIT_ANY")/wc, g("SQ_WAIT_INST_LDS")/wc, g("SQ_LDS_BANK_CONFLICT")/max(g("SQ_LDS_IDX_ACTIVE"), 1.0)))
