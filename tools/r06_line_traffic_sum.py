"""Summary of tools/r06_line_traffic.sh: per kernel family of tools/chain_lab.py at C4 (8 maps per launch), the counter traffic per launch
(FETCH_SIZE x the calibration of tools/pmc_traffic_sum.py: 16-byte-per-lane row reads report half their bytes; WRITE_SIZE as is; both
in KiB) against the algorithmic bytes (arrays in + out once)."""
import csv, sys, re, collections, json
O = sys.argv[1]
def load(path, name):
	out = collections.defaultdict(lambda: [0, 0.0])
	try:
		with open(path) as f:
			for row in csv.DictReader(f):
				if row["Counter_Name"] != name: continue
				k = re.sub(r"[<(].*", "", row["Kernel_Name"]).replace("void ", "")
				if "chain_kernel" in row["Kernel_Name"]:
					m = re.search(r"chain_kernel<pxs::(\w+(?:<\d>)?)", row["Kernel_Name"]); k = "chain_kernel<%s>" % (m.group(1) if m else "?")
				o = out[k]; o[0] += 1; o[1] += float(row["Counter_Value"])*1024
	except FileNotFoundError: pass
	return out
nmaps, nm, nr, ncc, nphi = 8, 4001, 5400, 4033, 10800
alg = {"to_cc": nmaps*nm*(nr + ncc)*16, "from_cc_adjoint": nmaps*nm*(nr + ncc)*16, "h2map": nmaps*(nr*nm*16 + nr*nphi*8)}
res = {}
for v in ("1", "0"):
	f = load("%s/FETCH_SIZE_line%s.csv" % (O, v), "FETCH_SIZE"); w = load("%s/WRITE_SIZE_line%s.csv" % (O, v), "WRITE_SIZE")
	fam = {"theta": 0.0, "h2map": 0.0}
	rows = {}
	for k in sorted(set(f) | set(w)):
		if not (k.startswith("pxs::") or k.startswith("chain_kernel")): continue
		n = max(f.get(k, [0, 0])[0], w.get(k, [0, 0])[0], 1)
		rows[k] = dict(launches=n, fetch_bytes_raw_per_launch=f.get(k, [0, 0])[1]/n, write_bytes_per_launch=w.get(k, [0, 0])[1]/n)
	res["line_engines_on" if v == "1" else "stage_chains"] = rows
res["algorithmic_bytes_per_launch"] = alg
res["note"] = "tools/chain_lab.py c4: every launch is 8 maps; to_cc and from_cc_adjoint each one launch of theta_line_kernel (its two instantiations) or 5 / 3 chain stages; h2map one ring_line_kernel or StRingS1 + StRingS2. fetch bytes raw: 16-byte-per-lane row reads report half (x2 for those kernels: theta_line_kernel, ring_line_kernel, StResize, StSigma, StSplit, StRingS2)."
json.dump(res, open(O + "/line_traffic.json", "w"), indent=1)
for sec in ("line_engines_on", "stage_chains"):
	print("==", sec)
	for k, r in res[sec].items(): print("  %-48s n=%3d fetch(raw) %8.1f MB  write %8.1f MB" % (k, r["launches"], r["fetch_bytes_raw_per_launch"]/1e6, r["write_bytes_per_launch"]/1e6))
print("algorithmic (in + out once), MB per launch:", {k: round(v/1e6, 1) for k, v in alg.items()})
