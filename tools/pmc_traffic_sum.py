"""Summarise FETCH_SIZE / WRITE_SIZE counter passes (tools/pmc_traffic.sh) per kernel.
FETCH_SIZE and WRITE_SIZE are reported in KB by rocprofv3; on gfx950 FETCH_SIZE counts a wide
coalesced read at half its bytes (MI355X_MICROARCH.md, HBM section), so both the raw and the x2
figure are given.  Writes profiles/<tag>_traffic.json."""
import csv, json, sys, collections, re
def load(path, name):
	out = collections.defaultdict(lambda: [0, 0.0, 0.0])
	with open(path) as f:
		for row in csv.DictReader(f):
			if row["Counter_Name"] != name: continue
			k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
			if len(k) > 60: k = k[:60]
			o = out[k]; o[0] += 1; o[1] += float(row["Counter_Value"]); o[2] += (int(row["End_Timestamp"])-int(row["Start_Timestamp"]))*1e-6
	return out
# Calibration of FETCH_SIZE per access pattern (the guide: "calibrate on a known byte count in your own access pattern").  The
# fused FFT-chain kernels read arrays of exactly known size once, which calibrates the counter: kernels whose loads are 16 bytes
# per lane along a contiguous row (element-fast: StResize, StSigma, StSplit, StRingA2, StRingS2) report HALF their bytes (x2),
# kernels that load 8-byte reals (StRingA1) or 16-byte points in 128-byte runs across rows (line-fast: StFirst, StRingS1) report
# them in full (x1).  WRITE_SIZE matches the known bytes everywhere.  Legendre kernels keep the guide's x2 as an upper bound.
FETCH_FACTOR = {"StFirst": 1.0, "StRingA1": 1.0, "StRingS1": 1.0}
def fetch_factor(k):
	for key, f in FETCH_FACTOR.items():
		if key in k: return f
	return 2.0
cfg, tag = sys.argv[1], sys.argv[2]
ROUND_TRIPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3   # bench.py --steps 1 --warmup 0 with PXS_BENCH_NO_WEIGHTS=1: two round trips in the setup (cold call, second round trip) + one timed
f = load(f"gpurun_out/pmc_fetch_{cfg}/f_counter_collection.csv", "FETCH_SIZE")
w = load(f"gpurun_out/pmc_write_{cfg}/w_counter_collection.csv", "WRITE_SIZE")
res = {}
for k in sorted(set(f) | set(w)):
	nf, kbf, msf = f.get(k, [0, 0, 0]); nw, kbw, _ = w.get(k, [0, 0, 0])
	if not k.startswith("pxs::"): continue
	res[k] = {"launches": nf, "launches_per_round_trip": nf/ROUND_TRIPS, "ms_total_under_pmc": round(msf, 3), "fetch_MB_per_launch_raw": round(kbf/1024/max(nf, 1), 3),
		"fetch_MB_per_launch_x2": round(2*kbf/1024/max(nf, 1), 3), "fetch_factor_calibrated": fetch_factor(k),
		"fetch_MB_per_launch_calibrated": round(fetch_factor(k)*kbf/1024/max(nf, 1), 3), "write_MB_per_launch": round(kbw/1024/max(nw, 1), 3)}
json.dump({"config": cfg, "round_trips_in_trace": ROUND_TRIPS, "units": "the *_MB_per_launch fields are MiB (2^20 bytes: rocprofv3 reports KiB, divided by 1024 here); bench.py converts them to bytes with 2^20", "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KB counters / 1024); x2 = gfx950 wide-read correction of MI355X_MICROARCH.md; calibrated = per-kernel factor from the known array sizes of the FFT-chain kernels (see tools/pmc_traffic_sum.py)", "kernels": res}, open(f"profiles/{tag}_traffic_{cfg}.json", "w"), indent=1)
for k, v in res.items(): print(f"{k:62s} n={v['launches']:4d} ms={v['ms_total_under_pmc']:9.2f} fetch={v['fetch_MB_per_launch_raw']:10.2f} (x2 {v['fetch_MB_per_launch_x2']:10.2f}) write={v['write_MB_per_launch']:10.2f} MB/launch")
