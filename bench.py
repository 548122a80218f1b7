#!/usr/bin/env python
"""bench.py -- map2alm + alm2map round trips/s on MI355X (BASELINE.json metric).

One step = curvedsky.map2alm(map, lmax, spin) followed by curvedsky.alm2map(alm, map, spin) on device-resident
float64 data.  Inputs are synthetic band-limited Gaussian maps made on the GPU before the timed region.

  --config c3 (default)  3x(21600x43200) T/Q/U, lmax 10000: the configuration the metric is quoted on.  With N > 1 every
                         rank transforms its own map (weak scaling, no data-path collective) and the alm of each step are
                         all-gathered over RCCL on a side stream, overlapped with the synthesis and the next step's analysis (two alm buffers).
  --config c4            BASELINE config 4: 64 independent 1x(5400x10800) maps, lmax 4000, sharded contiguously over the
                         ranks (strong scaling: 64/N maps per GPU, one batched call per direction), RCCL all-gather of the alm.
  --config c5            BASELINE config 5: 100 Gaussian realisations at lmax 6000 on 10800x21600, sharded over the ranks; per
                         realisation rand_alm (device generator) -> alm2map -> enmap.fft("phys") -> calc_ps2d -> lbin and map2alm -> alm2cl;
                         only the spectra are gathered.  Reports realisations/s with a per-stage breakdown.
  --config c2 / c1 / ref 3x(5400x10800) lmax 4000 / 1x(1024x2048) lmax 512 / the reference's benchmark shape 1x(900x1800) lmax 750

Prints ONE JSON line on rank 0 (driver contract); human-readable detail goes to stderr.  Besides the contract fields:
  roofline      the dominant kernel family (Legendre, FP64 FMA) from hipEvent stage timers inside the library
  fft_chain     the HBM-bound kernel family of the step (ring FFTs + theta resampling): algorithmic bytes / time / 8 TB/s
  fft           enmap.fft / ifft of one map component on the GPU with numpy.fft.fftn on the host cores beside it (N = 1)
  h2d_inclusive the round trip with the PCIe transfers of map and alm added (measured pinned-memory rate; never `value`)
  cpu_baseline  oracle/sht_port.c (+ scipy.fft) on a bounded sample on the host cores, or ducc0 itself when importable
"""
import argparse, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path: sys.path.insert(0, ROOT)

CONFIGS = {
	"c1":  dict(ncomp=1, shape=(1024, 2048),   lmax=512,   spin=[0],    name="C1 1x(1024x2048) lmax=512 spin0"),
	"c2":  dict(ncomp=3, shape=(5400, 10800),  lmax=4000,  spin=[0, 2], name="C2 3x(5400x10800) T/Q/U lmax=4000 spin0/2"),
	"c3":  dict(ncomp=3, shape=(21600, 43200), lmax=10000, spin=[0, 2], name="C3 3x(21600x43200) T/Q/U lmax=10000 spin0/2"),
	"c4":  dict(ncomp=1, shape=(5400, 10800),  lmax=4000,  spin=[0],    name="C4 64x[1x(5400x10800)] independent maps, lmax=4000 spin0", nbatch_total=64),
	"c5":  dict(ncomp=1, shape=(10800, 21600), lmax=6000,  spin=[0],    name="C5 100x[rand_alm -> alm2map -> enmap.fft -> ps2d -> lbin | map2alm -> alm2cl], 1x(10800x21600), lmax=6000", nreal_total=100),
	"tqu": dict(ncomp=3, shape=(5400, 10800),  lmax=4000,  spin=[0, 2], name="16x[3x(5400x10800)] T/Q/U maps in one call, lmax=4000 spin0/2 (batched polarisation sims; not a BASELINE configuration)", nbatch_total=16),
	"ref": dict(ncomp=1, shape=(900, 1800),    lmax=750,   spin=[0],    name="reference benchmark shape 1x(900x1800) lmax=750"),
}
FRAC_DEFINITION = "frac: credited flops of one direction over the kernel family's time / 78.6 TFLOP/s, never above 1 (frac_convention_exceeds_peak says if it was clipped) -- VALU kernels: SURVEY 8(d) F_alg = (4 n0 + 12 n2) R nalm; FP64-MFMA kernels (batched scalar maps): the GEMM, 2 flop per (l, m, ring, map) = F_alg / 2; frac_hw: the FP64 flops the kernels executed (recurrence + accumulation FMAs / MFMAs of the steps the waves ran, counted in the kernels) over the same time"
FP64_PEAK_TFLOPS = 78.6     # MI355X FP64 vector = FP64 matrix peak (AMD spec; 256 CU x 4 SIMD x 16 FMA lanes x 2 x 2.4 GHz)
# what a kernel of nothing but independent v_fma_f64 with three VGPR operands (the form of the analysis accumulation and of every recurrence step) sustains
# on the gpurun boxes: 54-65 TFLOP/s by hipEvent time at ANY occupancy from 1 to 8 waves per SIMD -- one wave alone issues one every 4.4 s_memtime counts, and
# the counts per ms fall from 1.85e6 to 1.1e6 as waves are added (tools/dp_rate.hip, profiles/r05_dp_rate_and_k_waves.txt); with one scalar operand (the synthesis accumulation) 74.6-76.1
# (tools/fma_peak.hip, round 3); a v_mfma_f64_16x16x4_f64 stream 65.7 (profiles/r05_mfma_f64_rate.txt).  Context for frac_hw; `peak` stays the nominal figure.
FP64_SUSTAINED_TFLOPS = 64.9
def credited_flops(F_survey, mm, n0_only=True):
	"""flops one direction is credited with in `roofline.achieved / frac`.  VALU kernels: SURVEY 8(d)'s F_alg (4 flop per (l, m, ring) for spin 0,
	12 for a Q/U pair).  FP64-MFMA kernels (batched scalar maps): the GEMM the algorithm needs, 2 flop per (l, m, ring, map) -- half of
	SURVEY's scalar credit, which prices the two FMAs of a VALU accumulation and made `frac` exceed 1 for kernels that issue one MFMA
	multiply-add per (l, m, ring, map, side).  The recurrence shared by the maps of a batch is not credited."""
	return 0.5*F_survey if mm else F_survey

def roof_frac(achieved_tflops):
	"""(frac, flag): a fraction of peak is reported as one -- a convention that credits more than the pipe can do is clipped and flagged"""
	f = achieved_tflops/FP64_PEAK_TFLOPS
	return (round(min(f, 1.0), 4), f > 1.0)

def sustained_note(frac_hw_both):
	return dict(fp64_fma_stream_TFLOPs=FP64_SUSTAINED_TFLOPS, source="tools/dp_rate.hip: v_fma_f64 with three VGPR operands only, best of 1-8 waves per SIMD (profiles/r05_dp_rate_and_k_waves.txt); 74.6-76.1 with one scalar operand (tools/fma_peak.hip); 78.6 nominal",
		frac_hw_of_stream={k: round(v*FP64_PEAK_TFLOPS/FP64_SUSTAINED_TFLOPS, 4) for k, v in frac_hw_both.items()})
HBM_PEAK_GBS = 8000.0

def log(*a):
	print(*a, file=sys.stderr, flush=True)

def emit(res):
	"""the ONE line on stdout: whatever native libraries left in libc's stdout buffer goes out first"""
	try:
		import ctypes; ctypes.CDLL(None).fflush(None)
	except Exception: pass
	print(json.dumps(res), flush=True)

def nalm(lmax): return (lmax+1)*(lmax+2)//2

def alg_flops_direction(cfg, R, nmaps=1):
	"""F_alg = (4 n0 + 12 n2) R nalm per direction (per map)"""
	n0 = sum(1 for s in cfg["spin"] if s == 0); n2 = sum(1 for s in cfg["spin"] if s != 0)
	return nmaps*(4*n0+12*n2)*R*nalm(cfg["lmax"])

def chain_alg_bytes(cfg, ncc, nmaps=1):
	"""algorithmic bytes of the FFT stages per direction (complex128 ring spectra): ring FFT = map + leg on the map's rings,
	theta resampling = leg on the map's rings + leg on the CC grid (when the plan resamples)"""
	ny, nx = cfg["shape"]; nm = cfg["lmax"]+1; nc = cfg["ncomp"]*nmaps
	ring = nc*(ny*nx*8+nm*ny*16)
	theta = nc*(nm*ny*16+nm*ncc*16) if ncc < ny else 0
	return ring+theta

def make_alm(cfg, seed, device, ncomp=None):
	"""white Gaussian alm with C_l = 1/(l+1)^2 (T), 0.01 C_l (E,B); m=0 real (SURVEY 8d recipe), on the GPU"""
	import torch
	lmax = cfg["lmax"]; n = nalm(lmax); ncomp = cfg["ncomp"] if ncomp is None else ncomp
	g = torch.Generator(device=device); g.manual_seed(seed)
	re = torch.randn((ncomp, n), generator=g, device=device, dtype=torch.float64)
	im = torch.randn((ncomp, n), generator=g, device=device, dtype=torch.float64)
	alm = torch.complex(re, im)/np.sqrt(2)
	m_of = torch.repeat_interleave(torch.arange(lmax+1, device=device), torch.arange(lmax+1, 0, -1, device=device))
	l_of = torch.arange(n, device=device)-(m_of*(2*lmax+1-m_of))//2
	alm = alm/(l_of+1.0)
	alm[:, :lmax+1] = alm[:, :lmax+1].real*np.sqrt(2)+0j
	if len(cfg["spin"]) > 1 and ncomp == 3:
		alm[1:] *= 0.1
		alm[1:, l_of < 2] = 0
	return alm

def probe_ducc0():
	"""say whether the real reference kernel library is on the box"""
	try:
		import ducc0
		return dict(available=True, version=getattr(ducc0, "__version__", "?"))
	except Exception as e:
		return dict(available=False, reason="%s: %s" % (type(e).__name__, e))

def ducc0_accuracy():
	"""accuracy.vs_ducc0: the HIP path against ducc0 itself (when the box has it) on a C1-sized grid -- white noise through analysis_2d (where
	the quadrature of the default form shows), a band-limited map through synthesis_2d; tests/test_ducc0_live.py is the full comparison"""
	if not probe_ducc0()["available"]: return dict(available=False, note="ducc0 not importable on this box: the default analysis form stays pinned to the oracle's restatement only (tests/test_ducc0_live.py skips)")
	import ducc0
	from pixell_amd import sht
	nt, nph, lmax = 1024, 2048, 512
	ms = sht.tri_mstart(lmax, lmax); na = int(ms[-1])+lmax+1
	rng = np.random.default_rng(5)
	out = dict(available=True, version=getattr(ducc0, "__version__", "?"), grid="F1 %dx%d lmax %d" % (nt, nph, lmax))
	rel = lambda a, b: float(np.sqrt(np.mean(np.abs(a-b)**2)/np.mean(np.abs(b)**2)))
	for spin in (0, 2):
		nc = 1 if spin == 0 else 2
		kw = dict(spin=spin, lmax=lmax, mmax=lmax, geometry="F1", phi0=0.0, mstart=ms)
		noise = rng.standard_normal((nc, nt, nph))
		a_ref = np.zeros((nc, na), complex); ducc0.sht.experimental.analysis_2d(alm=a_ref, map=noise, nthreads=0, **kw)
		a_got = np.zeros((nc, na), complex); sht.analysis_2d(alm=a_got, map=noise, **kw)
		m_ref = np.zeros((nc, nt, nph)); ducc0.sht.experimental.synthesis_2d(alm=a_ref, map=m_ref, nthreads=0, **kw)
		m_got = np.zeros((nc, nt, nph)); sht.synthesis_2d(alm=a_ref, map=m_got, **kw)
		out["spin%d" % spin] = dict(analysis_2d_white_noise_rel_rms=rel(a_got, a_ref), synthesis_2d_rel_rms=rel(m_got, m_ref))
	return out

def ducc0_baseline(cfg, budget_s=30.0):
	"""ducc0.sht.experimental.analysis_2d + synthesis_2d on all host cores (only when ducc0 imports and one round trip fits the budget)"""
	import ducc0
	ny, nx = cfg["shape"]; lmax = cfg["lmax"]
	if 1e-9*cfg["ncomp"]*ny*lmax*lmax > 50*budget_s: return None          # rough flop estimate: skip configs that would take minutes
	rng = np.random.default_rng(0); nthr = os.cpu_count()
	t_tot = 0.0
	for s in cfg["spin"]:
		nc = 1 if s == 0 else 2
		alm = (rng.standard_normal((nc, nalm(lmax)))+1j*rng.standard_normal((nc, nalm(lmax))))
		alm[:, :lmax+1] = alm[:, :lmax+1].real
		m = np.zeros((nc, ny, nx))
		t0 = time.perf_counter()
		ducc0.sht.experimental.synthesis_2d(alm=alm, map=m, spin=s, lmax=lmax, geometry="F1", nthreads=nthr)
		ducc0.sht.experimental.analysis_2d(alm=alm, map=m, spin=s, lmax=lmax, geometry="F1", nthreads=nthr)
		t_tot += time.perf_counter()-t0
	return dict(value=round(1.0/t_tot, 6), unit="round-trips/s", cores=nthr, kind="reference", seconds_per_round_trip=round(t_tot, 3),
		sample="ducc0 %s synthesis_2d + analysis_2d, full workload, nthreads=%d" % (getattr(ducc0, "__version__", "?"), nthr))

def cpu_baseline(cfg, budget_s=20.0):
	"""ducc0 when it is importable (kind 'reference'), otherwise the CPU port (the oracle, kind 'port') on a bounded sample"""
	if probe_ducc0()["available"]:
		try:
			r = ducc0_baseline(cfg)
			if r is not None: return r
		except Exception as e: log("ducc0 baseline failed: %r" % (e,))
	from oracle import sht_port
	return sht_port.time_sample(cfg, budget_s)

def cpu_baseline_full(name, cfg, nmaps=1):
	"""cpu_baseline of a secondary configuration: ONE FULL round trip of the CPU port on the host cores, nothing extrapolated
	(oracle.sht_port.time_full), outside every timed region.  Batched configurations (independent maps): one map measured, times the maps.
	The reference's benchmark shape also with one thread, as scripts/benchmark_pixell.py:17-21 sets OMP_NUM_THREADS=1."""
	from oracle import sht_port
	one = dict(cfg); one["ncomp"] = cfg["ncomp"]
	try:
		r = sht_port.time_full(one, None, max_seconds=45.0)
		if r is None: return dict(kind="port", value=None, note="one full round trip of the CPU port would take more than 45 s on this host: not run (the headline's cpu_baseline is the sampled one)")
		if nmaps > 1:
			r = dict(r); r["seconds_per_round_trip_one_map"] = r["seconds_per_round_trip"]; r["maps"] = nmaps
			r["seconds_per_round_trip"] = round(r["seconds_per_round_trip"]*nmaps, 4); r["value"] = round(1.0/r["seconds_per_round_trip"], 6)
			r["unit"] = "round-trips/s of the whole batch"; r["sample"] += " One map measured in full; the batch is %d independent maps (x %d)." % (nmaps, nmaps)
		if name == "ref":
			r1 = sht_port.time_full(one, 1, max_seconds=45.0)
			if r1: r["one_thread"] = {k: r1[k] for k in ("value", "unit", "cores", "seconds_per_round_trip", "legendre_s", "ring_fft_s", "theta_resampling_s")}
		return r
	except Exception as e:
		return dict(kind="port", value=None, error=repr(e))

def measured_traffic(config, dom):
	"""HBM bytes of a kernel family per step from the committed PMC passes of this bench command (tools/pmc_traffic.sh ->
	profiles/r02_traffic_<config>.json; FETCH_SIZE corrected per access pattern -- x2 for 16-byte-per-lane row reads as the gfx950 note
	of MI355X_MICROARCH.md prescribes, x1 where the known array sizes of the chain kernels show full counting -- WRITE_SIZE as reported).
	Counters cannot be read from inside this process: null when no profile of this config is committed."""
	for tag in ("r06b", "r06", "r05c", "r05b", "r05", "r04b", "r04", "r03", "r02", "r01"):
		path = os.path.join(ROOT, "profiles", "%s_traffic_%s.json" % (tag, config))
		if os.path.exists(path): break
	else: return dict(traffic=None)
	try:
		d = json.load(open(path)); tot = 0.0; raw = 0.0
		for k, v in d["kernels"].items():
			if not any(k.startswith("pxs::"+p) for p in dom): continue
			n = v["launches_per_round_trip"]
			tot += n*(v.get("fetch_MB_per_launch_calibrated", v["fetch_MB_per_launch_x2"])+v["write_MB_per_launch"]); raw += n*(v["fetch_MB_per_launch_raw"]+v["write_MB_per_launch"])
		return dict(traffic=round(tot*2**20), traffic_unit="bytes per step, all launches of the kernel family",
			traffic_uncorrected=round(raw*2**20), traffic_source=os.path.basename(path))
	except Exception as e:
		log("traffic profile unreadable: %r" % (e,)); return dict(traffic=None)

def fft_block(dmap_comp, enmap, torch, reps=3):
	"""enmap.fft / ifft (2-D, last two axes) of one map component on the GPU; numpy.fft.fftn (the reference's always-available
	numpy engine, pixell/fft.py:8-31) on a bounded sample on the host beside it.  B = bytes of the user arrays read + written."""
	ny, nx = dmap_comp.shape[-2:]
	def timed(fn):
		out = fn(); torch.cuda.synchronize()
		t0 = time.perf_counter()
		for _ in range(reps): out = fn()
		torch.cuda.synchronize()
		return (time.perf_counter()-t0)/reps, out
	fbuf = enmap.dmap(torch.empty(dmap_comp.shape, dtype=torch.complex128, device=dmap_comp.tensor.device), dmap_comp.wcs)
	gbuf = enmap.dmap(torch.empty_like(fbuf.tensor), dmap_comp.wcs)     # outputs preallocated: the timing is the transform, not hipMalloc
	t_r2c, f = timed(lambda: enmap.fft(dmap_comp, omap=fbuf))
	b_r2c = ny*nx*(8+16)
	t_c2c, g = timed(lambda: enmap.ifft(f, omap=gbuf))
	b_c2c = ny*nx*32
	err = float((g.tensor.real-dmap_comp.tensor).abs().max()/dmap_comp.tensor.abs().max())
	del f, g, fbuf, gbuf
	# host: numpy.fft.fftn of a block of the same aspect ratio (a few seconds of work at most)
	sy, sx = max(8, ny//8), max(16, nx//8)
	x = np.random.default_rng(0).standard_normal((sy, sx))
	t0 = time.perf_counter(); np.fft.fftn(x, axes=(-2, -1)); t_np = time.perf_counter()-t0
	return dict(shape=[ny, nx], real_to_complex_ms=round(t_r2c*1e3, 3), real_to_complex_GBps=round(b_r2c/t_r2c/1e9, 1),
		complex_to_complex_ms=round(t_c2c*1e3, 3), complex_to_complex_GBps=round(b_c2c/t_c2c/1e9, 1),
		frac_of_8TBps=round(b_r2c/t_r2c/1e9/HBM_PEAK_GBS, 4), frac_of_8TBps_complex_to_complex=round(b_c2c/t_c2c/1e9/HBM_PEAK_GBS, 4), roundtrip_max_error=err,
		note="frac_of_8TBps is enmap.fft of a real map (what the path transforms): real rows go two per complex line through the chain stages and only the Hermitian half runs through the column passes, 4.5 N x 16 bytes of traffic for (8 + 16) N algorithmic bytes; complex input (enmap.ifft) goes through the same stage kinds (rows into a transposed intermediate, columns back): both axes exceed the LDS (160 KiB = 10240 complex128 points), so each axis is a two-pass four-step transform, 4 reads + 4 writes of the array against the 1 + 1 the algorithmic count credits",
		cpu_baseline=dict(kind="reference", engine="numpy.fft.fftn (pixell.fft numpy engine)", cores=1, sample="%dx%d float64 block" % (sy, sx),
			seconds=round(t_np, 4), GBps=round(sy*sx*24/t_np/1e9, 3)))

def pcie_rates(torch, device, nbytes=1 << 31):
	"""pinned host <-> device copy rates in GB/s"""
	host = torch.empty(nbytes//8, dtype=torch.float64).pin_memory()
	dev = torch.empty(nbytes//8, dtype=torch.float64, device=device)
	dev.copy_(host, non_blocking=True); torch.cuda.synchronize()
	t0 = time.perf_counter(); dev.copy_(host, non_blocking=True); torch.cuda.synchronize(); h2d = nbytes/(time.perf_counter()-t0)/1e9
	t0 = time.perf_counter(); host.copy_(dev, non_blocking=True); torch.cuda.synchronize(); d2h = nbytes/(time.perf_counter()-t0)/1e9
	return h2d, d2h

def _free_port():
	import socket
	s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p

def spawn_ranks(ngpus, argv):
	"""`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) through torch.distributed.run on this node,
	exactly as the driver does, and hand back its exit code.  Rank 0 of the child job prints the JSON line."""
	import subprocess
	cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus), "--master-addr", "127.0.0.1",
		"--master-port", str(_free_port()), os.path.abspath(__file__)]+list(argv)
	log("bench.py: --gpus %d without a launcher: %s" % (ngpus, " ".join(cmd)))
	env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env.setdefault("OMP_NUM_THREADS", "1")
	return subprocess.call(cmd, env=env)

def dry_run(args, rank, world, backend):
	"""--dry: the launch / sharding / gather skeleton of the N-rank bench without any transform (GPU-less rehearsal of the
	rank plumbing: tests/test_bench_launch.py).  The line carries "dry": true and no performance claim."""
	import torch, torch.distributed as dist
	from pixell_amd import dist as pdist
	cfg = CONFIGS[args.config]; batched = "nbatch_total" in cfg
	ntot = int(os.environ.get("PXS_BENCH_NBATCH", cfg.get("nbatch_total", 0))) if batched else 0
	lo, hi = pdist.shard_range(ntot, rank, world) if batched else (rank, rank+1)
	nmaps = hi-lo; ncomp = cfg["ncomp"]; nelem = 37
	alm = torch.full((ncomp*nmaps, nelem), float(rank+1), dtype=torch.complex128)
	rows = [ncomp*n for n in pdist.shard_sizes(ntot, world)] if batched else [ncomp]*world
	ranks_seen = 1; gather = None
	if world > 1:
		gather = pdist.AlmGather(alm, rows, "cpu", backend)
		seen = [None]*world; dist.all_gather_object(seen, rank); ranks_seen = len(set(seen))
	t0 = time.perf_counter()
	for _ in range(args.warmup+args.steps):
		if gather is not None: gather.run(alm)
	if world > 1: dist.barrier()
	dt = time.perf_counter()-t0
	if gather is not None:
		for r, blk in enumerate(gather.result()): assert blk.shape[0] == rows[r] and bool((blk.real == r+1).all()), "gather returned the wrong block for rank %d" % r
	maps_total = ntot if batched else world
	if rank == 0:
		print(json.dumps(dict(metric="map2alm+alm2map round-trips/sec", value=None, unit="round-trips/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
			ms_per_step=round(dt/max(1, args.steps)*1e3, 3), higher_is_better=True, scaling="strong" if batched else "weak", dry=True, backend=backend,
			rccl_ranks_seen=ranks_seen, config=dict(workload=cfg["name"], maps_per_gpu=nmaps, maps_total=maps_total))), flush=True)

def run_c5(args, torch, dist, rank, world, local, device, backend):
	"""BASELINE config 5 (SURVEY 8d recipe): the mixed SHT + 2-D FFT power-spectrum pipeline over independent realisations.
	One step = every rank works through its contiguous shard of the realisations, PXS_BENCH_C5_BATCH (8) at a time: the two SHTs of a
	batch are ONE call each (the batched Legendre kernels; the reference loops over the maps, curvedsky.py:763-765, 1038-1046);
	nothing but the spectra leaves a rank.  Returns the result line (rank 0 emits it)."""
	from pixell_amd import curvedsky, enmap, sht, dist as pdist
	cfg = CONFIGS["c5"]; lmax = cfg["lmax"]; ny, nx = cfg["shape"]
	ntot = int(os.environ.get("PXS_BENCH_NREAL", cfg["nreal_total"]))
	lo, hi = pdist.shard_range(ntot, rank, world); nloc = hi-lo
	nbc = max(1, min(int(os.environ.get("PXS_BENCH_C5_BATCH", "8")), nloc))
	shape, wcs = enmap.fullsky_geometry(shape=(ny, nx))
	ainfo = curvedsky.alm_info(lmax)
	cl_in = 1.0/(np.arange(lmax+1)+1.0)**2
	m = enmap.dmap(torch.zeros((nbc, ny, nx), dtype=torch.float64, device=device), wcs)
	fbuf = enmap.dmap(torch.empty((1, ny, nx), dtype=torch.complex128, device=device), wcs)
	alm_in = torch.zeros((nbc, nalm(lmax)), dtype=torch.complex128, device=device)
	alm_out = torch.zeros((nbc, nalm(lmax)), dtype=torch.complex128, device=device)
	stages = ["rand_alm", "alm2map", "enmap_fft", "ps2d_lbin", "map2alm", "alm2cl"]
	minfo = curvedsky.analyse_geometry(m.shape, wcs)
	plan = sht.grid_plan(minfo.ducc_geo.name, ny, nx, minfo.phi0, minfo.flip, lmax, lmax, ainfo.mstart, 1)
	def batch(i0, n, ev=None):
		"""realisations i0 .. i0 + n of this rank's shard; returns (binned 2-D spectra, bin centres, C_l [n, lmax + 1])"""
		def mark():
			if ev is not None: e = torch.cuda.Event(enable_timing=True); e.record(); ev.append(e)
		mark()
		for j in range(n): alm_in[j] = curvedsky.rand_alm(cl_in, ainfo=ainfo, seed=200+i0+j, rng="device")
		mark(); curvedsky.alm2map(alm_in[:n], enmap.dmap(m.tensor[:n], wcs), spin=[0], ainfo=ainfo)
		mark()
		bs = []; t_fft = []
		for j in range(n):
			e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
			e0.record(); f = enmap.fft(enmap.dmap(m.tensor[j:j+1], wcs), omap=fbuf, normalize="phys")
			e1.record(); ps = enmap.calc_ps2d(f); b, l = enmap.lbin(ps); e2.record()
			bs.append(b); t_fft.append((e0, e1, e2))
		mark()      # (the FFT / spectrum stages of the n maps alternate: their split comes from the inner events)
		curvedsky.map2alm(enmap.dmap(m.tensor[:n], wcs), alm=alm_out[:n], spin=[0], ainfo=ainfo)
		mark(); cl = torch.stack([curvedsky.alm2cl(alm_out[j], ainfo=ainfo) for j in range(n)])
		mark()
		if ev is not None: ev.append(t_fft)
		return bs, l, cl
	t0 = time.time()
	bs, l, cl = batch(lo, nbc)                       # builds the plans; checks below
	torch.cuda.synchronize()
	rt_err = float((alm_out-alm_in).abs().pow(2).mean().sqrt()/alm_in.abs().pow(2).mean().sqrt())
	cl_ref = curvedsky.alm2cl(alm_in[0], ainfo=ainfo)
	cl_err = float(((cl[0]-cl_ref).abs().max()/cl_ref.abs().max()).item())
	b = bs[0]
	ok = np.isfinite(b[0]) & (l > 200) & (l < 0.5*lmax)
	flat_ratio = float(np.median(b[0][ok]/np.interp(l[ok], np.arange(lmax+1), cl_in)))       # flat-sky estimate of a full-sky CAR map: order of magnitude only
	log("[rank %d] c5 setup %.1fs; %d realisation(s) per step in batches of %d; round-trip rms %.2e, alm2cl invariance %.2e, binned 2-D spectrum / C_l (median) %.2f"
		% (rank, time.time()-t0, nloc, nbc, rt_err, cl_err, flat_ratio))
	if not (rt_err < 1e-8 and cl_err < 1e-9): raise SystemExit("bench.py c5: pipeline check failed (round trip %.3e, alm2cl %.3e)" % (rt_err, cl_err))
	cls = torch.zeros((max(pdist.shard_sizes(ntot, world)), lmax+1), dtype=torch.float64, device=device)
	gathered = torch.zeros((world,)+tuple(cls.shape), dtype=torch.float64, device=device) if world > 1 else None
	def step(ev=None):
		for k in range(0, nloc, nbc):
			n = min(nbc, nloc-k)
			_, _, cl = batch(lo+k, n, ev)
			cls[k:k+n] = cl
		if gathered is not None:
			if backend == "nccl": dist.all_gather_into_tensor(gathered.view(world, -1), cls.view(-1))
			else:
				host = [torch.empty(cls.shape, dtype=cls.dtype) for _ in range(world)]; dist.all_gather(host, cls.cpu())
	for _ in range(args.warmup): step()
	torch.cuda.synchronize()
	if world > 1: dist.barrier()
	plan.profile(True)
	torch.cuda.synchronize()
	evs = []
	t0 = time.perf_counter()
	for _ in range(args.steps): step(evs)
	torch.cuda.synchronize()
	if world > 1: dist.barrier()
	torch.cuda.synchronize()
	dt = time.perf_counter()-t0
	prof = plan.profile_read(reset=True); fl_syn, fl_ana = plan.profile_flops(reset=True); plan.profile(False)
	if world > 1:
		t = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
		dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
	stage_ms = {k: 0.0 for k in stages}
	for r in range(len(evs)//7):      # per batch: six marks (before rand_alm, after rand_alm, after alm2map, after the FFT / spectrum loop, after map2alm, after alm2cl) + the inner FFT events
		e = evs[7*r:7*r+7]
		stage_ms["rand_alm"] += e[0].elapsed_time(e[1]); stage_ms["alm2map"] += e[1].elapsed_time(e[2])
		stage_ms["map2alm"] += e[3].elapsed_time(e[4]); stage_ms["alm2cl"] += e[4].elapsed_time(e[5])
		for e0, e1, e2 in e[6]: stage_ms["enmap_fft"] += e0.elapsed_time(e1); stage_ms["ps2d_lbin"] += e1.elapsed_time(e2)
	nre = max(1, args.steps*nloc)
	stage_ms = {k: round(v/nre, 3) for k, v in stage_ms.items()}
	ms_step = dt/args.steps*1e3
	# algorithmic bytes / flops per realisation: SHT round trip 2 F_alg, 2 (map + alm) bytes; the 2-D FFT reads the real map and writes the complex one
	R_alg = min(ny, lmax+2); F_survey = 4*R_alg*nalm(lmax); F_dir = credited_flops(F_survey, nbc >= 4); B_fft = ny*nx*(8+16)
	leg = {k: prof[k][0]/nre for k in ("leg_syn", "leg_ana")}      # Legendre kernel ms per realisation, hipEvents inside the library (pxs_profile)
	dom = "leg_ana" if leg["leg_ana"] >= leg["leg_syn"] else "leg_syn"
	exe = {"leg_syn": fl_syn/nre, "leg_ana": fl_ana/nre}
	frac_hw = {k: (round(exe[k]/(leg[k]*1e-3)/1e12/FP64_PEAK_TFLOPS, 4) if leg[k] > 0 else 0.0) for k in leg}
	res = dict(metric="power-spectrum pipeline realisations/sec (rand_alm + alm2map + enmap.fft + ps2d/lbin + map2alm + alm2cl)", value=round(ntot*args.steps/dt, 4), unit="realisations/s",
		n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_step, 3), higher_is_better=True, scaling="strong", vs_baseline=None, dtype="f64", data="synthetic",
		config=dict(workload=cfg["name"] if ntot == cfg["nreal_total"] else cfg["name"].replace("100x", "%dx" % ntot), geometry="CAR fejer1 %dx%d" % (ny, nx), lmax=lmax,
			realisations_total=ntot, realisations_per_gpu=nloc, realisations_per_call=nbc, rng="device (torch Philox); the legacy-numpy recipe pinned to the reference costs ~0.5 s per realisation on the host",
			parallelism=("independent realisations sharded contiguously over the ranks; RCCL all-gather of the C_l only" if world > 1 else "single GPU")),
		ms_per_realisation=round(ms_step/max(nloc, 1), 3), stage_ms_per_realisation=stage_ms,
		roofline=dict(bound="mfma" if (dom == "leg_ana" and nbc >= 4) else "fp64_valu", kernel="leg_ana_* (Legendre analysis of a batch)" if dom == "leg_ana" else "leg_syn_* (Legendre synthesis of a batch)",
			achieved=round(F_dir/(leg[dom]*1e-3)/1e12, 3) if leg[dom] > 0 else 0.0, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s",
			frac=roof_frac(F_dir/(leg[dom]*1e-3)/1e12)[0] if leg[dom] > 0 else 0.0, frac_convention_exceeds_peak=roof_frac(F_dir/(leg[dom]*1e-3)/1e12)[1] if leg[dom] > 0 else False,
			frac_hw=frac_hw[dom], frac_hw_both=frac_hw, frac_definition=FRAC_DEFINITION, traffic=None,
			kernel_ms_per_realisation={k: round(v, 3) for k, v in leg.items()}, algorithmic_flops_per_realisation_direction=F_dir, survey_flops_per_realisation_direction=F_survey, executed_flops_per_realisation=exe,
			sustained=sustained_note(frac_hw)),
		fft=dict(bound="hbm", kernel="enmap.fft real -> complex, 10800x21600", ms=stage_ms["enmap_fft"], achieved=round(B_fft/(stage_ms["enmap_fft"]*1e-3)/1e9, 1) if stage_ms["enmap_fft"] > 0 else 0.0,
			peak=HBM_PEAK_GBS, unit="GB/s", frac=round(B_fft/(stage_ms["enmap_fft"]*1e-3)/1e9/HBM_PEAK_GBS, 4) if stage_ms["enmap_fft"] > 0 else 0.0),
		checks=dict(roundtrip_rms_error=rt_err, alm2cl_invariance=cl_err, binned_ps2d_over_cl_median=flat_ratio), ducc0=probe_ducc0())
	if world > 1: res["rccl_ranks_seen"] = world
	del m, fbuf, alm_in, alm_out
	return res

def run_sht(args, ctx):
	"""One SHT configuration (args.config, args.steps, args.warmup): K timed round trips of map2alm + alm2map on device-resident maps; returns the
	result line.  args.primary (default True): with the legs that belong to the headline line only (other analysis forms; host arrays, 2-D FFT and
	CPU baseline unless args.no_cpu)."""
	from pixell_amd import curvedsky, enmap, sht, dist as pdist
	torch, dist, rank, world, local, device, backend, force_pg = ctx.torch, ctx.dist, ctx.rank, ctx.world, ctx.local, ctx.device, ctx.backend, ctx.force_pg
	primary = getattr(args, "primary", True)
	cfg = CONFIGS[args.config]
	batched = "nbatch_total" in cfg
	ntot = int(os.environ.get("PXS_BENCH_NBATCH", cfg.get("nbatch_total", 0))) if batched else 0     # (smaller batches for rehearsals)
	# memory check: fall back to the largest configuration that fits
	free, total = torch.cuda.mem_get_info()
	need = {"c3": 130e9, "c4": 1.5e9*max(1, ntot//world)+12e9, "tqu": 4.5e9*max(1, ntot//world)+40e9, "c2": 12e9, "c1": 1e9, "ref": 1e9, "c5": 40e9}[args.config]
	if free < need:
		log("config %s needs ~%.0f GB, only %.0f GB free: falling back to c2" % (args.config, need/1e9, free/1e9))
		cfg = CONFIGS["c2"]; args.config = "c2"; batched = False
	lmax = cfg["lmax"]; ny, nx = cfg["shape"]; ncomp = cfg["ncomp"]
	if batched:
		lo, hi = pdist.shard_range(ntot, rank, world)       # contiguous shard of the batch axis
		nmaps = hi-lo
	else: lo, nmaps = rank, 1
	shape, wcs = enmap.fullsky_geometry(shape=(ny, nx))
	ainfo = curvedsky.alm_info(lmax)
	t0 = time.time()
	ncomp_all = ncomp*nmaps
	alm_in = torch.cat([make_alm(cfg, 1000+lo+i, device) for i in range(nmaps)], 0) if batched else make_alm(cfg, 1000+rank, device)
	dmap = enmap.dmap(torch.zeros((ncomp_all, ny, nx), dtype=torch.float64, device=device), wcs)
	if batched and ncomp > 1:      # a stack of T/Q/U maps: [map][component][...], one call per spin group over all maps
		if world > 1: raise SystemExit("bench.py: --config %s is a single-GPU configuration" % args.config)
		alm_in = alm_in.view(nmaps, ncomp, -1); dmap = enmap.dmap(dmap.tensor.view(nmaps, ncomp, ny, nx), wcs)
	alm_out = torch.zeros_like(alm_in)
	torch.cuda.synchronize()
	# ---- the cold call, piece by piece (a script that transforms once on a geometry pays all of it; the timed steps below pay none):
	# plan object (ring tables, FFT-chain plans, theta-resampling tables), recurrence tables of every spin (host long double, threaded),
	# the first transform of each direction (scratch allocation, recurrence seeds recorded), the second round trip (seeds loaded)
	def timed(fn):
		torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return (time.perf_counter()-t)*1e3, r
	mem0 = sht.memory()
	minfo = curvedsky.analyse_geometry(dmap.shape, wcs)
	t_plan, plan = timed(lambda: sht.grid_plan(minfo.ducc_geo.name, ny, nx, minfo.phi0, minfo.flip, lmax, lmax, ainfo.mstart, 1))
	t_tab, _ = timed(lambda: [plan.set_option("build_tables", s) for s in sorted(set(cfg["spin"]))])
	t_syn1, _ = timed(lambda: curvedsky.alm2map(alm_in, dmap, spin=cfg["spin"], ainfo=ainfo))      # synthetic band-limited input
	t_ana1, _ = timed(lambda: curvedsky.map2alm(dmap, alm=alm_out, spin=cfg["spin"], ainfo=ainfo))
	rt_err = float((alm_out-alm_in).abs().pow(2).mean().sqrt()/alm_in.abs().pow(2).mean().sqrt())
	t_rt2, _ = timed(lambda: (curvedsky.map2alm(dmap, alm=alm_out, spin=cfg["spin"], ainfo=ainfo), curvedsky.alm2map(alm_out, dmap, spin=cfg["spin"], ainfo=ainfo)))
	mem1 = sht.memory()
	cold = dict(plan_build_ms=round(t_plan+t_tab, 1), plan_object_ms=round(t_plan, 1), recurrence_tables_ms=round(t_tab, 1),
		hipMalloc_ms=round(mem1["malloc_ms"]-mem0["malloc_ms"], 1), hipMalloc_GB=round((mem1["malloc_bytes"]-mem0["malloc_bytes"])/1e9, 2), hipMalloc_calls=mem1["malloc_calls"]-mem0["malloc_calls"],
		arena_hits=mem1["arena_hits"]-mem0["arena_hits"], arena_GB_kept=round(mem1["arena_bytes"]/1e9, 2),
		first_alm2map_ms=round(t_syn1, 1), first_map2alm_ms=round(t_ana1, 1), first_roundtrip_ms=round(t_plan+t_tab+t_syn1+t_ana1, 1), second_roundtrip_ms=round(t_rt2, 1),
		note="first_roundtrip_ms = plan_build_ms + the first transform of each direction on the fresh plan (scratch allocation, recurrence seeds recorded where the plan uses them); steady_ms is the timed step. "
			"hipMalloc_ms is the part of it the library spent inside hipMalloc (pxs_memory): the device scratch of a plan is mapped at a box-dependent rate; scratch released by earlier plans of the process is reused (arena_hits) instead")
	log("[rank %d] setup %.1fs; %d map(s); round-trip rms error %.2e; cold call: %s" % (rank, time.time()-t0, nmaps, rt_err, {k: v for k, v in cold.items() if k != "note"}))
	# north_star: alm must come back to < 1e-8 relative rms.  A throughput number of a transform that does not is worthless.
	if not (rt_err < 1e-8) and not os.environ.get("PXS_BENCH_NOCHECK"): raise SystemExit("bench.py: round-trip rms error %.3e exceeds 1e-8 -- refusing to time a wrong transform" % rt_err)
	info = plan.info()
	# what map2alm integrates (include/pxsht.h, pxs_plan_option "analysis"): the default is ducc0's route as published
	aform = dict(form={0: "interpolant", 1: "weights", 2: "ducc0"}[plan.query("analysis_form")], ncc_circle=plan.query("ncc_circle"), ducc_ncc_circle=plan.query("ducc_ncc_circle"),
		note="ducc0 = the route of ducc0's analysis_2d as published (theta-interpolant, low-passed to |k| < N_cc, on the CC grid of N_cc + 1 rings, that grid's weights, transposed upsampling to the N_cc/2 + 1 rings of the Legendre stage); ncc_circle is the N_cc this plan runs, ducc_ncc_circle = 2 good_size_complex(lmax + 1)")
	gather = None; side = None; ranks_seen = None
	if (world > 1 or force_pg) and not args.no_gather:
		rows = [ncomp*(pdist.shard_range(ntot, r, world)[1]-pdist.shard_range(ntot, r, world)[0]) for r in range(world)] if batched else [ncomp]*world
		gather = pdist.AlmGather(alm_out, rows, device, backend)
		side = torch.cuda.Stream(device=device)
		seen = [None]*world; dist.all_gather_object(seen, (rank, torch.cuda.get_device_name(local)))
		ranks_seen = len({r for r, _ in seen})

	# N > 1: the alm of a step go out on the side stream while its synthesis AND the next step's analysis run: two alm buffers take turns, and a step
	# only waits for the gather that read ITS buffer two steps ago (one buffer made every step wait for its own gather before the next analysis could
	# overwrite the alm: at 8 ranks the 7 GB a rank receives take longer than the synthesis of its 8 maps)
	alm_pp = [alm_out, torch.zeros_like(alm_out)] if gather is not None else [alm_out]
	gathered_ev = [None]*len(alm_pp); nstep = [0]
	def step():
		b = nstep[0] % len(alm_pp); nstep[0] += 1
		a = alm_pp[b]
		if gathered_ev[b] is not None: torch.cuda.current_stream().wait_event(gathered_ev[b])
		curvedsky.map2alm(dmap, alm=a, spin=cfg["spin"], ainfo=ainfo)
		if gather is not None:
			ev = torch.cuda.Event(); ev.record()
			side.wait_event(ev)
			with torch.cuda.stream(side):
				gather.run(a, side)
				gathered_ev[b] = torch.cuda.Event(); gathered_ev[b].record(side)
		curvedsky.alm2map(a, dmap, spin=cfg["spin"], ainfo=ainfo)

	for _ in range(args.warmup): step()
	torch.cuda.synchronize()
	if world > 1 or force_pg: dist.barrier()
	plan.profile(True)
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(args.steps): step()
	torch.cuda.synchronize()
	if world > 1 or force_pg: dist.barrier()
	torch.cuda.synchronize()
	dt = time.perf_counter()-t0
	prof = plan.profile_read(reset=True); fl_syn, fl_ana = plan.profile_flops(reset=True); plan.profile(False)
	if world > 1 or force_pg:
		t = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
		dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
	ms_step = dt/args.steps*1e3
	maps_total = ntot if batched else world
	value = maps_total*args.steps/dt
	cold["steady_ms"] = round(ms_step, 3)
	# ---- the other forms of the analysis (pxs_plan_option "analysis", include/pxsht.h), reported beside the default (ducc0's route as
	# published: the fine-CC form); never `value`.  All give the same alm on the band-limited input of this bench.
	#   "interpolant": exact quadrature of the full theta-interpolant (the default of rounds 1-3; a fine circle of M > N + 2 lmax points);
	#   "weights": ring quadrature weights + adjoint synthesis, the reference's cyl route (curvedsky.py:852-861, 1068-1084), where the
	#              grid has >= 2 lmax + 2 rings: three theta-resampling stages instead of five.
	ana_forms = None
	if primary and world == 1 and not batched and not os.environ.get("PXS_BENCH_NO_WEIGHTS"):
		ana_forms = {}
		for form in ["interpolant"]+(["weights"] if ny >= 2*lmax+2 else []):
			try:
				aw = torch.zeros_like(alm_out)
				def wstep():
					curvedsky.map2alm(dmap, alm=aw, spin=cfg["spin"], ainfo=ainfo, analysis=form)
					curvedsky.alm2map(aw, dmap, spin=cfg["spin"], ainfo=ainfo)
				wstep(); torch.cuda.synchronize()
				w_err = float((aw-alm_out).abs().pow(2).mean().sqrt()/alm_out.abs().pow(2).mean().sqrt())
				plan.profile(True); torch.cuda.synchronize(); tw = time.perf_counter()
				for _ in range(args.steps): wstep()
				torch.cuda.synchronize(); dtw = time.perf_counter()-tw
				pw = plan.profile_read(reset=True); plan.profile_flops(reset=True); plan.profile(False)
				ana_forms[form] = dict(ms_per_step=round(dtw/args.steps*1e3, 3), value=round(args.steps/dtw, 4), unit="round-trips/s", stage_ms_per_step={k: round(v[0]/args.steps, 3) for k, v in pw.items()},
					alm_rms_difference_from_default=w_err)
				del aw
			except Exception as e: log("analysis form %s: leg failed: %r" % (form, e))
		ana_forms["note"] = "map2alm(..., analysis=<form>) + alm2map; options, not the default (ducc0's route): on maps that are not band-limited the forms differ"

	# ---- roofline of the dominant kernel family (Legendre; FP64 FMA bound) ----
	R_syn, R_ana = info["nring_syn"], info["nring_ana"]
	R_alg = min(ny, lmax+2)
	dom = "leg_ana" if prof["leg_ana"][0] >= prof["leg_syn"][0] else "leg_syn"
	flops_dir = alg_flops_direction(cfg, R_alg, nmaps)             # all spin groups / maps of one direction on this rank
	dom_ms_per_step = prof[dom][0]/args.steps
	exe = (fl_ana if dom == "leg_ana" else fl_syn)/max(args.steps, 1)        # flops the kernels of that family executed per step (counted in the kernels)
	# batched scalar maps: the analysis of 4 or more maps per call is the FP64-MFMA kernel (leg_ana_s0_mm); everything else is plain v_fma_f64
	mm = batched and nmaps >= 4      # (analysis and synthesis, scalar maps and Q/U pairs alike)
	flops_survey = flops_dir; flops_dir = credited_flops(flops_survey, mm)
	achieved = flops_dir/(dom_ms_per_step*1e-3)/1e12 if dom_ms_per_step > 0 else 0.0
	frac_, clipped_ = roof_frac(achieved)
	roof = dict(bound="mfma" if mm else "fp64_valu",
		pipe=("FP64 MFMA (v_mfma_f64_16x16x4_f64): the maps of a batch share one recurrence per ring pair, rings are the K dimension, 16 columns = 4 maps x 4 real right-hand sides; its dense peak equals the vector peak on MI355X"
			if mm else "FP64 vector FMA (v_fma_f64): the contraction is 4 right-hand sides wide per map, too narrow for the 16x16x4 f64 MFMA, whose dense peak equals the vector peak; no MFMA instruction is issued"),
		kernel="leg_ana_* (Legendre analysis, all launches of a step)" if dom == "leg_ana" else "leg_syn_* (Legendre synthesis, all launches of a step)",
		achieved=round(achieved, 3), peak=FP64_PEAK_TFLOPS, unit="TFLOP/s", frac=frac_, frac_convention_exceeds_peak=clipped_,
		frac_hw=round(exe/(dom_ms_per_step*1e-3)/1e12/FP64_PEAK_TFLOPS, 4) if dom_ms_per_step > 0 else 0.0, frac_definition=FRAC_DEFINITION, traffic=None,
		algorithmic_flops_per_step_direction=flops_dir, survey_flops_per_step_direction=flops_survey, executed_flops_per_step_direction=exe,
		executed_flops_both={"leg_syn": fl_syn/max(args.steps, 1), "leg_ana": fl_ana/max(args.steps, 1)},
		frac_hw_both={k: (round(v/max(args.steps, 1)/(prof[k][0]/args.steps*1e-3)/1e12/FP64_PEAK_TFLOPS, 4) if prof[k][0] > 0 else 0.0) for k, v in (("leg_syn", fl_syn), ("leg_ana", fl_ana))},
		kernel_ms_per_step=round(dom_ms_per_step, 3),
		launches_per_step=prof[dom][1]//max(args.steps, 1), R_algorithmic=R_alg, R_actual_syn=R_syn, R_actual_ana=R_ana)
	roof["sustained"] = sustained_note(roof["frac_hw_both"])
	full_workload = not batched or ntot == cfg["nbatch_total"]      # (the committed counter passes are of the full configuration: a share of the batch has no profile)
	if full_workload: roof.update(measured_traffic(args.config, [dom]))
	# ---- the HBM-bound family: ring FFTs + theta resampling (fused chains, csrc/fftchain.hip) ----
	chain_ms = (prof["ring_fft"][0]+prof["resample"][0])/args.steps
	chain_bytes = 2*chain_alg_bytes(cfg, R_ana, nmaps)           # both directions
	chain = dict(bound="hbm", kernel="chain_kernel<*> + theta_line_kernel / ring_line_kernel where a line fits a CU (ring FFTs map<->leg and theta resampling leg<->leg_cc, both directions)",
		achieved=round(chain_bytes/(chain_ms*1e-3)/1e9, 1) if chain_ms > 0 else 0.0, peak=HBM_PEAK_GBS, unit="GB/s",
		algorithmic_bytes_per_step=chain_bytes, kernel_ms_per_step=round(chain_ms, 3))
	chain["frac"] = round(chain["achieved"]/HBM_PEAK_GBS, 4)
	if full_workload: chain.update(measured_traffic(args.config, ["chain_kernel", "theta_line_kernel", "ring_line_kernel", "transpose", "fft_lds", "split_pair", "unpack", "fold"]))
	else: chain["traffic"] = None
	map_bytes = ncomp_all*ny*nx*8; alm_bytes = ncomp_all*nalm(lmax)*16
	tot_bytes = (ntot*ncomp*(ny*nx*8+nalm(lmax)*16)) if batched else world*(map_bytes+alm_bytes)
	hbm_gbs = 2*tot_bytes/(ms_step*1e-3)/1e9
	res = dict(metric="map2alm+alm2map round-trips/sec", value=round(value, 4), unit="round-trips/s", n_gpus=world, steps=args.steps,
		warmup=args.warmup, ms_per_step=round(ms_step, 3), higher_is_better=True, scaling="strong" if batched else "weak", vs_baseline=None, dtype="f64",
		data="synthetic", config=dict(workload=cfg["name"] if not batched or ntot == cfg["nbatch_total"] else cfg["name"].replace("64x", "%dx" % ntot),
			geometry="CAR fejer1 %dx%d" % (ny, nx), lmax=lmax, spin=cfg["spin"], maps_per_gpu=nmaps, maps_total=maps_total,
			parallelism=("independent maps sharded contiguously over the ranks; RCCL all-gather of alm" if world > 1 else "single GPU")),
		roofline=roof, fft_chain=chain,
		stage_ms_per_step={k: round(v[0]/args.steps, 3) for k, v in prof.items()},
		stage_note="stages run back to back on one stream (hipEvent-bracketed inside the library)",
		plan_state="data-independent tables built with the plan or by its first transform (untimed warm-up): recurrence coefficients, twiddles, CC quadrature, and the recurrence seeds of DESIGN.md section 4; every timed step runs the full transform on the resident map",
		hbm_algorithmic_GBps=round(hbm_gbs, 1), hbm_frac_of_8TBps=round(hbm_gbs/HBM_PEAK_GBS, 5),
		cold_start=cold, analysis_form=aform, analysis_other_forms=ana_forms,
		roundtrip_rms_error=rt_err, ducc0=probe_ducc0())
	if ranks_seen is not None: res["rccl_ranks_seen"] = ranks_seen; res["collective"] = gather.describe()
	if rank == 0:
		log("stage ms/step:", res["stage_ms_per_step"], " total %.1f ms/step" % ms_step)
		if not args.no_cpu and world == 1:
			try:
				h2d, d2h = pcie_rates(torch, device)
				t_io = (map_bytes+alm_bytes)/(h2d*1e9)+(map_bytes+alm_bytes)/(d2h*1e9)    # map in + alm out, alm in + map out
				res["h2d_inclusive"] = dict(ms_per_step=round(ms_step+t_io*1e3, 1), value=round(maps_total/((ms_step*1e-3)+t_io), 4), unit="round-trips/s",
					pcie_h2d_GBps=round(h2d, 1), pcie_d2h_GBps=round(d2h, 1), note="GPU step + (map + alm bytes) in each direction at the pinned-memory copy rate measured on 2 GiB; host-buffer callers pay this, it is never `value`")
			except Exception as e: log("pcie probe failed: %r" % (e,))
			try:
				# the same round trip with HOST arrays (numpy, pageable), measured: pixell_amd/hostio.py moves them in pinned double-buffered
				# slabs and pipelines the spin groups of a call (T transforms under the upload of Q and U, maps come back while the next group runs)
				hm = enmap.ndmap(dmap.tensor.cpu().numpy(), wcs); ha = alm_out.cpu().numpy()
				def hstep():
					curvedsky.map2alm(hm, alm=ha, spin=cfg["spin"], ainfo=ainfo)
					curvedsky.alm2map(ha, hm, spin=cfg["spin"], ainfo=ainfo)
				hstep(); torch.cuda.synchronize()
				nh = 2; th = time.perf_counter()
				for _ in range(nh): hstep()
				torch.cuda.synchronize(); th = (time.perf_counter()-th)/nh
				h_err = float(np.sqrt(np.mean(np.abs(ha-alm_in.cpu().numpy())**2)/np.mean(np.abs(ha)**2)))
				res["h2d_inclusive"].update(measured_ms_per_step=round(th*1e3, 1), measured_value=round(maps_total/th, 4), measured_alm_rms_error=h_err,
					measured_note="curvedsky.map2alm + alm2map on numpy arrays (pageable host memory), %d round trips after one warm-up; ms_per_step / value above are the model (resident step + PCIe time at the pinned copy rate)" % nh)
				log("host-array round trip: %.1f ms (%.3f round trips/s), alm rms error %.2e" % (th*1e3, maps_total/th, h_err))
				del hm, ha
			except Exception as e: log("host-array leg failed: %r" % (e,))
			try:
				del alm_in; torch.cuda.empty_cache()
				res["fft"] = fft_block(enmap.dmap(dmap.tensor[:1], wcs), enmap, torch)
			except Exception as e: log("fft block failed: %r" % (e,)); res["fft"] = None
			# (the driver's parsed line keeps `roofline` whole: the HBM-bound families' fractions ride inside it as well)
			res["roofline"]["hbm_families"] = dict(
				fft_chain=dict(frac=res["fft_chain"]["frac"], achieved_GBps=res["fft_chain"]["achieved"], kernel_ms_per_step=res["fft_chain"]["kernel_ms_per_step"], traffic=res["fft_chain"].get("traffic")),
				enmap_fft=({k: res["fft"][k] for k in ("frac_of_8TBps", "frac_of_8TBps_complex_to_complex") if k in res["fft"]} if res.get("fft") else None))
			try: res["accuracy"] = dict(vs_ducc0=ducc0_accuracy())
			except Exception as e: res["accuracy"] = dict(vs_ducc0=dict(error=repr(e)))
			try:
				res["cpu_baseline"] = cpu_baseline(dict(cfg, ncomp=ncomp))
			except Exception as e:   # the baseline must never take the GPU number down with it
				log("cpu_baseline failed: %r" % (e,)); res["cpu_baseline"] = None
	del dmap, alm_out, alm_pp, gather
	return res


class Ctx: pass

LEG_KEYS = ("ms_per_step", "value", "unit", "steps", "stage_ms_per_step", "roundtrip_rms_error")
def compact_leg(res):
	"""what a secondary configuration contributes to the one line: its time, throughput, stage split, roofline fractions and round-trip error"""
	out = {k: res[k] for k in LEG_KEYS if k in res}
	out["workload"] = res["config"]["workload"]
	if "ms_per_realisation" in res: out.update(ms_per_realisation=res["ms_per_realisation"], stage_ms_per_realisation=res["stage_ms_per_realisation"], roundtrip_rms_error=res["checks"]["roundtrip_rms_error"], realisations_per_call=res["config"]["realisations_per_call"])
	r = res["roofline"]
	out["roofline"] = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_convention_exceeds_peak", "frac_hw", "frac_hw_both", "frac_definition", "sustained", "kernel_ms_per_step", "kernel_ms_per_realisation", "traffic", "traffic_unit", "traffic_source") if k in r}
	if "fft_chain" in res: out["fft_chain"] = {k: res["fft_chain"][k] for k in ("bound", "achieved", "unit", "frac", "kernel_ms_per_step", "algorithmic_bytes_per_step", "traffic", "traffic_source") if k in res["fft_chain"]}
	if "fft" in res and res["fft"] and "frac" in res["fft"]: out["fft"] = res["fft"]
	return out

def secondary_legs(args, ctx):
	"""After the timed headline loop (never inside it, never in `value`): the other BASELINE configurations and the reference's own benchmark
	shape, a few steps each, so that every configuration has a driver-run number in the one line.  N > 1: the batched configuration (C4, 64
	maps sharded over the ranks -- strong scaling, north_star's "batched-map scaling") beside the weak-scaling headline."""
	torch = ctx.torch
	from pixell_amd import sht
	legs = {}
	plan = [("c4", dict(config="c4", steps=5, warmup=2))] if ctx.world > 1 else [
		("c1", dict(config="c1", steps=20, warmup=2)),
		("c2", dict(config="c2", steps=3, warmup=1)),
		("c4_share", dict(config="c4", steps=3, warmup=1, nbatch=8)),
		("c4", dict(config="c4", steps=2, warmup=1)),
		("c5", dict(config="c5", steps=1, warmup=0, nreal=8)),
		("ref", dict(config="ref", steps=40, warmup=2))]
	for name, kw in plan:
		t0 = time.time()
		leg = argparse.Namespace(**vars(args)); leg.config = kw["config"]; leg.steps = kw["steps"]; leg.warmup = kw["warmup"]; leg.no_cpu = True; leg.primary = False
		try:
			sht.clear_plans(); torch.cuda.empty_cache()
			if "nbatch" in kw: os.environ["PXS_BENCH_NBATCH"] = str(kw["nbatch"])
			if "nreal" in kw: os.environ["PXS_BENCH_NREAL"] = str(kw["nreal"])
			res = run_c5(leg, ctx.torch, ctx.dist, ctx.rank, ctx.world, ctx.local, ctx.device, ctx.backend) if kw["config"] == "c5" else run_sht(leg, ctx)
			legs[name] = compact_leg(res); legs[name]["leg_wall_s"] = round(time.time()-t0, 1)
			if ctx.rank == 0 and ctx.world == 1 and not args.no_cpu:      # (after the leg's timed loop; rank 0 at N = 1 only, as the headline's)
				t1 = time.time()
				cfgl = CONFIGS[kw["config"]]
				nm = int(kw.get("nbatch", cfgl.get("nbatch_total", 1))) if kw["config"] == "c4" else 1
				legs[name]["cpu_baseline"] = cpu_baseline_full(name, cfgl, nm)
				if kw["config"] == "c5": legs[name]["cpu_baseline"]["note_c5"] = "the map2alm + alm2map pair of one realisation only (the 2-D FFT / spectrum steps of the pipeline are not in it)"
				legs[name]["cpu_baseline_wall_s"] = round(time.time()-t1, 1)
			if name == "ref": legs[name]["note"] = "the reference's own benchmark shape (scripts/benchmark_pixell_runner.py:13-27: 900x1800, lmax 750, 40 iterations): wall time per round trip including the Python layer"
		except Exception as e:
			if ctx.world > 1: raise          # (a rank that skipped a leg's collectives would hang the others)
			log("secondary leg %s failed: %r" % (name, e)); legs[name] = dict(error=repr(e))
		finally:
			os.environ.pop("PXS_BENCH_NBATCH", None) if "nbatch" in kw else None
			os.environ.pop("PXS_BENCH_NREAL", None) if "nreal" in kw else None
	return legs

def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=1)
	ap.add_argument("--steps", type=int, default=3)
	ap.add_argument("--warmup", type=int, default=1)
	ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))      # (tqu: 16 T/Q/U maps in one call -- the batched spin-2 Legendre kernels; not a BASELINE configuration)
	ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / fft / h2d legs")
	ap.add_argument("--no-legs", action="store_true", help="skip the secondary configurations (`configs` block) after the headline loop")
	ap.add_argument("--no-gather", action="store_true", help="skip the alm all-gather (N>1)")
	ap.add_argument("--dry", action="store_true", help="rank plumbing only (launch, sharding, gather), no transforms: runs without a GPU over gloo")
	args = ap.parse_args()
	if args.gpus < 1: raise SystemExit("bench.py: --gpus must be >= 1")
	if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
		sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))        # one process per GPU; the children come back here with WORLD_SIZE set
	import torch
	import torch.distributed as dist
	rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
	if world != args.gpus:
		raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d rank(s): refusing to report a line whose n_gpus is not what was asked for" % (args.gpus, world))
	# PXS_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share devices, the gather goes
	# through host memory); the driver's runs use nccl (= RCCL) with one rank per GPU
	backend = os.environ.get("PXS_BENCH_BACKEND", "gloo" if args.dry else "nccl")
	if args.dry:
		if world > 1:
			os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
			dist.init_process_group(backend)
		try: dry_run(args, rank, world, backend)
		finally:
			if world > 1: dist.destroy_process_group()
		return
	assert torch.cuda.is_available(), "bench.py needs a GPU"
	if backend == "nccl" and torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
		raise SystemExit("bench.py: %d ranks on this node but only %d GPU(s) visible: RCCL needs one GPU per rank (PXS_BENCH_BACKEND=gloo shares devices for rehearsals)"
			% (int(os.environ.get("LOCAL_WORLD_SIZE", world)), torch.cuda.device_count()))
	if backend != "nccl": local = local % torch.cuda.device_count()
	torch.cuda.set_device(local)             # before the process group: RCCL binds the communicator to the current device
	device = torch.device("cuda", local)
	# PXS_BENCH_FORCE_PG=1: a process group (and the alm gather) even with one rank -- exercises the RCCL calls of the N > 1 path on a one-GPU box
	force_pg = world == 1 and os.environ.get("PXS_BENCH_FORCE_PG") == "1"
	if force_pg:
		os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(_free_port()))
		os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
	if world > 1 or force_pg:
		os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
		# RCCL logs to stdout (this image sets NCCL_DEBUG=VERSION: a banner, from libc's buffer, i.e. at process exit, AFTER the JSON
		# line): send its log to stderr and keep stdout to the one line the driver parses
		os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
		dist.init_process_group(backend)
		dist.barrier()                               # creates the communicator now; anything it printed leaves libc's buffer before the timed part
		try:
			import ctypes; ctypes.CDLL(None).fflush(None)
		except Exception: pass
	ctx = Ctx(); ctx.torch, ctx.dist, ctx.rank, ctx.world, ctx.local, ctx.device, ctx.backend, ctx.force_pg = torch, dist, rank, world, local, device, backend, force_pg
	try:
		t_all = time.time()
		res = run_c5(args, torch, dist, rank, world, local, device, backend) if args.config == "c5" else run_sht(args, ctx)
		# the other BASELINE configurations, after the headline loop: only beside the default headline (c3), so that `--config X` stays a
		# short single-configuration run for profiling
		# (PXS_BENCH_REHEARSE_LEGS=1: the legs behind another headline configuration -- two-rank rehearsals of the N > 1 path on a one-GPU box)
		if ((args.config == "c3" and res["config"]["lmax"] == CONFIGS["c3"]["lmax"]) or os.environ.get("PXS_BENCH_REHEARSE_LEGS") == "1") and not args.no_legs and not os.environ.get("PXS_BENCH_NO_LEGS"):
			res["configs"] = secondary_legs(args, ctx)
			res["configs_note"] = "secondary configurations run after the timed loop of the headline configuration, each on freshly built plans; never part of `value`"
		res["bench_wall_s"] = round(time.time()-t_all, 1)
		if rank == 0: emit(res)
	finally:
		if world > 1 or force_pg: dist.destroy_process_group()

if __name__ == "__main__":
	main()
