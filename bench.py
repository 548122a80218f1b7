#!/usr/bin/env python
"""bench.py -- map2alm + alm2map round trips/s on MI355X (BASELINE.json metric).

One step = curvedsky.map2alm(map, lmax, spin=[0,2]) followed by curvedsky.alm2map(alm, map, spin=[0,2])
on a device-resident T/Q/U CAR (Fejer-1) map.  Inputs are synthetic band-limited Gaussian maps made
on the GPU before the timed region.  With N>1 every rank transforms its own map (weak
scaling, no data-path collective) and the per-step alm are all-gathered over RCCL on a side stream,
overlapped with the next step.

Prints ONE JSON line on rank 0 (see the driver contract); human-readable detail goes to stderr.
  --config c3 (default): 3x(21600x43200), lmax=10000   -- the configuration the metric is quoted on
  --config c2: 3x(5400x10800), lmax=4000;  c1: 1x(1024x2048), lmax=512;  ref: 1x(900x1800), lmax=750
"""
import argparse, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path: sys.path.insert(0, ROOT)

CONFIGS = {
	"c1":  dict(ncomp=1, shape=(1024, 2048),   lmax=512,   spin=[0],    name="C1 1x(1024x2048) lmax=512 spin0"),
	"c2":  dict(ncomp=3, shape=(5400, 10800),  lmax=4000,  spin=[0, 2], name="C2 3x(5400x10800) T/Q/U lmax=4000 spin0/2"),
	"c3":  dict(ncomp=3, shape=(21600, 43200), lmax=10000, spin=[0, 2], name="C3 3x(21600x43200) T/Q/U lmax=10000 spin0/2"),
	"ref": dict(ncomp=1, shape=(900, 1800),    lmax=750,   spin=[0],    name="reference benchmark shape 1x(900x1800) lmax=750"),
}
FP64_PEAK_TFLOPS = 78.6     # MI355X FP64 vector = FP64 matrix peak (AMD spec; 256 CU x 4 SIMD x 16 FMA lanes x 2 x 2.4 GHz)
HBM_PEAK_GBS = 8000.0

def log(*a):
	print(*a, file=sys.stderr, flush=True)

def nalm(lmax): return (lmax+1)*(lmax+2)//2

def alg_flops_direction(cfg, R):
	"""F_alg = (4 n0 + 12 n2) R nalm per direction"""
	n0 = sum(1 for s in cfg["spin"] if s == 0); n2 = sum(1 for s in cfg["spin"] if s != 0)
	return (4*n0+12*n2)*R*nalm(cfg["lmax"])

def make_alm(cfg, seed, device):
	"""white Gaussian alm with C_l = 1/(l+1)^2 (T), 0.01 C_l (E,B); m=0 real (SURVEY 8d recipe), on the GPU"""
	import torch
	lmax = cfg["lmax"]; n = nalm(lmax)
	g = torch.Generator(device=device); g.manual_seed(seed)
	re = torch.randn((cfg["ncomp"], n), generator=g, device=device, dtype=torch.float64)
	im = torch.randn((cfg["ncomp"], n), generator=g, device=device, dtype=torch.float64)
	alm = torch.complex(re, im)/np.sqrt(2)
	# l of each element in the triangular layout
	m_of = torch.repeat_interleave(torch.arange(lmax+1, device=device), torch.arange(lmax+1, 0, -1, device=device))
	mstart = (m_of*(2*lmax+1-m_of))//2
	l_of = torch.arange(n, device=device)-mstart
	alm = alm/(l_of+1.0)
	alm[:, :lmax+1] = alm[:, :lmax+1].real*np.sqrt(2)+0j
	if cfg["ncomp"] == 3:
		alm[1:] *= 0.1
		alm[1:, l_of < 2] = 0
	return alm

def cpu_baseline(cfg, budget_s=20.0):
	"""CPU port (the oracle) timed on a bounded sample of the same workload on the host cores."""
	from oracle import sht_port
	return sht_port.time_sample(cfg, budget_s)

def measured_traffic(config, dom):
	"""HBM bytes of the dominant Legendre direction per step, from the committed PMC passes
	(tools/pmc_traffic.sh -> profiles/r01_traffic_<config>.json; FETCH_SIZE doubled per the gfx950 note in
	MI355X_MICROARCH.md, WRITE_SIZE as reported).  Counters cannot be read from inside this process, so this
	is the figure of the profiled run of the same command, or null when no profile of this config is committed."""
	path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_traffic_%s.json" % config)
	if not os.path.exists(path): return dict(traffic=None)
	try:
		d = json.load(open(path)); tot = 0.0; raw = 0.0
		for k, v in d["kernels"].items():
			if not k.startswith("pxs::"+dom): continue
			n = v["launches_per_round_trip"]
			tot += n*(v["fetch_MB_per_launch_x2"]+v["write_MB_per_launch"]); raw += n*(v["fetch_MB_per_launch_raw"]+v["write_MB_per_launch"])
		return dict(traffic=round(tot*2**20), traffic_unit="bytes per step, all launches of the kernel family",
			traffic_uncorrected=round(raw*2**20), traffic_source=os.path.basename(path))
	except Exception as e:
		log("traffic profile unreadable: %r" % (e,)); return dict(traffic=None)

def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=1)
	ap.add_argument("--steps", type=int, default=3)
	ap.add_argument("--warmup", type=int, default=1)
	ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
	ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
	ap.add_argument("--no-gather", action="store_true", help="skip the final alm all-gather (N>1)")
	args = ap.parse_args()
	import torch
	import torch.distributed as dist
	rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
	assert torch.cuda.is_available(), "bench.py needs a GPU"
	# PXS_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share devices, the gather goes
	# through host memory); the driver's runs use nccl (= RCCL) with one rank per GPU
	backend = os.environ.get("PXS_BENCH_BACKEND", "nccl")
	if backend != "nccl": local = local % torch.cuda.device_count()
	torch.cuda.set_device(local)             # before the process group: RCCL binds the communicator to the current device
	device = torch.device("cuda", local)
	if world > 1:
		os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
		dist.init_process_group(backend)
	from pixell_amd import curvedsky, enmap, sht
	cfg = CONFIGS[args.config]
	lmax = cfg["lmax"]; ny, nx = cfg["shape"]; ncomp = cfg["ncomp"]
	# memory check: fall back to the largest configuration that fits
	free, total = torch.cuda.mem_get_info()
	need = {"c3": 150e9, "c2": 12e9, "c1": 1e9, "ref": 1e9}[args.config]
	if free < need:
		log("config %s needs ~%.0f GB, only %.0f GB free: falling back to c2" % (args.config, need/1e9, free/1e9))
		cfg = CONFIGS["c2"]; args.config = "c2"; lmax = cfg["lmax"]; ny, nx = cfg["shape"]; ncomp = cfg["ncomp"]
	shape, wcs = enmap.fullsky_geometry(shape=(ny, nx))
	ainfo = curvedsky.alm_info(lmax)
	t0 = time.time()
	alm_in = make_alm(cfg, 1000+rank, device)
	dmap = enmap.dmap(torch.zeros((ncomp, ny, nx), dtype=torch.float64, device=device), wcs)
	curvedsky.alm2map(alm_in, dmap, spin=cfg["spin"], ainfo=ainfo)      # synthetic band-limited input (also builds plans)
	alm_out = torch.zeros_like(alm_in)
	curvedsky.map2alm(dmap, alm=alm_out, spin=cfg["spin"], ainfo=ainfo)
	torch.cuda.synchronize()
	rt_err = float((alm_out-alm_in).abs().pow(2).mean().sqrt()/alm_in.abs().pow(2).mean().sqrt())
	log("[rank %d] setup %.1fs; round-trip rms error %.2e" % (rank, time.time()-t0, rt_err))
	# north_star: alm must come back to < 1e-8 relative rms.  A throughput number of a transform that does not is worthless.
	if not (rt_err < 1e-8) and not os.environ.get("PXS_BENCH_NOCHECK"): raise SystemExit("bench.py: round-trip rms error %.3e exceeds 1e-8 -- refusing to time a wrong transform" % rt_err)
	minfo = curvedsky.analyse_geometry(dmap.shape, wcs)
	# curvedsky runs the spin groups of a map on two streams with a plan each ("lanes"): stage timers are summed over both
	nlanes = 2 if (len(list(enmap.spin_helper(cfg["spin"], ncomp))) > 1 and os.environ.get("PIXELL_AMD_LANES", "1") != "0") else 1
	plans = [sht.grid_plan(minfo.ducc_geo.name, ny, nx, minfo.phi0, minfo.flip, lmax, lmax, ainfo.mstart, 1, lane=l) for l in range(nlanes)]   # (lane 1: scalar group, lane 0: spin group)
	plan = plans[0]
	info = plan.info()
	gather_buf = None; side = None
	if world > 1 and not args.no_gather:
		gather_buf = torch.empty((world,)+tuple(alm_out.shape), dtype=alm_out.dtype, device=device)
		side = torch.cuda.Stream(device=device)

	def step():
		curvedsky.map2alm(dmap, alm=alm_out, spin=cfg["spin"], ainfo=ainfo)
		if gather_buf is not None:
			ev = torch.cuda.Event(); ev.record()
			side.wait_event(ev)
			with torch.cuda.stream(side):
				if backend == "nccl":
					dist.all_gather_into_tensor(gather_buf.view(torch.float64).view(world, -1), alm_out.view(torch.float64).view(-1))
				else:   # rehearsal: gloo has no device all-gather
					side.synchronize()
					host = [torch.empty(alm_out.shape, dtype=alm_out.dtype) for _ in range(world)]
					dist.all_gather(host, alm_out.cpu())
					for r in range(world): gather_buf[r].copy_(host[r])
		curvedsky.alm2map(alm_out, dmap, spin=cfg["spin"], ainfo=ainfo)
		if gather_buf is not None: torch.cuda.current_stream().wait_stream(side)   # alm_out is rewritten by the next step

	for _ in range(args.warmup): step()
	torch.cuda.synchronize()
	if world > 1: dist.barrier()
	for p_ in plans: p_.profile(True)
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(args.steps): step()
	torch.cuda.synchronize()
	if world > 1: dist.barrier()
	torch.cuda.synchronize()
	dt = time.perf_counter()-t0
	prof = {}
	for p_ in plans:
		for k_, v_ in p_.profile_read(reset=True).items():
			a_ = prof.get(k_, (0.0, 0)); prof[k_] = (a_[0]+v_[0], a_[1]+v_[1])
		p_.profile(False)
	if world > 1:
		t = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
		dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
	ms_step = dt/args.steps*1e3
	value = world*args.steps/dt

	# ---- roofline of the dominant kernel (Legendre; FP64 FMA bound) ----
	R_syn, R_ana = info["nring_syn"], info["nring_ana"]
	R_alg = min(ny, lmax+2)
	stages = {k: (v[0]/max(v[1], 1), v[1], v[0]) for k, v in prof.items()}
	# per launch algorithmic flops: a Legendre launch handles one spin group (1 comp spin 0 / 2 comps spin s)
	n_launch_per_step = len(cfg["spin"])
	dom = "leg_ana" if prof["leg_ana"][0] >= prof["leg_syn"][0] else "leg_syn"
	dom_ms_total = prof[dom][0]
	flops_dir = alg_flops_direction(cfg, R_alg)                 # all spin groups of one direction
	dom_ms_per_step = dom_ms_total/args.steps
	achieved = flops_dir/(dom_ms_per_step*1e-3)/1e12 if dom_ms_per_step > 0 else 0.0
	roof = dict(bound="mfma", pipe="fp64 vector FMA (f64 MFMA dense peak is the same 78.6 TF; kernel uses v_fma_f64)",
		kernel="leg_ana_* (Legendre analysis, all spin groups of a step)" if dom == "leg_ana" else "leg_syn_* (Legendre synthesis, all spin groups of a step)",
		achieved=round(achieved, 3), peak=FP64_PEAK_TFLOPS, unit="TFLOP/s", frac=round(achieved/FP64_PEAK_TFLOPS, 4), traffic=None,
		algorithmic_flops_per_step_direction=flops_dir, kernel_ms_per_step=round(dom_ms_per_step, 3),
		launches_per_step=prof[dom][1]//max(args.steps, 1), R_algorithmic=R_alg, R_actual_syn=R_syn, R_actual_ana=R_ana)
	roof.update(measured_traffic(args.config, dom))
	map_bytes = ncomp*ny*nx*8; alm_bytes = ncomp*nalm(lmax)*16
	hbm_gbs = 2*(map_bytes+alm_bytes)/(ms_step*1e-3)/1e9
	res = dict(metric="map2alm+alm2map round-trips/sec", value=round(value, 4), unit="round-trips/s", n_gpus=world, steps=args.steps,
		warmup=args.warmup, ms_per_step=round(ms_step, 3), higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64",
		data="synthetic", config=dict(workload=cfg["name"], geometry="CAR fejer1 %dx%d" % (ny, nx), lmax=lmax, spin=cfg["spin"],
			maps_per_gpu=1, parallelism="independent maps per GPU; RCCL all-gather of alm" if world > 1 else "single GPU"),
		roofline=roof,
		stage_ms_per_step={k: round(v[2]/args.steps, 3) for k, v in stages.items()},
		stage_note=("the scalar and the spin group of the map run on two streams: event-bracketed stage times overlap and add up to more than ms_per_step"
			if nlanes > 1 else "stages run back to back on one stream"),
		hbm_algorithmic_GBps=round(hbm_gbs, 1), hbm_frac_of_8TBps=round(hbm_gbs/HBM_PEAK_GBS, 5),
		roundtrip_rms_error=rt_err)
	if rank == 0:
		log("stage ms/step:", res["stage_ms_per_step"], " total %.1f ms/step" % ms_step)
		if not args.no_cpu and world == 1:
			try:
				res["cpu_baseline"] = cpu_baseline(cfg)
			except Exception as e:   # the baseline must never take the GPU number down with it
				log("cpu_baseline failed: %r" % (e,)); res["cpu_baseline"] = None
		print(json.dumps(res), flush=True)
	if world > 1: dist.destroy_process_group()

if __name__ == "__main__":
	main()
