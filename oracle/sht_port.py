"""Python driver of oracle/sht_port.c -- TEST / BASELINE INFRASTRUCTURE ONLY.

build()        gcc -O3 -march=native -fopenmp -> oracle/libsht_port.so (git-ignored, travels with gpurun)
leg_s0/leg_spin  thin ctypes wrappers (used by tests/test_oracle_port.py to pin the port to the oracle)
time_sample()  bench.py's cpu_baseline leg: times the port on a bounded sample of the benchmark
               workload on the host cores and extrapolates to one full map2alm+alm2map round trip.
"""
import ctypes, os, subprocess, time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libsht_port.so")
_lib = None

def build(force=False):
	src = os.path.join(HERE, "sht_port.c")
	if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
		subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-shared", "-fPIC", src, "-o", LIB, "-lm"])
	return LIB

def lib():
	global _lib
	if _lib is None:
		_lib = ctypes.CDLL(build())
		vp, i = ctypes.c_void_p, ctypes.c_int
		_lib.sht_port_leg_s0.argtypes = [i, i, vp, i, vp, vp, vp, vp, vp, vp, i]
		_lib.sht_port_leg_spin.argtypes = [i, i, i, vp, i, vp, vp, vp, vp, vp, vp, vp, vp, i]
		_lib.sht_port_threads.restype = i
	return _lib

def pairs_from_theta(theta):
	"""north/south pairing of an ascending symmetric ring list -> (idx_n, idx_s(-1), cth, sth, sh2, ch2)"""
	th = np.asarray(theta, np.longdouble); n = len(th)
	idx_n, idx_s = [], []
	for i in range((n+1)//2):
		j = n-1-i
		if j != i and abs(th[j]-(np.pi-th[i])) < 1e-12: idx_n.append(i); idx_s.append(j)
		else: idx_n.append(i); idx_s.append(-1)
		if j != i and idx_s[-1] == -1: idx_n.append(j); idx_s.append(-1)
	idx_n = np.array(idx_n, np.int32); idx_s = np.array(idx_s, np.int32)
	t = th[idx_n]
	return idx_n, idx_s, np.cos(t).astype(float), np.sin(t).astype(float), np.sin(t/2).astype(float), np.cos(t/2).astype(float)

def _p(a): return a.ctypes.data

def leg(spin, lmax, msel, theta, alm=None, leg=None):
	"""alm[nmsel, nc, lmax+1] <-> leg[nmsel, nc, nring] for the listed m (synthesis if leg is None, else adjoint).
	Only rings with theta <= pi/2 or paired rings are supported (all grids used here)."""
	msel = np.ascontiguousarray(msel, np.int32); nms = len(msel)
	idx_n, idx_s, cth, sth, sh2, ch2 = pairs_from_theta(theta); np_ = len(idx_n); nr = len(theta)
	has_s = np.ascontiguousarray(idx_s >= 0, np.int32)
	nc = 1 if spin == 0 else 2
	direction = 0 if leg is None else 1
	if direction == 0:
		A = np.ascontiguousarray(alm, np.complex128).reshape(nms, nc, lmax+1).copy()
		ln = np.zeros((nms, nc, np_), np.complex128); ls = np.zeros((nms, nc, np_), np.complex128)
	else:
		A = np.zeros((nms, nc, lmax+1), np.complex128)
		L = np.asarray(leg, np.complex128).reshape(nms, nc, nr)
		ln = np.ascontiguousarray(L[:, :, idx_n]); ls = np.ascontiguousarray(L[:, :, np.maximum(idx_s, 0)])
	if spin == 0: lib().sht_port_leg_s0(lmax, nms, _p(msel), np_, _p(cth), _p(sth), _p(has_s), _p(A), _p(ln), _p(ls), direction)
	else: lib().sht_port_leg_spin(spin, lmax, nms, _p(msel), np_, _p(cth), _p(sth), _p(sh2), _p(ch2), _p(has_s), _p(A), _p(ln), _p(ls), direction)
	if direction == 1: return A
	out = np.zeros((nms, nc, nr), np.complex128)
	out[:, :, idx_n] = ln
	sel = idx_s >= 0
	out[:, :, idx_s[sel]] = ls[:, :, sel]
	return out

def time_sample(cfg, budget_s=20.0):
	"""Time the port on a bounded sample and extrapolate to one full round trip of `cfg`
	(dict with shape, lmax, spin, ncomp as in bench.py), all stages with real threads on the host cores:
	  Legendre        synthesis + adjoint on the minimal CC grid (lmax+2 rings, the ring count the GPU path iterates) for a
	                  subset of m, OpenMP over m, extrapolated by sum(lmax - m + 1);
	  ring FFTs       scipy.fft rfft + irfft (pocketfft, workers = all cores) on a subset of rings;
	  theta resampling  oracle/sht_fast.theta_resample (the FFT-based exact |sin| integration, scipy.fft workers) on a subset of
	                  m columns, counted once per direction (its transpose costs the same).
	This is a C/numpy restatement of the same algorithm -- NOT ducc0 (absent from this image); bench.py reports whether
	`import ducc0` works on the box and times ducc0 instead when it does."""
	import scipy.fft as sfft
	L = lib(); ncores = L.sht_port_threads()
	lmax = cfg["lmax"]; ny, nx = cfg["shape"]
	R = min(ny, lmax+2)
	theta = np.arange(R)*np.pi/(R-1)
	rng = np.random.default_rng(0)
	def run(spin, msel):
		nc = 1 if spin == 0 else 2
		alm = rng.standard_normal((len(msel), nc, lmax+1))+1j*rng.standard_normal((len(msel), nc, lmax+1))
		t0 = time.perf_counter()
		lg = leg(spin, lmax, msel, theta, alm=alm)
		leg(spin, lmax, msel, theta, leg=lg)
		return time.perf_counter()-t0
	def weight(ms, spin): return float(np.sum(lmax-np.maximum(np.asarray(ms), spin)+1))
	allm = np.arange(lmax+1)
	spins = list(cfg["spin"])
	t_leg = 0.0; nsel_used = {}
	for spin in spins:
		n0 = min(lmax+1, max(2*ncores, 8))
		msel = np.unique(np.linspace(0, lmax, n0).astype(int))
		t = run(spin, msel)                                   # calibration (also warms the threads)
		share = budget_s*0.5/len(spins)
		n1 = int(min(lmax+1, max(n0, n0*share/max(t, 1e-3))))
		msel = np.unique(np.linspace(0, lmax, n1).astype(int))
		t = run(spin, msel)
		t_leg += t*weight(allm, spin)/weight(msel, spin)
		nsel_used[spin] = len(msel)
	def timed_scaled(fn, n0, nmax, share):
		"""run fn(n) on a calibration sample n0, then on the sample size that fills `share` seconds; returns (seconds, n)"""
		t0 = time.perf_counter(); fn(n0); t = time.perf_counter()-t0
		n1 = int(min(nmax, max(n0, n0*share/max(t, 1e-3))))
		if n1 <= n0: return t, n0
		t0 = time.perf_counter(); fn(n1); return time.perf_counter()-t0, n1
	# ring FFTs, threaded
	def ring_fft(n):
		x = rng.standard_normal((n, nx))
		t0 = time.perf_counter(); h = sfft.rfft(x, axis=1, workers=ncores); sfft.irfft(h, n=nx, axis=1, workers=ncores)
		ring_fft.t = time.perf_counter()-t0
	t_, nr = timed_scaled(ring_fft, min(cfg["ncomp"]*ny, 2*ncores), min(cfg["ncomp"]*ny, int(4e8/nx)), budget_s*0.15)
	t_fft_full = ring_fft.t*(cfg["ncomp"]*ny/nr)
	# theta resampling (only when the map has more rings than the CC grid): every host thread resamples its own block of columns
	# (pocketfft through scipy with workers=1, numpy for the rest: both release the GIL on arrays this size), the tables that do not
	# depend on the data are built once before the clock starts.  Timed at TWO sample sizes so that what does not scale with the
	# number of columns (thread start-up) is not multiplied up: full = fixed + per-column x all columns.
	t_res_full = 0.0; res_info = None
	if ny > R:
		from concurrent.futures import ThreadPoolExecutor
		from . import sht_fast
		sht_fast._resample_tables("F1", ny, lmax)
		ncol_all = cfg["ncomp"]*(lmax+1)
		def resample(n):
			Lc = rng.standard_normal((n, ny))+1j*rng.standard_normal((n, ny))
			blocks = [b for b in np.array_split(np.arange(n), ncores) if len(b)]
			t0 = time.perf_counter()
			with ThreadPoolExecutor(max_workers=ncores) as ex:
				list(ex.map(lambda idx: sht_fast.theta_resample(Lc[idx], idx % 2, "F1", ny, lmax, workers=1), blocks))
			return time.perf_counter()-t0
		n1 = int(min(ncol_all, 2*ncores))
		resample(n1)                                              # warm-up (thread pool, page faults of the tables)
		ta = resample(n1)
		per_col = ta/n1
		n2 = int(min(ncol_all, max(2*n1, (budget_s*0.15)/max(per_col, 1e-6))))
		n2 = min(n2, int(2e9//(16*8*(ny+lmax))))                  # bound the temporaries (~8 complex arrays of the padded length per column)
		n2 = max(n2, n1)
		tb = resample(n2) if n2 > n1 else ta
		slope = (tb-ta)/(n2-n1) if n2 > n1 and tb > ta else per_col
		fixed = max(0.0, ta-slope*n1)
		t_res_full = 2*(fixed+slope*ncol_all)
		res_info = dict(columns_timed=[n1, n2], seconds=[round(ta, 4), round(tb, 4)], per_column_s=slope, fixed_s=round(fixed, 4), columns_total=ncol_all, directions=2)
	total = t_leg+t_fft_full+t_res_full
	return dict(value=round(1.0/total, 6), unit="round-trips/s", cores=ncores, kind="port",
		seconds_per_round_trip=round(total, 3), legendre_s=round(t_leg, 3), ring_fft_s=round(t_fft_full, 3), theta_resampling_s=round(t_res_full, 3),
		extrapolation=dict(legendre={"m_values_timed": {str(k): v for k, v in nsel_used.items()}, "of": lmax+1, "scaled_by": "sum over m of (lmax - max(m, spin) + 1)"},
			ring_fft={"rings_timed": nr, "of": cfg["ncomp"]*ny, "scaled_by": "rings"}, theta_resampling=res_info),
		sample="oracle/sht_port.c (C, f64, OpenMP x%d, -O3 -march=native): Legendre synthesis+adjoint on the CC grid of %d rings for %s of %d m values "
			"(extrapolated by sum(lmax-m+1)); scipy.fft rfft+irfft (workers=%d) on %d of %d rings; theta resampling in the full-interpolant form (oracle/sht_fast.py; the fine-CC form of the default analysis runs the same transforms on a shorter middle circle: "
			"pocketfft via scipy, one block of columns per host thread) timed at two sample sizes, fixed + per-column cost extrapolated to %d columns, both directions. "
			"A restatement of the same algorithm on the CPU, NOT ducc0." % (ncores, R, str(nsel_used), lmax+1, ncores, nr, cfg["ncomp"]*ny, cfg["ncomp"]*(lmax+1)))

def time_full(cfg, nthreads=None, max_seconds=60.0):
	"""One FULL round trip of `cfg` on the host, nothing extrapolated: every m of the Legendre stage (synthesis + adjoint on the CC grid
	of lmax + 2 rings), every ring FFT, every column of the theta resampling in both directions -- the same three stages as time_sample,
	for configurations small enough to run whole (the reference's benchmark shape, BASELINE C1 and C2).  nthreads: host threads (None:
	all; 1 = the OMP_NUM_THREADS=1 setting of the reference's own benchmark script).  Returns None if a calibration step says the whole
	would take longer than max_seconds.  A restatement of the same algorithm on the CPU, NOT ducc0."""
	import scipy.fft as sfft
	L = lib(); L.sht_port_set_threads.restype = ctypes.c_int
	nthr = int(nthreads) if nthreads else L.sht_port_threads()
	old = L.sht_port_set_threads(nthr)
	try:
		lmax = cfg["lmax"]; ny, nx = cfg["shape"]
		R = min(ny, lmax+2)
		theta = np.arange(R)*np.pi/(R-1)
		rng = np.random.default_rng(0)
		allm = np.arange(lmax+1)
		# calibration on 2 % of the m values
		msel = np.unique(np.linspace(0, lmax, max(8, (lmax+1)//50)).astype(int))
		t0 = time.perf_counter()
		for spin in cfg["spin"]:
			nc = 1 if spin == 0 else 2
			a = rng.standard_normal((len(msel), nc, lmax+1))+0j
			leg(spin, lmax, msel, theta, leg=leg(spin, lmax, msel, theta, alm=a))
		if (time.perf_counter()-t0)*(lmax+1)/len(msel) > max_seconds: return None      # (also the warm-up of the OpenMP threads)
		t_leg = 0.0
		for spin in cfg["spin"]:
			nc = 1 if spin == 0 else 2
			alm = rng.standard_normal((lmax+1, nc, lmax+1))+1j*rng.standard_normal((lmax+1, nc, lmax+1))
			t0 = time.perf_counter()
			lg = leg(spin, lmax, allm, theta, alm=alm)
			leg(spin, lmax, allm, theta, leg=lg)
			t_leg += time.perf_counter()-t0
			del alm, lg
		x = rng.standard_normal((cfg["ncomp"]*ny, nx))
		t0 = time.perf_counter(); h = sfft.rfft(x, axis=1, workers=nthr); sfft.irfft(h, n=nx, axis=1, workers=nthr); t_fft = time.perf_counter()-t0
		del x, h
		t_res = 0.0
		if ny > R:
			from concurrent.futures import ThreadPoolExecutor
			from . import sht_fast
			sht_fast._resample_tables("F1", ny, lmax)
			ncol = cfg["ncomp"]*(lmax+1)
			with ThreadPoolExecutor(max_workers=nthr) as ex:      # warm-up: thread start, page faults of the tables
				list(ex.map(lambda k: sht_fast.theta_resample(rng.standard_normal((2, ny))+0j, np.arange(2) % 2, "F1", ny, lmax, workers=1), range(nthr)))
			t0 = time.perf_counter()
			for c0 in range(0, ncol, 4096):      # (blocks: the temporaries of a column are ~8 padded complex lines)
				n = min(4096, ncol-c0)
				Lc = rng.standard_normal((n, ny))+1j*rng.standard_normal((n, ny))
				blocks = [b for b in np.array_split(np.arange(n), nthr) if len(b)]
				with ThreadPoolExecutor(max_workers=nthr) as ex:
					list(ex.map(lambda idx: sht_fast.theta_resample(Lc[idx], idx % 2, "F1", ny, lmax, workers=1), blocks))
			t_res = 2*(time.perf_counter()-t0)      # both directions (the transpose costs the same; the data generation above is inside: < 2 %)
		total = t_leg+t_fft+t_res
		return dict(value=round(1.0/total, 6), unit="round-trips/s", cores=nthr, kind="port", full=True, seconds_per_round_trip=round(total, 4),
			legendre_s=round(t_leg, 4), ring_fft_s=round(t_fft, 4), theta_resampling_s=round(t_res, 4),
			sample="oracle/sht_port.c (C, f64, OpenMP x%d) + scipy.fft: ONE FULL round trip, nothing extrapolated -- Legendre synthesis + adjoint for all %d m on the CC grid of %d rings, "
				"rfft + irfft of all %d rings, theta resampling of all %d columns (timed one way, counted twice). A restatement of the same algorithm on the CPU, NOT ducc0." % (nthr, lmax+1, R, cfg["ncomp"]*ny, cfg["ncomp"]*(lmax+1) if ny > R else 0))
	finally:
		L.sht_port_set_threads(old)

if __name__ == "__main__":
	import json, sys
	cfg = dict(shape=(5400, 10800), lmax=4000, spin=[0, 2], ncomp=3)
	print(json.dumps(time_sample(cfg, float(sys.argv[1]) if len(sys.argv) > 1 else 10.0), indent=1))
