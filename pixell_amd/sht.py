"""`ducc0.sht.experimental`-shaped entry points backed by the HIP kernels (include/pxsht.h).

Same keyword interface as the calls pixell/curvedsky.py makes (curvedsky.py:907-924, 936-960,
1032-1046, 1068-1084, 501): arrays may be numpy (staged through device memory) or torch CUDA
tensors (used in place, no copies).  Output arrays are mutated in place and returned, as ducc does.
Extra keyword `flip=(flip_y, flip_x)` (ours): the map is given in its native pixel order and the
flips of curvedsky.map2buffer/buffer2map (curvedsky.py:1384-1411) are folded into kernel
addressing; phi0 is still that of the flipped map (analyse_geometry().phi0).
"""
import ctypes, os, contextlib, threading
import numpy as np
from . import _lib

_DT = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.complex64): 2, np.dtype(np.complex128): 3}

def _torch():
	import torch
	return torch

def _is_tensor(x):
	return type(x).__module__.startswith("torch")

def _np_dtype(x):
	if _is_tensor(x):
		torch = _torch()
		return np.dtype({torch.float32: np.float32, torch.float64: np.float64, torch.complex64: np.complex64, torch.complex128: np.complex128}[x.dtype])
	return np.dtype(x.dtype)

_cuda_ready = False
def device_index():
	if _lib.is_hostsim(): return 0
	torch = _torch()
	if not torch.cuda.is_available():
		raise RuntimeError("pixell_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
	global _cuda_ready
	if not _cuda_ready:
		# make torch create its HIP context before libpxsht.so makes its first runtime call: a process whose first HIP
		# call came from the library (numpy-only callers) got "no ROCm-capable device" from hipSetDevice on the GPU box
		torch.cuda.init(); torch.zeros(1, device="cuda"); _cuda_ready = True
	return torch.cuda.current_device()

def current_stream():
	if _lib.is_hostsim(): return None
	return ctypes.c_void_p(_torch().cuda.current_stream().cuda_stream)

# host-array route (pixell_amd/hostio.py): the Pipeline of the API call in progress ON THIS THREAD, if any.  Per thread: a call on
# another thread must neither see this one's pipeline (its write-backs would complete on a queue this call may already have closed)
# nor be kept from opening its own.
_tls = threading.local()
def _pipe(): return getattr(_tls, "pipe", None)
@contextlib.contextmanager
def host_pipeline(inputs=(), outputs=()):
	"""for the duration of one API call: numpy `inputs` are uploaded by a background thread, in order, starting now; numpy outputs
	of the transforms issued inside are downloaded in the background; everything is complete when the block exits"""
	from . import hostio
	if _pipe() is not None or _lib.is_hostsim() or not any(hostio.eligible(a) for a in list(inputs)+list(outputs)):
		yield; return
	_tls.pipe = hostio.Pipeline(); _tls.pipe.prefetch(inputs)
	try: yield
	finally:
		p = _tls.pipe; _tls.pipe = None; p.close()

class _Buf:
	"""device view of a numpy array or torch tensor (contiguous), with optional write-back.  overwrite: the call writes every
	element (an output whose old content need not travel to the device)"""
	def __init__(self, arr, writeback=False, overwrite=False):
		self.arr = arr; self.writeback = writeback; self.tmp = None; self.keep = None; self.slab = False
		if _is_tensor(arr):
			if not arr.is_cuda and not _lib.is_hostsim(): raise ValueError("torch tensors passed to pixell_amd must live on the GPU")
			if not arr.is_contiguous():
				if writeback: raise ValueError("output tensors must be contiguous")
				arr = arr.contiguous()
			self.keep = arr; self.ptr = arr.data_ptr()
		else:
			if _lib.is_hostsim():
				if arr.flags.c_contiguous and arr.dtype.isnative: self.keep = arr
				else: self.keep = np.ascontiguousarray(arr); self.tmp = self.keep
				self.ptr = self.keep.ctypes.data
			else:
				torch = _torch()
				from . import hostio
				pipe = _pipe(); got = pipe.take(arr) if pipe is not None else None
				self.slab = hostio.eligible(arr)
				if got is not None:                      # uploaded in the background since the call began
					self.tmp, ev = got; torch.cuda.current_stream().wait_event(ev)
					self.tmp.record_stream(torch.cuda.current_stream())      # (allocated by the background thread: tell the allocator which stream consumes it)
				elif self.slab and writeback and overwrite:
					self.tmp = torch.empty(arr.shape, dtype=getattr(torch, arr.dtype.name), device="cuda")
				elif self.slab:
					self.tmp, ev = hostio.upload(arr); torch.cuda.current_stream().wait_event(ev)
				else: self.tmp = torch.from_numpy(np.ascontiguousarray(arr)).cuda()
				self.ptr = self.tmp.data_ptr()
	def finish(self):
		if not self.writeback or self.tmp is None: return
		if _lib.is_hostsim(): self.arr[...] = self.tmp
		elif self.slab:
			from . import hostio
			ev = _torch().cuda.Event(); ev.record()
			pipe = _pipe()
			if pipe is not None: pipe.writeback(self.tmp, self.arr, ev)      # complete when the host_pipeline block (of this thread) exits
			else: hostio.download(self.tmp, self.arr, after=ev)
		else: self.arr[...] = self.tmp.cpu().numpy()

class Plan:
	"""RAII wrapper of pxs_plan"""
	def __init__(self, handle):
		self.handle = handle; self.lock = threading.RLock()      # (option + call pairs on a shared cached plan are atomic per thread)
		self._det_default = int(os.environ.get("PXS_DETERMINISTIC", "0") not in ("", "0"))      # what the library gave the plan when it was made
	def __del__(self):
		try:
			if self.handle: _lib.load().pxs_plan_destroy(self.handle); self.handle = None
		except Exception: pass
	def profile(self, enable=True):
		_lib.check(_lib.load().pxs_profile(self.handle, int(bool(enable))))
	def profile_read(self, reset=True):
		"""{stage: (total ms, launches)} from the hipEvent stage timers (pxsht.h PXS_STAGE_*)"""
		ms = (ctypes.c_double*4)(); cnt = (ctypes.c_int*4)()
		_lib.check(_lib.load().pxs_profile_read(self.handle, ms, cnt, int(bool(reset))))
		names = ["leg_syn", "leg_ana", "ring_fft", "resample"]
		return {n: (ms[i], cnt[i]) for i, n in enumerate(names)}
	def profile_flops(self, reset=True):
		"""FP64 flops the Legendre kernels executed while profiling was on: (synthesis, analysis) (pxs_profile_flops)"""
		f = (ctypes.c_double*2)()
		_lib.check(_lib.load().pxs_profile_flops(self.handle, f, int(bool(reset))))
		return f[0], f[1]
	def set_option(self, name, value):
		"""pxs_plan_option: "analysis" = 2 (ducc0's route, the default) | 0 (full theta-interpolant) | 1 (ring weights + adjoint synthesis where ntheta >= 2 lmax + 2);
		"deterministic" = 0 | 1 (ordered, bitwise repeatable analysis sums; see set_deterministic)"""
		_lib.check(_lib.load().pxs_plan_option(self.handle, name.encode(), int(value)))
	def query(self, name):
		"""pxs_plan_query: "analysis_form", "ncc_circle", "ducc_ncc_circle", "theta_line" """
		v = ctypes.c_int64()
		_lib.check(_lib.load().pxs_plan_query(self.handle, name.encode(), ctypes.byref(v)))
		return int(v.value)
	def info(self):
		a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
		_lib.check(_lib.load().pxs_plan_info(self.handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
		return dict(nring_syn=a.value, nring_ana=b.value, scratch_bytes=c.value)

class _PlanCache:
	"""least-recently-used cache of pxs_plan handles: a plan owns device scratch (tens of GB at the largest configurations), so a
	sweep over geometries or band limits must not keep every plan alive.  PIXELL_AMD_MAX_PLANS (default 4) bounds the count; an
	evicted plan is destroyed (its kernels are stream-ordered before the free)."""
	def __init__(self):
		import collections
		self.d = collections.OrderedDict()
		self.cap = max(1, int(os.environ.get("PIXELL_AMD_MAX_PLANS", "4")))
	def get(self, key):
		p = self.d.get(key)
		if p is not None: self.d.move_to_end(key)
		return p
	def __setitem__(self, key, plan):
		self.d[key] = plan; self.d.move_to_end(key)
		while len(self.d) > self.cap: self.d.popitem(last=False)
	def clear(self): self.d.clear()
	def __len__(self): return len(self.d)
_plans = _PlanCache()
def clear_plans(release=False):
	"""drop the cached plans.  Their device scratch goes to the library's arena (pxs_memory) for the next plans; release=True returns it to the driver."""
	_plans.clear()
	if release: memory(release=True)

def memory(release=False):
	"""pxs_memory: dict(malloc_ms, malloc_calls, malloc_bytes, arena_hits, arena_bytes, live_bytes); release=True empties the arena first"""
	import gc; gc.collect()      # (plans dropped a moment ago have released their buffers)
	st = (ctypes.c_double*6)()
	_lib.check(_lib.load().pxs_memory(int(bool(release)), st))
	return dict(malloc_ms=st[0], malloc_calls=int(st[1]), malloc_bytes=int(st[2]), arena_hits=int(st[3]), arena_bytes=int(st[4]), live_bytes=int(st[5]))

_deterministic = None
def set_deterministic(on=True):
	"""Bitwise repeatable transforms (pxs_plan_option "deterministic", include/pxsht.h): by default the Legendre analysis adds the
	contributions of the ring chunks of an m with atomic adds in arrival order, so map2alm / alm2map_adjoint repeat from run to run
	to ~1e-14 only.  on=True applies the ordered sums to every plan used from now on (cached plans included); None restores the
	default (PXS_DETERMINISTIC as the plan was made)."""
	global _deterministic
	_deterministic = None if on is None else bool(on)
def _apply_mode(plan):
	"""the process-wide deterministic switch onto a (cached, shared) plan.  Called with plan.lock held, in the same critical section as the
	transform it applies to (_run_syn / _run_ana): a toggle by another thread cannot land between an option and its call."""
	want = _deterministic
	if want is None and getattr(plan, "_det", None) is not None:      # a cached plan that was switched: back to the default it was made with
		plan.set_option("deterministic", plan._det_default); plan._det = None
	elif want is not None and getattr(plan, "_det", None) != want:
		plan.set_option("deterministic", int(want)); plan._det = want
	return plan

def tri_mstart(lmax, mmax=None):
	if mmax is None: mmax = lmax
	m = np.arange(mmax+1, dtype=np.int64)
	return (m*(2*lmax+1-m)//2).astype(np.uint64)

def grid_plan(geometry, ntheta, nphi, phi0, flip, lmax, mmax, mstart, lstride=1):
	"""cached pxs_plan (a plan owns its device scratch: one call at a time per plan)"""
	ms = np.ascontiguousarray(np.asarray(mstart)[:mmax+1], dtype=np.uint64)
	key = ("g", geometry, int(ntheta), int(nphi), float(phi0), bool(flip[0]), bool(flip[1]), int(lmax), int(mmax), ms.tobytes(), int(lstride), device_index())
	p = _plans.get(key)
	if p is None:
		h = ctypes.c_void_p()
		_lib.check(_lib.load().pxs_plan_grid2d(ctypes.byref(h), geometry.encode(), int(ntheta), int(nphi), float(phi0),
			int(bool(flip[0])), int(bool(flip[1])), int(lmax), int(mmax), ms.ctypes.data, int(lstride), device_index()))
		p = Plan(h); _plans[key] = p
	return p

def ring_plan(theta, nphi, phi0, ringstart, lmax, mmax, mstart, lstride=1, pixstride=1):
	th = np.ascontiguousarray(theta, dtype=np.float64); nph = np.ascontiguousarray(nphi, dtype=np.uint64)
	p0 = np.ascontiguousarray(phi0, dtype=np.float64); rs = np.ascontiguousarray(ringstart, dtype=np.uint64)
	ms = np.ascontiguousarray(np.asarray(mstart)[:mmax+1], dtype=np.uint64)
	key = ("r", th.tobytes(), nph.tobytes(), p0.tobytes(), rs.tobytes(), int(pixstride), int(lmax), int(mmax), ms.tobytes(), int(lstride), device_index())
	p = _plans.get(key)
	if p is None:
		h = ctypes.c_void_p()
		_lib.check(_lib.load().pxs_plan_rings(ctypes.byref(h), len(th), th.ctypes.data, nph.ctypes.data, p0.ctypes.data, rs.ctypes.data,
			int(pixstride), int(lmax), int(mmax), ms.ctypes.data, int(lstride), device_index()))
		p = Plan(h); _plans[key] = p
	return p

def _ncomp(spin, mode):
	if mode == "DERIV1": return 1, 2
	return (1, 1) if spin == 0 else (2, 2)

def _check_pair(alm, map, spin, mode, pixdims):
	nca, ncm = _ncomp(spin, mode)
	if mode not in ("STANDARD", "DERIV1"): raise ValueError("unknown mode '%s'" % str(mode))
	if alm.ndim == 3 and map.ndim == 2+pixdims:          # a batch of independent maps (ours; ducc takes one map per call)
		if alm.shape[0] != map.shape[0]: raise ValueError("batched call: alm and map disagree on the batch size")
		alm, map = alm[0], map[0]
	if alm.ndim != 2 or alm.shape[0] != nca: raise ValueError("alm must have shape [%d,nelem] for spin %d mode %s" % (nca, spin, mode))
	if map.ndim != 1+pixdims or map.shape[0] != ncm: raise ValueError("map must have %d components" % ncm)
	ad, md = _np_dtype(alm), _np_dtype(map)
	if ad not in (np.complex64, np.complex128) or md not in (np.float32, np.float64): raise TypeError("alm must be complex, map real")
	return ad, md

class _View:
	"""device pointer + (batch, component) strides of alm [nb?, nc, nelem] / map [nb?, nc, pixels...].  CUDA tensors are used in
	place whenever their inner axes are contiguous (the batch and component axes may be strided views, e.g. one spin group of
	many maps); numpy arrays and other tensors are staged contiguously (and written back for outputs)."""
	def __init__(self, arr, inner_ndim, batched, writeback, overwrite=False):
		self.buf = None
		if _is_tensor(arr) and (arr.is_cuda or _lib.is_hostsim()):
			st = list(arr.stride()); sh = list(arr.shape)
			inner_ok = all(st[-k] == int(np.prod(sh[len(sh)-k+1:], dtype=np.int64)) for k in range(1, inner_ndim+1))
			if inner_ok:
				self.keep = arr; self.ptr = arr.data_ptr()
				self.cstride = st[-inner_ndim-1]; self.bstride = st[0] if batched else 0
				return
		self.buf = _Buf(arr, writeback=writeback, overwrite=overwrite); self.ptr = self.buf.ptr
		inner = int(np.prod(arr.shape[arr.ndim-inner_ndim:], dtype=np.int64))
		self.cstride = inner; self.bstride = inner*arr.shape[-inner_ndim-1] if batched else 0
	def finish(self):
		if self.buf is not None: self.buf.finish()

def _run_syn(plan, alm, map, spin, mode, adjoint, map_overwrite=False):
	"""alm [nca, nelem], map [ncm, ...] -- or a batch of independent maps alm [nb, nca, nelem], map [nb, ncm, ...] in ONE library call"""
	ad, md = _np_dtype(alm), _np_dtype(map)
	batched = alm.ndim == 3
	nb = alm.shape[0] if batched else 1
	av = _View(alm, 1, batched, bool(adjoint)); mv = _View(map, map.ndim-(2 if batched else 1), batched, not adjoint, overwrite=map_overwrite and not adjoint)
	with plan.lock:      # (the adjoint synthesis runs the Legendre analysis: the deterministic option and the call as one step)
		_apply_mode(plan)
		_lib.check(_lib.load().pxs_synthesis(plan.handle, int(spin), 1 if mode == "DERIV1" else 0, int(bool(adjoint)), int(nb),
			av.ptr, _DT[ad], av.cstride, av.bstride, mv.ptr, _DT[md], mv.cstride, mv.bstride, current_stream()))
	av.finish(); mv.finish()

ANALYSIS_MODES = {"ducc0": 2, "interpolant": 0, "weights": 1}
def _analysis_mode(analysis):
	"""analysis=None: PIXELL_AMD_ANALYSIS or "ducc0" (the route of ducc0's analysis_2d as published, see include/pxsht.h pxs_plan_option)"""
	if analysis is None: analysis = os.environ.get("PIXELL_AMD_ANALYSIS", "ducc0")
	if analysis not in ANALYSIS_MODES: raise ValueError("analysis must be 'ducc0', 'interpolant' or 'weights', not %r" % (analysis,))
	return ANALYSIS_MODES[analysis]

def analysis_form(geometry, ntheta, nphi, lmax, mmax=None, mstart=None, phi0=0.0, lstride=1, flip=(False, False), analysis=None):
	"""What analysis_2d does on this grid with this option (pxs_plan_query): dict(form="ducc0" (its fine-CC form) | "weights" |
	"interpolant", ncc_circle=N_cc of the Legendre-stage CC grid (0: none), ducc_ncc_circle=ducc0's own N_cc for this lmax)"""
	if mmax is None: mmax = lmax
	if mstart is None: mstart = tri_mstart(lmax, mmax)
	plan = grid_plan(geometry, ntheta, nphi, phi0, flip, lmax, mmax, mstart, lstride)
	plan.set_option("analysis", _analysis_mode(analysis))
	return dict(form={0: "interpolant", 1: "weights", 2: "ducc0"}[plan.query("analysis_form")], ncc_circle=plan.query("ncc_circle"), ducc_ncc_circle=plan.query("ducc_ncc_circle"))

def _run_ana(plan, map, alm, spin, adjoint, analysis=None, alm_dense=False):
	ad, md = _np_dtype(alm), _np_dtype(map)
	batched = alm.ndim == 3
	nb = alm.shape[0] if batched else 1
	av = _View(alm, 1, batched, not adjoint, overwrite=alm_dense and not adjoint); mv = _View(map, map.ndim-(2 if batched else 1), batched, bool(adjoint), overwrite=bool(adjoint))      # (a grid plan writes every pixel)
	with plan.lock:      # the options and the call they apply to, as one step: plans are cached and shared between threads
		_apply_mode(plan)
		plan.set_option("analysis", _analysis_mode(analysis))
		_lib.check(_lib.load().pxs_analysis(plan.handle, int(spin), int(bool(adjoint)), int(nb), mv.ptr, _DT[md], mv.cstride, mv.bstride,
			av.ptr, _DT[ad], av.cstride, av.bstride, current_stream()))
	av.finish(); mv.finish()

def _grid_args(alm, map, spin, lmax, mmax, mstart, geometry, phi0, lstride, mode, flip):
	_check_pair(alm, map, spin, mode, 2)
	if mmax is None: mmax = lmax
	if mstart is None: mstart = tri_mstart(lmax, mmax)
	nt, nph = map.shape[-2:]
	return grid_plan(geometry, nt, nph, phi0, flip, lmax, mmax, mstart, lstride)

def synthesis_2d(*, alm, map, spin, lmax, geometry, mmax=None, mstart=None, phi0=0.0, nthreads=0, lstride=1, mode="STANDARD", flip=(False, False), return_plan=False):
	"""ducc0.sht.experimental.synthesis_2d as called at curvedsky.py:907-924"""
	plan = _grid_args(alm, map, spin, lmax, mmax, mstart, geometry, phi0, lstride, mode, flip)
	_run_syn(plan, alm, map, spin, mode, False, map_overwrite=True)      # (a grid plan writes every pixel of the map)
	return plan if return_plan else map

def adjoint_synthesis_2d(*, alm, map, spin, lmax, geometry, mmax=None, mstart=None, phi0=0.0, nthreads=0, lstride=1, mode="STANDARD", flip=(False, False), return_plan=False):
	plan = _grid_args(alm, map, spin, lmax, mmax, mstart, geometry, phi0, lstride, mode, flip)
	_run_syn(plan, alm, map, spin, mode, True)
	return plan if return_plan else alm

def analysis_2d(*, alm, map, spin, lmax, geometry, mmax=None, mstart=None, phi0=0.0, nthreads=0, lstride=1, flip=(False, False), return_plan=False, analysis=None):
	"""ducc0.sht.experimental.analysis_2d as called at curvedsky.py:1032-1046.
	analysis (ours): "ducc0" (default): the route ducc0's analysis_2d takes as published -- the theta-interpolant of the rings, low-passed
	to |k| < N_cc where the grid is finer, evaluated on the CC grid of N_cc + 1 rings and integrated with that grid's weights (a CC
	grid with ntheta >= 2 lmax + 2: its own weights directly) | "interpolant": exact quadrature of the full interpolant |
	"weights": ring quadrature weights + adjoint synthesis, the reference's cyl route (curvedsky.py:852-861, 1068-1084), on grids
	with ntheta >= 2 lmax + 2.  The same alm for band-limited maps (include/pxsht.h, pxs_plan_option)."""
	plan = _grid_args(alm, map, spin, lmax, mmax, mstart, geometry, phi0, lstride, "STANDARD", flip)
	# (an alm array that is exactly the triangular layout is written in full: a host array of it need not be uploaded first)
	dense = lstride == 1 and (mmax is None or mmax == lmax) and alm.shape[-1] == (lmax+1)*(lmax+2)//2 and (mstart is None or int(np.asarray(mstart)[-1]) + lmax + 1 == alm.shape[-1])
	_run_ana(plan, map, alm, spin, False, analysis, alm_dense=dense)
	return plan if return_plan else alm

def adjoint_analysis_2d(*, alm, map, spin, lmax, geometry, mmax=None, mstart=None, phi0=0.0, nthreads=0, lstride=1, flip=(False, False), return_plan=False, analysis=None):
	plan = _grid_args(alm, map, spin, lmax, mmax, mstart, geometry, phi0, lstride, "STANDARD", flip)
	_run_ana(plan, map, alm, spin, True, analysis)
	return plan if return_plan else map

def _ring_args(alm, map, theta, nphi, phi0, ringstart, lmax, mmax, mstart, spin, lstride, pixstride, mode):
	if mmax is None: mmax = lmax
	if mstart is None: mstart = tri_mstart(lmax, mmax)
	return ring_plan(theta, nphi, phi0, ringstart, lmax, mmax, mstart, lstride, pixstride), mmax, mstart

def synthesis(*, alm, theta, nphi, phi0, ringstart, lmax, mmax=None, mstart=None, spin=0, map=None, lstride=1, pixstride=1, nthreads=0, mode="STANDARD"):
	"""ducc0.sht.experimental.synthesis as called at curvedsky.py:936-960 (map[nc, npix])"""
	plan, mmax, mstart = _ring_args(alm, map, theta, nphi, phi0, ringstart, lmax, mmax, mstart, spin, lstride, pixstride, mode)
	nca, ncm = _ncomp(spin, mode)
	if map is None:
		rs_ = np.asarray(ringstart).astype(np.int64); last_ = rs_+(np.asarray(nphi).astype(np.int64)-1)*pixstride
		npix = int(max(rs_.max(), last_.max())+1)
		rdt = np.float32 if _np_dtype(alm) == np.complex64 else np.float64
		map = _torch().zeros((ncm, npix), dtype=getattr(_torch(), np.dtype(rdt).name), device=alm.device) if _is_tensor(alm) else np.zeros((ncm, npix), rdt)
	_check_pair(alm, map, spin, mode, 1)
	_run_syn(plan, alm, map, spin, mode, False)
	synthesis.last_plan = plan
	return map

def adjoint_synthesis(*, map, theta, nphi, phi0, ringstart, lmax, mmax=None, mstart=None, spin=0, alm=None, lstride=1, pixstride=1, nthreads=0, mode="STANDARD"):
	"""ducc0.sht.experimental.adjoint_synthesis as called at curvedsky.py:1068-1084"""
	plan, mmax, mstart = _ring_args(alm, map, theta, nphi, phi0, ringstart, lmax, mmax, mstart, spin, lstride, pixstride, mode)
	nca, ncm = _ncomp(spin, mode)
	if alm is None:
		nelem = int(np.max(np.asarray(mstart).astype(np.int64))+lmax*lstride+1)
		cdt = np.complex64 if _np_dtype(map) == np.float32 else np.complex128
		alm = _torch().zeros((nca, nelem), dtype=getattr(_torch(), np.dtype(cdt).name), device=map.device) if _is_tensor(map) else np.zeros((nca, nelem), cdt)
	_check_pair(alm, map, spin, mode, 1)
	_run_syn(plan, alm, map, spin, mode, True)
	adjoint_synthesis.last_plan = plan
	return alm

_gridweights_cache = {}
def get_gridweights(geometry, ntheta):
	"""ducc0.sht.experimental.get_gridweights (curvedsky.py:501, 855); sum = 4 pi.  The last few results are kept: quad_weights asks
	for the same grid on every map2alm of a declination band"""
	key = (str(geometry), int(ntheta))
	out = _gridweights_cache.get(key)
	if out is None:
		out = np.zeros(int(ntheta), np.float64)
		_lib.check(_lib.load().pxs_gridweights(geometry.encode(), int(ntheta), out.ctypes.data))
		if len(_gridweights_cache) >= 8: _gridweights_cache.pop(next(iter(_gridweights_cache)))
		_gridweights_cache[key] = out
	return out.copy()

def grid_maxlmax(geometry, ntheta):
	return int(_lib.load().pxs_grid_maxlmax(geometry.encode(), int(ntheta)))
