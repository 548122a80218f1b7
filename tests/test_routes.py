"""Replay of the integration routes recorded by tests/golden/make_routes.py (which ran the REFERENCE's own curvedsky / fft
drivers in the build container on top of pixell_amd and on top of the oracle / numpy engine and asserted equality there).

Route B2: every call pixell.curvedsky made into `ducc0.sht.experimental` -- keyword names and values exactly as recorded at
the call sites (pixell/curvedsky.py:907-924, 936-960, 1032-1046, 1068-1084, 501) -- is issued to pixell_amd.sht and compared
with what the oracle returned for it; the top-level calls (alm2map / map2alm / adjoints / deriv on 2d, cyl and partial
geometries) are issued to pixell_amd.curvedsky and compared with reference-over-oracle results.
Route B3: the engine protocol object fft.engines["hip"].FFTW(a, b, axes, direction, threads, flags)(normalise_idft) driven the
way pixell.fft.fft / ifft / rfft / irfft drive it (pixell/fft.py:133-209), against the reference's numpy engine outputs,
including caller-supplied non-contiguous output views (np.shares_memory)."""
import os, json, types
import numpy as np
import pytest
from pixell_amd import sht, curvedsky, enmap, fft as pfft
from pixell_amd.wcs import CarWCS

def _real_m0(a, lmax):
	a = np.array(a); a[..., :lmax+1] = a[..., :lmax+1].real; return a

def b2_boundary(golden_dir):
	d = np.load(os.path.join(golden_dir, "routes_b2.npz"))
	calls = json.loads(str(d["calls_json"]))
	assert len(calls) >= 40
	seen = set()
	for i, c in enumerate(calls):
		name, kw = c["name"], dict(c["scalars"])
		ref = d["call%03d_out" % i]
		seen.add(name)
		if name == "get_gridweights":
			out = sht.get_gridweights(kw["geometry"], kw["ntheta"])
			assert np.max(np.abs(out-ref)) < 1e-13; continue
		for k in c["arrays"]: kw[k] = np.array(d["call%03d_in_%s" % (i, k)])
		out = getattr(sht, name)(**kw)                       # keywords exactly as pixell passes them (incl. nthreads, mode, mstart, ...)
		assert out.shape == ref.shape and out.dtype == ref.dtype, name
		if np.iscomplexobj(ref): out, ref = _real_m0(out, kw["lmax"]), _real_m0(ref, kw["lmax"])
		assert np.max(np.abs(out-ref)) < 1e-11*max(np.max(np.abs(ref)), 1e-300), (name, c["scalars"])
	assert {"synthesis_2d", "adjoint_synthesis_2d", "analysis_2d", "adjoint_analysis_2d", "synthesis", "adjoint_synthesis", "get_gridweights"} <= seen

def b2_toplevel(golden_dir):
	d = np.load(os.path.join(golden_dir, "routes_b2.npz"))
	for case in json.loads(str(d["cases_json"])):
		g = lambda k: d["case_%s__%s" % (case, k)]
		shape = tuple(int(v) for v in g("shape")); wcs = CarWCS(g("cdelt"), g("crval"), g("crpix"))
		ref = g("out"); kind = case.rsplit("_", 1)[0]
		if kind == "alm2map":
			out = curvedsky.alm2map(np.array(g("alm")), enmap.zeros((3,)+shape, wcs), spin=list(g("spin")))
		elif kind == "alm2map_adjoint":
			out = curvedsky.alm2map_adjoint(enmap.ndmap(np.array(g("map")), wcs), spin=list(g("spin")), ainfo=curvedsky.alm_info(int(g("lmax"))))
		elif kind == "map2alm":
			kw = dict(niter=int(g("niter"))) if "case_%s__niter" % case in d else {}
			out = curvedsky.map2alm(enmap.ndmap(np.array(g("map")), wcs), lmax=int(g("lmax")), spin=list(g("spin")), **kw)
		elif kind == "map2alm_adjoint":
			out = curvedsky.map2alm_adjoint(np.array(g("alm")), enmap.zeros((3,)+shape, wcs), spin=list(g("spin")))
		elif kind == "deriv":
			out = curvedsky.alm2map(np.array(g("alm")), enmap.zeros((2,)+shape, wcs), deriv=True)
		else: raise AssertionError(case)
		out = np.asarray(out)
		assert out.shape == ref.shape, case
		if np.iscomplexobj(ref): out, ref = _real_m0(out, 20), _real_m0(ref, 20)
		assert np.max(np.abs(out-ref)) < 1e-10*np.max(np.abs(ref)), case

def b3_engine(golden_dir):
	d = np.load(os.path.join(golden_dir, "routes_b3.npz"))
	fake = types.SimpleNamespace(engines={}, engine="numpy")
	fake.set_engine = lambda e: setattr(fake, "engine", e)
	eng = pfft.register(fake)                                # what a pixell maintainer calls on pixell.fft
	assert fake.engines["hip"] is eng and fake.engine == "hip"
	e = eng.empty_aligned((3, 5), np.complex128, n=32)
	assert isinstance(e, np.ndarray) and e.shape == (3, 5) and e.dtype == np.complex128
	for i, c in enumerate(json.loads(str(d["cases_json"]))):
		a = np.array(d["b3_%02d_in" % i]); ref = d["b3_%02d_out" % i]
		fun, kw = c["fun"], c["kw"]; axes = kw.get("axes", [-1])
		# the output array the reference's driver would allocate or be handed (pixell/fft.py:147-155, 176-183, 190-193, 203-207)
		if c["view"] is not None:
			full = np.full(c["view"]["shape"], -7.0, dtype=c["view"]["dtype"])
			b = full[tuple(slice(*s) for s in c["view"]["sel"])]
		else:
			if fun == "fft":     b = eng.empty_aligned(a.shape, np.result_type(a.dtype, 0j))
			elif fun == "ifft":  b = eng.empty_aligned(a.shape, a.dtype)
			elif fun == "rfft":  b = eng.empty_aligned(pfft.rfft_shape(a.shape, axes), np.result_type(a.dtype, 0j))
			else:                b = eng.empty_aligned(pfft.irfft_shape(a.shape, axes, kw.get("n")), np.zeros([], a.dtype).real.dtype)
			full = b
		direction = "FFTW_FORWARD" if fun in ("fft", "rfft") else "FFTW_BACKWARD"
		plan = eng.FFTW(a, b, axes=axes, direction=direction, threads=4, flags=["FFTW_ESTIMATE"])
		if direction == "FFTW_FORWARD": res = plan()
		else: res = plan(normalise_idft=bool(kw.get("normalize", False)))
		assert np.shares_memory(b, full)
		tol = 1e-5 if a.dtype in (np.complex64, np.float32) else 1e-12
		assert full.shape == ref.shape and np.max(np.abs(full-ref)) < tol*np.max(np.abs(ref)), (i, c)
	# r2r through the protocol (FFTW direction list, pixell/fft.py:211-267): DCT-II / DST-III against scipy
	import scipy.fft as sfft
	x = np.random.default_rng(1).standard_normal((4, 18)); y = np.empty_like(x)
	eng.FFTW(x, y, axes=[-1], direction=["FFTW_REDFT10"], threads=1)()
	assert np.max(np.abs(y-sfft.dct(x, type=2, axis=-1))) < 1e-12
	eng.FFTW(x, y, axes=[-1], direction=["FFTW_RODFT01"], threads=1)()
	assert np.max(np.abs(y-sfft.dst(x, type=3, axis=-1))) < 1e-12
	with pytest.raises(ValueError): eng.FFTW(x, y, axes=[-2, -1], direction=["FFTW_REDFT10", "FFTW_RODFT10"])

@pytest.mark.hostsim
def test_route_b2_boundary_hostsim(golden_dir): b2_boundary(golden_dir)
@pytest.mark.gpu
def test_route_b2_boundary_gpu(golden_dir): b2_boundary(golden_dir)
@pytest.mark.hostsim
def test_route_b2_toplevel_hostsim(golden_dir): b2_toplevel(golden_dir)
@pytest.mark.gpu
def test_route_b2_toplevel_gpu(golden_dir): b2_toplevel(golden_dir)
@pytest.mark.hostsim
def test_route_b3_engine_hostsim(golden_dir): b3_engine(golden_dir)
@pytest.mark.gpu
def test_route_b3_engine_gpu(golden_dir): b3_engine(golden_dir)
