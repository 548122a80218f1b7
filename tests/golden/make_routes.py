"""Exercise the two integration routes of INTEGRATION.md against the REFERENCE in this container and record fixtures.

Run:  python tests/golden/make_routes.py     (needs /root/reference; never run on the GPU box)

Route B2 -- stock pixell.curvedsky over pixell_amd.sht:  the reference's own curvedsky module is imported (stub astropy
  WCS, see _ref_harness.py) TWICE: once with the long-double oracle and once with pixell_amd.sht (kernels running in the
  test-only host simulator) mounted as `ducc0.sht.experimental`.  curvedsky.alm2map / map2alm / their adjoints / deriv are
  run on 2d, cyl and partial geometries.  Every call that reaches the ducc boundary is recorded: function name, the keyword
  names and scalar values exactly as pixell passes them, the array arguments and the result.  The script asserts that the
  product gives what the oracle gives at the boundary and at the top level, then saves
      routes_b2.npz   boundary calls (kwargs + arrays in/out from the ORACLE run) and top-level inputs/outputs
  so that the -m gpu test replays the same keyword calls through pixell_amd.sht on the MI355X and the same top-level calls
  through pixell_amd.curvedsky.
Route B3 -- pixell.fft.engines["hip"]:  the reference's pixell.fft is imported, pixell_amd.fft.register() installs the engine,
  and the reference's OWN fft.fft / ifft / rfft / irfft drivers (pixell/fft.py:133-209) are run with engine="hip" and with
  engine="numpy" on the same inputs, including caller-supplied non-contiguous output views.  Saved as routes_b3.npz
  (inputs, call descriptions, numpy-engine outputs) for the GPU replay through the engine protocol object.
Only arrays and call descriptions are stored; no reference code.
"""
import sys, os, json, types
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "hostsim"))
import build_hostsim; build_hostsim.build()
sys.path.insert(0, os.path.join(HERE, ".."))
import conftest; conftest.use_hostsim()      # the build container has no GPU: the kernels run in the test-only simulator
from oracle import sht_oracle as so
from pixell_amd import sht as psht, fft as pfft_amd
import _ref_harness as H

NAMES = ["synthesis_2d", "adjoint_synthesis_2d", "analysis_2d", "adjoint_analysis_2d", "synthesis", "adjoint_synthesis", "get_gridweights"]

class Recorder(types.ModuleType):
	"""ducc0.sht.experimental stand-in: forwards to `backend` and logs every call"""
	def __init__(self, backend):
		types.ModuleType.__init__(self, "ducc0.sht.experimental")
		self.calls = []; self.backend = backend
		for n in NAMES: setattr(self, n, self._wrap(n))
	def _wrap(self, name):
		fn = getattr(self.backend, name)
		def call(*args, **kw):
			rec = dict(name=name, scalars={}, arrays_in={})
			if name == "get_gridweights":
				rec["scalars"] = dict(geometry=str(args[0]), ntheta=int(args[1]))
				out = fn(*args); rec["out"] = np.array(out); self.calls.append(rec); return out
			assert not args, "pixell calls ducc with keywords only"
			for k, v in kw.items():
				if isinstance(v, np.ndarray): rec["arrays_in"][k] = np.array(v)
				else: rec["scalars"][k] = v if isinstance(v, str) else (float(v) if isinstance(v, float) else int(v))
			out = fn(**kw)
			rec["out"] = np.array(out); self.calls.append(rec)
			return out
		return call

def run_cases(ns, log):
	"""the same top-level calls for both backends; returns {case: dict(inputs..., out=...)}"""
	enmap, curvedsky = ns.enmap, ns.curvedsky
	rng = np.random.default_rng(11)
	res = {}
	def geo(kind):
		if kind == "f1":   return enmap.fullsky_geometry(shape=(26, 48))
		if kind == "cc":   return enmap.fullsky_geometry(shape=(25, 48), variant="cc")
		if kind == "band": return enmap.band_geometry(np.deg2rad(40), shape=(30, 60))
		if kind == "patch":
			s, w = enmap.fullsky_geometry(shape=(24, 48))
			w = w.deepcopy(); w.wcs.crpix[0] -= 5; w.wcs.crpix[1] -= 3
			return (14, 30), w
	lmax = 20
	def wcsd(w): return dict(cdelt=np.array(w.wcs.cdelt), crval=np.array(w.wcs.crval), crpix=np.array(w.wcs.crpix))
	alm3 = so.rand_alm_simple(lmax, 3, 4, spin=(0, 2)); alm1 = so.rand_alm_simple(lmax, 1, 5, spin=(0,))
	for kind in ["f1", "cc", "band", "patch"]:
		shape, wcs = geo(kind); shape = tuple(int(v) for v in shape[-2:])
		m = enmap.zeros((3,)+shape, wcs)
		log.append("alm2map:"+kind); out = curvedsky.alm2map(alm3.copy(), m, spin=[0, 2])
		res["alm2map_"+kind] = dict(alm=alm3, shape=np.array(shape), spin=np.array([0, 2]), out=np.array(out), **wcsd(wcs))
		pix = enmap.ndmap(rng.standard_normal((3,)+shape), wcs)
		log.append("alm2map_adjoint:"+kind); at = curvedsky.alm2map_adjoint(pix.copy(), spin=[0, 2], ainfo=curvedsky.alm_info(lmax))
		res["alm2map_adjoint_"+kind] = dict(map=np.array(pix), shape=np.array(shape), spin=np.array([0, 2]), lmax=lmax, out=np.array(at), **wcsd(wcs))
		if kind in ("f1", "cc"):
			log.append("map2alm:"+kind); a = curvedsky.map2alm(pix.copy(), lmax=lmax, spin=[0, 2])
			res["map2alm_"+kind] = dict(map=np.array(pix), shape=np.array(shape), spin=np.array([0, 2]), lmax=lmax, out=np.array(a), **wcsd(wcs))
			log.append("map2alm_adjoint:"+kind); ma = curvedsky.map2alm_adjoint(alm3.copy(), enmap.zeros((3,)+shape, wcs), spin=[0, 2])
			res["map2alm_adjoint_"+kind] = dict(alm=alm3, shape=np.array(shape), spin=np.array([0, 2]), out=np.array(ma), **wcsd(wcs))
		else:
			log.append("map2alm_cyl:"+kind); a = curvedsky.map2alm(pix.copy(), lmax=lmax, spin=[0, 2], niter=1)
			res["map2alm_"+kind] = dict(map=np.array(pix), shape=np.array(shape), spin=np.array([0, 2]), lmax=lmax, niter=1, out=np.array(a), **wcsd(wcs))
		md = enmap.zeros((2,)+shape, wcs)
		log.append("deriv:"+kind); d = curvedsky.alm2map(alm1[0].copy(), md, deriv=True)
		res["deriv_"+kind] = dict(alm=alm1[0], shape=np.array(shape), out=np.array(d), **wcsd(wcs))
	return res

def route_b2():
	rec_o = Recorder(so); ns = H.load_reference(rec_o)
	log_o = []; res_o = run_cases(ns, log_o)
	# second import of the reference modules with the product mounted
	for k in [k for k in sys.modules if k == "pixell" or k.startswith("pixell.")]: del sys.modules[k]
	rec_p = Recorder(psht); ns2 = H.load_reference(rec_p)
	log_p = []; res_p = run_cases(ns2, log_p)
	assert log_o == log_p and len(rec_o.calls) == len(rec_p.calls)
	worst = 0.0
	for a, b in zip(rec_o.calls, rec_p.calls):
		assert a["name"] == b["name"] and a["scalars"] == b["scalars"] and sorted(a["arrays_in"]) == sorted(b["arrays_in"]), (a["name"], a["scalars"], b["scalars"])
		scale = max(np.max(np.abs(a["out"])), 1e-300)
		oa, ob = a["out"], b["out"]
		if a["name"] in ("adjoint_synthesis_2d", "adjoint_synthesis", "analysis_2d"):
			oa = oa.copy(); ob = ob.copy()
			l = a["scalars"]["lmax"]; oa[..., :l+1] = oa[..., :l+1].real; ob[..., :l+1] = ob[..., :l+1].real   # Im a_l0 is not defined by a real map
		err = np.max(np.abs(oa-ob))/scale; worst = max(worst, err)
		assert err < 1e-10, (a["name"], a["scalars"], err)
	for k in res_o:
		oa, ob = res_o[k]["out"], res_p[k]["out"]
		if np.iscomplexobj(oa):
			l = 20; oa = oa.copy(); ob = ob.copy(); oa[..., :l+1] = oa[..., :l+1].real; ob[..., :l+1] = ob[..., :l+1].real
		err = np.max(np.abs(oa-ob))/max(np.max(np.abs(oa)), 1e-300); worst = max(worst, err)
		assert err < 1e-10, (k, err)
	out = {}
	meta = []
	for i, c in enumerate(rec_o.calls):
		meta.append(dict(name=c["name"], scalars=c["scalars"], arrays=sorted(c["arrays_in"])))
		for k, v in c["arrays_in"].items(): out["call%03d_in_%s" % (i, k)] = v
		out["call%03d_out" % i] = c["out"]
	out["calls_json"] = np.array(json.dumps(meta))
	out["cases_json"] = np.array(json.dumps(sorted(res_o)))
	for k, d in res_o.items():
		for kk, v in d.items(): out["case_%s__%s" % (k, kk)] = np.asarray(v)
	np.savez_compressed(os.path.join(HERE, "routes_b2.npz"), **out)
	print("route B2: %d boundary calls, %d top-level cases; product (host simulator) vs oracle under the reference: max rel diff %.2e" % (len(rec_o.calls), len(res_o), worst))
	return ns2

def route_b3(ns):
	rfft = ns.fft
	eng = pfft_amd.register(rfft, make_default=False)
	assert rfft.engines["hip"] is eng and rfft.engine != "hip"
	rng = np.random.default_rng(3)
	cases = []; out = {}
	def add(fun, a, kw, view=None):
		"""fun in {fft, ifft, rfft, irfft}; view: (full shape, slicing) of a caller-supplied non-contiguous output"""
		i = len(cases)
		def outbuf(engine):
			if view is None: return None
			full = np.full(view[0], -7.0, dtype=view[2]); return full, full[view[1]]
		res = {}
		for engine in ("numpy", "hip"):
			ob = outbuf(engine)
			args = dict(kw); args["engine"] = engine
			if ob is not None:
				if fun in ("fft", "rfft"): r = getattr(rfft, fun)(a.copy(), ob[1], **args)
				else: r = getattr(rfft, fun)(a.copy(), ob[1], **args)
				assert np.shares_memory(r, ob[0])
				res[engine] = ob[0]
			else: res[engine] = getattr(rfft, fun)(a.copy(), **args)
		scale = max(np.max(np.abs(res["numpy"])), 1e-300)
		tol = 1e-5 if a.dtype in (np.complex64, np.float32) else 1e-12
		assert res["hip"].shape == res["numpy"].shape and res["hip"].dtype == res["numpy"].dtype and np.max(np.abs(res["hip"]-res["numpy"]))/scale < tol, (fun, kw)
		cases.append(dict(fun=fun, kw={k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in kw.items()},
			view=None if view is None else dict(shape=list(view[0]), sel=[[s.start, s.stop, s.step] for s in view[1]], dtype=np.dtype(view[2]).name)))
		out["b3_%02d_in" % i] = a; out["b3_%02d_out" % i] = res["numpy"]
	c = rng.standard_normal((6, 20, 24))+1j*rng.standard_normal((6, 20, 24))
	r = rng.standard_normal((6, 20, 24))
	add("fft", c, dict(axes=[-1]))
	add("fft", c, dict(axes=[-2, -1]))
	add("fft", r, dict(axes=[-2, -1]))                              # real input, complex output of the same shape
	add("ifft", c, dict(axes=[-2, -1], normalize=True))
	add("ifft", c, dict(axes=[0], normalize=False))
	add("rfft", r, dict(axes=[-1]))
	add("rfft", r, dict(axes=[-2, -1]))
	h = np.fft.rfft(r, axis=-1)
	add("irfft", h, dict(n=24, axes=[-1], normalize=True))
	add("irfft", h, dict(n=24, axes=[-1], normalize=False))
	add("fft", c.astype(np.complex64), dict(axes=[-1]))
	sl = (slice(None), slice(None), slice(0, 48, 2))
	add("fft", c, dict(axes=[-1]), view=((6, 20, 48), sl, np.complex128))                 # strided caller-supplied output
	add("ifft", c, dict(axes=[-2, -1], normalize=True), view=((6, 20, 48), sl, np.complex128))
	sl2 = (slice(None), slice(None), slice(1, 14, 1))
	add("rfft", r, dict(axes=[-1]), view=((6, 20, 16), sl2, np.complex128))                # offset view
	out["cases_json"] = np.array(json.dumps(cases))
	np.savez_compressed(os.path.join(HERE, "routes_b3.npz"), **out)
	print("route B3: %d cases through the reference's fft drivers with engines['hip'] == engines['numpy']" % len(cases))

if __name__ == "__main__":
	ns = route_b2()
	route_b3(ns)
