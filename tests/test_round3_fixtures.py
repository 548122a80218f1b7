"""Round-3 fixtures recorded from the reference's own functions (tests/golden/make_round3.py): enmap.slice_geometry for |step| > 1,
curvedsky.quad_weights and the weight handling of the cyl path (curvedsky.py:492-505, 843-873) on bands stored south-to-north and
north-to-south -- the reference's row conventions, reproduced result for result."""
import os
import numpy as np
import pytest
from pixell_amd import curvedsky, enmap, wcs as wcsutils

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "round3.npz")

def _wcs(d, pre):
	_, w = enmap.fullsky_geometry(shape=(4, 8)); w = w.deepcopy()
	w.wcs.cdelt[:] = d[pre+"cdelt"]; w.wcs.crval[:] = d[pre+"crval"]; w.wcs.crpix[:] = d[pre+"crpix"]
	return w

def test_slice_geometry_matches_reference():
	d = np.load(GOLD)
	shape = tuple(int(v) for v in d["slice_in_shape"]); wcs = _wcs(d, "slice_in_")
	for i in range(int(d["slice_n"])):
		sel = tuple(slice(*[None if v == -9999 else int(v) for v in r]) for r in d["slice_%d_sel" % i])
		s2, w2 = wcsutils.slice_geometry(shape, wcs, sel)
		assert tuple(s2) == tuple(int(v) for v in d["slice_%d_shape" % i])
		assert np.allclose(w2.wcs.crpix, d["slice_%d_crpix" % i], rtol=0, atol=1e-12) and np.allclose(w2.wcs.cdelt, d["slice_%d_cdelt" % i], rtol=1e-15)
		m = enmap.ndmap(np.zeros(shape), wcs)[sel]
		assert np.allclose(m.wcs.wcs.crpix, d["slice_%d_crpix" % i], rtol=0, atol=1e-12)
	assert isinstance(enmap.ndmap(np.ones((2, 3, 4)), wcs)[1, 2, 3], np.floating)       # full integer indexing: a numpy scalar

def quad_weights_body():
	d = np.load(GOLD)
	for k in str(d["geo_names"]).split(","):
		if "qw_"+k not in d: continue
		shape = tuple(int(v) for v in d["geo_%s_shape" % k]); wcs = _wcs(d, "geo_%s_" % k)
		mi = curvedsky.analyse_geometry(shape, wcs)
		assert [bool(f) for f in mi.flip] == [bool(f) for f in d["geo_%s_flip" % k]] and mi.case == str(d["geo_%s_case" % k])
		w = curvedsky.quad_weights(shape, wcs)
		assert np.allclose(w, d["qw_"+k], rtol=1e-12, atol=0), k
		wm = curvedsky.quad_weights(shape, wcs, row_order="map")
		assert np.allclose(wm, w if mi.flip[0] else w[::-1], rtol=1e-15)

def cyl_weights_body(names=None, tol=1e-10):
	d = np.load(GOLD); lmax = int(d["lmax"])
	for k in (names or str(d["geo_names"]).split(",")):
		shape = tuple(int(v) for v in d["geo_%s_shape" % k]); wcs = _wcs(d, "geo_%s_" % k)
		m = enmap.ndmap(d["cyl_%s_map" % k].copy(), wcs)
		def cmp(a, ref, what):
			a = np.array(a); ref = np.array(ref)
			a[..., :lmax+1] = a[..., :lmax+1].real; ref[..., :lmax+1] = ref[..., :lmax+1].real     # Im a_l0 is not defined by a real map
			err = np.max(np.abs(a-ref))/np.max(np.abs(ref))
			assert err < tol, (k, what, err)
		for niter in (0, 1):
			cmp(curvedsky.map2alm(m.copy(), lmax=lmax, spin=[0, 2], niter=niter, method="cyl"), d["cyl_%s_default_niter%d" % (k, niter)], "default weights niter %d" % niter)
		cmp(curvedsky.map2alm(m.copy(), lmax=lmax, spin=[0, 2], weights=d["cyl_%s_weights" % k].copy(), method="cyl"), d["cyl_%s_explicit" % k], "caller-supplied weights")

@pytest.mark.hostsim
def test_quad_weights_hostsim(): quad_weights_body()
@pytest.mark.gpu
def test_quad_weights_gpu(): quad_weights_body()
@pytest.mark.hostsim
def test_cyl_weights_hostsim(): cyl_weights_body(["band_asym_n2s", "shift_asym"])
@pytest.mark.gpu
def test_cyl_weights_gpu(): cyl_weights_body()
