"""CPU: host-side mirror of the reference's geometry / layout bookkeeping against fixtures generated
from the reference itself (tests/golden/make_golden.py)."""
import os, json
import numpy as np
import pytest
from pixell_amd import curvedsky, enmap
from pixell_amd.wcs import CarWCS

@pytest.fixture(scope="module")
def geo(golden_dir):
	return json.load(open(os.path.join(golden_dir, "geometry.json")))

KEYS = ["c1_1024x2048", "c2_5400x10800", "c3_21600x43200", "c5_10800x21600", "rt_32x61", "ref_bench_900x1800",
	"cc_181x360", "f1_6x12", "cc_7x12", "patch_cc", "patch_gen_cyl", "band_30deg"]

@pytest.mark.parametrize("key", KEYS)
def test_analyse_geometry_matches_reference(geo, key):
	"""curvedsky.analyse_geometry / get_ducc_geo / get_method (curvedsky.py:1252-1353, 478-488)"""
	d = geo[key]
	wcs = CarWCS(d["cdelt"], d["crval"], d["crpix"])
	mi = curvedsky.analyse_geometry(tuple(d["shape"]), wcs)
	assert mi.case == d["case"]
	assert [bool(f) for f in mi.flip] == d["flip"]
	assert abs(mi.phi0-d["phi0"]) < 1e-12
	assert [int(v) for v in mi.ypad] == d["ypad"] and [int(v) for v in mi.xpad] == d["xpad"]
	assert curvedsky.get_method(tuple(d["shape"]), wcs) == d["method"]
	if "name" in d:
		g = mi.ducc_geo
		assert (g.name, int(g.ny), int(g.nx), int(g.yoff), int(g.lmax)) == (d["name"], d["ny"], d["nx"], d["yoff"], d["lmax"])
	else:
		assert mi.ducc_geo is None
	if "theta_first" in d:
		ri = curvedsky.get_ring_info(tuple(d["shape"]), wcs)
		assert abs(ri.theta[0]-d["theta_first"]) < 1e-12 and abs(ri.theta[-1]-d["theta_last"]) < 1e-12
		assert abs(np.exp(1j*ri.phi0[0])-np.exp(1j*d["ring_phi0"])) < 1e-12   # equal modulo 2 pi (the reference unwinds RA)

def test_fullsky_geometry_matches_reference(geo):
	"""enmap.fullsky_geometry (enmap.py:1713-1740); reference test tests/test_pixell.py:560-566"""
	shape, wcs = enmap.fullsky_geometry(res=np.deg2rad(0.5/60))
	assert list(shape) == geo["fullsky_0.5arcmin_shape"] == [21600, 43200]
	assert abs(enmap.area(shape, wcs)-4*np.pi) < 1e-6
	for key in ["c1_1024x2048", "rt_32x61"]:
		d = geo[key]
		shape, wcs = enmap.fullsky_geometry(shape=tuple(d["shape"]))
		assert np.allclose(wcs.wcs.cdelt, d["cdelt"], rtol=0, atol=1e-13) and np.allclose(wcs.wcs.crval, d["crval"], rtol=0, atol=1e-13)
		assert np.allclose(wcs.wcs.crpix, d["crpix"], rtol=0, atol=1e-13)
	shape, wcs = enmap.fullsky_geometry(res=np.deg2rad(1.0), variant="CC")
	assert tuple(shape) == (181, 360) and np.allclose(wcs.wcs.crpix, geo["cc_181x360"]["crpix"])
	shape, wcs = enmap.band_geometry(np.deg2rad(30), res=np.deg2rad(0.5))
	assert list(shape) == geo["band_30deg"]["shape"] and np.allclose(wcs.wcs.crpix, geo["band_30deg"]["crpix"])

def test_alm_info_layouts(geo):
	"""curvedsky.alm_info (curvedsky.py:409-447); reference test tests/test_pixell.py:760-824"""
	ai = curvedsky.alm_info(lmax=10, mmax=7)
	assert ai.nelem == geo["alm_info_10_7"]["nelem"] and [int(v) for v in ai.mstart] == geo["alm_info_10_7"]["mstart"]
	ai = curvedsky.alm_info(lmax=6, layout="rect")
	assert ai.nelem == geo["alm_info_rect_6"]["nelem"] and [int(v) for v in ai.mstart] == geo["alm_info_rect_6"]["mstart"]
	ai = curvedsky.alm_info(nalm=66)
	assert ai.lmax == 10 and ai.mstart.dtype == np.uint64 and ai.lm2ind(3, 2) == int(ai.mstart[2])+3
	with pytest.raises(AssertionError):
		curvedsky.alm_info(lmax=10, mmax=7, nalm=66)

def test_spin_helper(geo):
	"""enmap.spin_helper (enmap.py:3378-3388)"""
	assert [[int(a), int(b), int(c)] for a, b, c in enmap.spin_helper([0, 2], 3)] == geo["spin_helper_02_3"]
	assert [[int(a), int(b), int(c)] for a, b, c in enmap.spin_helper([0, 1, 2], 5)] == geo["spin_helper_012_5"]
	assert [[int(a), int(b), int(c)] for a, b, c in enmap.spin_helper(1, 6)] == geo["spin_helper_1_6"]
	with pytest.raises(IndexError):
		list(enmap.spin_helper([0, 2], 2))

def test_prepare_alm_errors():
	"""curvedsky.prepare_alm (curvedsky.py:1413-1427)"""
	with pytest.raises(ValueError):
		curvedsky.prepare_alm()
	alm, ai = curvedsky.prepare_alm(lmax=5, pre=(3,), dtype=np.float32)
	assert alm.shape == (3, 21) and alm.dtype == np.complex64 and ai.lmax == 5
	with pytest.raises(ValueError):
		curvedsky.prepare_alm(alm=np.zeros(21, np.complex64), dtype=np.float64)
	alm, ai = curvedsky.prepare_alm(alm=np.zeros(21, np.complex64), dtype=np.float64, convert=True)
	assert alm.dtype == np.complex128

def test_slicing_moves_the_geometry():
	"""ndmap[..., y-slice, x-slice] carries the wcs of the slice (enmap.ndmap.__getitem__ / slice_geometry, enmap.py:140-163, 264-285)"""
	shape, wcs = enmap.fullsky_geometry(shape=(12, 24))
	m = enmap.ndmap(np.arange(3*12*24, dtype=float).reshape(3, 12, 24), wcs)
	for sel in [(Ellipsis, slice(2, 9), slice(4, 20)), (slice(None), slice(None, None, -1), slice(None)), (1, slice(3, None, 2), slice(None, None, -3)), (Ellipsis, slice(5, 7))]:
		sub = m[sel]
		assert isinstance(sub, enmap.ndmap) and np.array_equal(np.asarray(sub), np.asarray(m)[sel])
		# enmap.slice_geometry's convention (enmap.py:264-285, fixtures in test_round3_fixtures.py): the new pixels tile the area of
		# the selected ones -- new pixel p spans old pixel start + p*step and the |step| - 1 old pixels after it in walking
		# direction, so its centre lies (|step| - 1)/2 old pixels beyond the centre of old pixel start + p*step
		yy, xx = np.mgrid[:12, :24]
		src_y = np.broadcast_to(yy, m.shape)[sel]; src_x = np.broadcast_to(xx, m.shape)[sel]
		src_y = src_y.reshape((-1,)+src_y.shape[-2:])[0].astype(float); src_x = src_x.reshape((-1,)+src_x.shape[-2:])[0].astype(float)
		full = [s for s in sel if isinstance(s, slice)][-2:] if sel[0] is not Ellipsis else list(sel[1:])
		full = [slice(None)]*(2-len(full))+full if sel[0] is not Ellipsis else full+[slice(None)]*(2-len(full))
		sy, sx = [(s.step or 1) for s in full]
		src_y += np.sign(sy)*(abs(sy)-1)/2; src_x += np.sign(sx)*(abs(sx)-1)/2
		py, px = np.mgrid[:sub.shape[-2], :sub.shape[-1]]
		assert np.allclose(enmap.pix2sky(sub.shape, sub.wcs, [py, px]), enmap.pix2sky(m.shape, m.wcs, [src_y, src_x]), atol=1e-13)
	assert not isinstance(m[0, 3], enmap.ndmap)               # a pixel axis indexed away: plain array
	assert m[1].wcs is m.wcs
	# a flipped full-sky map is still a "2d" geometry, with the flips reversed
	mi = curvedsky.analyse_geometry(m[..., ::-1, ::-1].shape, m[..., ::-1, ::-1].wcs)
	assert mi.case == "2d" and [bool(f) for f in mi.flip] == [False, False]

def test_non_car_cylindrical_is_general():
	"""CEA / MER maps are not transformed with CAR ring positions"""
	shape, wcs = enmap.fullsky_geometry(shape=(12, 24))
	w = wcs.deepcopy(); w.wcs.ctype = ["RA---CEA", "DEC--CEA"]
	assert curvedsky.analyse_geometry(shape, w).case == "general" and curvedsky.get_method(shape, w) == "general"

def test_plan_cache_is_bounded():
	from pixell_amd import sht
	assert sht._plans.cap >= 1
	c = sht._PlanCache(); c.cap = 2
	for k in "abc": c[k] = object()
	assert len(c) == 2 and c.get("a") is None and c.get("c") is not None

def test_prefetch_pipeline_close_does_not_hang(monkeypatch):
	"""with more uploads queued than prefetch slots, a call that ends before taking them (an exception in a transform) must
	still be able to close its pipeline -- the worker's wait for a slot is cancellable, and a failed upload gives its slot back."""
	import threading, time, types
	from pixell_amd import hostio
	class OOM(Exception): pass
	fake = types.SimpleNamespace(cuda=types.SimpleNamespace(current_device=lambda: 0, set_device=lambda d: None, OutOfMemoryError=OOM, empty_cache=lambda: None))
	monkeypatch.setattr(hostio, "_torch", lambda: fake)
	monkeypatch.setattr(hostio, "eligible", lambda a: True)
	calls = []
	def fake_upload(a):
		calls.append(a.shape)
		if a.shape[0] == 3: raise ValueError("upload failed")      # not an out-of-memory error: the slot must come back all the same
		return (a, None)
	monkeypatch.setattr(hostio, "upload", fake_upload)
	p = hostio.Pipeline()
	arrs = [np.zeros((k + 1, 4)) for k in range(5)]
	p.prefetch(arrs)
	assert p.take(arrs[0]) is not None      # frees one slot; the third upload fails, the rest wait for slots nobody will free
	t0 = time.time()
	closer = threading.Thread(target=lambda: [None for _ in [0] if not _close(p)], daemon=True)
	def _close(pp):
		try: pp.close()
		except ValueError: pass
		return True
	closer.start(); closer.join(5.0)
	assert not closer.is_alive(), "Pipeline.close() hangs on the prefetch semaphore"
	assert time.time() - t0 < 5.0
