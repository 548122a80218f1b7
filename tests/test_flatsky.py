"""Flat-sky harmonic helpers around enmap.fft against outputs of the reference's enmap functions run with its
numpy FFT engine (tests/golden/flatsky.npz, made by tests/golden/make_golden.py).  Same bodies on the GPU (-m gpu, through
the C ABI) and on the test-only host simulator."""
import os
import numpy as np
import pytest
from pixell_amd import enmap
from pixell_amd.wcs import CarWCS

def _load(golden_dir):
	d = np.load(os.path.join(golden_dir, "flatsky.npz"))
	return d, CarWCS(d["cdelt"], d["crval"], d["crpix"])

def test_geometry_host(golden_dir):
	d, wcs = _load(golden_dir); shape = d["map"].shape[-2:]
	np.testing.assert_allclose(enmap.extent(shape, wcs), d["extent"], rtol=1e-14)
	np.testing.assert_allclose(enmap.extent(shape, wcs, signed=True), d["extent_signed"], rtol=1e-14)
	ly, lx = enmap.laxes(shape, wcs)
	np.testing.assert_allclose(ly, d["ly"], rtol=1e-14, atol=1e-12); np.testing.assert_allclose(lx, d["lx"], rtol=1e-14, atol=1e-12)
	np.testing.assert_allclose(np.asarray(enmap.modlmap(shape, wcs)), d["modlmap"], rtol=1e-14, atol=1e-12)
	np.testing.assert_allclose(enmap.pixsize(shape, wcs), d["pixsize"], rtol=1e-14)

def harm_body(golden_dir):
	d, wcs = _load(golden_dir)
	m = enmap.ndmap(d["map"].copy(), wcs); tol = dict(rtol=1e-12, atol=1e-12)
	h = enmap.map2harm(m)
	assert h.dtype == np.complex128 and h.shape == m.shape
	np.testing.assert_allclose(np.asarray(h), d["harm"], **tol)
	assert np.array_equal(np.asarray(m), d["map"])                       # input untouched
	np.testing.assert_allclose(np.asarray(enmap.map2harm(m, normalize="phys")), d["harm_phys"], rtol=1e-12, atol=1e-14)
	np.testing.assert_allclose(np.asarray(enmap.map2harm(m, iau=True)), d["harm_iau"], **tol)
	np.testing.assert_allclose(np.asarray(enmap.map2harm(m[:2], spin=1)), d["harm_spin1"], **tol)
	np.testing.assert_allclose(np.asarray(enmap.harm2map_adjoint(m, normalize="phys")), d["harm_adj"], rtol=1e-12, atol=1e-14)
	c = enmap.ndmap(d["cplx"].copy(), wcs)
	b = enmap.harm2map(c); assert b.dtype == np.float64
	np.testing.assert_allclose(np.asarray(b), d["back"], **tol)
	assert np.array_equal(np.asarray(c), d["cplx"])
	bk = enmap.harm2map(c, normalize="phys", keep_imag=True); assert bk.dtype == np.complex128
	np.testing.assert_allclose(np.asarray(bk), d["back_phys_keep"], rtol=1e-12, atol=1e-9)
	np.testing.assert_allclose(np.asarray(enmap.map2harm_adjoint(c)), d["back_adj"], **tol)
	# round trip
	np.testing.assert_allclose(np.asarray(enmap.harm2map(enmap.map2harm(m))), d["map"], rtol=1e-12, atol=1e-12)

def ps_body(golden_dir):
	d, wcs = _load(golden_dir)
	h = enmap.ndmap(d["harm"].copy(), wcs)
	ps = enmap.calc_ps2d(h[:, None], h[None, :])
	assert ps.shape == d["ps2d"].shape and ps.dtype == np.float64
	np.testing.assert_allclose(np.asarray(ps), d["ps2d"], rtol=1e-13, atol=1e-14)
	np.testing.assert_allclose(np.asarray(enmap.calc_ps2d(h[0], enmap.ndmap(d["cplx"][1], wcs))), d["ps2d_cross"], rtol=1e-13, atol=1e-13)
	sp = enmap.calc_ps2d(enmap.ndmap(d["harm"][0].astype(np.complex64), wcs)); assert sp.dtype == np.float32
	np.testing.assert_allclose(np.asarray(sp), d["ps2d_sp"], rtol=2e-6, atol=1e-6)
	b, l, nhit = enmap.lbin(enmap.ndmap(d["ps2d"][0, 0], wcs), return_nhit=True)
	assert np.array_equal(nhit, d["lbin_nhit"])
	ok = d["lbin_nhit"] > 0                                              # empty bins are nan on both sides
	np.testing.assert_allclose(b[ok], d["lbin_b"][ok], rtol=1e-12); np.testing.assert_allclose(l[ok], d["lbin_l"][ok], rtol=1e-12)
	assert np.all(np.isnan(b[~ok])) and np.all(np.isnan(d["lbin_b"][~ok]))
	# a second call on the geometry takes the |l| sums and pixel counts of the bins from the first (they depend on the geometry alone)
	b2, l2, nhit2 = enmap.lbin(enmap.ndmap(d["ps2d"][0, 0]*2, wcs), return_nhit=True)
	assert np.array_equal(nhit2, nhit) and np.array_equal(l2[ok], l[ok]); np.testing.assert_allclose(b2[ok], 2*b[ok], rtol=1e-14)
	b3, l3 = enmap.lbin(enmap.ndmap(d["ps2d"][:, 0], wcs), brel=2.5, return_bins=True)
	assert b3.shape == d["lbin3_b"].shape and l3.shape == d["lbin3_l"].shape
	np.testing.assert_allclose(b3, d["lbin3_b"], rtol=1e-12, equal_nan=True); np.testing.assert_allclose(l3, d["lbin3_l"], rtol=1e-12, equal_nan=True)

@pytest.mark.hostsim
def test_harm_hostsim(golden_dir): harm_body(golden_dir)
@pytest.mark.hostsim
def test_ps_hostsim(golden_dir): ps_body(golden_dir)
@pytest.mark.gpu
def test_harm_gpu(golden_dir): harm_body(golden_dir)
@pytest.mark.gpu
def test_ps_gpu(golden_dir): ps_body(golden_dir)

def lbin_edges_body(shape2):
	"""Pixels ON bin edges: on a full-sky CAR map the default bin width is the l step of the x axis, so the whole ly = 0 row sits on edges
	(and the lx = 0 column for brel = ly step / lx step): the bin of each pixel is floor(|l| / bsize) as numpy's float64 expression gives it
	(enmap._bin_helper, enmap.py:2533-2556) -- the kernel takes a fast square root / reciprocal and must fall back to the exact expression there."""
	shape, wcs = enmap.fullsky_geometry(shape=shape2)
	ly, lx = enmap.laxes(shape, wcs)
	rng = np.random.default_rng(5)
	m = rng.standard_normal(shape)
	for brel in (1.0, abs(ly[1])/abs(lx[1]), 0.37):
		bsize = min(abs(lx[1]), abs(ly[1]))*brel
		l = np.sqrt(ly[:, None]**2+lx[None, :]**2)
		n = int(float(np.sqrt(np.max(ly**2)+np.max(lx**2)))/bsize)
		ib = np.floor(l/bsize).astype(np.int64).ravel()
		keep = ib < n
		hit_ref = np.bincount(ib[keep], minlength=n)[:n]
		sum_ref = np.bincount(ib[keep], weights=m.ravel()[keep], minlength=n)[:n]
		b, lc, nhit = enmap.lbin(enmap.ndmap(m, wcs), brel=brel, return_nhit=True)
		assert len(nhit) == n and np.array_equal(nhit, hit_ref), "pixels in the wrong bin (brel %.3f): %d bins differ" % (brel, int(np.sum(nhit != hit_ref)))
		ok = hit_ref > 0
		np.testing.assert_allclose(b[ok]*hit_ref[ok], sum_ref[ok], rtol=1e-10, atol=1e-10)
		b2, _ = enmap.lbin(enmap.ndmap(m, wcs), brel=brel)      # (second call: counts from the cache, one atomic add per pixel)
		np.testing.assert_allclose(b2[ok], b[ok], rtol=1e-12, atol=1e-13)
@pytest.mark.hostsim
def test_lbin_edges_hostsim(): lbin_edges_body((90, 180))
@pytest.mark.gpu
def test_lbin_edges_gpu(): lbin_edges_body((90, 180)); lbin_edges_body((1080, 2160)); lbin_edges_body((1350, 2048))

@pytest.mark.gpu
def test_flatsky_device_resident(golden_dir):
	"""dmap in -> dmap out: the whole map -> T,E,B harmonics -> 2-D spectra -> binned spectrum chain stays in HBM"""
	import torch
	d, wcs = _load(golden_dir)
	m = enmap.dmap(torch.from_numpy(d["map"]).cuda(), wcs)
	h = enmap.map2harm(m, normalize="phys"); assert isinstance(h, enmap.dmap) and h.tensor.is_cuda
	np.testing.assert_allclose(h.tensor.cpu().numpy(), d["harm_phys"], rtol=1e-12, atol=1e-14)
	ps = enmap.calc_ps2d(h[:, None], h[None, :]); assert isinstance(ps, enmap.dmap) and ps.shape == (3, 3)+m.shape[-2:]
	b, l = enmap.lbin(ps[0, 0]); assert len(b) == len(l) > 0
	back = enmap.harm2map(h, normalize="phys"); assert isinstance(back, enmap.dmap)
	np.testing.assert_allclose(back.tensor.cpu().numpy(), d["map"], rtol=1e-12, atol=1e-12)

@pytest.mark.gpu
def test_flatsky_large_property():
	"""Parseval at 3 x 1080 x 2160: sum |harm|^2 == sum map^2 (unitary normalisation), E/B rotation preserves Q^2+U^2 power"""
	import torch
	shape, wcs = enmap.fullsky_geometry(shape=(1080, 2160))
	t = torch.randn((3,)+tuple(shape), dtype=torch.float64, device="cuda")
	h = enmap.map2harm(enmap.dmap(t, wcs))
	p_map = (t**2).sum(dim=(-2, -1)).cpu().numpy(); p_h = (h.tensor.abs()**2).sum(dim=(-2, -1)).cpu().numpy()
	np.testing.assert_allclose(p_h[0], p_map[0], rtol=1e-12)
	np.testing.assert_allclose(p_h[1]+p_h[2], p_map[1]+p_map[2], rtol=1e-12)
	back = enmap.harm2map(h)
	assert float((back.tensor-t).abs().max()) < 1e-11
