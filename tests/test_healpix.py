"""The healpix layer of pixell_amd.curvedsky (ring tables with per-ring nphi / phi0 through the general ring path of sht.hip)
against tests/golden/healpix.npz, which tests/golden/make_healpix.py recorded from the REFERENCE's own
get_ring_info_healpix / get_ring_info_radial / alm2map_healpix / map2alm_healpix (pixell/curvedsky.py:312-403, 1192-1234) over the
long-double oracle."""
import os
import numpy as np
import pytest
from pixell_amd import curvedsky, enmap, uharm, fft as pfft
from pixell_amd.wcs import CarWCS

def _real_m0(a, lmax):
	a = np.array(a); a[..., :lmax+1] = a[..., :lmax+1].real; return a

def test_ring_tables(golden_dir):
	d = np.load(os.path.join(golden_dir, "healpix.npz"))
	for nside in (1, 2, 5, 8):
		r = curvedsky.get_ring_info_healpix(nside)
		assert r.npix == 12*nside**2 and r.nrow == 4*nside-1 and int(np.sum(r.nphi)) == r.npix
		for k in ("theta", "phi0"): np.testing.assert_allclose(r[k], d["rings%d_%s" % (nside, k)], rtol=0, atol=2e-15)
		for k in ("nphi", "offsets"): assert r[k].dtype == np.uint64 and np.array_equal(r[k], d["rings%d_%s" % (nside, k)])
	r = curvedsky.get_ring_info_healpix(4, d["sub_rings"])
	for k in ("theta", "phi0"): np.testing.assert_allclose(r[k], d["sub_%s" % k], rtol=0, atol=2e-15)
	for k in ("nphi", "offsets"): assert np.array_equal(r[k], d["sub_%s" % k])
	r = curvedsky.get_ring_info_radial(np.array([0.1, 0.5, 2.0]))
	for k in ("theta", "nphi", "phi0", "offsets"): assert np.array_equal(r[k], d["rad_%s" % k])
	assert curvedsky.npix2nside(12*64**2) == 64
	w = curvedsky.apply_minfo_theta_lim(curvedsky.get_ring_info_healpix(4), 0.6, 2.2)
	assert len(w.theta) == len(w.nphi) == len(w.offsets) < 15 and w.theta.min() >= 0.6 and w.theta.max() <= 2.2

def healpix_body(golden_dir, to_dev=lambda x: x, to_host=np.asarray):
	d = np.load(os.path.join(golden_dir, "healpix.npz"))
	nside, lmax = (int(v) for v in d["meta"]); npix = 12*nside**2
	alm, pix = np.array(d["alm"]), np.array(d["pix"])
	def close(a, b, tol=1e-11): assert a.shape == b.shape and np.max(np.abs(a-b)) < tol*np.max(np.abs(b))
	close(to_host(curvedsky.alm2map_healpix(to_dev(alm.copy()), nside=nside, spin=[0, 2])), d["alm2map"])
	close(to_host(curvedsky.alm2map_healpix(to_dev(alm.copy()), to_dev(np.full((3, npix), 7.0)), spin=[0, 2])), d["alm2map"])      # overwrites a given map
	at = curvedsky.alm2map_healpix(to_dev(np.zeros_like(alm)), to_dev(pix.copy()), spin=[0, 2], adjoint=True)
	close(_real_m0(to_host(at), lmax), _real_m0(d["alm2map_adjoint"], lmax))
	close(to_host(curvedsky.alm2map_healpix(to_dev(alm[0].copy()), to_dev(np.zeros((2, npix))), deriv=True)), d["deriv"])
	w = to_host(curvedsky.alm2map_healpix(to_dev(alm.copy()), to_dev(np.full((3, npix), 7.0)), spin=[0, 2], theta_min=0.6, theta_max=2.2))
	close(w, d["alm2map_window"]); assert np.any(w == 0)                                             # rings outside the window are zeroed
	for niter in (0, 2):
		a = curvedsky.map2alm_healpix(to_dev(pix.copy()), lmax=lmax, spin=[0, 2], niter=niter)
		close(_real_m0(to_host(a), lmax), _real_m0(d["map2alm_niter%d" % niter], lmax), 1e-10)
	a = curvedsky.map2alm_healpix(to_dev(np.array(d["alm2map"])), lmax=lmax, spin=[0, 2], niter=3)
	close(_real_m0(to_host(a), lmax), _real_m0(d["map2alm_roundtrip"], lmax), 1e-10)
	ma = curvedsky.map2alm_healpix(to_dev(np.zeros((3, npix))), alm=to_dev(alm.copy()), spin=[0, 2], adjoint=True, niter=1)
	close(to_host(ma), d["map2alm_adjoint"], 1e-10)
	with pytest.raises(NotImplementedError): curvedsky.map2alm_healpix(to_dev(np.zeros((2, npix))), lmax=lmax, deriv=True)
	with pytest.raises(ValueError): curvedsky.alm2map_healpix(to_dev(alm.copy()), to_dev(np.zeros((2, npix))), spin=[0, 2])

def profile_body(golden_dir):
	"""profile2harm / harm2profile (m = 0 transforms on one-pixel rings, curvedsky.py:510-554) and the UHT profile methods
	(uharm.py:127-138, 191-207) against the reference's outputs"""
	d = np.load(os.path.join(golden_dir, "healpix.npz"))
	rr, br, r2 = d["prof_r"], d["prof_br"], d["prof_r2"]
	def close(a, b, tol=1e-11): assert a.shape == b.shape and np.max(np.abs(a-b)) < tol*np.max(np.abs(b))
	close(curvedsky.profile2harm(br, rr, lmax=150), d["prof_bl"])
	close(curvedsky.profile2harm(br[0], rr), d["prof_bl_auto"])
	close(curvedsky.profile2harm(br[1, 20:], rr[20:], lmax=90, left=2.0, right=0.0), d["prof_bl_lr"])
	close(curvedsky.harm2profile(d["prof_bl"], r2), d["prof_back"])
	shape, wcs = enmap.fullsky_geometry(shape=(46, 90))
	uht = uharm.UHT(shape, wcs, mode="curved", lmax=40)
	close(uht.rprof2hprof(br[0], rr), d["uht_rprof2hprof"]); close(uht.hprof2rprof(d["uht_hprof"], r2), d["uht_hprof2rprof"])
	close(uht.hprof_rpow(d["uht_hprof"], 2.0), d["uht_rpow"], 1e-10)

def helpers_body(golden_dir):
	"""the reference's helper entry points by name: map2buffer / buffer2map / flip_* / pad_geometry, the raw transforms on buffers,
	alm real <-> complex, get_ducc_maxlmax, chebt / ichebt (curvedsky.py:900-1086, 1236-1250, 1349-1353, 1384-1473; fft.py:307-317)"""
	d = np.load(os.path.join(golden_dir, "healpix.npz"))
	def close(a, b, tol=1e-11): assert a.shape == b.shape and np.max(np.abs(a-b)) <= tol*np.max(np.abs(b))
	def real_m0(a, lmax=12): a = np.array(a); a[..., :lmax+1] = a[..., :lmax+1].real; return a
	w = d["buf_in_wcs"]; mb = enmap.ndmap(np.array(d["buf_in"]), CarWCS(w[0], w[1], w[2]))
	flip, pad = [bool(v) for v in d["buf_flip"]], d["buf_pad"]
	info = curvedsky.analyse_geometry(mb.shape, mb.wcs)
	assert list(info.flip) == flip and np.array_equal(np.array([info.ypad, info.xpad]).T, pad)
	buf = curvedsky.map2buffer(mb, flip, pad)
	assert np.array_equal(np.asarray(buf), d["buf_out"])
	bw = d["buf_wcs"]; np.testing.assert_allclose([buf.wcs.wcs.cdelt, buf.wcs.wcs.crval, buf.wcs.wcs.crpix], bw, rtol=0, atol=1e-12)
	assert np.array_equal(np.asarray(curvedsky.buffer2map(buf, flip, pad)), d["buf_back"])
	assert curvedsky.map2buffer(mb, flip, pad, obuf=True).sum() == 0
	a12 = np.array(d["raw_alm"])
	close(np.asarray(curvedsky.alm2map_raw_2d(a12.copy(), enmap.zeros((3,)+buf.shape[-2:], buf.wcs), spin=[0, 2])), d["raw_alm2map_2d"])
	close(real_m0(curvedsky.map2alm_raw_2d(buf.copy(), alm=np.zeros_like(a12), spin=[0, 2])), real_m0(d["raw_map2alm_2d"]), 1e-10)
	with pytest.raises(ValueError): curvedsky.alm2map_raw_2d(a12.copy(), enmap.zeros(mb.shape, mb.wcs), spin=[0, 2])       # a band is not a complete grid
	cw = d["raw_cyl_wcs"]; cyl = enmap.ndmap(np.array(d["raw_cyl_in"]), CarWCS(cw[0], cw[1], cw[2]))
	close(np.asarray(curvedsky.alm2map_raw_cyl(a12.copy(), enmap.zeros(cyl.shape, cyl.wcs), spin=[0, 2])), d["raw_alm2map_cyl"])
	close(real_m0(curvedsky.map2alm_raw_cyl(cyl.copy(), alm=np.zeros_like(a12), spin=[0, 2], niter=1, weights=d["raw_cyl_weights"])), real_m0(d["raw_map2alm_cyl"]), 1e-10)
	# quad_weights of this map -- an asymmetric band stored north to south: the reference's order (south first for every map; pinned by
	# tests/golden/round3.npz) is the default, row_order="map" follows the map's rings
	th = curvedsky.get_ring_info(cyl.shape, cyl.wcs).theta
	wq = curvedsky.quad_weights(cyl.shape, cyl.wcs, row_order="map"); assert np.argmin(wq) == np.argmin(np.sin(th))
	assert np.array_equal(curvedsky.quad_weights(cyl.shape, cyl.wcs), wq[::-1])
	assert np.array_equal(curvedsky.alm_complex2real(a12), d["c2r"]) and np.array_equal(curvedsky.alm_real2complex(d["c2r"][0]), d["r2c"])
	got = np.array([[curvedsky.get_ducc_maxlmax(n, k) for k in (8, 9, 30)] for n in ("CC", "F1", "MW", "MWflip", "DH", "F2")])
	assert np.array_equal(got, d["maxlmax"])
	assert curvedsky.dangerous_dtype(np.dtype(">f8")) and not curvedsky.dangerous_dtype(np.dtype("=f8"))
	xx = np.array(d["cheb_in"])
	close(pfft.chebt(xx.copy()), d["cheb"], 1e-13); close(pfft.ichebt(np.array(d["cheb"])), d["icheb"], 1e-13); close(d["icheb"], xx, 1e-13)
	assert pfft.get_engine("auto") == "hip" and pfft.get_engine("numpy") == "numpy" and pfft.empty((2, 3), np.complex64).shape == (2, 3)
	with pytest.raises(KeyError): pfft.set_engine("fftw")
	np.testing.assert_allclose(pfft.ind2freq(9, np.arange(9), 0.5), np.fft.fftfreq(9, 0.5)); np.testing.assert_allclose(pfft.freq2ind(9, np.fft.fftfreq(9, 0.5), 0.5), np.arange(9))

@pytest.mark.hostsim
def test_helpers_hostsim(golden_dir): helpers_body(golden_dir)
@pytest.mark.gpu
def test_helpers_gpu(golden_dir): helpers_body(golden_dir)

@pytest.mark.hostsim
def test_healpix_hostsim(golden_dir): healpix_body(golden_dir)
@pytest.mark.hostsim
def test_profiles_hostsim(golden_dir): profile_body(golden_dir)
@pytest.mark.gpu
def test_profiles_gpu(golden_dir): profile_body(golden_dir)

@pytest.mark.gpu
def test_healpix_gpu(golden_dir):
	healpix_body(golden_dir)
	import torch
	healpix_body(golden_dir, to_dev=lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda(), to_host=lambda x: x.cpu().numpy())

@pytest.mark.gpu
@pytest.mark.parametrize("nside,lmax", [(256, 512), (1024, 700)])
def test_healpix_large_gpu(nside, lmax):
	"""nside 256 (786 432 pixels, 1023 rings of 256 different lengths, lengths with prime factors up to 251 -> Bluestein), lmax 512,
	and nside 1024 (ring lengths up to 4096: two-pass transforms): adjointness <Y a, p> = <a, Y^T p> to rounding and convergence of
	the Jacobi iteration on a band-limited map"""
	npix = 12*nside**2
	from oracle import sht_oracle as so
	rng = np.random.default_rng(3)
	alm = so.rand_alm_simple(lmax, 3, 2, spin=(0, 2))
	m = curvedsky.alm2map_healpix(alm.copy(), nside=nside, spin=[0, 2])
	pix = rng.standard_normal((3, npix))
	at = curvedsky.alm2map_healpix(np.zeros_like(alm), pix.copy(), spin=[0, 2], adjoint=True)
	wgt = np.full(alm.shape[-1], 2.0); wgt[:lmax+1] = 1                                             # real-field inner product on m >= 0 storage
	lhs = np.sum(m*pix); rhs = np.sum(wgt*(alm.real*at.real+alm.imag*at.imag))
	assert abs(lhs-rhs) < 1e-11*abs(lhs)
	back = curvedsky.map2alm_healpix(m, lmax=lmax, spin=[0, 2], niter=3)
	assert np.sqrt(np.mean(np.abs(back-alm)**2)/np.mean(np.abs(alm)**2)) < 2e-3
