"""The reference's own curvedsky tests, run against pixell_amd.curvedsky (same names and
semantics), plus the golden alm2map vector.  Reference: tests/test_pixell.py:870-965 (round trip),
1028-1046 (dtype conversion), 1051-1085 (adjointness), 351-360 + tests/data/MM_unlensed_071123.fits."""
import os
import numpy as np
import pytest
from pixell_amd import curvedsky, enmap, sht

def roundtrip_body(lmax=30):
	for use_oalm in [False, True]:
		ainfo = curvedsky.alm_info(lmax)
		shape, wcs = enmap.fullsky_geometry(shape=(lmax+2, 2*lmax+1))
		i = ainfo.lm2ind(lmax, lmax)
		for dt, ct in [(np.float64, np.complex128), (np.float32, np.complex64)]:
			alm = np.zeros(ainfo.nelem, ct); alm[i] = 1+1j
			omap = enmap.zeros(shape, wcs, dt)
			curvedsky.alm2map(alm, omap, spin=0)
			out = curvedsky.map2alm(omap, alm=np.zeros_like(alm) if use_oalm else None, spin=0, ainfo=ainfo)
			np.testing.assert_array_almost_equal(out, alm)
			alm = np.zeros((2, ainfo.nelem), ct); alm[0, i] = 1+1j; alm[1, i] = 2-2j
			omap = enmap.zeros((2,)+shape, wcs, dt)
			curvedsky.alm2map(alm, omap, spin=1)
			out = curvedsky.map2alm(omap, alm=np.zeros_like(alm) if use_oalm else None, spin=1, ainfo=ainfo)
			np.testing.assert_array_almost_equal(out, alm)
			alm = np.zeros((3, 2, ainfo.nelem), ct)
			alm[0, 0, i] = 1+1j; alm[0, 1, i] = 2-2j; alm[1, 0, i] = 3+3j; alm[1, 1, i] = 4-4j; alm[2, 0, i] = 5+5j; alm[2, 1, i] = 6-6j
			omap = enmap.zeros((3, 2)+shape, wcs, dt)
			curvedsky.alm2map(alm, omap, spin=1)
			out = curvedsky.map2alm(omap, alm=np.zeros_like(alm) if use_oalm else None, spin=1, ainfo=ainfo)
			np.testing.assert_array_almost_equal(out, alm)

def alm_conversion_body():
	lmax = 30; ainfo = curvedsky.alm_info(lmax)
	alm = np.zeros(ainfo.nelem, np.complex64); alm[ainfo.lm2ind(lmax, lmax)] = 1+1j
	shape, wcs = enmap.fullsky_geometry(shape=(lmax+2, 2*lmax+1))
	m = enmap.zeros(shape, wcs, np.float64)
	curvedsky.alm2map(alm, m, spin=0)
	assert np.abs(m).max() > 0
	with pytest.raises(ValueError):
		curvedsky.map2alm(m, alm, spin=0)
	with pytest.raises(NotImplementedError):
		curvedsky.map2alm(m, lmax=lmax, spin=0, deriv=True)
	with pytest.raises(ValueError):
		curvedsky.alm2map(alm, m, method="nonsense")

def zip_alm(alm, ainfo):
	n = ainfo.lm2ind(1, 1)
	return np.concatenate([alm[..., :n].real, alm[..., n:].view(curvedsky.real_dtype(alm.dtype))*2**0.5], -1)
def unzip_alm(z, ainfo):
	n = ainfo.lm2ind(1, 1)
	o = np.zeros(z.shape[:-1]+(ainfo.nelem,), curvedsky.complex_dtype(z.dtype))
	o[..., :n] = z[..., :n]; o[..., n:] = z[..., n:].view(o.dtype)/2**0.5
	return o
def map_bash(fun, shape, wcs, ncomp, lmax, dtype):
	ainfo = curvedsky.alm_info(lmax); nz = int(2*ainfo.nelem-ainfo.lm2ind(1, 1))
	umap = enmap.zeros((ncomp,)+shape, wcs, dtype); oalm = np.zeros((ncomp, ainfo.nelem), curvedsky.complex_dtype(dtype))
	mat = np.zeros((ncomp, nz, ncomp)+shape, dtype)
	for I in np.ndindex(*((ncomp,)+shape)):
		umap[I] = 1; oalm[:] = 0
		fun(map=umap, alm=oalm, ainfo=ainfo)
		mat[(slice(None), slice(None))+I] = zip_alm(oalm, ainfo); umap[I] = 0
	return mat
def alm_bash(fun, shape, wcs, ncomp, lmax, dtype):
	ainfo = curvedsky.alm_info(lmax); nz = int(2*ainfo.nelem-ainfo.lm2ind(1, 1))
	z = np.zeros((ncomp, nz), dtype); omap = enmap.zeros((ncomp,)+shape, wcs, dtype)
	mat = np.zeros((ncomp, nz, ncomp)+shape, dtype)
	for ci in range(ncomp):
		for k in range(nz):
			z[ci, k] = 1; omap[:] = 0
			fun(alm=unzip_alm(z, ainfo), map=omap, ainfo=ainfo)
			mat[ci, k] = omap; z[ci, k] = 0
	return mat

def adjointness_body(variants=("fejer1", "cc"), ncomps=(1, 3), dtypes=(np.float64,), do_analysis=True):
	"""alm2map_adjoint == alm2map^T and map2alm_adjoint == map2alm^T as explicit matrices"""
	res = 30*np.pi/180
	for dtype in dtypes:
		for variant in variants:
			shape, wcs = enmap.fullsky_geometry(res=res, variant=variant)
			lmax = 5
			for ncomp in ncomps:
				m1 = alm_bash(curvedsky.alm2map, shape, wcs, ncomp, lmax, dtype)
				m2 = map_bash(curvedsky.alm2map_adjoint, shape, wcs, ncomp, lmax, dtype)
				np.testing.assert_array_almost_equal(m1, m2)
				if do_analysis:
					m1 = map_bash(curvedsky.map2alm, shape, wcs, ncomp, lmax, dtype)
					m2 = alm_bash(curvedsky.map2alm_adjoint, shape, wcs, ncomp, lmax, dtype)
					np.testing.assert_array_almost_equal(m1, m2)

def golden_body(golden_dir):
	"""alm2map of the reference's rand_alm(seed=1) == reference golden map, through the full
	curvedsky.alm2map path (flip=[True,True] folded into strides)"""
	d = np.load(os.path.join(golden_dir, "lens_unlensed.npz"))
	from pixell_amd.wcs import CarWCS
	wcs = CarWCS(d["cdelt"], d["crval"], d["crpix"])
	gold = d["map"]
	omap = enmap.zeros(gold.shape, wcs, np.float64)
	curvedsky.alm2map(d["alm"], omap, spin=[0, 2], ainfo=curvedsky.alm_info(lmax=int(d["lmax"])))
	assert np.all(np.isclose(np.asarray(omap), gold))
	assert np.max(np.abs(np.asarray(omap)-gold))/np.sqrt(np.mean(gold**2)) < 1e-10
	return omap

def cyl_body():
	"""band map (case 'cyl'): alm2map agrees with the rows of the full-sky map; map2alm with Jacobi
	iterations approaches the input alm (curvedsky.py:843-873, 1122-1136)"""
	from oracle import sht_oracle as so
	lmax = 14
	fshape, fwcs = enmap.fullsky_geometry(shape=(24, 32))
	alm = so.rand_alm_simple(lmax, 3, 4, spin=(0, 2))
	full = enmap.zeros((3,)+fshape, fwcs); curvedsky.alm2map(alm, full, spin=[0, 2])
	w = fwcs.deepcopy(); w.wcs.crpix[1] -= 5
	band = enmap.zeros((3, 14, 32), w)
	assert curvedsky.analyse_geometry(band.shape, band.wcs).case == "cyl"
	curvedsky.alm2map(alm, band, spin=[0, 2])
	assert np.max(np.abs(np.asarray(band)-np.asarray(full)[:, 5:19])) < 1e-12
	a0 = curvedsky.map2alm(full, lmax=lmax, spin=[0, 2], method="cyl", niter=0)
	a3 = curvedsky.map2alm(full, lmax=lmax, spin=[0, 2], method="cyl", niter=3)
	e0 = np.std(a0-alm)/np.std(alm); e3 = np.std(a3-alm)/np.std(alm)
	assert e3 <= e0*1.0001 and e3 < 1e-6
	# adjoint consistency of the cyl path
	m = np.random.default_rng(0).standard_normal((1, 14, 32)); bm = enmap.ndmap(m, w)
	at = curvedsky.alm2map_adjoint(bm, spin=0, ainfo=curvedsky.alm_info(lmax))
	sa = enmap.zeros((1, 14, 32), w); curvedsky.alm2map(alm[:1], sa, spin=0)
	wt = np.full(alm.shape[1], 2.0); wt[:lmax+1] = 1
	at[:, :lmax+1] = at[:, :lmax+1].real
	assert abs(np.sum(np.asarray(sa)*m)-np.sum(wt*(alm[:1].real*at.real+alm[:1].imag*at.imag))) < 1e-10

def padding_body():
	"""patches that need padding (curvedsky.py:766-772, 786-792, 832-841, 866-871): a band of a full-sky grid through
	method="2d" (ypad), and a patch with partial rows (case "partial", xpad) through "cyl": synthesis must equal the
	crop of the full-sky map, analysis/adjoints must equal the same call on the zero-padded full map."""
	from oracle import sht_oracle as so
	lmax = 10
	fshape, fwcs = enmap.fullsky_geometry(shape=(20, 24))
	alm = so.rand_alm_simple(lmax, 3, 7, spin=(0, 2))
	full = enmap.zeros((3,)+fshape, fwcs); curvedsky.alm2map(alm, full, spin=[0, 2])
	fullv = np.asarray(full)
	# band rows 3..16 of the full grid, forced through the 2d method
	wb = fwcs.deepcopy(); wb.wcs.crpix[1] -= 3
	band = enmap.zeros((3, 14, 24), wb)
	mi = curvedsky.analyse_geometry(band.shape, band.wcs)
	assert mi.case == "cyl" and tuple(int(v) for v in mi.ypad) != (0, 0)
	curvedsky.alm2map(alm, band, spin=[0, 2], method="2d")
	assert np.max(np.abs(np.asarray(band)-fullv[:, 3:17])) < 1e-12
	padded = np.zeros_like(fullv); padded[:, 3:17] = fullv[:, 3:17]
	a_band = curvedsky.map2alm(enmap.ndmap(fullv[:, 3:17].copy(), wb), lmax=lmax, spin=[0, 2], method="2d")
	a_pad  = curvedsky.map2alm(enmap.ndmap(padded, fwcs), lmax=lmax, spin=[0, 2], method="2d")
	assert np.max(np.abs(a_band-a_pad)) < 1e-13
	# patch: rows 3..16, columns 5..19 -> case "partial"
	wp = wb.deepcopy(); wp.wcs.crpix[0] -= 5
	patch = enmap.zeros((3, 14, 15), wp)
	mi = curvedsky.analyse_geometry(patch.shape, patch.wcs)
	assert mi.case == "partial" and curvedsky.get_method(patch.shape, patch.wcs) == "cyl"
	curvedsky.alm2map(alm, patch, spin=[0, 2])
	assert np.max(np.abs(np.asarray(patch)-fullv[:, 3:17, 5:20])) < 1e-12
	rowpad = np.zeros((3, 14, 24)); rowpad[:, :, 5:20] = fullv[:, 3:17, 5:20]
	a_patch = curvedsky.map2alm(enmap.ndmap(fullv[:, 3:17, 5:20].copy(), wp), lmax=lmax, spin=[0, 2], niter=1)
	a_rows  = curvedsky.map2alm(enmap.ndmap(rowpad, wb), lmax=lmax, spin=[0, 2], niter=1)
	assert np.max(np.abs(a_patch-a_rows)) < 1e-13
	# adjoint of synthesis on the patch = adjoint on the zero-padded rows
	m = np.random.default_rng(1).standard_normal((1, 14, 15))
	rp = np.zeros((1, 14, 24)); rp[:, :, 5:20] = m
	at1 = curvedsky.alm2map_adjoint(enmap.ndmap(m, wp), spin=0, ainfo=curvedsky.alm_info(lmax))
	at2 = curvedsky.alm2map_adjoint(enmap.ndmap(rp, wb), spin=0, ainfo=curvedsky.alm_info(lmax))
	assert np.max(np.abs(at1-at2)) < 1e-13

@pytest.mark.hostsim
def test_padding_hostsim(): padding_body()
@pytest.mark.gpu
def test_padding_gpu(): padding_body()

@pytest.mark.hostsim
def test_roundtrip_hostsim(): roundtrip_body(8)
@pytest.mark.hostsim
def test_alm_conversion_hostsim(): alm_conversion_body()
@pytest.mark.hostsim
def test_adjointness_hostsim(): adjointness_body(variants=("fejer1",), ncomps=(1,), do_analysis=False)   # (analysis adjoint vs oracle: test_sht_parity)
@pytest.mark.hostsim
def test_cyl_hostsim(): cyl_body()

@pytest.mark.gpu
def test_roundtrip_gpu(): roundtrip_body(30)
@pytest.mark.gpu
def test_alm_conversion_gpu(): alm_conversion_body()
@pytest.mark.gpu
def test_adjointness_gpu(): adjointness_body(do_analysis=True, dtypes=(np.float64, np.float32))
@pytest.mark.gpu
def test_golden_unlensed_gpu(golden_dir): golden_body(golden_dir)
@pytest.mark.gpu
def test_cyl_gpu(): cyl_body()

@pytest.mark.gpu
def test_device_resident_gpu():
	"""torch tensors stay on the GPU: dmap + tensor alm give the same result as numpy staging"""
	import torch
	from oracle import sht_oracle as so
	lmax = 64; shape, wcs = enmap.fullsky_geometry(shape=(80, 160))
	alm = so.rand_alm_simple(lmax, 3, 9, spin=(0, 2))
	ref = enmap.zeros((3,)+shape, wcs); curvedsky.alm2map(alm, ref, spin=[0, 2])
	talm = torch.from_numpy(alm).cuda(); tmap = enmap.dmap(torch.zeros((3,)+shape, dtype=torch.float64, device="cuda"), wcs)
	curvedsky.alm2map(talm, tmap, spin=[0, 2])
	assert np.max(np.abs(tmap.tensor.cpu().numpy()-np.asarray(ref))) < 1e-13
	out = curvedsky.map2alm(tmap, lmax=lmax, spin=[0, 2])
	assert out.is_cuda and np.max(np.abs(out.cpu().numpy()-alm)) < 1e-12

@pytest.mark.gpu
def test_config5_pipeline_gpu():
	"""BASELINE config 5 at reduced size, device resident end to end: rand_alm -> alm2map -> (enmap.map2harm phys -> calc_ps2d
	-> lbin) and (map2alm -> alm2cl).  Checks: alm2cl of the recovered alm equals alm2cl of the input (1e-10); the binned
	flat-sky spectrum of the band around the equator follows the input C_l to the sample variance expected of it."""
	import torch
	lmax = 300
	cl_in = 1.0/(np.arange(lmax+1)+10.0)**2
	alm, ainfo = curvedsky.rand_alm(cl_in, lmax=lmax, seed=11, return_ainfo=True)
	shape, wcs = enmap.fullsky_geometry(shape=(lmax+20, 2*lmax+40))
	dalm = torch.from_numpy(alm[None]).cuda()
	m = enmap.dmap(torch.zeros((1,)+tuple(shape), dtype=torch.float64, device="cuda"), wcs)
	curvedsky.alm2map(dalm, m, spin=0, ainfo=ainfo)
	back = curvedsky.map2alm(m, lmax=lmax, spin=0)
	assert back.is_cuda
	c0 = curvedsky.alm2cl(dalm[0], ainfo=ainfo).cpu().numpy(); c1 = curvedsky.alm2cl(back[0], ainfo=ainfo).cpu().numpy()
	np.testing.assert_allclose(c1, c0, rtol=1e-9, atol=1e-14)
	# flat-sky side on an equatorial band of the same map (rows within +-15 degrees)
	ny = shape[0]; r0 = int(ny*75/180); r1 = ny-r0
	wb = wcs.deepcopy(); wb.wcs.crpix[1] -= r0
	band = enmap.dmap(m.tensor[:, r0:r1].contiguous(), wb)
	h = enmap.map2harm(band, normalize="phys", spin=[0])
	ps = enmap.calc_ps2d(h[0])
	b, l = enmap.lbin(ps, brel=4)
	ok = (l > 30) & (l < 0.8*lmax) & np.isfinite(b)
	ratio = b[ok]/np.interp(l[ok], np.arange(lmax+1), cl_in)
	assert 0.7 < np.median(ratio) < 1.3

def edge_body():
	"""edge cases of the reference interface: lmax 0 and 1, explicit alm layouts (stride 2, rectangular), all-scalar
	spin list on 3 components, unpaired spin component (IndexError, enmap.py:3384), zero-size pre-dimension"""
	from oracle import sht_oracle as so
	shape, wcs = enmap.fullsky_geometry(shape=(12, 20))       # nx > 2 lmax: no aliased m in the round trips
	for lmax in (0, 1):
		ai = curvedsky.alm_info(lmax)
		alm = (np.arange(ai.nelem)+1.0).astype(np.complex128)
		m = enmap.zeros(shape, wcs); curvedsky.alm2map(alm, m, spin=0, ainfo=ai)
		if lmax == 0: assert np.allclose(np.asarray(m), 1/np.sqrt(4*np.pi), rtol=1e-13)
		back = curvedsky.map2alm(m, ainfo=ai, spin=0)
		np.testing.assert_allclose(back, alm, atol=1e-12)
	lmax = 8
	tri = curvedsky.alm_info(lmax)
	alm = so.rand_alm_simple(lmax, 3, 9, spin=(0, 2))
	ref = enmap.zeros((3,)+shape, wcs); curvedsky.alm2map(alm, ref, spin=[0, 2], ainfo=tri)
	for ai in (curvedsky.alm_info(lmax, stride=2), curvedsky.alm_info(lmax, layout="rect"), curvedsky.alm_info(lmax=lmax, mmax=lmax, layout=tri.mstart[::-1].copy()*0+tri.mstart)):
		a2 = curvedsky.transfer_alm(tri, alm, ai)
		m2 = enmap.zeros((3,)+shape, wcs); curvedsky.alm2map(a2, m2, spin=[0, 2], ainfo=ai)
		np.testing.assert_allclose(np.asarray(m2), np.asarray(ref), atol=1e-12)
		b2 = curvedsky.map2alm(ref, ainfo=ai, spin=[0, 2])
		np.testing.assert_allclose(curvedsky.transfer_alm(ai, b2, tri), alm, atol=1e-11)
	m3 = enmap.zeros((3,)+shape, wcs); curvedsky.alm2map(alm, m3, spin=[0], ainfo=tri)          # three scalar maps
	for i in range(3):
		mi = enmap.zeros(shape, wcs); curvedsky.alm2map(alm[i], mi, spin=0, ainfo=tri)
		np.testing.assert_allclose(np.asarray(m3)[i], np.asarray(mi), atol=1e-13)
	with pytest.raises(IndexError):
		curvedsky.alm2map(alm[:2], enmap.zeros((2,)+shape, wcs), spin=[0, 2], ainfo=tri)          # spin-2 needs a pair
	e = curvedsky.alm2map(np.zeros((0, 3, tri.nelem), complex), enmap.zeros((0, 3)+shape, wcs), spin=[0, 2], ainfo=tri)
	assert e.shape == (0, 3)+tuple(shape)

@pytest.mark.hostsim
def test_edges_hostsim(): edge_body()
@pytest.mark.gpu
def test_edges_gpu(): edge_body()

def deriv_cyl_body():
	"""map2alm(deriv=True) on the cyl path (curvedsky.py:1067-1076) is Y_grad^T W: on the gradient maps of
	alm2map(deriv=True) it returns l(l+1) a_lm up to the accuracy of the ring weights for the 1/sin(theta) terms of the
	gradient (exact to 1e-6 for l <= 6 on this grid, per cent level at l ~ lmax)"""
	from oracle import sht_oracle as so
	lmax = 12
	shape, wcs = enmap.fullsky_geometry(shape=(20, 32))
	alm = so.rand_alm_simple(lmax, 1, 3, spin=(0,))[0]
	ai = curvedsky.alm_info(lmax)
	g = enmap.zeros((2,)+shape, wcs); curvedsky.alm2map(alm, g, deriv=True)
	with pytest.raises(NotImplementedError): curvedsky.map2alm(g, lmax=lmax, deriv=True, method="2d")
	back = curvedsky.map2alm(g, lmax=lmax, deriv=True, method="cyl")
	assert back.shape == alm.shape
	l = np.concatenate([np.arange(m, lmax+1) for m in range(lmax+1)])
	lo = l <= 6
	np.testing.assert_allclose(back[lo], (l*(l+1)*alm)[lo], atol=1e-5)
	assert np.max(np.abs(back-l*(l+1)*alm)) < 0.05*np.max(np.abs(l*(l+1)*alm))
	# adjoint pair: <map2alm_adjoint(a), m> == <a, map2alm(m)> with niter = 0
	rng = np.random.default_rng(2)
	m = enmap.ndmap(rng.standard_normal((2,)+shape), wcs)
	a1 = curvedsky.map2alm(m, lmax=lmax, deriv=True, method="cyl")
	mt = enmap.zeros((2,)+shape, wcs); curvedsky.map2alm(mt, alm=alm.copy(), lmax=lmax, deriv=True, method="cyl", adjoint=True)
	wt = np.full(alm.shape, 2.0); wt[:lmax+1] = 1
	a1[:lmax+1] = a1[:lmax+1].real
	lhs = np.sum(np.asarray(mt)*np.asarray(m)); rhs = np.sum(wt*(alm.real*a1.real+alm.imag*a1.imag))
	assert abs(lhs-rhs) < 1e-10*max(1.0, abs(rhs))

@pytest.mark.hostsim
def test_deriv_cyl_hostsim(): deriv_cyl_body()
@pytest.mark.gpu
def test_deriv_cyl_gpu(): deriv_cyl_body()

def filter_randmap_body():
	"""curvedsky.filter and rand_map (curvedsky.py:17-37, 654-671): filtering a band-limited map with 1 returns it, with
	a top-hat it removes exactly the other multipoles; rand_map reproduces alm2map(rand_alm(seed))"""
	lmax = 8
	shape, wcs = enmap.fullsky_geometry(shape=(12, 20))
	cl = 1.0/(np.arange(lmax+1)+1.0)**2
	m = curvedsky.rand_map((3,)+tuple(shape), wcs, np.array([[cl, 0*cl, 0*cl], [0*cl, cl, 0*cl], [0*cl, 0*cl, cl]]), lmax=lmax, seed=4)
	alm = curvedsky.rand_alm(np.array([[cl, 0*cl, 0*cl], [0*cl, cl, 0*cl], [0*cl, 0*cl, cl]]), lmax=lmax, seed=4)
	ref = enmap.zeros((3,)+tuple(shape), wcs); curvedsky.alm2map(alm, ref, spin=[0, 2])
	np.testing.assert_allclose(np.asarray(m), np.asarray(ref), atol=1e-13)
	t = enmap.ndmap(np.asarray(m)[0].copy(), wcs)
	np.testing.assert_allclose(np.asarray(curvedsky.filter(t, lambda l: 1+0*l, lmax=lmax)), np.asarray(t), atol=1e-12)
	lo = curvedsky.filter(t, lambda l: 1.0*(l <= 4), lmax=lmax)
	a_lo = curvedsky.map2alm(lo, lmax=lmax, spin=0); a_t = curvedsky.map2alm(t, lmax=lmax, spin=0)
	ai = curvedsky.alm_info(lmax); l = np.concatenate([np.arange(mm, lmax+1) for mm in range(lmax+1)])
	np.testing.assert_allclose(a_lo, np.where(l <= 4, a_t, 0), atol=1e-12)

@pytest.mark.hostsim
def test_filter_randmap_hostsim(): filter_randmap_body()
@pytest.mark.gpu
def test_filter_randmap_gpu(): filter_randmap_body()

@pytest.mark.gpu
def test_repeatable_and_cyl_equals_2d_gpu(monkeypatch):
	"""repeated device-resident calls on one plan give the same results (no stale scratch, no ordering hazards between the
	fused FFT kernels of consecutive calls) -- bit for bit with PXS_DETERMINISTIC=1, to rounding in the default mode, where the
	waves of one m add their moments with atomics in no fixed order -- and the explicit-ring path agrees with the named-grid path"""
	import torch
	from oracle import sht_oracle as so
	lmax = 700                      # 365 ring pairs: two waves per m in the spin-2 analysis
	shape, wcs = enmap.fullsky_geometry(shape=(lmax+30, 2*lmax+60))
	alm = torch.from_numpy(so.rand_alm_simple(lmax, 3, 12, spin=(0, 2))).cuda()
	def run():
		res = []
		for rep in range(2):
			m = enmap.dmap(torch.zeros((3,)+tuple(shape), dtype=torch.float64, device="cuda"), wcs)
			for _ in range(3):
				curvedsky.alm2map(alm, m, spin=[0, 2])
				back = curvedsky.map2alm(m, lmax=lmax, spin=[0, 2])
			at = curvedsky.alm2map_adjoint(m, spin=[0, 2], ainfo=curvedsky.alm_info(lmax))
			mc = enmap.dmap(torch.zeros((3,)+tuple(shape), dtype=torch.float64, device="cuda"), wcs)
			curvedsky.alm2map(alm, mc, spin=[0, 2], method="cyl")                       # explicit-ring path
			atc = curvedsky.alm2map_adjoint(mc, spin=[0, 2], method="cyl", ainfo=curvedsky.alm_info(lmax))
			torch.cuda.synchronize()
			res.append((m.tensor.cpu().numpy().copy(), back.cpu().numpy().copy(), at.cpu().numpy().copy(), mc.tensor.cpu().numpy().copy(), atc.cpu().numpy().copy()))
		return res
	monkeypatch.setattr(sht, "_deterministic", True)
	res = run()
	for a, b in zip(res[0], res[1]): assert np.array_equal(a, b)
	monkeypatch.setattr(sht, "_deterministic", None)
	res2 = run()
	for a, b, c in zip(res2[0], res2[1], res[0]):
		assert np.max(np.abs(a-b)) <= 1e-13*np.max(np.abs(a))
		assert np.max(np.abs(a-c)) <= 1e-13*np.max(np.abs(a))                                # both schemes agree to rounding
	assert np.max(np.abs(res[0][1]-alm.cpu().numpy())) < 1e-11
	assert np.max(np.abs(res[0][3]-res[0][0])) < 1e-11                                  # cyl and 2d agree
	assert np.max(np.abs(res[0][4]-res[0][2])) < 1e-11*np.max(np.abs(res[0][2]))

@pytest.mark.gpu
def test_host_array_route_gpu(monkeypatch):
	"""numpy maps / alm through the slab route (pixell_amd/hostio.py: pinned double-buffered transfers, the spin groups of a call
	pipelined in the background) give the bytes the device-resident call gives; small slabs so that many of them are in play"""
	import torch
	from pixell_amd import hostio
	monkeypatch.setattr(hostio, "SLAB_BYTES", 24 << 20); monkeypatch.setattr(hostio, "MIN_BYTES", 16 << 20)
	monkeypatch.setattr(sht, "_deterministic", True)      # (the default analysis adds with atomics: equal to rounding, not bit for bit, even between two device-resident calls)
	lmax = 1500
	shape, wcs = enmap.fullsky_geometry(shape=(2000, 4000))
	ainfo = curvedsky.alm_info(lmax)
	rng = np.random.default_rng(4)
	alm = (rng.standard_normal((3, ainfo.nelem))+1j*rng.standard_normal((3, ainfo.nelem)))/(1.0+np.arange(ainfo.nelem) % 97)
	alm[:, :lmax+1] = alm[:, :lmax+1].real
	# device-resident reference
	d_alm = torch.from_numpy(alm).cuda()
	d_map = enmap.dmap(torch.zeros((3,)+shape, dtype=torch.float64, device="cuda"), wcs)
	curvedsky.alm2map(d_alm, d_map, spin=[0, 2], ainfo=ainfo)
	d_back = torch.zeros_like(d_alm); curvedsky.map2alm(d_map, alm=d_back, spin=[0, 2], ainfo=ainfo)
	# host arrays (every array here is above MIN_BYTES except the T alm)
	h_map = enmap.ndmap(np.full((3,)+shape, np.nan), wcs)
	curvedsky.alm2map(alm.copy(), h_map, spin=[0, 2], ainfo=ainfo)
	assert np.array_equal(np.asarray(h_map), d_map.tensor.cpu().numpy())
	h_back = np.full_like(alm, np.nan); curvedsky.map2alm(h_map, alm=h_back, spin=[0, 2], ainfo=ainfo)
	assert np.array_equal(h_back, d_back.cpu().numpy())
	# the ducc-shaped route (one call per spin group, no pipeline) through the same slabs

	mi = curvedsky.analyse_geometry(h_map.shape, wcs)
	kw = dict(lmax=lmax, mstart=ainfo.mstart, geometry=mi.ducc_geo.name, phi0=mi.phi0, flip=tuple(bool(f) for f in mi.flip))
	m2 = np.full((2,)+shape, np.nan); sht.synthesis_2d(alm=alm[1:], map=m2, spin=2, **kw)
	assert np.array_equal(m2, d_map.tensor[1:].cpu().numpy())
	a2 = np.zeros_like(alm[1:]); sht.analysis_2d(alm=a2, map=m2, spin=2, **kw)
	assert np.array_equal(a2, d_back[1:].cpu().numpy())
	# float32 maps, adjoint directions
	x = rng.standard_normal((1,)+shape).astype(np.float32)
	ax_h = np.zeros((1, ainfo.nelem), np.complex64); curvedsky.alm2map_adjoint(enmap.ndmap(x, wcs), alm=ax_h, spin=[0], ainfo=ainfo)
	ax_d = torch.zeros((1, ainfo.nelem), dtype=torch.complex64, device="cuda"); curvedsky.alm2map_adjoint(enmap.dmap(torch.from_numpy(x).cuda(), wcs), alm=ax_d, spin=[0], ainfo=ainfo)
	assert np.array_equal(ax_h, ax_d.cpu().numpy())
	sht.clear_plans(); torch.cuda.empty_cache()

def prof2alm_body(golden_dir):
	"""curvedsky.prof2alm (curvedsky.py:556-580) without the rotation: the m = 0 alm of the reference (tests/golden/make_prof2alm.py), their
	expansion to the full layout for the default direction (the identity rotation), NotImplementedError for any other"""
	d = np.load(os.path.join(golden_dir, "prof2alm.npz"))
	a = curvedsky.prof2alm(d["prof_cc"], norot=True)
	assert a.shape == d["alm_cc"].shape and np.abs(a-d["alm_cc"]).max() < 1e-12*np.abs(d["alm_cc"]).max()
	b = curvedsky.prof2alm(d["prof_f1"], spin=[0, 2], geometry="F1", norot=True)
	assert b.shape == d["alm_f1"].shape and np.abs(b-d["alm_f1"]).max() < 1e-12*np.abs(d["alm_f1"]).max()
	full = curvedsky.prof2alm(d["prof_cc"])
	lmax = d["alm_cc"].shape[-1]-1
	assert full.shape == ((lmax+1)*(lmax+2)//2,) and np.array_equal(full[:lmax+1], a) and not full[lmax+1:].any()
	with pytest.raises(NotImplementedError): curvedsky.prof2alm(d["prof_cc"], dir=[0.3, 0.2])

@pytest.mark.hostsim
def test_prof2alm_hostsim(golden_dir): prof2alm_body(golden_dir)
@pytest.mark.gpu
def test_prof2alm_gpu(golden_dir): prof2alm_body(golden_dir)
