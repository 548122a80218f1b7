"""Generate the committed golden fixtures by running the REFERENCE in this container.

Run:  python tests/golden/make_golden.py      (needs /root/reference; never run on the GPU box)

What it does
  1. imports the reference through tests/golden/_ref_harness.py with the repo's CPU oracle
     mounted as `ducc0.sht.experimental`;
  2. PINS THE ORACLE: reference `curvedsky.alm2map(lensing.rand_alm(seed=1))` on the CC 1-degree
     grid must reproduce the reference's golden file tests/data/MM_unlensed_071123.fits
     (tests/test_pixell.py:208-217,351-360), and the reference's own round-trip / adjointness /
     dtype tests (tests/test_pixell.py:870-965, 1028-1085) must pass on top of the oracle;
  3. writes fixtures (inputs + expected outputs only) next to this script:
       lens_unlensed.npz   alm input (reference rand_alm, legacy RNG) + expected map (the FITS data)
       geometry.json       analyse_geometry / get_ring_info / alm_info results for the path's geometries
       fft_golden.npz      pixell.fft / enmap.fft outputs from the reference's numpy engine
       alm_ops.npz         cmisc alm2cl / lmul outputs (next-row f1)
       fft_ops.npz         pixell.fft.shift / resample / resample_fft (f4)
       flatsky.npz         enmap.map2harm / harm2map / calc_ps2d / lbin / laxes / extent (f3)
       alm_rand.npz        rand_alm / rand_alm_white / transpose_alm / lmul matrix form / alm2cl dtypes (f1)
"""
import sys, os, json, types
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
from oracle import sht_oracle as so
import _ref_harness as H

def read_fits_f64(fname):
	with open(fname, "rb") as f: raw = f.read()
	hdr = raw[:2880].decode("ascii")
	cards = {hdr[i:i+8].strip(): hdr[i+10:i+80].split("/")[0].strip() for i in range(0, 2880, 80)}
	assert int(cards["BITPIX"]) == -64
	shape = tuple(int(cards["NAXIS%d" % i]) for i in range(int(cards["NAXIS"]), 0, -1))
	n = int(np.prod(shape))
	return np.frombuffer(raw[2880:2880+8*n], ">f8").reshape(shape).astype(np.float64), cards

def alm_rand_fixture(curvedsky):
	"""rand_alm / rand_alm_white / alm2cl / lmul (matrix and broadcasting forms) of the reference -> alm_rand.npz"""
	rng = np.random.default_rng(77)
	lmax = 24; nl = lmax+1
	A = rng.standard_normal((3, 3, nl)); ps3 = np.einsum("ikl,jkl->ijl", A, A)+0.1*np.eye(3)[:, :, None]
	ps1 = rng.random(nl)+0.1
	ps2 = np.stack([ps3[0, 0], ps3[1, 1], ps3[2, 2], ps3[0, 1]])       # healpy 'diag' order, truncated
	out = dict(lmax=lmax, ps3=ps3, ps1=ps1, ps2=ps2)
	out["alm3"] = curvedsky.rand_alm(ps3, seed=5)
	out["alm1"] = curvedsky.rand_alm(ps1, seed=6)
	out["alm2"] = curvedsky.rand_alm(ps2, seed=7)
	out["alm3_sp"] = curvedsky.rand_alm(ps3, seed=5, dtype=np.complex64)
	out["alm_lmax"] = curvedsky.rand_alm(ps1, lmax=30, seed=8)          # spectrum padded with zeros
	ai = curvedsky.alm_info(lmax)
	out["white"] = curvedsky.rand_alm_white(ai, pre=[2], seed=9)
	out["white_lmajor"] = curvedsky.rand_alm_white(ai, pre=[2], seed=9, m_major=False)
	ai2 = curvedsky.alm_info(lmax=lmax, mmax=10)
	w = rng.standard_normal((2, ai2.nelem))+1j*rng.standard_normal((2, ai2.nelem))
	out["mmax_alm"] = w; out["mmax_tr"] = ai2.transpose_alm(w)
	out["mmax_cl"] = ai2.alm2cl(w[:, None], w[None, :])
	lm = rng.standard_normal((2, 3, nl-4))                              # short filter: zero beyond its end
	out["lmat"] = lm; out["lmatmul"] = ai.lmul(out["alm3"], lm)
	lb = rng.standard_normal((3, nl))
	out["lbro"] = lb; out["lmul_bro"] = ai.lmul(out["alm3"], lb)
	sp = out["alm3_sp"]
	out["cl_sp"] = ai.alm2cl(sp[:, None], sp[None, :]); out["cl_sp_dp"] = ai.alm2cl(sp[:, None], sp[None, :], dtype=np.float64)
	out["cl_cross"] = ai.alm2cl(out["alm3"][:, None], out["white"][None, :])
	out["xfl_fun"] = curvedsky.almxfl(out["alm1"], lambda l: 1/(1+l)**2)
	out["transfer_down"] = curvedsky.transfer_alm(ai, out["alm3"], curvedsky.alm_info(lmax=16, mmax=9))
	out["transfer_up"] = curvedsky.transfer_alm(ai2, w, curvedsky.alm_info(lmax=30, layout="rect"))
	out["transfer_add"] = curvedsky.transfer_alm(ai2, w, ai, oalm=out["white"].copy(), op=lambda a, b: a+b)
	np.savez_compressed(os.path.join(HERE, "alm_rand.npz"), **out)

def flatsky_fixture(enmap):
	"""enmap.map2harm / harm2map / calc_ps2d / lbin / laxes / extent of the reference (numpy FFT engine) -> flatsky.npz"""
	rng = np.random.default_rng(31)
	shape, wcs = enmap.band_geometry(np.deg2rad(20), res=np.deg2rad(2.0))
	shape = tuple(int(v) for v in shape)
	m = enmap.ndmap(rng.standard_normal((3,)+shape), wcs)
	out = dict(map=np.asarray(m), cdelt=np.array(wcs.wcs.cdelt), crval=np.array(wcs.wcs.crval), crpix=np.array(wcs.wcs.crpix))
	out["extent"] = enmap.extent(shape, wcs); out["extent_signed"] = enmap.extent(shape, wcs, signed=True)
	ly, lx = enmap.laxes(shape, wcs); out["ly"] = ly; out["lx"] = lx
	out["modlmap"] = np.asarray(enmap.modlmap(shape, wcs)); out["pixsize"] = enmap.pixsize(shape, wcs)
	h = enmap.map2harm(m); out["harm"] = np.asarray(h)
	out["harm_phys"] = np.asarray(enmap.map2harm(m, normalize="phys"))
	out["harm_iau"] = np.asarray(enmap.map2harm(m, iau=True))
	out["harm_spin1"] = np.asarray(enmap.map2harm(m[:2], spin=1))
	out["harm_adj"] = np.asarray(enmap.harm2map_adjoint(m, normalize="phys"))
	c = enmap.ndmap(rng.standard_normal((3,)+shape)+1j*rng.standard_normal((3,)+shape), wcs)
	out["cplx"] = np.asarray(c)
	out["back"] = np.asarray(enmap.harm2map(c)); out["back_phys_keep"] = np.asarray(enmap.harm2map(c, normalize="phys", keep_imag=True))
	out["back_adj"] = np.asarray(enmap.map2harm_adjoint(c))
	ps = enmap.calc_ps2d(h[:, None], h[None, :]); out["ps2d"] = np.asarray(ps)
	out["ps2d_cross"] = np.asarray(enmap.calc_ps2d(h[0], c[1]))
	out["ps2d_sp"] = np.asarray(enmap.calc_ps2d(h[0].astype(np.complex64)))
	b, l, nhit = enmap.lbin(ps[0, 0], return_nhit=True); out["lbin_b"] = b; out["lbin_l"] = l; out["lbin_nhit"] = nhit
	b3, l3 = enmap.lbin(enmap.ndmap(np.asarray(ps)[:, 0], wcs), brel=2.5, return_bins=True); out["lbin3_b"] = b3; out["lbin3_l"] = l3
	np.savez_compressed(os.path.join(HERE, "flatsky.npz"), **out)

def fftops_fixture(pfft):
	"""pixell.fft.shift / resample / resample_fft with the reference's numpy engine -> fft_ops.npz"""
	rng = np.random.default_rng(41)
	a = rng.standard_normal((3, 12, 20)); c = a+1j*rng.standard_normal(a.shape)
	out = dict(a=a, c=c)
	out["shift1"] = pfft.shift(a, 2.5); out["shift2"] = pfft.shift(a, [1.25, -3.0]); out["shiftc"] = pfft.shift(c, [0.5], axes=[1])
	out["shift_deriv"] = pfft.shift(a, [0.0, 0.0], deriv=1)
	out["res_up"] = pfft.resample(a, 31); out["res_dn"] = pfft.resample(a, (7, 9)); out["res_c"] = pfft.resample(c, 33, axes=[1])
	fa = np.fft.fft2(c)
	out["rfft_a"] = pfft.resample_fft(fa, (16, 11), axes=(-2, -1), norm=0.5)
	out["rfft_op"] = pfft.resample_fft(fa, 25, out=np.ones((3, 12, 25), complex), op=lambda x, y: x+y)
	np.savez_compressed(os.path.join(HERE, "fft_ops.npz"), **out)

def main():
	sht = types.ModuleType("sht_exp")
	for name in ["synthesis_2d", "adjoint_synthesis_2d", "analysis_2d", "adjoint_analysis_2d",
			"synthesis", "adjoint_synthesis", "get_gridweights"]:
		setattr(sht, name, getattr(so, name))
	ns = H.load_reference(sht)
	enmap, curvedsky, powspec, lensing, utils, pfft = ns.enmap, ns.curvedsky, ns.powspec, ns.lensing, ns.utils, ns.fft
	data = "/root/reference/tests/data/"
	if "--only-alm-rand" in sys.argv:
		alm_rand_fixture(curvedsky); print("alm_rand.npz written"); return
	if "--only-fftops" in sys.argv:
		pfft.set_engine("numpy"); fftops_fixture(pfft); print("fft_ops.npz written"); return
	if "--only-flatsky" in sys.argv:
		pfft.set_engine("numpy"); flatsky_fixture(enmap); print("flatsky.npz written"); return

	# ---- 1. the one true golden vector -------------------------------------------------
	shape, wcs = enmap.fullsky_geometry(res=np.deg2rad(1.0), variant="CC")
	shape = (3,)+tuple(int(s) for s in shape)
	ps_cmb, ps_lens = powspec.read_camb_scalar(data+"test_scalCls.dat")
	ps_lensinput = np.zeros((4, 4, ps_cmb.shape[-1]))
	ps_lensinput[0, 0] = ps_lens
	ps_lensinput[1:, 1:] = ps_cmb
	phi_alm, cmb_alm, ainfo = lensing.rand_alm(ps_lensinput, lmax=400, seed=1, ncomp=3)
	omap = enmap.zeros(shape, wcs, np.float64)
	curvedsky.alm2map(cmb_alm, omap, spin=[0, 2])
	gold, cards = read_fits_f64(data+"MM_unlensed_071123.fits")
	ok = np.isclose(np.asarray(omap), gold)
	rel = np.max(np.abs(np.asarray(omap)-gold), axis=(1, 2))/np.sqrt(np.mean(gold**2, axis=(1, 2)))
	print("MM_unlensed: all isclose =", bool(ok.all()), " max|d|/rms per comp =", rel)
	assert ok.all(), "oracle does not reproduce the reference's golden alm2map output"
	minfo = curvedsky.analyse_geometry(shape, wcs)
	np.savez_compressed(os.path.join(HERE, "lens_unlensed.npz"), alm=cmb_alm, map=gold,
		mstart=ainfo.mstart, lmax=ainfo.lmax, phi0=float(minfo.phi0), flip=np.array(minfo.flip),
		geometry=str(minfo.ducc_geo.name), cdelt=np.array(wcs.wcs.cdelt), crval=np.array(wcs.wcs.crval),
		crpix=np.array(wcs.wcs.crpix))

	# ---- 2. the reference's own invariants on top of the oracle ------------------------
	# test_alm2map_2d_roundtrip (tests/test_pixell.py:870-965), condensed
	lmax = 30; ai = curvedsky.alm_info(lmax)
	shp, w = enmap.fullsky_geometry(shape=(lmax+2, 2*lmax+1))
	i = ai.lm2ind(lmax, lmax)
	for dt, ct in [(np.float64, np.complex128), (np.float32, np.complex64)]:
		for use_oalm in [False, True]:
			alm = np.zeros(ai.nelem, ct); alm[i] = 1+1j
			m = enmap.zeros(shp, w, dt); curvedsky.alm2map(alm, m, spin=0)
			out = curvedsky.map2alm(m, alm=np.zeros_like(alm) if use_oalm else None, spin=0, ainfo=ai)
			np.testing.assert_array_almost_equal(out, alm)
			alm = np.zeros((3, 2, ai.nelem), ct)
			alm[0, 0, i] = 1+1j; alm[0, 1, i] = 2-2j; alm[1, 0, i] = 3+3j; alm[1, 1, i] = 4-4j; alm[2, 0, i] = 5+5j; alm[2, 1, i] = 6-6j
			m = enmap.zeros((3, 2)+tuple(shp), w, dt); curvedsky.alm2map(alm, m, spin=1)
			out = curvedsky.map2alm(m, alm=np.zeros_like(alm) if use_oalm else None, spin=1, ainfo=ai)
			np.testing.assert_array_almost_equal(out, alm)
	print("reference round-trip test passes on the oracle")
	# test_alm_conversion (tests/test_pixell.py:1028-1046)
	alm = np.zeros(ai.nelem, np.complex64); alm[i] = 1+1j
	m = enmap.zeros(shp, w, np.float64); curvedsky.alm2map(alm, m, spin=0)
	try:
		curvedsky.map2alm(m, alm, spin=0); raise AssertionError("expected ValueError")
	except ValueError: pass
	# test_adjointness (tests/test_pixell.py:1051-1085), fullsky fejer1 + cc, via explicit matrices
	sys.path.insert(0, "/root/reference/tests")
	def zip_alm(alm, ainfo):
		n = ainfo.lm2ind(1, 1)
		return np.concatenate([alm[..., :n].real, alm[..., n:].view(utils.real_dtype(alm.dtype))*2**0.5], -1)
	def unzip_alm(z, ainfo):
		n = ainfo.lm2ind(1, 1)
		o = np.zeros(z.shape[:-1]+(ainfo.nelem,), utils.complex_dtype(z.dtype))
		o[..., :n] = z[..., :n]; o[..., n:] = z[..., n:].view(o.dtype)/2**0.5
		return o
	def map_bash(fun, shape, wcs, ncomp, lmax):
		ainfo = curvedsky.alm_info(lmax); nz = int(2*ainfo.nelem-ainfo.lm2ind(1, 1))
		umap = enmap.zeros((ncomp,)+shape, wcs); oalm = np.zeros((ncomp, ainfo.nelem), complex)
		mat = np.zeros((ncomp, nz, ncomp)+shape)
		for I in utils.nditer((ncomp,)+shape):
			umap[I] = 1; oalm[:] = 0
			fun(map=umap, alm=oalm, ainfo=ainfo)
			mat[(slice(None), slice(None))+I] = zip_alm(oalm, ainfo); umap[I] = 0
		return mat
	def alm_bash(fun, shape, wcs, ncomp, lmax):
		ainfo = curvedsky.alm_info(lmax); nz = int(2*ainfo.nelem-ainfo.lm2ind(1, 1))
		z = np.zeros((ncomp, nz)); omap = enmap.zeros((ncomp,)+shape, wcs)
		mat = np.zeros((ncomp, nz, ncomp)+shape)
		for ci in range(ncomp):
			for k in range(nz):
				z[ci, k] = 1; omap[:] = 0
				fun(alm=unzip_alm(z, ainfo), map=omap, ainfo=ainfo)
				mat[ci, k] = omap; z[ci, k] = 0
		return mat
	res = 30*utils.degree
	for variant in ["fejer1", "cc"]:
		shp2, w2 = enmap.fullsky_geometry(res=res, variant=variant)
		shp2 = tuple(int(s) for s in shp2)
		_, wcc = enmap.fullsky_geometry(res=res, variant="cc")
		lm = 7-2
		for ncomp in [1, 3]:
			m1 = alm_bash(curvedsky.alm2map, shp2, w2, ncomp, lm)
			m2 = map_bash(curvedsky.alm2map_adjoint, shp2, w2, ncomp, lm)
			np.testing.assert_array_almost_equal(m1, m2)
			m1 = map_bash(curvedsky.map2alm, shp2, w2, ncomp, lm)
			m2 = alm_bash(curvedsky.map2alm_adjoint, shp2, w2, ncomp, lm)
			np.testing.assert_array_almost_equal(m1, m2)
	print("reference adjointness test (fullsky fejer1, cc) passes on the oracle")

	# ---- 3. geometry fixtures ---------------------------------------------------------
	geo = {}
	def add_geo(key, shape, wcs):
		mi = curvedsky.analyse_geometry(shape, wcs)
		d = dict(shape=[int(s) for s in shape[-2:]], cdelt=list(map(float, wcs.wcs.cdelt)),
			crval=list(map(float, wcs.wcs.crval)), crpix=list(map(float, wcs.wcs.crpix)),
			case=mi.case, flip=[bool(f) for f in mi.flip], phi0=float(mi.phi0),
			ypad=[int(v) for v in mi.ypad], xpad=[int(v) for v in mi.xpad],
			method=curvedsky.get_method(shape, wcs))
		if mi.ducc_geo is not None:
			d.update(name=mi.ducc_geo.name, ny=int(mi.ducc_geo.ny), nx=int(mi.ducc_geo.nx),
				yoff=int(mi.ducc_geo.yoff), lmax=int(mi.ducc_geo.lmax))
		if shape[-2] <= 2048:
			ri = curvedsky.get_ring_info(shape, wcs)
			d.update(theta_first=float(ri.theta[0]), theta_last=float(ri.theta[-1]), ring_phi0=float(ri.phi0[0]))
		geo[key] = d
	for key, shp in [("c1_1024x2048", (1024, 2048)), ("c2_5400x10800", (5400, 10800)),
			("c3_21600x43200", (21600, 43200)), ("c5_10800x21600", (10800, 21600)),
			("rt_32x61", (32, 61)), ("ref_bench_900x1800", (900, 1800))]:
		s, w = enmap.fullsky_geometry(shape=shp); add_geo(key, s, w)
	s, w = enmap.fullsky_geometry(res=np.deg2rad(1.0), variant="CC"); add_geo("cc_181x360", s, w)
	s, w = enmap.fullsky_geometry(res=30*utils.degree, variant="fejer1"); add_geo("f1_6x12", s, w)
	s, w = enmap.fullsky_geometry(res=30*utils.degree, variant="cc"); add_geo("cc_7x12", s, w)
	g = enmap.Geometry(s, w)[3:-3, 3:-3]; add_geo("patch_cc", g.shape, g.wcs)
	w3 = g.wcs.deepcopy(); w3.wcs.crpix += 0.123; add_geo("patch_gen_cyl", g.shape, w3)
	s, w = enmap.band_geometry(np.deg2rad(30), res=np.deg2rad(0.5)); add_geo("band_30deg", s, w)
	s, w = enmap.fullsky_geometry(res=np.deg2rad(0.5/60)); geo["fullsky_0.5arcmin_shape"] = [int(v) for v in s]
	# alm_info / spin_helper
	ai = curvedsky.alm_info(lmax=10, mmax=7)
	geo["alm_info_10_7"] = dict(nelem=int(ai.nelem), mstart=[int(v) for v in ai.mstart])
	ai = curvedsky.alm_info(lmax=6, layout="rect")
	geo["alm_info_rect_6"] = dict(nelem=int(ai.nelem), mstart=[int(v) for v in ai.mstart])
	geo["spin_helper_02_3"] = [[int(a), int(b), int(c)] for a, b, c in enmap.spin_helper([0, 2], 3)]
	geo["spin_helper_012_5"] = [[int(a), int(b), int(c)] for a, b, c in enmap.spin_helper([0, 1, 2], 5)]
	geo["spin_helper_1_6"] = [[int(a), int(b), int(c)] for a, b, c in enmap.spin_helper(1, 6)]
	with open(os.path.join(HERE, "geometry.json"), "w") as f: json.dump(geo, f, indent=1, sort_keys=True)

	# ---- 4. FFT fixtures from the reference's numpy engine ------------------------------
	pfft.set_engine("numpy")   # the reference's always-available engine (fft.py:8-31,78-83)
	rng = np.random.default_rng(7)
	fx = {}
	a = rng.standard_normal((3, 12, 20)); fx["r_3x12x20"] = a
	fx["fft_r_3x12x20_axes-2-1"] = pfft.fft(a, axes=[-2, -1])
	fx["fft_r_3x12x20_axes-1"] = pfft.fft(a, axes=[-1])
	fx["rfft_r_3x12x20_axes-1"] = pfft.rfft(a, axes=[-1])
	c = rng.standard_normal((2, 9, 61))+1j*rng.standard_normal((2, 9, 61)); fx["c_2x9x61"] = c
	fx["fft_c_2x9x61_axes-2-1"] = pfft.fft(c, axes=[-2, -1])
	fx["ifft_c_2x9x61_axes-2-1"] = pfft.ifft(c, axes=[-2, -1])
	fx["ifft_c_2x9x61_axes-2-1_norm"] = pfft.ifft(c, axes=[-2, -1], normalize=True)
	h = pfft.rfft(a, axes=[-1]); fx["irfft_of_rfft"] = pfft.irfft(h, n=20, axes=[-1], normalize=True)
	shp, w = enmap.geometry(pos=(0, 0), shape=(3, 100, 100), res=0.01)
	im = enmap.enmap(rng.random(shp), w); fx["enmap_in"] = np.asarray(im)
	fx["enmap_fft"] = np.asarray(enmap.fft(im)); fx["enmap_fft_phys"] = np.asarray(enmap.fft(im, normalize="phys"))
	fx["enmap_ifft_of_fft"] = np.asarray(enmap.ifft(enmap.fft(im)))
	fx["enmap_pixsize"] = np.array(float(im.pixsize()))
	fx["enmap_cdelt"] = np.array(w.wcs.cdelt); fx["enmap_crval"] = np.array(w.wcs.crval); fx["enmap_crpix"] = np.array(w.wcs.crpix)
	np.savez_compressed(os.path.join(HERE, "fft_golden.npz"), **fx)

	# ---- 5. alm helper fixtures (cmisc) -------------------------------------------------
	lmax = 20; ai = curvedsky.alm_info(lmax)
	al = (rng.standard_normal((3, ai.nelem))+1j*rng.standard_normal((3, ai.nelem)))
	cl = ai.alm2cl(al[:, None, :], al[None, :, :])
	fl = rng.random(lmax+1)
	np.savez_compressed(os.path.join(HERE, "alm_ops.npz"), alm=al, cl=cl, fl=fl, almxfl=curvedsky.almxfl(al, fl),
		mstart=ai.mstart, lmax=lmax)
	alm_rand_fixture(curvedsky)
	flatsky_fixture(enmap)
	fftops_fixture(pfft)
	print("fixtures written to", HERE)

if __name__ == "__main__":
	main()
