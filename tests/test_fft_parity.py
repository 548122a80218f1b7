"""Parity of the HIP FFT path (pixell_amd.fft / enmap.fft -> pxf_fft_nd) with the reference's numpy
engine (pixell/fft.py:8-31): golden vectors generated from the reference (tests/golden/fft_golden.npz),
the reference's known-answer shape tests (tests/test_pixell.py:373-486) and numpy.fft itself."""
import os
import numpy as np
import pytest
from pixell_amd import fft as pfft, enmap
from pixell_amd.wcs import CarWCS

def rel(a, b): return np.max(np.abs(a-b))/max(np.max(np.abs(b)), 1e-300)
TOL = 1e-13

def check_golden(golden_dir):
	g = np.load(os.path.join(golden_dir, "fft_golden.npz"))
	a = g["r_3x12x20"]; c = g["c_2x9x61"]
	assert rel(pfft.fft(a, axes=[-2, -1]), g["fft_r_3x12x20_axes-2-1"]) < TOL
	assert rel(pfft.fft(a, axes=[-1]), g["fft_r_3x12x20_axes-1"]) < TOL
	assert rel(pfft.rfft(a, axes=[-1]), g["rfft_r_3x12x20_axes-1"]) < TOL
	assert rel(pfft.fft(c, axes=[-2, -1]), g["fft_c_2x9x61_axes-2-1"]) < TOL
	assert rel(pfft.ifft(c, axes=[-2, -1]), g["ifft_c_2x9x61_axes-2-1"]) < TOL
	assert rel(pfft.ifft(c, axes=[-2, -1], normalize=True), g["ifft_c_2x9x61_axes-2-1_norm"]) < TOL
	h = pfft.rfft(a, axes=[-1])
	assert rel(pfft.irfft(h, n=20, axes=[-1], normalize=True), g["irfft_of_rfft"]) < TOL
	# enmap.fft / ifft incl. "phys" normalisation (enmap.py:1307-1337); geometry from the fixture
	im = g["enmap_in"]; cd = g["enmap_cdelt"]
	wcs = CarWCS(cdelt=cd, crval=g["enmap_crval"], crpix=g["enmap_crpix"])
	m = enmap.ndmap(im, wcs)
	assert abs(m.pixsize()/float(g["enmap_pixsize"])-1) < 1e-6
	f = enmap.fft(m)
	assert rel(np.asarray(f), g["enmap_fft"]) < TOL
	assert rel(np.asarray(enmap.ifft(f)).real, im) < TOL            # reference test_fft (tests/test_pixell.py:373-378)
	fp = enmap.fft(m, normalize="phys")
	assert rel(np.asarray(fp)/m.pixsize()**0.5, g["enmap_fft_phys"]/float(g["enmap_pixsize"])**0.5) < TOL
	assert rel(np.asarray(enmap.ifft(fp, normalize="phys")).real, im) < TOL

def check_known_answers():
	"""constant inputs -> DC-only outputs over last / middle / non-contiguous axes, output written
	into a caller-supplied view (reference tests/test_pixell.py:380-486)"""
	signal = np.ones((1, 2, 5, 10))
	out = pfft.fft(signal, axes=[-1])
	exp = np.zeros((1, 2, 5, 10), complex); exp[..., 0] = 10
	assert np.allclose(out, exp)
	out = pfft.fft(signal, axes=[-2, -1]); exp[:] = 0; exp[..., 0, 0] = 50
	assert np.allclose(out, exp)
	out = pfft.fft(signal, axes=[-3, -2, -1]); exp[:] = 0; exp[..., 0, 0, 0] = 100
	assert np.allclose(out, exp)
	# middle axis, non-contiguous input view, output into a view of a bigger array
	big = np.ones((1, 2, 5, 20))[..., ::2]
	out = pfft.fft(big, axes=[-2]); exp = np.zeros((1, 2, 5, 10), complex); exp[..., 0, :] = 5
	assert np.allclose(out, exp)
	obig = np.zeros((1, 2, 5, 20), complex); ov = obig[..., ::2]
	res = pfft.fft(signal, ov, axes=[-2, -1])
	assert np.shares_memory(res, obig) and np.allclose(ov[..., 0, 0], 50) and np.allclose(obig[..., 1::2], 0)
	sig = np.zeros((1, 2, 5, 10), complex); sig[..., 0, 0] = 50
	out = pfft.ifft(sig, axes=[-2, -1], normalize=True)
	assert np.allclose(out, 1)
	out = pfft.ifft(sig, axes=[-2, -1])
	assert np.allclose(out, 50)

def check_lengths(lengths, batch=3):
	rng = np.random.default_rng(5)
	for n in lengths:
		a = rng.standard_normal((batch, n))+1j*rng.standard_normal((batch, n))
		assert rel(pfft.fft(a), np.fft.fft(a, axis=-1)) < 5e-13, n
		assert rel(pfft.ifft(a), np.fft.ifft(a, axis=-1)*n) < 5e-13, n
		r = a.real.copy()
		assert rel(pfft.rfft(r), np.fft.rfft(r, axis=-1)) < 5e-13, n
		assert rel(pfft.irfft(np.fft.rfft(r, axis=-1), n=n, normalize=True), r) < 5e-13, n

@pytest.mark.hostsim
def test_fft_golden_hostsim(golden_dir): check_golden(golden_dir)
@pytest.mark.hostsim
def test_fft_known_answers_hostsim(): check_known_answers()
@pytest.mark.hostsim
def test_fft_lengths_hostsim(): check_lengths([1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 18, 36, 54, 61, 72, 100, 122, 135, 150, 200, 216, 320, 432, 1000, 2048, 2304, 4320, 10800])

@pytest.mark.gpu
def test_fft_golden_gpu(golden_dir): check_golden(golden_dir)
@pytest.mark.gpu
def test_fft_known_answers_gpu(): check_known_answers()
@pytest.mark.gpu
def test_fft_lengths_gpu():
	check_lengths([1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 16, 25, 27, 49, 61, 64, 100, 121, 122, 125, 216, 243, 360, 1000, 1024, 2048,
		2304, 4096, 4320, 8100, 10800, 19200, 20250, 21600, 43200, 64000], batch=5)

@pytest.mark.gpu
def test_fft_2d_large_gpu():
	"""2-D c2c / r2c over the last two axes at map-like sizes incl. four-step columns"""
	rng = np.random.default_rng(6)
	a = rng.standard_normal((2, 2700, 5400))
	assert rel(pfft.fft(a, axes=[-2, -1]), np.fft.fftn(a, axes=(-2, -1))) < 1e-12
	h = pfft.rfft(a, axes=[-2, -1])
	assert rel(h, np.fft.rfftn(a, axes=(-2, -1))) < 1e-12
	assert rel(pfft.irfft(h, n=5400, axes=[-2, -1], normalize=True), a) < 1e-12

def check_fft2_real(shapes, monkeypatch, tol=1e-12):
	"""real map -> complex spectrum over the last two axes through the chain stages (FftChain::fft2_real: two rows per complex line,
	Hermitian half through the column passes) against numpy, forward and backward, float64 and float32 input, and against the
	generic engine (PXS_FFT2_FAST_MINPIX=-1) that it replaces for dense arrays"""
	rng = np.random.default_rng(12)
	for shp in shapes:
		a = rng.standard_normal(shp)
		monkeypatch.setenv("PXS_FFT2_FAST_MINPIX", "0")
		f = pfft.fft(a, axes=[-2, -1]); b = pfft.ifft(a, np.zeros(shp, complex), axes=[-2, -1])
		f32 = pfft.fft(a.astype(np.float32), np.zeros(shp, np.complex128), axes=[-2, -1])
		monkeypatch.setenv("PXS_FFT2_FAST_MINPIX", "-1")
		g = pfft.fft(a, axes=[-2, -1])
		monkeypatch.delenv("PXS_FFT2_FAST_MINPIX")
		ref = np.fft.fftn(a, axes=(-2, -1))
		assert rel(f, ref) < tol and rel(g, ref) < tol and rel(b, np.conj(ref)) < tol, shp
		assert rel(f32, np.fft.fftn(a.astype(np.float32).astype(np.float64), axes=(-2, -1))) < tol, shp
		# complex input (enmap.ifft, fft of complex maps: FftChain::fft2_c2c), forward, backward, in place
		z = a+1j*rng.standard_normal(shp); refz = np.fft.fftn(z, axes=(-2, -1))
		monkeypatch.setenv("PXS_FFT2_FAST_MINPIX", "0")
		fz = pfft.fft(z, axes=[-2, -1]); bz = pfft.ifft(z, axes=[-2, -1], normalize=True)
		iz = z.copy(); pfft.fft(iz, iz, axes=[-2, -1])
		monkeypatch.setenv("PXS_FFT2_FAST_MINPIX", "-1")
		gz = pfft.fft(z, axes=[-2, -1])
		monkeypatch.delenv("PXS_FFT2_FAST_MINPIX")
		assert rel(fz, refz) < tol and rel(gz, refz) < tol and rel(iz, refz) < tol and rel(bz, np.fft.ifftn(z, axes=(-2, -1))) < tol, shp

@pytest.mark.hostsim
def test_fft2_real_hostsim(monkeypatch): check_fft2_real([(2, 24, 40), (32, 25), (3, 15, 27), (1, 20, 18), (2, 45, 64)], monkeypatch)
@pytest.mark.gpu
def test_fft2_real_gpu(monkeypatch): check_fft2_real([(2, 24, 40), (32, 25), (3, 15, 27), (3, 100, 100), (2, 1350, 2700), (1, 2700, 5400), (2, 1000, 1024), (1, 1215, 2025)], monkeypatch)

@pytest.mark.gpu
def test_enmap_fft_config3_shape_gpu():
	"""enmap.fft at the C3 map shape (21600 x 43200) against the DFT sum itself on sampled bins: for 12 columns kx the sum over x
	is a host matrix product map @ e^{-2 pi i kx x / nx}, for 12 rows ky the sum over y of that -- 144 bins of the 9.3e8, none of
	them through an FFT.  (only <= 2700 x 5400 had been compared with an independent transform.)"""
	import torch
	from pixell_amd import enmap
	ny, nx = 21600, 43200
	shape, wcs = enmap.fullsky_geometry(shape=(ny, nx))
	g = torch.Generator(device="cuda"); g.manual_seed(3)
	m = torch.randn((1, ny, nx), generator=g, dtype=torch.float64, device="cuda")
	f = enmap.fft(enmap.dmap(m, wcs), normalize=False).tensor
	kxs = np.array([0, 1, 2, 17, 1000, 10799, 10800, 21599, 21600, 21601, 32400, 43199]); kys = np.array([0, 1, 3, 50, 5399, 5400, 10799, 10800, 10801, 16200, 21598, 21599])
	x = np.arange(nx); y = np.arange(ny)
	ang = 2*np.pi*((kxs[None, :]*x[:, None]) % nx)/nx
	host = m[0].cpu().numpy()
	R = host @ np.cos(ang) - 1j*(host @ np.sin(ang))                  # [ny, nkx]
	angy = 2*np.pi*((kys[:, None]*y[None, :]) % ny)/ny
	ref = (np.cos(angy)-1j*np.sin(angy)) @ R                           # [nky, nkx]
	got = f[0][torch.as_tensor(kys, device="cuda")][:, torch.as_tensor(kxs, device="cuda")].cpu().numpy()
	assert np.max(np.abs(got-ref))/np.sqrt(ny*nx) < 1e-11               # bins of white noise have rms sqrt(npix)
	back = enmap.ifft(enmap.dmap(f, wcs), normalize=False).tensor
	assert float((back.real/(ny*nx)-m).abs().max()) < 1e-11

def check_bluestein():
	"""lengths with a prime factor > 2048 (numpy, the reference's fallback engine, takes any n): chirp-z through two 5-smooth FFTs"""
	from pixell_amd._lib import PxsError
	rng = np.random.default_rng(8)
	for n in (2053, 4099, 2*4099):
		x = rng.standard_normal((3, n))+1j*rng.standard_normal((3, n))
		assert rel(pfft.fft(x), np.fft.fft(x, axis=-1)) < 1e-12
		assert rel(pfft.ifft(x, normalize=True), np.fft.ifft(x, axis=-1)) < 1e-12
		r = rng.standard_normal((2, n))
		assert rel(pfft.rfft(r), np.fft.rfft(r, axis=-1)) < 1e-12
	a = rng.standard_normal((2, 2053, 24))                       # prime length on a strided axis, 2-D with one awkward axis
	assert rel(pfft.fft(a+0j, axes=[-2]), np.fft.fft(a, axis=-2)) < 1e-12
	assert rel(pfft.fft(a+0j, axes=[-2, -1]), np.fft.fft2(a)) < 1e-12
	assert rel(pfft.rfft(a.transpose(0, 2, 1).copy(), axes=[-2, -1]), np.fft.rfft2(a.transpose(0, 2, 1))) < 1e-12
	for n in (2053, 2*4099):                                     # c2r along such an axis: Hermitian extension inside the chirp transform
		r = rng.standard_normal((3, n)); h = np.fft.rfft(r, axis=-1)
		assert rel(pfft.irfft(h, n=n, normalize=True), r) < 1e-12
		assert rel(pfft.irfft(h, n=n), np.fft.irfft(h, n=n, axis=-1)*n) < 1e-12
	r2 = rng.standard_normal((2, 12, 2053)); h2 = np.fft.rfftn(r2, axes=(-2, -1))
	assert rel(pfft.irfft(h2, n=2053, axes=[-2, -1], normalize=True), r2) < 1e-12
	assert pfft.fft_len(4099, "above") >= 4099

@pytest.mark.hostsim
def test_bluestein_hostsim(): check_bluestein()
@pytest.mark.gpu
def test_bluestein_gpu(): check_bluestein()

def check_dct():
	"""DCT-I (FFTW_REDFT00; pixell.fft.dct / idct / redft00 and enmap.fft(dct=True), fft.py:211-307, enmap.py:1314-1342)
	against scipy.fft.dct(type=1), which has the same unnormalised definition; the reference's numpy engine has no DCT"""
	import scipy.fft as sfft
	from pixell_amd import enmap
	rng = np.random.default_rng(5)
	for n in (2, 3, 9, 17, 33, 101, 1025, 1537, 4097):
		x = rng.standard_normal((3, n))
		np.testing.assert_allclose(pfft.dct(x), sfft.dct(x, type=1, axis=-1), rtol=1e-12, atol=1e-11*np.sqrt(n))
	x = rng.standard_normal((2, 33, 49))
	ref = sfft.dctn(x, type=1, axes=(-2, -1))
	np.testing.assert_allclose(pfft.dct(x, axes=[-2, -1]), ref, rtol=1e-12, atol=1e-10)
	np.testing.assert_allclose(pfft.dct(x, axes=[0]), sfft.dct(x, type=1, axis=0), rtol=1e-12, atol=1e-11)
	np.testing.assert_allclose(pfft.idct(ref, axes=[-2, -1], normalize=True), x, rtol=1e-12, atol=1e-12)
	np.testing.assert_allclose(pfft.redft00(x, normalize=True), sfft.dct(x, type=1, axis=-1)/(2*48), rtol=1e-12, atol=1e-13)
	xf = x.astype(np.float32); o = pfft.dct(xf, axes=[-1]); assert o.dtype == np.float32
	np.testing.assert_allclose(o, sfft.dct(xf.astype(np.float64), type=1, axis=-1), rtol=2e-5, atol=2e-4)
	# all eight FFTW r2r kinds against scipy (same unnormalised definitions), 1-D and 2-D, and their inverses
	for name, fun, ty in [("DCT-I", sfft.dct, 1), ("DCT-II", sfft.dct, 2), ("DCT-III", sfft.dct, 3), ("DCT-IV", sfft.dct, 4),
			("DST-I", sfft.dst, 1), ("DST-II", sfft.dst, 2), ("DST-III", sfft.dst, 3), ("DST-IV", sfft.dst, 4)]:
		big = {"DCT-I": 3001, "DST-I": 2999}.get(name, 3000)          # extended length 6000: four-step path
		for n in (2, 5, 16, 31, 100, 1200, big):
			y = rng.standard_normal((2, n))
			np.testing.assert_allclose(pfft.dct(y, type=name), fun(y, type=ty, axis=-1), rtol=1e-11, atol=1e-10*np.sqrt(n), err_msg="%s n=%d" % (name, n))
		y = rng.standard_normal((3, 20, 27))
		f2 = pfft.dct(y, axes=[-2, -1], type=name)
		np.testing.assert_allclose(f2, fun(fun(y, type=ty, axis=-1), type=ty, axis=-2), rtol=1e-11, atol=1e-10, err_msg=name)
		np.testing.assert_allclose(pfft.idct(f2, axes=[-2, -1], type=name, normalize=True), y, rtol=1e-11, atol=1e-11, err_msg=name+" inverse")
	with pytest.raises(ValueError): pfft.dct(x, type="DCT-V")
	shape, wcs = enmap.fullsky_geometry(shape=(33, 64))
	m = enmap.ndmap(rng.standard_normal((2,)+tuple(shape)), wcs)
	d = enmap.dct(m); norm = np.prod(2*np.array(shape)-1)**0.5
	np.testing.assert_allclose(np.asarray(d), sfft.dctn(np.asarray(m), type=1, axes=(-2, -1))/norm, rtol=1e-12, atol=1e-12)
	np.testing.assert_allclose(np.asarray(enmap.idct(d)), sfft.dctn(np.asarray(d), type=1, axes=(-2, -1))/norm, rtol=1e-12, atol=1e-12)

@pytest.mark.hostsim
def test_dct_hostsim(): check_dct()
@pytest.mark.gpu
def test_dct_gpu(): check_dct()

def check_fftops(golden_dir):
	"""fft.shift / resample / resample_fft against the reference's own functions run with its numpy engine (fft_ops.npz)"""
	d = np.load(os.path.join(golden_dir, "fft_ops.npz"))
	a, c = d["a"], d["c"]; tol = dict(rtol=1e-12, atol=1e-12)
	s1 = pfft.shift(a, 2.5); assert s1.dtype == np.float64
	np.testing.assert_allclose(s1, d["shift1"], **tol)
	np.testing.assert_allclose(pfft.shift(a, [1.25, -3.0]), d["shift2"], **tol)
	np.testing.assert_allclose(pfft.shift(c, [0.5], axes=[1]), d["shiftc"], **tol)
	np.testing.assert_allclose(pfft.shift(a, [0.0, 0.0], deriv=1), d["shift_deriv"], **tol)
	np.testing.assert_allclose(pfft.resample(a, 31), d["res_up"], **tol)
	np.testing.assert_allclose(pfft.resample(a, (7, 9)), d["res_dn"], **tol)
	np.testing.assert_allclose(pfft.resample(c, 33, axes=[1]), d["res_c"], **tol)
	fa = np.fft.fft2(c)
	assert np.array_equal(pfft.resample_fft(fa, (16, 11), axes=(-2, -1), norm=0.5), d["rfft_a"])
	assert np.array_equal(pfft.resample_fft(fa, 25, out=np.ones((3, 12, 25), complex), op=lambda x, y: x+y), d["rfft_op"])
	with pytest.raises(ValueError): pfft.resample_fft(fa, 25, out=np.ones((3, 12, 24), complex))

@pytest.mark.hostsim
def test_fftops_hostsim(golden_dir): check_fftops(golden_dir)
@pytest.mark.gpu
def test_fftops_gpu(golden_dir): check_fftops(golden_dir)
