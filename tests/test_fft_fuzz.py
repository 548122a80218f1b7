"""Randomised check of the FFT engine behind pixell_amd.fft (fft / ifft / rfft / irfft: pixell/fft.py:133-209 -- kind from the shapes and dtypes, forward
and backward unnormalised unless normalize=True, any axes, float32 / float64) against numpy.fft: random rank, lengths (smooth, prime, longer than the
LDS), axis subsets.  The 2-D transforms of the hot path (enmap.fft) are pinned elsewhere; this walks the generic N-d engine."""
import numpy as np
import pytest
from pixell_amd import fft

def rand_len(rng, big):
	k = rng.integers(0, 6)
	if k == 0: return int(rng.integers(1, 9))
	if k == 1: return int(rng.choice([7, 11, 13, 31, 61, 97, 127, 131, 251]))            # primes: direct DFT passes, Bluestein above 128
	if k == 2: return int(rng.choice([16, 27, 36, 48, 60, 64, 75, 100, 120, 128]))
	if k == 3 and big: return int(rng.choice([1024, 2187, 2400, 3125, 4096, 4320, 5400, 11200, 16384]))   # (11200, 16384 > the 10240 points an LDS line holds)
	return int(rng.integers(2, 200))

def run_fft_fuzz(ncases, seed, big=True):
	rng = np.random.default_rng(seed)
	worst = 0.0
	for case in range(ncases):
		nd = int(rng.integers(1, 4))
		shape = [rand_len(rng, False) for _ in range(nd)]
		shape[-1 if rng.random() < 0.7 else int(rng.integers(0, nd))] = rand_len(rng, big)
		while np.prod(shape) > 4e6: shape[int(np.argmax(shape))] //= 2
		shape = [max(1, int(n)) for n in shape]
		nax = int(rng.integers(1, nd+1))
		axes = sorted(rng.choice(nd, nax, replace=False).tolist())
		if rng.random() < 0.3: axes = [a-nd for a in axes]
		f64 = rng.random() < 0.75
		rdt, cdt = (np.float64, np.complex128) if f64 else (np.float32, np.complex64)
		tol = 2e-13 if f64 else 2e-5
		kind = str(rng.choice(["fft", "ifft", "ifft_norm", "rfft", "irfft"]))
		what = (case, kind, shape, axes, rdt.__name__)
		nax_t = tuple(a % nd for a in axes)
		if kind in ("fft", "ifft", "ifft_norm"):
			x = (rng.standard_normal(shape)+1j*rng.standard_normal(shape)).astype(cdt)
			if kind == "fft": got = fft.fft(x, axes=axes); ref = np.fft.fftn(x.astype(np.complex128), axes=nax_t)
			else:
				got = fft.ifft(x, axes=axes, normalize=(kind == "ifft_norm"))
				ref = np.fft.ifftn(x.astype(np.complex128), axes=nax_t)
				if kind == "ifft": ref = ref*np.prod([shape[a] for a in nax_t])
		elif kind == "rfft":
			x = rng.standard_normal(shape).astype(rdt)
			got = fft.rfft(x, axes=axes); ref = np.fft.rfftn(x.astype(np.float64), axes=nax_t)
		else:
			n = shape[nax_t[-1]]
			hs = list(shape); hs[nax_t[-1]] = n//2+1
			full = rng.standard_normal(shape)
			spec = np.fft.rfftn(full, axes=nax_t).astype(cdt)                  # a Hermitian-consistent half spectrum
			got = fft.irfft(spec, n=n, axes=axes)
			ref = np.fft.irfftn(spec.astype(np.complex128), s=[shape[a] for a in nax_t], axes=nax_t)*np.prod([shape[a] for a in nax_t])
		got = np.asarray(got)
		assert got.shape == ref.shape, ("shape", what, got.shape, ref.shape)
		d = float(np.max(np.abs(got-ref))/max(np.max(np.abs(ref)), 1e-300))
		if f64: worst = max(worst, d)
		assert d < tol, ("fft engine against numpy", what, d)
	return worst

def run_enmap_fft_fuzz(ncases, seed, nmax):
	"""enmap.fft / ifft (enmap.py:1307-1337): 2-D transform over the last two axes, unitary with normalize=True; real and complex maps, leading
	axes, float32 / float64, shapes with even, odd and prime sides (real maps of 2-3-5-smooth shape take the fused chain stages of the hot path)"""
	from pixell_amd import enmap
	rng = np.random.default_rng(seed)
	worst = 0.0
	for case in range(ncases):
		ny, nx = int(rng.integers(2, nmax)), int(rng.integers(2, nmax))
		if rng.random() < 0.5: ny, nx = int(rng.choice([24, 45, 60, 96, 100, 135, 150, 240])), int(rng.choice([32, 48, 54, 90, 128, 160, 250, 270]))
		pre = tuple(int(v) for v in rng.integers(1, 4, int(rng.integers(0, 3))))
		f64 = rng.random() < 0.75; real = rng.random() < 0.6
		rdt, cdt = (np.float64, np.complex128) if f64 else (np.float32, np.complex64)
		x = rng.standard_normal(pre+(ny, nx))
		if not real: x = x+1j*rng.standard_normal(pre+(ny, nx))
		x = x.astype(rdt if real else cdt)
		_, wcs = enmap.fullsky_geometry(shape=(max(ny, 2), max(nx, 2)))
		normalize = bool(rng.random() < 0.7)
		what = (case, pre, ny, nx, "real" if real else "complex", rdt.__name__, normalize)
		f = enmap.fft(enmap.ndmap(x, wcs), normalize=normalize)
		ref = np.fft.fftn(x.astype(np.complex128), axes=(-2, -1))/(np.sqrt(ny*nx) if normalize else 1.0)
		tol = 2e-13 if f64 else 2e-5
		d = float(np.max(np.abs(np.asarray(f)-ref))/np.max(np.abs(ref))); worst = max(worst, d if f64 else 0.0)
		assert np.asarray(f).shape == ref.shape and d < tol, ("enmap.fft against numpy", what, d)
		b = enmap.ifft(f, normalize=normalize)
		refb = x.astype(np.complex128)*(1.0 if normalize else ny*nx)
		d = float(np.max(np.abs(np.asarray(b)-refb))/np.max(np.abs(refb))); worst = max(worst, d if f64 else 0.0)
		assert d < tol, ("enmap.ifft(enmap.fft(x)) against x", what, d)
	return worst

@pytest.mark.hostsim
def test_fft_fuzz_hostsim(): run_fft_fuzz(25, 3, big=False); run_enmap_fft_fuzz(8, 5, 40)

@pytest.mark.gpu
def test_fft_fuzz_gpu():
	w = run_fft_fuzz(150, 4)
	w2 = run_enmap_fft_fuzz(60, 6, 700)
	print("\n[fft fuzz] 150 random transforms against numpy.fft: worst relative error (f64) %.2e; 60 random enmap.fft / ifft pairs: %.2e" % (w, w2))
