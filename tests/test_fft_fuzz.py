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

@pytest.mark.hostsim
def test_fft_fuzz_hostsim(): run_fft_fuzz(25, 3, big=False)

@pytest.mark.gpu
def test_fft_fuzz_gpu():
	w = run_fft_fuzz(150, 4)
	print("\n[fft fuzz] 150 random transforms against numpy.fft: worst relative error (f64) %.2e" % w)
