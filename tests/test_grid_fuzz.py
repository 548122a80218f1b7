"""Randomised check of the single-map hot path on named grids (synthesis_2d / analysis_2d as curvedsky.py:907-924, 1032-1046 call ducc0): random grid
name, ring count, ring length, band limit, spin, first-column azimuth.  Pixels of the synthesised map -- the rings next to both poles, the equator, random
rings -- against direct summation on the CPU (oracle/sht_fast.py), and the round trip.  The planner (theta circle and ring FFT splits, radix 7, CC-grid
detour of the synthesis) sees sizes here that no BASELINE configuration has."""
import time
import numpy as np
import pytest
from pixell_amd import sht
from oracle import sht_oracle as so, sht_fast as sf
from test_mm_fuzz import tri, nalm, smooth

def synth_rings_any(alm, spin, lmax, th):
	"""sf.synth_rings for rings that need not come in mirror pairs (MW / MWflip): the CPU port is evaluated on the mirror-symmetric closure of the list
	-- it works from cos(theta), which for a LONE ring next to the south pole carries theta - pi to ~1e-11 only (measured: 1.4e-11 of the map rms
	against 4e-14 for the same ring as the partner of its mirror image), while a mirror pair is evaluated at its northern member."""
	th = np.asarray(th, float)
	t = np.where(th <= np.pi/2, th, np.pi-th)
	north = np.unique(t)
	full = np.concatenate([north, (np.pi-north)[::-1]])
	leg = sf.synth_rings(alm, spin, lmax, full)
	idx = np.searchsorted(north, t)
	sel = np.where(th <= np.pi/2, idx, len(full)-1-idx)
	return leg[:, :, sel]

def run_grid_fuzz(ncases, seed, lmax_hi, npts=10):
	rng = np.random.default_rng(seed)
	worst = dict(pix=0.0, rt=0.0)
	for case in range(ncases):
		geometry = str(rng.choice(["F1", "CC", "MW", "MWflip"]))
		lmax = int(rng.integers(8, lmax_hi))
		over = float(rng.choice([1.0, 1.0, 1.7, 3.0]))      # maps with many more rings than the band limit needs take the CC-grid detour
		lo = int(over*(2*lmax+4))
		if geometry == "F1": nt = smooth(rng, lo, lo+2*lmax+80, even=True)//2
		elif geometry == "CC": nt = smooth(rng, lo, lo+2*lmax+80, even=True)//2+1
		else: nt = (smooth(rng, lo+1, lo+2*lmax+81, odd=True)+1)//2
		nph = smooth(rng, 2*lmax+2, 6*lmax+40, even=True)
		spin = int(rng.choice([0, 0, 2, 2, 1, 3]))
		nc = 1 if spin == 0 else 2
		ne = nalm(lmax, lmax); ms = tri(lmax, lmax)
		l_of = np.concatenate([np.arange(m, lmax+1) for m in range(lmax+1)])
		alm = (rng.standard_normal((nc, ne))+1j*rng.standard_normal((nc, ne)))/(l_of+1.0)
		alm[:, :lmax+1] = alm[:, :lmax+1].real
		alm[:, l_of < spin] = 0
		phi0 = float(rng.uniform(-3, 3))
		kw = dict(spin=spin, lmax=lmax, mstart=ms, geometry=geometry, phi0=phi0)
		what = (case, geometry, lmax, nt, nph, spin)
		m = np.zeros((nc, nt, nph)); sht.synthesis_2d(alm=alm, map=m, **kw)
		theta = so.grid_theta(geometry, nt)
		rows = sf.symmetric_subset(nt, np.concatenate([np.arange(min(3, nt//2)), [nt//2-1, nt//3], rng.integers(0, nt//2, 4)]))
		xs = rng.integers(0, nph, (len(rows), npts))
		leg = synth_rings_any(alm, spin, lmax, theta[rows])
		ref = sf.pixels_on_rings(leg, phi0+2*np.pi*xs/nph)
		got = m[:, rows[:, None], xs]
		for c in range(nc):
			d = float(np.max(np.abs(got[c]-ref[c]))/np.sqrt(np.mean(m[c]**2))); worst["pix"] = max(worst["pix"], d)
			assert d < 6e-12, ("synthesis pixels against direct summation", what, c, d)
		back = np.zeros_like(alm); sht.analysis_2d(alm=back, map=m, **kw)
		d = float(np.sqrt(np.mean(np.abs(back-alm)**2))/np.sqrt(np.mean(np.abs(alm)**2))); worst["rt"] = max(worst["rt"], d)
		assert d < 1e-11, ("round trip", what, d)
		sht.clear_plans()
	return worst

def run_band_fuzz(ncases, seed, lmax_hi):
	"""declination bands as explicit rings (the cyl path, curvedsky.py:843-873, 928-962): rows r0 .. r0 + nr of a Fejer-1 or Clenshaw-Curtis grid of n rings,
	stored plain or flipped in both directions; synthesis, its DERIV1 mode and the adjoint against the oracle"""
	rng = np.random.default_rng(seed)
	worst = 0.0
	for case in range(ncases):
		lmax = int(rng.integers(6, lmax_hi))
		kind = str(rng.choice(["F1", "CC"]))
		n = smooth(rng, 2*lmax+4, 6*lmax+60, even=True)//2+(1 if kind == "CC" else 0)
		r0 = int(rng.integers(0, max(1, n//3))); nr = int(rng.integers(max(2, n//5), n-r0+1))
		nph = smooth(rng, 2*lmax+2, 5*lmax+40, even=True)
		th = (r0+np.arange(nr)+0.5)*np.pi/n if kind == "F1" else (r0+np.arange(nr))*np.pi/(n-1)
		ms = tri(lmax, lmax)
		spin, mode = [(0, "STANDARD"), (2, "STANDARD"), (1, "DERIV1"), (1, "STANDARD"), (3, "STANDARD")][int(rng.integers(0, 5))]
		nca = 1 if (spin == 0 or mode == "DERIV1") else 2
		alm = so.rand_alm_simple(lmax, nca, 100+case, spin=(spin if mode != "DERIV1" else 0,))
		flip = rng.random() < 0.5
		rs = (np.arange(nr)[::-1]*nph+nph-1).astype(np.uint64) if flip else np.arange(nr, dtype=np.uint64)*nph
		kw = dict(theta=th, nphi=np.full(nr, nph, np.uint64), phi0=np.full(nr, float(rng.uniform(-3, 3))), ringstart=rs, lmax=lmax, mstart=ms, pixstride=-1 if flip else 1)
		what = (case, kind, n, r0, nr, nph, lmax, spin, mode, flip)
		ref = so.synthesis(alm=alm, spin=spin, mode=mode, **kw); out = sht.synthesis(alm=alm, spin=spin, mode=mode, **kw)
		d = float(np.max(np.abs(out-ref))/np.max(np.abs(ref))); worst = max(worst, d)
		assert d < 1e-11, ("band synthesis against the oracle", what, d)
		pix = rng.standard_normal(ref.shape)
		ra = so.adjoint_synthesis(map=pix, spin=spin, mode=mode, **kw); oa = sht.adjoint_synthesis(map=pix, spin=spin, mode=mode, **kw)
		ra[:, :lmax+1] = ra[:, :lmax+1].real
		d = float(np.sqrt(np.mean(np.abs(oa-ra)**2))/np.sqrt(np.mean(np.abs(ra)**2))); worst = max(worst, d)
		assert d < 1e-11, ("band adjoint synthesis against the oracle", what, d)
		sht.clear_plans()
	return worst

@pytest.mark.hostsim
def test_band_fuzz_hostsim(): run_band_fuzz(4, 2, lmax_hi=16)

@pytest.mark.gpu
def test_band_fuzz_gpu():
	w = run_band_fuzz(24, 9, lmax_hi=90)
	print("\n[band fuzz] 24 random declination bands: worst error against the oracle %.2e" % w)

@pytest.mark.hostsim
def test_grid_fuzz_hostsim():
	run_grid_fuzz(4, 3, lmax_hi=20, npts=4)

@pytest.mark.gpu
def test_grid_fuzz_gpu():
	t0 = time.time()
	w = run_grid_fuzz(40, 11, lmax_hi=700)
	print("\n[grid fuzz] 40 random named grids in %.0f s: worst pixel error %.2e of the map rms, worst round trip %.2e" % (time.time()-t0, w["pix"], w["rt"]))
