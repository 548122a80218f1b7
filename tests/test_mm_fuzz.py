"""Randomised check of the batched FP64-MFMA Legendre kernels (leg_ana_s0_mm / leg_syn_s0_mm / leg_ana_spin_mm / leg_syn_spin_mm): random named grid,
band limit, mmax, spin, batch size (every remainder class of the 8- and 4-map workgroups) and precision.  Every map of a batched call against its own
single-map call (the VALU kernels), synthesis and analysis of maps that are not band-limited; at small band limits both against the CPU oracle, so that
"the batch equals the single-map call" cannot hide a common error.  Same keyword interface as ducc0's synthesis_2d / analysis_2d
(curvedsky.py:907-924, 1032-1046)."""
import time
import numpy as np
import pytest
from pixell_amd import sht
from oracle import sht_oracle as so

def tri(lmax, mmax):
	m = np.arange(mmax+1, dtype=np.int64); return (m*(2*lmax+1-m)//2).astype(np.uint64)
def nalm(lmax, mmax): return int((mmax+1)*(lmax+1)-mmax*(mmax+1)//2)

def smooth(rng, lo, hi, odd=False, even=False):
	while True:
		n = int(rng.integers(lo, hi)); k = n
		if (odd and n % 2 == 0) or (even and n % 2): continue
		for q in (2, 3, 5, 7):
			while k % q == 0: k //= q
		if k == 1: return n

def run_mm_fuzz(ncases, seed, lmax_hi, oracle_lmax=0, nb_hi=22):
	rng = np.random.default_rng(seed)
	worst = dict(syn=0.0, ana=0.0, syn_oracle=0.0, ana_oracle=0.0)
	for case in range(ncases):
		geometry = str(rng.choice(["F1", "CC", "MW", "MWflip"]))
		lmax = int(rng.integers(12, lmax_hi))
		# ring counts whose theta circle (F1: 2 nt, CC: 2 nt - 2, MW / MWflip: 2 nt - 1) and ring length are 2-3-5-7-smooth, as the grids of real maps are
		if geometry == "F1": nt = smooth(rng, 2*lmax+4, 4*lmax+80, even=True)//2
		elif geometry == "CC": nt = smooth(rng, 2*lmax+4, 4*lmax+80, even=True)//2+1
		else: nt = (smooth(rng, 2*lmax+5, 4*lmax+81, odd=True)+1)//2
		nph = smooth(rng, 2*lmax+2, 4*lmax+40, even=True)
		mmax = lmax if rng.random() < 0.6 else int(rng.integers(1, lmax+1))
		spin = int(rng.choice([0, 0, 2, 2, 1, 3]))
		nb = int(rng.integers(4, nb_hi))
		cdt, rdt = (np.complex128, np.float64) if rng.random() < 0.8 else (np.complex64, np.float32)
		f64 = rdt == np.float64
		nc = 1 if spin == 0 else 2
		ms = tri(lmax, mmax); ne = nalm(lmax, mmax)
		l_of = np.concatenate([np.arange(m, lmax+1) for m in range(mmax+1)])
		alm = (rng.standard_normal((nb, nc, ne))+1j*rng.standard_normal((nb, nc, ne)))/(l_of+1.0)
		alm[:, :, :lmax+1] = alm[:, :, :lmax+1].real
		alm[:, :, l_of < spin] = 0
		alm = alm.astype(cdt)
		kw = dict(spin=spin, lmax=lmax, mmax=mmax, mstart=ms, geometry=geometry, phi0=float(rng.uniform(-3, 3)))
		what = (case, geometry, lmax, mmax, nt, nph, spin, nb, cdt.__name__)
		maps = np.zeros((nb, nc, nt, nph), rdt); sht.synthesis_2d(alm=alm, map=maps, **kw)
		picks = sorted(set([0, nb-1, nb//2, int(rng.integers(0, nb))]))
		for i in picks:
			one = np.zeros((nc, nt, nph), rdt); sht.synthesis_2d(alm=alm[i], map=one, **kw)
			d = float(np.abs(one-maps[i]).max()/np.abs(one).max())
			if f64: worst["syn"] = max(worst["syn"], d)
			assert d < (2e-12 if f64 else 3e-5), ("synthesis: batch against single", what, i, d)
		noisy = (maps+0.05*rng.standard_normal(maps.shape)*np.abs(maps).max()).astype(rdt)      # not band-limited: every l of every m carries something
		back = np.zeros_like(alm); sht.analysis_2d(alm=back, map=noisy, **kw)
		for i in picks:
			one = np.zeros_like(alm[i]); sht.analysis_2d(alm=one, map=noisy[i], **kw)
			d = float(np.abs(one-back[i]).max()/np.sqrt(np.mean(np.abs(one)**2)))
			if f64: worst["ana"] = max(worst["ana"], d)
			# (1e-12 is reached on grids of a few hundred rings: a wave of the single-map kernels spans 256-512 ring pairs there and leaves the rings next to
			# the poles in the plain form of the recurrence (l^2 eps), the 64-pair waves of the batched kernels take the polar form; band-limited input: 2e-14 both)
			assert d < (1e-11 if f64 else 3e-4), ("analysis: batch against single", what, i, d)
		# the adjoints take the same kernels the other way round: adjoint_synthesis_2d = Legendre analysis without the quadrature, adjoint_analysis_2d = synthesis
		if case % 2 == 0:
			at = np.zeros_like(alm); sht.adjoint_synthesis_2d(alm=at, map=noisy, **kw)
			mt = np.zeros_like(maps); sht.adjoint_analysis_2d(alm=alm, map=mt, **kw)
			i = picks[-1]
			one = np.zeros_like(alm[i]); sht.adjoint_synthesis_2d(alm=one, map=noisy[i], **kw)
			# (unweighted sums over the rings: the low-m entries, fed by the polar rings, are ~100 x the rms: measured against the largest entry)
			d = float(np.abs(one-at[i]).max()/np.abs(one).max())
			assert d < (2e-12 if f64 else 3e-5), ("adjoint synthesis: batch against single", what, i, d)
			onem = np.zeros((nc, nt, nph), rdt); sht.adjoint_analysis_2d(alm=alm[i], map=onem, **kw)
			d = float(np.abs(onem-mt[i]).max()/np.abs(onem).max())
			assert d < (2e-12 if f64 else 3e-5), ("adjoint analysis: batch against single", what, i, d)
		if f64 and lmax <= oracle_lmax:
			i = picks[-1]
			ref = np.zeros((nc, nt, nph)); so.synthesis_2d(alm=alm[i].astype(complex), map=ref, **kw)
			d = float(np.abs(ref-maps[i]).max()/np.abs(ref).max()); worst["syn_oracle"] = max(worst["syn_oracle"], d)
			assert d < 1e-11, ("synthesis: batch against the oracle", what, i, d)
			aref = np.zeros((nc, ne), complex); so.analysis_2d(alm=aref, map=maps[i].astype(float), **kw)      # (band-limited input: the route of the analysis does not matter)
			bl = np.zeros_like(alm); sht.analysis_2d(alm=bl, map=maps, **kw)
			d = float(np.sqrt(np.mean(np.abs(bl[i]-aref)**2))/np.sqrt(np.mean(np.abs(aref)**2))); worst["ana_oracle"] = max(worst["ana_oracle"], d)
			assert d < 1e-11, ("analysis: batch against the oracle", what, i, d)
		sht.clear_plans()
	return worst

@pytest.mark.hostsim
def test_mm_fuzz_hostsim():
	run_mm_fuzz(1, 7, lmax_hi=20, oracle_lmax=20, nb_hi=9)

@pytest.mark.gpu
def test_mm_fuzz_gpu():
	t0 = time.time()
	w = run_mm_fuzz(40, 1, lmax_hi=900)
	w2 = run_mm_fuzz(12, 2, lmax_hi=60, oracle_lmax=60)
	print("\n[mm fuzz] 52 cases in %.0f s; worst batch-vs-single difference: synthesis %.2e of the map maximum, analysis %.2e of the alm rms; against the oracle %.2e / %.2e"
		% (time.time()-t0, max(w["syn"], w2["syn"]), max(w["ana"], w2["ana"]), w2["syn_oracle"], w2["ana_oracle"]))

def run_api_fuzz(ncases, seed, lmax_hi):
	"""curvedsky.alm2map / map2alm on maps with leading axes (curvedsky.py:763-765, 1038-1046 loop over them one ducc0 call at a time; here they go down
	as batched library calls): full-sky and band CAR maps [pre..., ncomp, ny, nx], spin = [0, 2]; the whole-array call against the loop over the leading axes"""
	from pixell_amd import curvedsky, enmap
	rng = np.random.default_rng(seed)
	for case in range(ncases):
		lmax = int(rng.integers(8, lmax_hi))
		ny = smooth(rng, 2*lmax+4, 4*lmax+40, even=True)//2; nx = smooth(rng, 2*lmax+2, 4*lmax+40, even=True)
		variant = str(rng.choice(["fejer1", "cc"]))
		if variant == "cc": ny += 1
		shape, wcs = enmap.fullsky_geometry(shape=(ny, nx), variant=variant)
		band = rng.random() < 0.35
		if band:      # a declination band: rows [y0, y1) of the full-sky map (the cyl path)
			y0 = int(rng.integers(0, ny//4)); y1 = int(rng.integers(3*ny//4, ny+1))
			wcs = wcs.deepcopy(); wcs.wcs.crpix[1] -= y0; shape = (y1-y0, nx)        # (as enmap.band_geometry cuts it)
		pre = [(), (2,), (5,), (2, 3), (7,), (9,)][int(rng.integers(0, 6))]
		ncomp = int(rng.choice([1, 3]))
		ainfo = curvedsky.alm_info(lmax)
		nel = ainfo.nelem
		l_of = np.concatenate([np.arange(m, lmax+1) for m in range(lmax+1)])
		alm = (rng.standard_normal(pre+(ncomp, nel))+1j*rng.standard_normal(pre+(ncomp, nel)))/(l_of+1.0)
		alm[..., :lmax+1] = alm[..., :lmax+1].real
		if ncomp == 3: alm[..., 1:, l_of < 2] = 0
		what = (case, variant, "band" if band else "full", tuple(shape), lmax, pre, ncomp)
		m = enmap.ndmap(np.zeros(pre+(ncomp,)+tuple(shape[-2:])), wcs)
		curvedsky.alm2map(alm, m, spin=[0, 2], ainfo=ainfo)
		flat_a = alm.reshape((-1, ncomp, nel)); flat_m = np.asarray(m).reshape((-1, ncomp)+tuple(shape[-2:]))
		for i in sorted(set([0, len(flat_a)-1, len(flat_a)//2])):
			one = enmap.ndmap(np.zeros((ncomp,)+tuple(shape[-2:])), wcs); curvedsky.alm2map(flat_a[i], one, spin=[0, 2], ainfo=ainfo)
			d = float(np.abs(np.asarray(one)-flat_m[i]).max()/np.abs(np.asarray(one)).max())
			assert d < 2e-12, ("alm2map: whole array against the loop", what, i, d)
		back = curvedsky.map2alm(m, lmax=lmax, spin=[0, 2], ainfo=ainfo)
		flat_b = np.asarray(back).reshape((-1, ncomp, nel))
		for i in sorted(set([0, len(flat_a)-1])):
			one = curvedsky.map2alm(enmap.ndmap(flat_m[i].copy(), wcs), lmax=lmax, spin=[0, 2], ainfo=ainfo)
			d = float(np.abs(np.asarray(one)-flat_b[i]).max()/np.abs(np.asarray(one)).max())
			assert d < 2e-11, ("map2alm: whole array against the loop", what, i, d)
		if not band:
			d = float(np.sqrt(np.mean(np.abs(np.asarray(back)-alm)**2))/np.sqrt(np.mean(np.abs(alm)**2)))
			assert d < 1e-11, ("round trip", what, d)
		curvedsky.sht.clear_plans()

@pytest.mark.hostsim
def test_api_fuzz_hostsim(): run_api_fuzz(3, 4, 14)

@pytest.mark.gpu
def test_api_fuzz_gpu(): run_api_fuzz(24, 8, 260)
