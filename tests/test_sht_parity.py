"""Parity of the HIP SHT path with the CPU oracle, through the C ABI (pixell_amd.sht ->
libpxsht.so).  Bodies are shared: small cases run in the GPU-less container on the test-only
host simulator (marker `hostsim`), the same and larger cases run on the MI355X (marker `gpu`).

Tolerances (float64): max |HIP - oracle| <= 1e-11 * max|oracle| for maps, and
rms(alm_hip - alm_ref)/rms(alm_ref) <= 1e-11 for alm (north_star asks for < 1e-8 rms).
The oracle is 80-bit with three-term recurrences; the HIP path is float64 with the Ishioka
recurrence, so agreement at this level is a numerical statement, not an identity."""
import numpy as np
import pytest
from pixell_amd import sht
from oracle import sht_oracle as so

TOL = 1e-11

def rel(a, b): return np.max(np.abs(a-b))/max(np.max(np.abs(b)), 1e-300)
def relrms(a, b): return np.sqrt(np.mean(np.abs(a-b)**2))/max(np.sqrt(np.mean(np.abs(b)**2)), 1e-300)

def oracle_form(geometry, nt, nph, lmax, analysis=None, **plan_kw):
	"""keywords that make the oracle's analysis_2d restate the form the product's analysis_2d takes on this grid (the default is ducc0's
	route; the product reports the form and the N_cc it realised, sht.analysis_form, and the oracle is asked for exactly that)"""
	f = sht.analysis_form(geometry, nt, nph, lmax, analysis=analysis, **plan_kw)
	return {"ducc0": dict(fine_cc=f["ncc_circle"]), "weights": dict(weights=True), "interpolant": dict(fine_cc=False)}[f["form"]]

def check_grid(geometry, nt, nph, lmax, spin, phi0=0.3, seed=3, random_map=True, mmax=None):
	nc = 1 if spin == 0 else 2
	mmax = lmax if mmax is None else mmax
	alm = so.rand_alm_simple(lmax, nc, seed, spin=(spin,))
	ms = so._tri_mstart(lmax, lmax)
	kw = dict(spin=spin, lmax=lmax, mmax=mmax, mstart=ms[:mmax+1], geometry=geometry, phi0=phi0)
	ref = np.zeros((nc, nt, nph)); so.synthesis_2d(alm=alm, map=ref, **kw)
	out = np.zeros((nc, nt, nph)); sht.synthesis_2d(alm=alm, map=out, **kw)
	assert rel(out, ref) < TOL, "synthesis_2d"
	rng = np.random.default_rng(seed)
	pix = rng.standard_normal((nc, nt, nph))
	ra = np.zeros_like(alm); so.adjoint_synthesis_2d(alm=ra, map=pix, **kw)
	oa = np.zeros_like(alm); sht.adjoint_synthesis_2d(alm=oa, map=pix, **kw)
	ra[:, :lmax+1] = ra[:, :lmax+1].real
	assert relrms(oa, ra) < TOL, "adjoint_synthesis_2d"
	if lmax <= so.grid_maxlmax(geometry, nt):
		oa = np.zeros_like(alm); sht.analysis_2d(alm=oa, map=ref, **kw)
		sel = np.zeros(alm.shape[1], bool)
		for m in range(mmax+1): sel[int(ms[m])+m:int(ms[m])+lmax+1] = True
		if mmax == lmax and 2*mmax < nph: assert relrms(oa[:, sel], alm[:, sel]) < TOL, "round trip"   # (no exact inverse when m aliases)
		ref2 = np.zeros((nc, nt, nph)); out2 = np.zeros((nc, nt, nph))
		from pixell_amd import _lib
		for form in (None, "interpolant") if (nt <= 24 or not _lib.is_hostsim()) else (None,):      # the default (ducc0's route) and the full-interpolant option (simulator: small grids only, for time)
			okw = oracle_form(geometry, nt, nph, lmax, analysis=form, mmax=mmax, mstart=ms[:mmax+1], phi0=phi0)
			if nt <= 64:
				so.adjoint_analysis_2d(alm=alm, map=ref2, **kw, **okw); sht.adjoint_analysis_2d(alm=alm, map=out2, analysis=form, **kw)
				assert rel(out2, ref2) < TOL, "adjoint_analysis_2d (%s)" % form
			if random_map:
				ra = np.zeros_like(alm); so.analysis_2d(alm=ra, map=pix, **kw, **okw)
				oa = np.zeros_like(alm); sht.analysis_2d(alm=oa, map=pix, analysis=form, **kw)
				assert relrms(oa, ra) < TOL, "analysis_2d on a non-band-limited map (%s)" % form

SMALL = [("F1", 20, 41, 19, 0), ("F1", 20, 41, 19, 2), ("F1", 32, 61, 30, 1), ("CC", 21, 40, 19, 0), ("CC", 21, 48, 19, 2),
	("MW", 16, 33, 15, 0), ("MWflip", 16, 33, 15, 2), ("F1", 24, 64, 12, 0), ("F1", 24, 64, 12, 3),
	("CC", 20, 12, 18, 0), ("F1", 20, 38, 19, 2),   # these two alias m onto the rings (mmax >= nphi/2), as the golden CC 181x360 lmax=400 case does
	("F1", 96, 200, 30, 0), ("F1", 96, 200, 30, 2), ("F1", 120, 240, 50, 1),   # many more rings than lmax: synthesis AND its adjoint go through the CC grid (from_cc / from_cc_adjoint)
	("CC", 97, 200, 30, 2), ("MW", 113, 200, 30, 1)]   # ... on grids with self-mirrored rings (counted double in the mirror extension of the transposed upsampling); CC: analysis by the grid's own weights

@pytest.mark.hostsim
@pytest.mark.parametrize("geometry,nt,nph,lmax,spin", SMALL)
def test_grid_small_hostsim(geometry, nt, nph, lmax, spin):
	check_grid(geometry, nt, nph, lmax, spin)

@pytest.mark.hostsim
@pytest.mark.parametrize("geometry,nt,nph,lmax,spin", [("F1", 32, 64, 30, 0), ("CC", 41, 80, 30, 2), ("MW", 36, 72, 34, 1), ("F1", 140, 280, 64, 2)])
def test_grid_default_k_hostsim(monkeypatch, geometry, nt, nph, lmax, spin):
	"""Ring sets of up to 512 pairs run the smallest compiled K (k_small_grid, legendre.hip); the host simulation only sees such grids, so this case switches
	the rule off (a switch of the simulation build alone) to walk the kernels of the large configurations -- leg_syn_s0<4>, leg_ana_s0<8>, leg_syn_spin<3>, leg_ana_spin<4>."""
	monkeypatch.setenv("PXS_K_SMALL_OFF", "1")
	sht.clear_plans()
	check_grid(geometry, nt, nph, lmax, spin)
	sht.clear_plans()

@pytest.mark.hostsim
@pytest.mark.parametrize("spin", [0, 2])
def test_deterministic_mode_hostsim(monkeypatch, spin):
	"""pxs_plan_option("deterministic", 1) (sht.set_deterministic): per-wave partial moments + ordered sum instead of atomic adds into the
	moments (legendre.hip, leg_analysis)"""
	sht.set_deterministic(True)
	try: check_grid("F1", 20, 41, 19, spin)
	finally: sht.set_deterministic(None)

@pytest.mark.gpu
def test_deterministic_mode_gpu(monkeypatch):
	"""several waves per m: the ordered scheme against the oracle (the default one is what every other test runs), the two
	schemes against each other at a size with 3-6 waves per m, and the ordered one repeats bit for bit"""
	try:
		sht.set_deterministic(True)      # the plan option, not an environment variable: a drop-in backend is configured through its API
		check_grid("CC", 514, 1200, 512, 2, random_map=False)
		nt, nph, lmax = 1500, 3000, 1400
		rng = np.random.default_rng(5); ms = so._tri_mstart(lmax, lmax)
		for spin, nc in ((0, 1), (2, 2)):
			pix = rng.standard_normal((nc, nt, nph))
			kw = dict(spin=spin, lmax=lmax, mstart=ms, geometry="F1", phi0=0.1)
			res = []
			for det in (True, True, False, False):      # (the same cached plan, switched between the calls)
				sht.set_deterministic(det)
				oa = np.zeros((nc, int(ms[-1])+lmax+1), complex); plan = sht.analysis_2d(alm=oa, map=pix, return_plan=True, **kw); res.append(oa)
			assert np.array_equal(res[0], res[1])
			assert relrms(res[2], res[0]) < 1e-14 and relrms(res[3], res[2]) < 1e-14
			assert plan.info()["scratch_bytes"] < 60e9
	finally: sht.set_deterministic(None)

def check_device_tables(lmax, nt, nph, monkeypatch):
	"""recurrence tables built on the GPU in double-double (LegTables::build) against the host long-double builder (build_host,
	PXS_TABLES_HOST=1; the rows agree to 2e-16 relative, tools/tables_probe.hip): transforms through either agree to the transform's own error level"""
	ms = so._tri_mstart(lmax, lmax); rng = np.random.default_rng(3)
	for spin, nc in ((0, 1), (2, 2), (1, 2)):
		alm = so.rand_alm_simple(lmax, nc, 8, spin=(spin,)); pix = rng.standard_normal((nc, nt, nph))
		kw = dict(spin=spin, lmax=lmax, mstart=ms, geometry="F1", phi0=0.1)
		res = []
		for host in ("1", "0"):
			monkeypatch.setenv("PXS_TABLES_HOST", host); sht.clear_plans()
			m = np.zeros((nc, nt, nph)); sht.synthesis_2d(alm=alm, map=m, **kw)
			a = np.zeros_like(alm); sht.adjoint_synthesis_2d(alm=a, map=pix, **kw)
			res.append((m, a))
		sht.clear_plans()
		# (the rows agree to an ulp; an ulp in a coefficient moves a recurrence of lmax steps by ~lmax ulp: the algorithm's own error level)
		tol = 2e-14 if lmax < 100 else 2e-12
		assert np.abs(res[0][0]-res[1][0]).max() < tol*np.abs(res[0][0]).max()
		assert np.abs(res[0][1]-res[1][1]).max() < tol*np.abs(res[0][1]).max()

@pytest.mark.hostsim
def test_device_tables_hostsim(monkeypatch): check_device_tables(40, 42, 84, monkeypatch)
@pytest.mark.gpu
def test_device_tables_gpu(monkeypatch): check_device_tables(1500, 1600, 3200, monkeypatch); check_device_tables(4000, 4100, 8200, monkeypatch)

@pytest.mark.hostsim
def test_scaled_recurrence_hostsim():
	"""large enough that sin^m(theta) needs the extended exponent near the poles (spin 0 and 2)"""
	check_grid("F1", 100, 200, 96, 0, random_map=False)      # (spin 2 with scaling: test_deep_scaling)

@pytest.mark.hostsim
def test_mmax_lt_lmax_hostsim():
	check_grid("F1", 24, 40, 16, 0, mmax=9)
	check_grid("F1", 24, 40, 16, 2, mmax=9)

@pytest.mark.gpu
@pytest.mark.parametrize("geometry,nt,nph,lmax,spin", SMALL+[("F1", 260, 512, 250, 0), ("CC", 300, 520, 255, 2),
	("F1", 513, 1040, 512, 0), ("F1", 400, 1024, 399, 1), ("CC", 514, 1200, 512, 2), ("F1", 700, 1400, 300, 2), ("F1", 640, 1280, 256, 0)])
def test_grid_gpu(geometry, nt, nph, lmax, spin):
	check_grid(geometry, nt, nph, lmax, spin, random_map=(nt <= 300))

def check_adjoint_analysis_fused(geometry, nt, nph, lmax, monkeypatch, nb=1):
	"""adjoint_analysis_2d through the fused transposed chain (FftChain::to_cc_adjoint, all maps of a call per launch) against the
	stage-by-stage transpose it replaces (PXS_ADJ_ANA_FUSED=0: dense rows, generic FFT engine, one map at a time), which the
	small grids pin to the oracle"""
	ms = so._tri_mstart(lmax, lmax)
	for spin in (0, 2):
		nc = 1 if spin == 0 else 2
		alm = np.stack([so.rand_alm_simple(lmax, nc, 50+i, spin=(spin,)) for i in range(nb)])
		kw = dict(spin=spin, lmax=lmax, mstart=ms, geometry=geometry, phi0=0.25, analysis="interpolant")      # (the only form with an unfused transpose)
		monkeypatch.setenv("PXS_ADJ_ANA_FUSED", "1")
		a = np.zeros((nb, nc, nt, nph)); sht.adjoint_analysis_2d(alm=alm, map=a, **kw)
		monkeypatch.setenv("PXS_ADJ_ANA_FUSED", "0")
		b = np.zeros((nb, nc, nt, nph)); sht.adjoint_analysis_2d(alm=alm, map=b, **kw)
		monkeypatch.delenv("PXS_ADJ_ANA_FUSED")
		assert rel(a, b) < TOL, (geometry, nt, spin)

@pytest.mark.hostsim
def test_adjoint_analysis_fused_hostsim(monkeypatch):
	check_adjoint_analysis_fused("F1", 36, 72, 35, monkeypatch); check_adjoint_analysis_fused("CC", 41, 80, 30, monkeypatch, nb=2)
@pytest.mark.gpu
def test_adjoint_analysis_fused_gpu(monkeypatch):
	for g, nt, nph, lmax, nb in [("F1", 36, 72, 35, 1), ("CC", 41, 80, 30, 2), ("F1", 130, 300, 128, 3), ("CC", 514, 1200, 512, 1), ("F1", 1350, 2700, 1000, 2), ("F1", 640, 1280, 639, 1)]:
		check_adjoint_analysis_fused(g, nt, nph, lmax, monkeypatch, nb=nb)

@pytest.mark.gpu
def test_config1_gpu():
	"""BASELINE config 1: 1x(1024x2048) F1 map, lmax=512, map2alm -> alm2map against the CPU oracle"""
	nt, nph, lmax = 1024, 2048, 512
	alm = so.rand_alm_simple(lmax, 1, 1, spin=(0,)); ms = so._tri_mstart(lmax, lmax)
	kw = dict(spin=0, lmax=lmax, mstart=ms, geometry="F1", phi0=-3.1400587)
	ref = np.zeros((1, nt, nph)); so.synthesis_2d(alm=alm, map=ref, **kw)
	out = np.zeros((1, nt, nph)); sht.synthesis_2d(alm=alm, map=out, **kw)
	assert rel(out, ref) < TOL
	oa = np.zeros_like(alm); sht.analysis_2d(alm=oa, map=out, **kw)
	assert relrms(oa, alm) < TOL

def check_flips_f32(geometry="F1", nt=32, nph=61, lmax=30, spin=2):
	alm = so.rand_alm_simple(lmax, 2, 3, spin=(spin,)); ms = so._tri_mstart(lmax, lmax)
	kw = dict(spin=spin, lmax=lmax, mstart=ms, geometry=geometry, phi0=0.1)
	ref = np.zeros((2, nt, nph)); so.synthesis_2d(alm=alm, map=ref, **kw)
	for flip in [(True, True), (True, False), (False, True)]:
		out = np.zeros((2, nt, nph)); sht.synthesis_2d(alm=alm, map=out, flip=flip, **kw)
		r2 = np.ascontiguousarray(ref[:, ::-1 if flip[0] else 1, ::-1 if flip[1] else 1])
		assert rel(out, r2) < TOL
		oa = np.zeros_like(alm); sht.analysis_2d(alm=oa, map=r2, flip=flip, **kw)
		assert relrms(oa, alm) < TOL
		pa = np.zeros_like(alm); sht.adjoint_synthesis_2d(alm=pa, map=r2, flip=flip, **kw)
		pb = np.zeros_like(alm); sht.adjoint_synthesis_2d(alm=pb, map=ref, **kw)
		assert relrms(pa, pb) < TOL
	a32 = alm.astype(np.complex64); o32 = np.zeros((2, nt, nph), np.float32)
	sht.synthesis_2d(alm=a32, map=o32, **kw)
	assert rel(o32, ref) < 1e-6
	oa32 = np.zeros_like(a32); sht.analysis_2d(alm=oa32, map=o32, **kw)
	assert relrms(oa32, alm) < 1e-6

@pytest.mark.hostsim
def test_flips_and_f32_hostsim(): check_flips_f32("F1", 16, 31, 15, 2)
@pytest.mark.gpu
def test_flips_and_f32_gpu(): check_flips_f32(); check_flips_f32("CC", 130, 300, 128, 1)

def check_rings():
	"""explicit rings incl. unpaired rings, unsorted order, DERIV1 (curvedsky.py:936-960)"""
	rng = np.random.default_rng(2)
	th = np.array([0.3, 0.9, 1.2, np.pi/2, np.pi-0.9, 2.9, 2.0]); nr = len(th); nph = 24; lmax = 10
	kw = dict(theta=th, nphi=np.full(nr, nph, np.uint64), phi0=np.full(nr, 0.2), ringstart=np.arange(nr, dtype=np.uint64)*nph,
		lmax=lmax, mstart=so._tri_mstart(lmax, lmax))
	for spin, mode in [(0, "STANDARD"), (2, "STANDARD"), (1, "STANDARD"), (1, "DERIV1")]:
		nca = 1 if (spin == 0 or mode == "DERIV1") else 2
		alm = so.rand_alm_simple(lmax, nca, 5, spin=(spin if mode != "DERIV1" else 0,))
		ref = so.synthesis(alm=alm, spin=spin, mode=mode, **kw); out = sht.synthesis(alm=alm, spin=spin, mode=mode, **kw)
		assert rel(out, ref) < TOL
		pix = rng.standard_normal(ref.shape)
		ra = so.adjoint_synthesis(map=pix, spin=spin, mode=mode, **kw); oa = sht.adjoint_synthesis(map=pix, spin=spin, mode=mode, **kw)
		ra[:, :lmax+1] = ra[:, :lmax+1].real
		assert relrms(oa, ra) < TOL

def healpix_rings(nside):
	"""the standard healpix ring layout (RING order): theta, nphi, phi0, ringstart"""
	i = np.arange(1, 4*nside); north = np.minimum(i, 4*nside-i)             # ring index counted from the nearer pole
	cap = north < nside
	nphi = np.where(cap, 4*north, 4*nside).astype(np.uint64)
	z = np.where(cap, 1-north**2/(3.0*nside**2), (4*nside-2.0*north)/(3*nside))*np.where(i <= 2*nside, 1, -1)
	phi0 = np.where(cap, np.pi/(4*north), np.where((north-nside) % 2 == 0, np.pi/(4*nside), 0.0))
	ringstart = np.concatenate([[0], np.cumsum(nphi)[:-1]]).astype(np.uint64)
	return np.arccos(z), nphi, phi0, ringstart

def check_general_rings(nside=4, lmax=14, big=False):
	"""ring sets with per-ring nphi / phi0 / ringstart (ducc synthesis / adjoint_synthesis as called for healpix maps and profile
	rings, curvedsky.py:328-349, 396-403, 537, 553): healpix; ragged rings incl. nphi = 1, a prime and heavy aliasing (nphi < mmax),
	gaps between rings, unsorted offsets, pixstride 2"""
	rng = np.random.default_rng(4)
	cases = [healpix_rings(nside)+(1,)]
	th = np.array([0.2, 0.7, 1.1, np.pi/2, np.pi-0.7, 2.6, 1.9, 0.05]); nph = np.array([5, 1, 13, 32, 7, 2, 24, 3], np.uint64)
	p0 = rng.uniform(-3, 3, len(th)); order = rng.permutation(len(th))
	rs = np.zeros(len(th), np.uint64); off = 3
	for r in order: rs[r] = off; off += 2*int(nph[r])+5            # pixstride 2, gaps, rings stored out of order
	if not big: cases.append((th, nph, p0, rs, 2))
	for th, nph, p0, rs, pstr in cases:
		kw = dict(theta=th, nphi=nph, phi0=p0, ringstart=rs, lmax=lmax, mstart=so._tri_mstart(lmax, lmax), pixstride=pstr)
		for spin, mode in [(0, "STANDARD"), (2, "STANDARD")]+([] if big else [(1, "DERIV1")]):
			nca = 1 if (spin == 0 or mode == "DERIV1") else 2
			alm = so.rand_alm_simple(lmax, nca, 5, spin=(spin if mode != "DERIV1" else 0,))
			ref = so.synthesis(alm=alm, spin=spin, mode=mode, **kw); out = sht.synthesis(alm=alm, spin=spin, mode=mode, **kw)
			used = np.zeros(ref.shape[-1], bool)
			for r in range(len(th)): used[int(rs[r])+pstr*np.arange(int(nph[r]))] = True
			assert rel(out[:, used], ref[:, used]) < TOL and np.all(out[:, ~used] == 0)
			pix = rng.standard_normal(ref.shape)
			ra = so.adjoint_synthesis(map=pix, spin=spin, mode=mode, **kw); oa = sht.adjoint_synthesis(map=pix, spin=spin, mode=mode, **kw)
			ra[:, :lmax+1] = ra[:, :lmax+1].real
			assert relrms(oa, ra) < TOL
			o32 = sht.synthesis(alm=alm.astype(np.complex64), spin=spin, mode=mode, **kw)
			assert o32.dtype == np.float32 and rel(o32[:, used], ref[:, used]) < 1e-5

def check_band_rings(n=96, r0=9, nr=70, nph=200, lmax=24):
	"""rows r0 .. r0+nr of an n-ring Fejer-1 grid as explicit rings (a declination band on the cyl path, curvedsky.py:843-873, 928-962):
	pxs_plan_rings recognises them and runs synthesis and its adjoint through the CC grid of the full grid; stored flipped in
	both directions (descending ringstart, pixstride -1) as pixell_amd.curvedsky passes band maps"""
	th = (r0+np.arange(nr)+0.5)*np.pi/n
	ms = so._tri_mstart(lmax, lmax); rng = np.random.default_rng(8)
	rs_plain = np.arange(nr, dtype=np.uint64)*nph
	rs_flip = (np.arange(nr)[::-1]*nph+nph-1).astype(np.uint64)
	for spin, mode in [(0, "STANDARD"), (2, "STANDARD"), (1, "DERIV1")]:
		nca = 1 if (spin == 0 or mode == "DERIV1") else 2
		alm = so.rand_alm_simple(lmax, nca, 6, spin=(spin if mode != "DERIV1" else 0,))
		for rs, pstr in ((rs_plain, 1), (rs_flip, -1)):
			kw = dict(theta=th, nphi=np.full(nr, nph, np.uint64), phi0=np.full(nr, -0.4), ringstart=rs, lmax=lmax, mstart=ms, pixstride=pstr)
			ref = so.synthesis(alm=alm, spin=spin, mode=mode, **kw); out = sht.synthesis(alm=alm, spin=spin, mode=mode, **kw)
			info = sht.synthesis.last_plan.info()
			assert info["nring_syn"] < nr, "the band plan did not take the CC grid"
			assert rel(out, ref) < TOL
			pix = rng.standard_normal(ref.shape)
			ra = so.adjoint_synthesis(map=pix, spin=spin, mode=mode, **kw); oa = sht.adjoint_synthesis(map=pix, spin=spin, mode=mode, **kw)
			ra[:, :lmax+1] = ra[:, :lmax+1].real
			assert relrms(oa, ra) < TOL

@pytest.mark.hostsim
def test_band_rings_hostsim(): check_band_rings()
@pytest.mark.gpu
def test_band_rings_gpu(): check_band_rings(); check_band_rings(n=1200, r0=150, nr=700, nph=2400, lmax=400)

def check_prime_rings():
	"""equal rings whose length has a prime factor > 2048 (no mixed-radix factorisation): general path + Bluestein"""
	th = np.array([0.4, 1.3, np.pi-1.3, 2.5]); nph = 2*2053; lmax = 12
	kw = dict(theta=th, nphi=np.full(4, nph, np.uint64), phi0=np.full(4, 0.1), ringstart=np.arange(4, dtype=np.uint64)*nph, lmax=lmax, mstart=so._tri_mstart(lmax, lmax))
	alm = so.rand_alm_simple(lmax, 2, 5, spin=(2,))
	ref = so.synthesis(alm=alm, spin=2, **kw); out = sht.synthesis(alm=alm, spin=2, **kw)
	assert rel(out, ref) < TOL
	ra = so.adjoint_synthesis(map=ref, spin=2, **kw); oa = sht.adjoint_synthesis(map=ref, spin=2, **kw)
	ra[:, :lmax+1] = ra[:, :lmax+1].real
	assert relrms(oa, ra) < TOL

@pytest.mark.hostsim
def test_prime_rings_hostsim(): check_prime_rings()
@pytest.mark.gpu
def test_prime_rings_gpu(): check_prime_rings()

@pytest.mark.hostsim
def test_general_rings_hostsim(): check_general_rings()
@pytest.mark.gpu
def test_general_rings_gpu(): check_general_rings(); check_general_rings(nside=16, lmax=40, big=True)

@pytest.mark.hostsim
def test_general_path_equals_uniform_hostsim(monkeypatch):
	"""PXS_GENERAL_RINGS=1 sends equal rings through the general path: same numbers as the paired ring FFTs"""
	monkeypatch.setenv("PXS_GENERAL_RINGS", "1"); sht.clear_plans()
	try: check_rings()
	finally: sht.clear_plans()

def check_seeds(lmax, nt, nph, reps=3):
	"""recurrence seeds (legendre.hip): the first transform of a kind on a plan records the state of the recurrences where their
	accumulation starts, later ones load it -- results must not change by a bit, and must equal those of a plan without seeds"""
	import os
	ms = so._tri_mstart(lmax, lmax); rng = np.random.default_rng(9)
	for spin, nc in ((0, 1), (2, 2)):
		alm = so.rand_alm_simple(lmax, nc, 3, spin=(spin,)); pix = rng.standard_normal((nc, nt, nph))
		kw = dict(spin=spin, lmax=lmax, mstart=ms, geometry="F1", phi0=0.2)
		def run():
			out = []
			for rep in range(reps):
				m = np.zeros((nc, nt, nph)); sht.synthesis_2d(alm=alm, map=m, **kw)
				a = np.zeros_like(alm); sht.analysis_2d(alm=a, map=pix, **kw)
				b = np.zeros_like(alm); sht.adjoint_synthesis_2d(alm=b, map=pix, **kw)
				out.append((m, a, b))
			return out
		sht.clear_plans(); res = run()
		for r in res[1:]:
			for x, y in zip(res[0], r): assert np.array_equal(x, y)
		os.environ["PXS_SEED_GB"] = "0"; sht.clear_plans()
		try: ref = run()
		finally: del os.environ["PXS_SEED_GB"]; sht.clear_plans()
		for x, y in zip(res[0], ref[0]): assert np.array_equal(x, y)

@pytest.mark.hostsim
def test_seeds_hostsim(monkeypatch):
	monkeypatch.setattr(sht, "_deterministic", True); monkeypatch.setenv("PXS_SEED_MIN_LMAX", "0")
	check_seeds(28, 30, 60, reps=2)
@pytest.mark.gpu
def test_seeds_gpu(monkeypatch):
	monkeypatch.setattr(sht, "_deterministic", True)           # (bitwise comparison of the analysis needs the ordered accumulation)
	monkeypatch.setenv("PXS_SEED_MIN_LMAX", "0")
	check_seeds(1500, 1600, 3100)

def check_deep_scaling(lmax=260):
	"""rings so close to the poles (1e-5 rad: sin^100 = 2^-1660) that sin^m(theta) needs two and three 2^-800 scale steps, next to equatorial rings in
	the same wave: lanes reach scale 0 at very different l (the ungated phase-B steps, data fetch / sum reset on arrival)"""
	rng = np.random.default_rng(4)
	th = np.array([1e-5, 3e-4, 0.004, 0.02, 0.3, 1.1, np.pi/2, np.pi-3e-4, np.pi-0.3, np.pi-1.1, 2.5]); nr = len(th); nph = 8
	kw = dict(theta=th, nphi=np.full(nr, nph, np.uint64), phi0=np.full(nr, 0.1), ringstart=np.arange(nr, dtype=np.uint64)*nph,
		lmax=lmax, mstart=so._tri_mstart(lmax, lmax))
	for spin in (0, 2):
		alm = so.rand_alm_simple(lmax, 1 if spin == 0 else 2, 6, spin=(spin,))
		ref = so.synthesis(alm=alm, spin=spin, **kw); out = sht.synthesis(alm=alm, spin=spin, **kw)
		assert rel(out, ref) < TOL
		pix = rng.standard_normal(ref.shape)
		ra = so.adjoint_synthesis(map=pix, spin=spin, **kw); oa = sht.adjoint_synthesis(map=pix, spin=spin, **kw)
		ra[:, :lmax+1] = ra[:, :lmax+1].real
		assert relrms(oa, ra) < TOL

def check_large_subset(lmax, use_port=False):
	"""full ring set of a CC-like grid at large lmax (several waves per m, every lane passing through the scaled phases
	at a different l) against the oracle (use_port: its C port, pinned to it by test_oracle_port.py, for the lmax the
	long-double Python oracle cannot reach in minutes) evaluated on a subset of the rings: synthesis ring by ring, adjoint synthesis
	with the map supported on the subset.  (A missing data fetch for lanes that reach scale 0 during phase A once gave
	4e-2 errors at lmax 4000 that no smaller case showed.)"""
	nr = lmax+2; nph = 8
	th = np.arange(nr)*np.pi/(nr-1); th[0] = 1e-4; th[-1] = np.pi-1e-4
	sub = np.unique(np.concatenate([np.arange(0, 8), np.arange(8, nr//2, max(1, nr//40)), nr-1-np.arange(0, 8), [nr//2]]))
	if use_port: sub = np.unique(np.concatenate([sub, nr-1-sub]))       # the port pairs rings north/south
	ms = so._tri_mstart(lmax, lmax)
	def ref_syn(alm, spin, t):
		if not use_port: return so.synthesis(alm=alm, spin=spin, **kw(t)).reshape(alm.shape[0], len(t), nph)
		from oracle import sht_fast as sf
		leg = sf.synth_rings(alm, spin, lmax, t)                          # [nm, nc, nsub]
		x = np.arange(nph)
		return sf.pixels_on_rings(leg, np.tile(0.1+2*np.pi*x/nph, (len(t), 1)))
	def ref_adj(pix, spin, t):
		if not use_port: return so.adjoint_synthesis(map=pix.reshape(pix.shape[0], -1), spin=spin, **kw(t))
		from oracle import sht_port
		L = np.fft.fft(pix, axis=2)[:, :, np.arange(lmax+1) % nph]*np.exp(-1j*np.arange(lmax+1)*0.1)[None, None, :]    # sum_x ring e^{-i m phi_x}
		cols = sht_port.leg(spin, lmax, np.arange(lmax+1), t, leg=np.transpose(L, (2, 0, 1)))
		out = np.zeros((pix.shape[0], so.nalm(lmax)), complex)
		for m in range(lmax+1): out[:, int(ms[m])+m:int(ms[m])+lmax+1] = cols[m, :, m:]
		return out
	def kw(t): return dict(theta=t, nphi=np.full(len(t), nph, np.uint64), phi0=np.full(len(t), 0.1), ringstart=np.arange(len(t), dtype=np.uint64)*nph, lmax=lmax, mstart=ms)
	rng = np.random.default_rng(1)
	for spin in (0, 2):
		nc = 1 if spin == 0 else 2
		alm = so.rand_alm_simple(lmax, nc, 6, spin=(spin,))
		out = sht.synthesis(alm=alm, spin=spin, **kw(th)).reshape(nc, nr, nph)
		ref = ref_syn(alm, spin, th[sub])
		assert rel(out[:, sub], ref) < TOL
		pix = np.zeros((nc, nr, nph)); pix[:, sub] = rng.standard_normal((nc, len(sub), nph))
		oa = sht.adjoint_synthesis(map=pix.reshape(nc, -1), spin=spin, **kw(th))
		ra = ref_adj(pix[:, sub], spin, th[sub])
		ra[:, :lmax+1] = ra[:, :lmax+1].real
		assert relrms(oa, ra) < TOL
		assert np.max(np.abs(oa-ra)) < 1e-8*np.sqrt(np.mean(np.abs(ra)**2))

@pytest.mark.gpu
def test_large_lmax_subset_gpu(): check_large_subset(2600)
@pytest.mark.gpu
@pytest.mark.parametrize("lmax", [4000, 10000])
def test_large_lmax_subset_port_gpu(lmax):
	"""the BASELINE band limits (a missing data fetch once gave 4e-2 errors at lmax 4000 that no smaller case showed)"""
	check_large_subset(lmax, use_port=True)
@pytest.mark.hostsim
def test_large_subset_logic_hostsim():
	check_large_subset(48); check_large_subset(48, use_port=True)

@pytest.mark.hostsim
def test_deep_scaling_hostsim(): check_deep_scaling(100)
@pytest.mark.gpu
def test_deep_scaling_gpu(): check_deep_scaling(700)

@pytest.mark.hostsim
def test_rings_hostsim(): check_rings()
@pytest.mark.gpu
def test_rings_gpu(): check_rings()

def check_gridweights_large(cases=(("F1", 2100), ("CC", 2161), ("MW", 2050))):
	"""grids beyond 2048 rings take the DFT route of pxs_gridweights: same numbers as the direct series of the oracle, sum 4 pi"""
	for geo, n in cases:
		w = sht.get_gridweights(geo, n); ref = so.get_gridweights(geo, n)
		assert abs(w.sum()-4*np.pi) < 1e-12 and np.max(np.abs(w-ref)) < 1e-15 and np.max(np.abs(w-ref)/ref) < 1e-9

@pytest.mark.hostsim
def test_gridweights_large_hostsim(): check_gridweights_large()
@pytest.mark.gpu
def test_gridweights_large_gpu(): check_gridweights_large(); check_gridweights_large((("F1", 5400), ("MWflip", 4000)))

def check_gridweights():
	for g in ["CC", "F1", "MW", "MWflip", "DH", "F2"]:
		for n in [7, 12, 33, 100]:
			assert np.max(np.abs(sht.get_gridweights(g, n)-so.get_gridweights(g, n))) < 1e-13
def test_gridweights(): check_gridweights()

def test_errors():
	"""error behaviour at the boundary: too few rings for analysis, aliasing mmax, unknown geometry"""
	from pixell_amd._lib import PxsError
	lmax = 20; ms = so._tri_mstart(lmax, lmax); n = so.nalm(lmax)
	with pytest.raises(PxsError):
		sht.analysis_2d(alm=np.zeros((1, n), complex), map=np.zeros((1, 10, 64)), spin=0, lmax=lmax, mstart=ms, geometry="F1")
	with pytest.raises(PxsError):
		sht.synthesis_2d(alm=np.zeros((1, n), complex), map=np.zeros((1, 30, 64)), spin=0, lmax=lmax, mstart=ms, geometry="GL")
	with pytest.raises(ValueError):
		sht.synthesis_2d(alm=np.zeros((1, n), complex), map=np.zeros((1, 30, 64)), spin=2, lmax=lmax, mstart=ms, geometry="F1")

def check_batched(nb=3, geometry="F1", nt=24, nph=48, lmax=20):
	"""nbatch > 1 at the C ABI (pxs_synthesis / pxs_analysis with batch strides): alm [nb, nc, nelem], map [nb, nc, nt, nph] in one
	call equals nb single calls (bit for bit) and the oracle; strided batches (one spin group out of T/Q/U stacks) included"""
	ms = so._tri_mstart(lmax, lmax)
	for spin in (0, 2):
		nc = 1 if spin == 0 else 2
		alm = np.stack([so.rand_alm_simple(lmax, nc, 20+i, spin=(spin,)) for i in range(nb)])
		kw = dict(spin=spin, lmax=lmax, mstart=ms, geometry=geometry, phi0=0.2)
		out = np.zeros((nb, nc, nt, nph)); sht.synthesis_2d(alm=alm, map=out, **kw)
		mm = nb >= 4      # the FP64-MFMA Legendre kernels of 4 or more maps per call sum in another order than the single-map kernels: rounding, not bits
		for i in range(nb):
			one = np.zeros((nc, nt, nph)); sht.synthesis_2d(alm=alm[i], map=one, **kw)
			if mm: assert np.abs(one-out[i]).max() < 1e-13*np.abs(one).max()
			else: assert np.array_equal(one, out[i])
			ref = np.zeros((nc, nt, nph)); so.synthesis_2d(alm=alm[i], map=ref, **kw)
			assert rel(out[i], ref) < TOL
		back = np.zeros_like(alm); sht.analysis_2d(alm=back, map=out, **kw)
		assert relrms(back, alm) < TOL
		adj = np.zeros_like(alm); sht.adjoint_synthesis_2d(alm=adj, map=out, **kw)
		one = np.zeros_like(alm[0]); sht.adjoint_synthesis_2d(alm=one, map=out[1], **kw)
		# spin 0, 4 or more maps: the Legendre analysis of the batch is the FP64-MFMA kernel (leg_ana_s0_mm), whose sum over the rings
		# runs in another order than the single-map kernel's -- equal to rounding, not bit for bit
		if mm: assert relrms(adj[1], one) < 1e-13 and np.abs(adj[1]-one).max() < 1e-12*np.abs(one).max()
		else: assert np.array_equal(one, adj[1])
		aa = np.zeros((nb, nc, nt, nph)); sht.adjoint_analysis_2d(alm=alm, map=aa, **kw)
		one = np.zeros((nc, nt, nph)); sht.adjoint_analysis_2d(alm=alm[nb-1], map=one, **kw)
		if mm: assert np.abs(one-aa[nb-1]).max() < 1e-13*np.abs(one).max()
		else: assert np.array_equal(one, aa[nb-1])

@pytest.mark.hostsim
def test_batched_hostsim(): check_batched()
# 2 x 26 rings = 4 x 13: the ring FFTs (64 pixels) are chained, the theta resampling is not -- a batched synthesis through the CC grid
# has to fall back to one map per pass there (round 2 raised "batched call on an unfused path"); the analysis takes ducc0's route
# through the generic FFT engine (N_cc = 28, radix 7 there too)
@pytest.mark.hostsim
def test_batched_unfused_theta_hostsim():
	assert sht.analysis_form("F1", 26, 64, 12, phi0=0.2) == dict(form="ducc0", ncc_circle=28, ducc_ncc_circle=28)
	check_batched(2, "F1", 26, 64, 12)
@pytest.mark.gpu
def test_batched_gpu():
	check_batched(); check_batched(5, "CC", 130, 300, 128); check_batched(2, "F1", 28, 64, 12); check_batched(3, "F1", 154, 320, 60)
	# device-resident strided batch: the Q/U group of a stack of T/Q/U maps, in place
	import torch
	from pixell_amd import curvedsky, enmap
	lmax = 64; shape, wcs = enmap.fullsky_geometry(shape=(70, 140))
	alm = torch.from_numpy(np.stack([so.rand_alm_simple(lmax, 3, 40+i, spin=(0, 2)) for i in range(4)])).cuda()
	m = enmap.dmap(torch.zeros((4, 3)+tuple(shape), dtype=torch.float64, device="cuda"), wcs)
	curvedsky.alm2map(alm, m, spin=[0, 2])
	for i in range(4):
		one = enmap.dmap(torch.zeros((3,)+tuple(shape), dtype=torch.float64, device="cuda"), wcs)
		curvedsky.alm2map(alm[i], one, spin=[0, 2])
		assert torch.equal(one.tensor, m.tensor[i])
	back = torch.zeros_like(alm); curvedsky.map2alm(m, alm=back, spin=[0, 2])
	assert float((back-alm).abs().max()) < 1e-11


def check_dh_f2(cases=(("DH", 22), ("F2", 21), ("DH", 41), ("F2", 40))):
	"""Driscoll-Healy and Fejer-2 grids (get_ducc_geo can return them, curvedsky.py:1329-1342): synthesis on their rings, analysis
	by Fejer's second rule, exact up to get_ducc_maxlmax = (n-2)//2 resp. (n-1)//2; adjoints; error beyond the limit"""
	from pixell_amd._lib import PxsError
	for g, nt in cases:
		lmax = so.grid_maxlmax(g, nt); assert sht.grid_maxlmax(g, nt) == lmax
		for spin in (0, 2):
			check_grid(g, nt, 2*lmax+3, lmax, spin)
		with pytest.raises(PxsError):
			sht.analysis_2d(alm=np.zeros((1, so.nalm(lmax+1)), complex), map=np.zeros((1, nt, 64)), spin=0, lmax=lmax+1, mstart=so._tri_mstart(lmax+1, lmax+1), geometry=g)

@pytest.mark.hostsim
def test_dh_f2_hostsim(): check_dh_f2((("DH", 22), ("F2", 21)))
@pytest.mark.gpu
def test_dh_f2_gpu(): check_dh_f2()

def check_weights_analysis(geometry, nt, nph, lmax, spin, nb=1, seed=5):
	"""analysis="weights" (pxs_plan_option: ring quadrature weights + adjoint synthesis, the reference's cyl route, curvedsky.py:852-861,
	1068-1084) against the oracle's restatement of the same composition -- on a band-limited map (where it must also equal the
	interpolant form and recover the alm) and on white noise (where the two forms differ) -- and its adjoint."""
	rng = np.random.default_rng(seed)
	nc = 1 if spin == 0 else 2
	assert nt >= 2*lmax+2
	ms = so._tri_mstart(lmax, lmax)
	kw = dict(spin=spin, lmax=lmax, geometry=geometry, phi0=0.2, mstart=ms)
	alm = so.rand_alm_simple(lmax, nc, seed, spin=(spin,))
	band = np.zeros((nc, nt, nph)); so.synthesis_2d(alm=alm, map=band, **kw)
	noise = rng.standard_normal((nc, nt, nph))
	for m, tag in ((band, "band-limited"), (noise, "noise")):
		ref = np.zeros_like(alm); so.analysis_2d(alm=ref, map=m, weights=True, **kw)
		got = np.zeros_like(alm); sht.analysis_2d(alm=got, map=m, analysis="weights", **kw)
		assert relrms(got, ref) < TOL, "weights analysis, %s map: %.3e" % (tag, relrms(got, ref))
		if tag == "band-limited":
			assert relrms(got, alm) < TOL
			itp = np.zeros_like(alm); sht.analysis_2d(alm=itp, map=m, **kw)
			assert relrms(got, itp) < TOL
		else:
			itp = np.zeros_like(alm); sht.analysis_2d(alm=itp, map=m, analysis="interpolant", **kw)
			assert relrms(got, itp) > 1e-6, "the two forms should differ on a map that is not band-limited"
	ref2 = np.zeros((nc, nt, nph)); so.adjoint_analysis_2d(alm=alm, map=ref2, weights=True, **kw)
	out2 = np.zeros((nc, nt, nph)); sht.adjoint_analysis_2d(alm=alm, map=out2, analysis="weights", **kw)
	assert rel(out2, ref2) < TOL, "adjoint of the weights analysis: %.3e" % rel(out2, ref2)
	if nb > 1:	# batched call == single calls
		maps = np.stack([noise*(i+1) for i in range(nb)]); a = np.zeros((nb,)+alm.shape, alm.dtype)
		sht.analysis_2d(alm=a, map=maps, analysis="weights", **kw)
		one = np.zeros_like(alm); sht.analysis_2d(alm=one, map=maps[nb-1], analysis="weights", **kw)
		assert rel(a[nb-1], one) < 1e-13
	# the option is per call: the default form (ducc0's route) is back
	chk = np.zeros_like(alm); sht.analysis_2d(alm=chk, map=noise, **kw)
	dflt = np.zeros_like(alm); so.analysis_2d(alm=dflt, map=noise, **kw, **oracle_form(geometry, nt, nph, lmax, phi0=0.2, mstart=ms))
	assert relrms(chk, dflt) < TOL
	with pytest.raises(ValueError): sht.analysis_2d(alm=chk, map=noise, analysis="quadrature", **kw)

WEIGHTS_CASES = [("F1", 64, 128, 30, 0), ("F1", 66, 128, 31, 2), ("CC", 65, 120, 30, 2), ("MW", 64, 100, 28, 0), ("F1", 120, 240, 50, 1)]
@pytest.mark.hostsim
@pytest.mark.parametrize("geometry,nt,nph,lmax,spin", WEIGHTS_CASES[:3])
def test_weights_analysis_hostsim(geometry, nt, nph, lmax, spin): check_weights_analysis(geometry, nt, nph, lmax, spin, nb=2 if spin == 0 else 1)
@pytest.mark.gpu
@pytest.mark.parametrize("geometry,nt,nph,lmax,spin", WEIGHTS_CASES)
def test_weights_analysis_gpu(geometry, nt, nph, lmax, spin): check_weights_analysis(geometry, nt, nph, lmax, spin, nb=3)

def check_ducc0_route(geometry, nt, nph, lmax, spin, expect_ducc_size=None, seed=7):
	"""The default analysis: ducc0's route as published (analysis_2d -> resample_to_prepared_CC), restated by the oracle as the direct
	sum over the CC grid of N_cc + 1 rings of (quadrature weight) x (theta-interpolant, low-passed to |k| < N_cc on finer grids) x
	lambda_lm.  White noise and a band-limited map, the adjoint, and -- where the plan realises ducc0's own N_cc =
	2 good_size_complex(lmax + 1), radix 7 included -- the oracle at ducc0's sizes (fine_cc=True)."""
	nc = 1 if spin == 0 else 2
	ms = so._tri_mstart(lmax, lmax)
	kw = dict(spin=spin, lmax=lmax, geometry=geometry, phi0=0.2, mstart=ms)
	f = sht.analysis_form(geometry, nt, nph, lmax, phi0=0.2, mstart=ms)
	assert f["form"] == "ducc0" and f["ducc_ncc_circle"] == 2*so.good_size_complex(lmax+1)
	if expect_ducc_size is not None: assert (f["ncc_circle"] == f["ducc_ncc_circle"]) == expect_ducc_size, f
	alm = so.rand_alm_simple(lmax, nc, seed, spin=(spin,))
	band = np.zeros((nc, nt, nph)); so.synthesis_2d(alm=alm, map=band, **kw)
	noise = np.random.default_rng(seed).standard_normal((nc, nt, nph))
	for m, tag in ((band, "band-limited"), (noise, "noise")):
		ref = np.zeros_like(alm); so.analysis_2d(alm=ref, map=m, fine_cc=f["ncc_circle"], **kw)
		got = np.zeros_like(alm); sht.analysis_2d(alm=got, map=m, **kw)
		assert relrms(got, ref) < TOL, "%s: %.3e" % (tag, relrms(got, ref))
		if tag == "band-limited": assert relrms(got, alm) < TOL
		if f["ncc_circle"] == f["ducc_ncc_circle"]:
			ref = np.zeros_like(alm); so.analysis_2d(alm=ref, map=m, fine_cc=True, **kw)
			assert relrms(got, ref) < TOL
	ref2 = np.zeros((nc, nt, nph)); so.adjoint_analysis_2d(alm=alm, map=ref2, fine_cc=f["ncc_circle"], **kw)
	out2 = np.zeros((nc, nt, nph)); sht.adjoint_analysis_2d(alm=alm, map=out2, **kw)
	assert rel(out2, ref2) < TOL

# (grid, rings, nphi, lmax, spin, ducc0's N_cc realised?)  N > 2 N_cc: low pass; N < 2 N_cc: zero padding; N = 2 N_cc; radix 7 in N_cc (lmax 20, 27)
DUCC0_CASES = [("F1", 120, 240, 30, 0, True), ("F1", 96, 192, 20, 2, True), ("F1", 40, 80, 30, 1, True), ("F1", 64, 128, 30, 2, True),
	("CC", 49, 100, 27, 0, True), ("MW", 63, 100, 20, 0, False), ("MWflip", 63, 100, 40, 1, None)]
@pytest.mark.hostsim
@pytest.mark.parametrize("geometry,nt,nph,lmax,spin,ds", DUCC0_CASES[:5])
def test_ducc0_route_hostsim(geometry, nt, nph, lmax, spin, ds): check_ducc0_route(geometry, nt, nph, lmax, spin, ds)
@pytest.mark.gpu
@pytest.mark.parametrize("geometry,nt,nph,lmax,spin,ds", DUCC0_CASES+[("F1", 300, 600, 127, 2, True), ("F1", 540, 1080, 250, 0, True), ("F1", 400, 1000, 399, 1, True), ("CC", 301, 800, 200, 2, None)])
def test_ducc0_route_gpu(geometry, nt, nph, lmax, spin, ds): check_ducc0_route(geometry, nt, nph, lmax, spin, ds)

def check_ducc0_route_cc_weights(geometry="CC", nt=65, nph=120, lmax=30, spin=2):
	"""a CC grid with at least 2 lmax + 2 rings: ducc0 multiplies it by its own weights (need_first_resample = false), i.e. the weights
	form; here through the transposed theta upsampling with the pole rings counted double in the mirror extension"""
	nc = 1 if spin == 0 else 2
	ms = so._tri_mstart(lmax, lmax); kw = dict(spin=spin, lmax=lmax, geometry=geometry, phi0=0.1, mstart=ms)
	assert sht.analysis_form(geometry, nt, nph, lmax, phi0=0.1, mstart=ms)["form"] == "weights"
	noise = np.random.default_rng(2).standard_normal((nc, nt, nph))
	ref = np.zeros((nc, so.nalm(lmax)), complex); so.analysis_2d(alm=ref, map=noise, weights=True, **kw)
	got = np.zeros_like(ref); sht.analysis_2d(alm=got, map=noise, **kw)
	assert relrms(got, ref) < TOL
@pytest.mark.hostsim
def test_ducc0_route_cc_weights_hostsim(): check_ducc0_route_cc_weights()
@pytest.mark.gpu
def test_ducc0_route_cc_weights_gpu(): check_ducc0_route_cc_weights(); check_ducc0_route_cc_weights("CC", 601, 1200, 250, 0)

def test_weights_analysis_small_grid_takes_the_default():
	"""below ntheta = 2 lmax + 2 ring weights are not exact: the option leaves such grids on the default form"""
	lmax, nt, nph = 30, 40, 80
	alm = so.rand_alm_simple(lmax, 1, 3, spin=(0,))
	kw = dict(spin=0, lmax=lmax, geometry="F1", phi0=0.0, mstart=so._tri_mstart(lmax, lmax))
	m = np.zeros((1, nt, nph)); so.synthesis_2d(alm=alm, map=m, **kw)
	got = np.zeros_like(alm); sht.analysis_2d(alm=got, map=m, analysis="weights", **kw)
	assert relrms(got, alm) < TOL
	assert sht.analysis_form("F1", nt, nph, lmax, analysis="weights")["form"] == sht.analysis_form("F1", nt, nph, lmax)["form"]

@pytest.mark.hostsim
def test_batched_shared_recurrence_hostsim():
	"""five scalar maps in one call: the analysis through the shared-recurrence FP64-MFMA kernel (leg_ana_s0_mm: one 8-map workgroup shape
	with three columns groups masked), the synthesis map by map"""
	check_batched(nb=5, nt=20, nph=40, lmax=16)

def check_mm_analysis(nbs=(4, 9, 12, 13), lmax=40, grid=None, spins=(0, 2)):
	"""Batched Legendre analysis and synthesis as FP64-MFMA GEMMs (leg_ana_s0_mm / leg_syn_s0_mm: one recurrence per ring pair shared by the maps, 8
	scalar maps per workgroup, a remainder of <= 4 in 4-map workgroups, a lone left-over map through the VALU kernel; leg_ana_spin_mm / leg_syn_spin_mm:
	the Q/U pairs of 4 maps per workgroup, two recurrences).  The batch equals its single-map calls (the VALU kernels, pinned to the oracle
	elsewhere) to rounding -- NOT bit for bit: the sums run in another order -- and the oracle.  Rings from 1e-5 rad of the poles to the equator:
	lanes of one wave reach scale 0 at very different l (masked P rows in phase B).  grid=(geometry, nt, nph): analysis_2d of a full grid
	(several ring chunks)."""
	rng = np.random.default_rng(11)
	for spin in spins:
		nc = 1 if spin == 0 else 2
		if grid is None:
			th = np.array([1e-5, 3e-4, 0.004, 0.02, 0.3, 1.1, np.pi/2, np.pi-3e-4, np.pi-0.3, np.pi-1.1, 2.5]); nr = len(th); nph = 8
			kw = dict(theta=th, nphi=np.full(nr, nph, np.uint64), phi0=np.full(nr, 0.1), ringstart=np.arange(nr, dtype=np.uint64)*nph,
				lmax=lmax, mstart=so._tri_mstart(lmax, lmax), spin=spin)
			for nb in nbs:
				pix = rng.standard_normal((nb, nc, nr*nph))
				out = sht.adjoint_synthesis(map=pix, alm=np.zeros((nb, nc, so.nalm(lmax)), complex), **kw)
				for i in sorted(set([0, 3, nb-1, nb//2])):
					one = sht.adjoint_synthesis(map=pix[i], **kw)
					assert relrms(out[i], one) < 1e-13 and np.abs(out[i]-one).max() < 1e-12*np.abs(one).max()
				ref = so.adjoint_synthesis(map=pix[nb-1], **kw); ref[:, :lmax+1] = ref[:, :lmax+1].real
				assert relrms(out[nb-1], ref) < TOL
				# the synthesis of the batch (accumulators for 64 ring pairs x the maps of a group in registers, steps = K dimension)
				alm = np.stack([so.rand_alm_simple(lmax, nc, 50+i, spin=(spin,)) for i in range(nb)])
				maps = sht.synthesis(alm=alm, map=np.zeros((nb, nc, nr*nph)), **kw)
				for i in sorted(set([0, 3, nb-1, nb//2])):
					one = sht.synthesis(alm=alm[i], **kw)
					assert np.abs(maps[i]-one).max() < 1e-12*np.abs(one).max()
				assert rel(maps[nb-1], so.synthesis(alm=alm[nb-1], **kw)) < TOL
		else:
			geometry, nt, nph = grid
			ms = so._tri_mstart(lmax, lmax); kw = dict(spin=spin, lmax=lmax, mstart=ms, geometry=geometry, phi0=0.3)
			for nb in nbs:
				alm = np.stack([so.rand_alm_simple(lmax, nc, 70+i, spin=(spin,)) for i in range(nb)])
				maps = np.zeros((nb, nc, nt, nph)); sht.synthesis_2d(alm=alm, map=maps, **kw)
				one = np.zeros((nc, nt, nph)); sht.synthesis_2d(alm=alm[nb-1], map=one, **kw)
				assert np.abs(maps[nb-1]-one).max() < 1e-12*np.abs(one).max()
				maps += 0.1*rng.standard_normal(maps.shape)            # not band-limited: every l of every m carries something
				back = np.zeros_like(alm); sht.analysis_2d(alm=back, map=maps, **kw)
				for i in sorted(set([0, nb-1, nb//2])):
					one = np.zeros_like(alm[i]); sht.analysis_2d(alm=one, map=maps[i], **kw)
					assert relrms(back[i], one) < 1e-13 and np.abs(back[i]-one).max() < 1e-11*np.sqrt(np.mean(np.abs(one)**2))

def check_mm_layouts(nb=5, nt=26, nph=48, lmax=20, mmax=13):
	"""the batched FP64-MFMA kernels behind the other alm layouts of the boundary: mmax < lmax, single precision (complex64 alm, float32 maps),
	an alm array with room to spare between the m blocks (mstart with gaps)"""
	ms = so._tri_mstart(lmax, mmax)+3*np.arange(mmax+1, dtype=np.uint64)      # three unused slots after every m
	nel = int(ms[-1])+lmax+1+3
	rng = np.random.default_rng(21)
	alm = np.zeros((nb, 1, nel), complex)
	for b in range(nb):
		for m in range(mmax+1):
			v = rng.standard_normal(lmax+1-m)+1j*rng.standard_normal(lmax+1-m)
			if m == 0: v = v.real+0j
			alm[b, 0, int(ms[m])+m:int(ms[m])+lmax+1] = v
	kw = dict(spin=0, lmax=lmax, mmax=mmax, mstart=ms, geometry="F1", phi0=0.1)
	maps = np.zeros((nb, 1, nt, nph)); sht.synthesis_2d(alm=alm, map=maps, **kw)
	for i in (0, nb-1):
		one = np.zeros((1, nt, nph)); sht.synthesis_2d(alm=alm[i], map=one, **kw)
		assert np.abs(one-maps[i]).max() < 1e-13*np.abs(one).max()
		ref = np.zeros((1, nt, nph)); so.synthesis_2d(alm=alm[i], map=ref, **kw)
		assert rel(maps[i], ref) < TOL
	back = np.zeros_like(alm); sht.analysis_2d(alm=back, map=maps, **kw)
	assert relrms(back, alm) < TOL
	a32 = alm.astype(np.complex64); m32 = np.zeros((nb, 1, nt, nph), np.float32); sht.synthesis_2d(alm=a32, map=m32, **kw)
	assert np.abs(m32-maps).max() < 2e-6*np.abs(maps).max()
	b32 = np.zeros_like(a32); sht.analysis_2d(alm=b32, map=m32, **kw)
	assert relrms(b32, alm) < 5e-6

@pytest.mark.hostsim
def test_mm_layouts_hostsim(): check_mm_layouts()
@pytest.mark.gpu
def test_mm_layouts_gpu(): check_mm_layouts(); check_mm_layouts(nb=9, nt=700, nph=1400, lmax=600, mmax=411)

@pytest.mark.hostsim
def test_mm_analysis_hostsim():
	# (the host simulation of the MFMA kernels is slow: the GPU test runs the full matrix of batch sizes and spins)
	check_mm_analysis(nbs=(4, 9), lmax=32, spins=(0,)); check_mm_analysis(nbs=(13,), lmax=32, spins=(2,)); check_mm_analysis(nbs=(5,), lmax=24, spins=(3,))
@pytest.mark.gpu
def test_mm_analysis_gpu():
	check_mm_analysis(lmax=700); check_mm_analysis(nbs=(5, 16), lmax=1500, grid=("CC", 1502, 3008)); check_mm_analysis(nbs=(8,), lmax=2100, grid=("F1", 2800, 5600))

@pytest.mark.hostsim
def test_batched_deterministic_and_seeded_hostsim(monkeypatch):
	"""batched calls on the two paths that launch maps one at a time: the ordered (bitwise repeatable) analysis (sht.set_deterministic), and
	the launch that records the recurrence seeds (the first map alone), forced on at toy size with PXS_SEED_MIN_LMAX=0"""
	small = dict(nb=2, nt=18, nph=36, lmax=14)
	monkeypatch.setattr(sht, "_deterministic", True); sht.clear_plans(); check_batched(**small)
	monkeypatch.setattr(sht, "_deterministic", None); monkeypatch.setenv("PXS_SEED_MIN_LMAX", "0"); sht.clear_plans(); check_batched(**small)
	monkeypatch.setattr(sht, "_deterministic", True); sht.clear_plans(); check_batched(**small)
	sht.clear_plans()

def check_baseline_analysis_forms():
	"""what map2alm integrates at the BASELINE configurations: ducc0's route, at ducc0's own N_cc = 2 good_size_complex(lmax + 1) wherever
	the grid's circle shares a usable modulus with it (C2 ... C5; C1's 2048-point circle does not: the planner's 1080 against 1050)"""
	for name, (nt, nph, lmax), realised in [("C1", (1024, 2048, 512), False), ("C2/C4", (5400, 10800, 4000), True), ("C3", (21600, 43200, 10000), True), ("C5", (10800, 21600, 6000), True)]:
		f = sht.analysis_form("F1", nt, nph, lmax)
		assert f["form"] == "ducc0" and f["ducc_ncc_circle"] == 2*so.good_size_complex(lmax+1), (name, f)
		assert (f["ncc_circle"] == f["ducc_ncc_circle"]) == realised, (name, f)
		assert sht.analysis_form("F1", nt, nph, lmax, analysis="interpolant")["form"] == "interpolant"
		assert sht.analysis_form("F1", nt, nph, lmax, analysis="weights")["form"] == ("weights" if nt >= 2*lmax+2 else "ducc0")
	sht.clear_plans()
@pytest.mark.hostsim
def test_baseline_analysis_forms_hostsim(): check_baseline_analysis_forms()
@pytest.mark.gpu
def test_baseline_analysis_forms_gpu(): check_baseline_analysis_forms()
