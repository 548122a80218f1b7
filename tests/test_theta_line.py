"""The single-kernel theta resampling engine (pixell_amd/csrc/thetaline.hip) against the five-stage chain it replaces
(FftChain::to_cc / from_cc_adjoint): same plan, same input, PXS_THETA_LINE switched per call.  The chain is pinned to the oracle by
tests/test_sht_parity.py; here the two device paths must agree to rounding.  In the GPU-less container the host simulator runs
the configuration compiled for it (360 rings, lmax 250: every feature of the C2 / C4 configuration at a fifteenth of its size)."""
import numpy as np, pytest
from pixell_amd import sht, _lib

def relrms(a, b): return float(np.sqrt(np.mean(np.abs(a - b)**2)/np.mean(np.abs(b)**2)))

def run_pair(nt, nph, lmax, spin, nb, monkeypatch, seed=3, mmax=None):
	nc = 1 if spin == 0 else 2
	if mmax is None: mmax = lmax
	ms = sht.tri_mstart(lmax, mmax); nalm = int(ms[-1]) + lmax + 1
	kw = dict(spin=spin, lmax=lmax, mmax=mmax, geometry="F1", phi0=0.1, mstart=ms)
	rng = np.random.default_rng(seed)
	shape = (nb, nc, nt, nph) if nb > 1 else (nc, nt, nph)
	noise = rng.standard_normal(shape)
	out = {}
	for line in ("1", "0"):
		monkeypatch.setenv("PXS_THETA_LINE", line)
		plan = sht.grid_plan("F1", nt, nph, 0.1, (False, False), lmax, mmax, ms, 1)
		assert plan.query("theta_line") == int(line)
		alm = np.zeros(shape[:-2] + (nalm,), complex)
		sht.analysis_2d(alm=alm, map=noise, **kw)                       # to_cc (the default, fine-CC form)
		alm2 = np.zeros_like(alm)
		sht.adjoint_synthesis_2d(alm=alm2, map=noise, **kw)             # from_cc_adjoint (grids with > 1.25 / 1.5 N_cc/2 rings)
		out[line] = (alm, alm2)
	assert relrms(out["1"][0], out["0"][0]) < 1e-13, relrms(out["1"][0], out["0"][0])
	assert relrms(out["1"][1], out["0"][1]) < 1e-13, relrms(out["1"][1], out["0"][1])
	assert np.abs(out["0"][0]).max() > 0 and np.abs(out["0"][1]).max() > 0

@pytest.mark.hostsim
@pytest.mark.parametrize("spin,nb,mmax", [(0, 1, 24), (1, 1, 24), (2, 1, 250), (0, 3, 250)])      # (mmax = 24: 25 columns, the last pair has one member; spin 1: the odd column first)
def test_line_engine_hostsim(monkeypatch, spin, nb, mmax):
	assert _lib.is_hostsim()
	run_pair(360, 720, 250, spin, nb, monkeypatch, mmax=mmax)

@pytest.mark.gpu
@pytest.mark.parametrize("spin,nb", [(0, 1), (2, 1), (0, 3)])
def test_line_engine_c2_size_gpu(monkeypatch, spin, nb):
	"""BASELINE C2 / C4 grid: 5400 x 10800, lmax 4000 (N = 10 800, N_cc = 8064, M = 16 128)"""
	run_pair(5400, 10800, 4000, spin, nb, monkeypatch)

def run_ring_pair(nt, nph, lmax, spin, monkeypatch, dtype=np.float64, flip=(False, False), mmax=None):
	"""synthesis_2d with the single-kernel ring FFT (ringline.hip) and with the two-stage chain: the same map"""
	nc = 1 if spin == 0 else 2
	if mmax is None: mmax = lmax
	ms = sht.tri_mstart(lmax, mmax); nalm = int(ms[-1]) + lmax + 1
	rng = np.random.default_rng(4)
	alm = rng.standard_normal((nc, nalm)) + 1j*rng.standard_normal((nc, nalm))
	for m0 in ms[:1]: alm[:, int(m0):int(m0) + lmax + 1] = alm[:, int(m0):int(m0) + lmax + 1].real
	out = {}
	for line in ("1", "0"):
		monkeypatch.setenv("PXS_RING_LINE", line)
		m = np.zeros((nc, nt, nph), dtype)
		sht.synthesis_2d(alm=alm, map=m, spin=spin, lmax=lmax, mmax=mmax, geometry="F1", phi0=0.2, mstart=ms, flip=flip)
		out[line] = m
	scale = np.abs(out["0"]).max()
	assert scale > 0 and np.abs(out["1"] - out["0"]).max() < (1e-13 if dtype == np.float64 else 1e-6)*scale

@pytest.mark.hostsim
def test_ring_line_hostsim(monkeypatch):
	run_ring_pair(360, 720, 250, 0, monkeypatch, mmax=60)      # (odd ring-pair handling: 360 rings = 180 pairs; 720 pixels = the simulator's configuration)
	run_ring_pair(181, 720, 200, 2, monkeypatch, mmax=40)      # an odd number of rings: the last pair has one ring
	run_ring_pair(360, 720, 250, 2, monkeypatch, dtype=np.float32, flip=(True, False))      # float32 maps, rows stored south to north
	run_ring_pair(360, 720, 250, 0, monkeypatch, flip=(False, True))                        # pixels stored east to west

def run_band(ny_full, nx, lmax, dec_cut_deg, monkeypatch):
	"""a declination band through the curvedsky interface (method "cyl": explicit rings): alm2map with the ring engine and with the chain"""
	from pixell_amd import curvedsky, enmap
	shape, wcs = enmap.band_geometry(np.deg2rad(dec_cut_deg), shape=None, res=np.pi/ny_full)
	assert shape[-1] == nx
	rng = np.random.default_rng(8)
	ainfo = curvedsky.alm_info(lmax)
	alm = rng.standard_normal((3, ainfo.nelem)) + 1j*rng.standard_normal((3, ainfo.nelem)); alm[:, :lmax + 1] = alm[:, :lmax + 1].real
	out = {}
	for line in ("1", "0"):
		monkeypatch.setenv("PXS_RING_LINE", line)
		sht.clear_plans()
		out[line] = np.array(curvedsky.alm2map(alm, enmap.zeros((3,) + tuple(shape[-2:]), wcs), spin=[0, 2], ainfo=ainfo))
	scale = np.abs(out["0"]).max()
	assert scale > 0 and np.abs(out["1"] - out["0"]).max() < 1e-13*scale

@pytest.mark.hostsim
def test_ring_line_band_hostsim(monkeypatch): run_band(360, 720, 200, 20.0, monkeypatch)
@pytest.mark.gpu
def test_ring_line_band_gpu(monkeypatch): run_band(5400, 10800, 3000, 15.0, monkeypatch)

@pytest.mark.gpu
@pytest.mark.parametrize("spin,dtype,flip", [(0, np.float64, (False, False)), (2, np.float64, (True, True)), (0, np.float32, (False, True))])
def test_ring_line_c2_size_gpu(monkeypatch, spin, dtype, flip):
	run_ring_pair(5400, 10800, 4000, spin, monkeypatch, dtype=dtype, flip=flip)

@pytest.mark.gpu
def test_ring_line_odd_rings_gpu(monkeypatch):
	run_ring_pair(2701, 10800, 2600, 0, monkeypatch)

@pytest.mark.gpu
def test_other_sizes_keep_the_chain_gpu():
	lmax = 300; ms = sht.tri_mstart(lmax, lmax)
	plan = sht.grid_plan("F1", 512, 1024, 0.0, (False, False), lmax, lmax, ms, 1)
	assert plan.query("theta_line") == 0
