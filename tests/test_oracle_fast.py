"""CPU: the large-size checkers of oracle/sht_fast.py agree with the long-double oracle at sizes both can do.
(They are what the -m gpu tests at the BASELINE sizes compare the HIP path with.)"""
import numpy as np
import pytest
from oracle import sht_oracle as so, sht_fast as sf

def relrms(a, b): return np.sqrt(np.mean(np.abs(a-b)**2))/max(np.sqrt(np.mean(np.abs(b)**2)), 1e-300)

@pytest.mark.parametrize("geometry,nt,nph,lmax,spin", [("F1", 20, 41, 19, 0), ("F1", 24, 50, 13, 2), ("CC", 21, 44, 19, 0), ("CC", 25, 52, 17, 2),
	("MW", 16, 33, 15, 1), ("MWflip", 16, 34, 15, 0), ("F1", 64, 130, 40, 2), ("F1", 41, 90, 40, 0)])
def test_analysis_columns_match_oracle(geometry, nt, nph, lmax, spin):
	"""non-band-limited random map: FFT-based theta resampling + C Legendre == the dense oracle, in the full-interpolant form and
	in the fine-CC form of ducc0's route (ducc0's N_cc, and one that makes the interpolant pass a low pass / a zero padding)"""
	rng = np.random.default_rng(5)
	nc = 1 if spin == 0 else 2
	pix = rng.standard_normal((nc, nt, nph)); phi0 = 0.37
	ms = so._tri_mstart(lmax, lmax)
	th, nphi, p0, rs = so._grid_rings(geometry, nt, nph, phi0)
	leg = so.map2leg(pix.reshape(nc, -1), nphi, p0, rs, lmax)              # [nc, nt, nm]
	msel = np.arange(lmax+1)
	N = so.grid_info(geometry, nt)["N"]
	for fine_cc in (False, 2*so.good_size_complex(lmax+1), 2*(lmax+1), 2*(N//4+lmax//2+1)):
		if fine_cc and geometry == "CC" and nt >= 2*lmax+2: continue      # (ducc0's route takes the grid's own weights there)
		ref = np.zeros((nc, so.nalm(lmax)), complex)
		so.analysis_2d(alm=ref, map=pix, spin=spin, lmax=lmax, mstart=ms, geometry=geometry, phi0=phi0, fine_cc=fine_cc)
		cols = sf.analysis_columns(np.transpose(leg, (2, 0, 1)), msel, spin, lmax, geometry, nt, nph, fine_cc=fine_cc or None)
		got = np.zeros_like(ref)
		for m in msel:
			l0 = max(m, spin)
			got[:, int(ms[m])+l0:int(ms[m])+lmax+1] = cols[m, :, l0:]
		assert relrms(got, ref) < 1e-11, fine_cc

@pytest.mark.parametrize("spin,lmax,nt,nph", [(0, 30, 40, 80), (2, 30, 33, 70)])
def test_pixels_on_rings_match_oracle(spin, lmax, nt, nph):
	nc = 1 if spin == 0 else 2
	alm = so.rand_alm_simple(lmax, nc, 3, spin=(spin,)); ms = so._tri_mstart(lmax, lmax); phi0 = -0.2
	ref = np.zeros((nc, nt, nph)); so.synthesis_2d(alm=alm, map=ref, spin=spin, lmax=lmax, mstart=ms, geometry="F1", phi0=phi0)
	sub = sf.symmetric_subset(nt, [0, 3, nt//2-1])
	th = so.grid_theta("F1", nt)
	leg = sf.synth_rings(alm, spin, lmax, th[sub], mchunk=7)
	xs = np.array([[0, 5, nph-1]]*len(sub))
	val = sf.pixels_on_rings(leg, phi0+2*np.pi*xs/nph)
	for i, r in enumerate(sub):
		assert np.max(np.abs(val[:, i]-ref[:, r, xs[i]])) < 1e-12*np.max(np.abs(ref))
