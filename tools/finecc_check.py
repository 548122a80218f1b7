#!/usr/bin/env python
"""The default form of the analysis (analysis="ducc0": ducc0's route, its fine-CC form) against the oracle's restatement, at the N_cc the plan chose, on band-limited
maps and white noise, and its adjoint.  usage: tools/finecc_check.py [small|big]"""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixell_amd import sht, _lib
from oracle import sht_oracle as so
def theta_plan(N, lmax):
	out = (ctypes.c_int64*10)(); _lib.load().pxs_debug_theta_plan(N, lmax, out); return list(out)
def relrms(a, b): return np.sqrt(np.sum(np.abs(a-b)**2)/np.sum(np.abs(b)**2))
cases = [("F1", 120, 240, 30, 0), ("F1", 96, 192, 20, 2), ("CC", 57, 100, 27, 0), ("F1", 64, 128, 30, 0), ("MW", 63, 100, 20, 0)]
if len(sys.argv) > 1 and sys.argv[1] == "big": cases += [("F1", 300, 600, 127, 2), ("F1", 540, 1080, 250, 0), ("F1", 400, 1000, 399, 1), ("CC", 421, 800, 200, 2)]
worst = 0
for geometry, nt, nph, lmax, spin in cases:
	N = so.grid_info(geometry, nt)["N"]; tp = theta_plan(N, lmax)
	if not tp[0]: print(geometry, nt, lmax, "no theta chain"); continue
	Ncc = tp[6]; nc = 1 if spin == 0 else 2
	kw = dict(spin=spin, lmax=lmax, geometry=geometry, phi0=0.2, mstart=so._tri_mstart(lmax, lmax))
	alm = so.rand_alm_simple(lmax, nc, 5, spin=(spin,))
	band = np.zeros((nc, nt, nph)); so.synthesis_2d(alm=alm, map=band, **kw)
	noise = np.random.default_rng(1).standard_normal((nc, nt, nph))
	res = []
	for m in (band, noise):
		ref = np.zeros_like(alm); so.analysis_2d(alm=ref, map=m, fine_cc=Ncc, **kw)
		got = np.zeros_like(alm); sht.analysis_2d(alm=got, map=m, analysis="ducc0", **kw)
		res.append(relrms(got, ref))
	ref2 = np.zeros((nc, nt, nph)); so.adjoint_analysis_2d(alm=alm, map=ref2, fine_cc=Ncc, **kw)
	out2 = np.zeros((nc, nt, nph)); sht.adjoint_analysis_2d(alm=alm, map=out2, analysis="ducc0", **kw)
	res.append(np.max(np.abs(out2-ref2))/np.max(np.abs(ref2)))
	worst = max(worst, max(res))
	print(geometry, nt, nph, lmax, spin, "N_cc %d (ducc0's: %d) g %d ac %d | band %.1e noise %.1e adjoint %.1e" % (Ncc, 2*so.good_size_complex(lmax+1), tp[1], tp[5], *res))
print("worst %.2e" % worst); assert worst < 1e-11
