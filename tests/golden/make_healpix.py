"""Fixtures for the healpix layer (pixell_amd.curvedsky.{get_ring_info_healpix, alm2map_healpix, map2alm_healpix}), made by
running the REFERENCE's own functions (pixell/curvedsky.py:312-403, 1192-1234) in this container with the long-double oracle
mounted as ducc0.sht.experimental.

Run:  python tests/golden/make_healpix.py      (needs /root/reference; never run on the GPU box)

Saved to healpix.npz: the ring tables of nside 1, 2, 8, 5 and of a ring subset; alm2map_healpix / its adjoint / deriv /
map2alm_healpix (niter 0 and 2, a theta window) outputs for seeded inputs.  Only arrays; no reference code."""
import sys, os, types
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..")); sys.path.insert(0, HERE)
from oracle import sht_oracle as so
import _ref_harness as H

def main():
	back = types.ModuleType("ducc0.sht.experimental")
	for n in ["synthesis_2d", "adjoint_synthesis_2d", "analysis_2d", "adjoint_analysis_2d", "synthesis", "adjoint_synthesis", "get_gridweights"]: setattr(back, n, getattr(so, n))
	ns = H.load_reference(back); cs = ns.curvedsky
	out = {}
	for nside in (1, 2, 5, 8):
		r = cs.get_ring_info_healpix(nside)
		for k in ("theta", "nphi", "phi0", "offsets"): out["rings%d_%s" % (nside, k)] = np.asarray(r[k])
	sub = np.array([0, 3, 4, 9, 14]); r = cs.get_ring_info_healpix(4, sub)
	out["sub_rings"] = sub
	for k in ("theta", "nphi", "phi0", "offsets"): out["sub_%s" % k] = np.asarray(r[k])
	rad = cs.get_ring_info_radial(np.array([0.1, 0.5, 2.0]))
	for k in ("theta", "nphi", "phi0", "offsets"): out["rad_%s" % k] = np.asarray(rad[k])
	nside, lmax = 4, 9
	rng = np.random.default_rng(21)
	alm = so.rand_alm_simple(lmax, 3, 8, spin=(0, 2)); out["alm"] = alm
	m = cs.alm2map_healpix(alm.copy(), nside=nside, spin=[0, 2]); out["alm2map"] = np.array(m)
	pix = rng.standard_normal((3, 12*nside**2)); out["pix"] = pix
	at = cs.alm2map_healpix(np.zeros_like(alm), pix.copy(), spin=[0, 2], adjoint=True); out["alm2map_adjoint"] = np.array(at)
	d = cs.alm2map_healpix(alm[0].copy(), np.zeros((2, 12*nside**2)), deriv=True); out["deriv"] = np.array(d)
	w = cs.alm2map_healpix(alm.copy(), nside=nside, spin=[0, 2], theta_min=0.6, theta_max=2.2); out["alm2map_window"] = np.array(w)
	for niter in (0, 2):
		a = cs.map2alm_healpix(pix.copy(), lmax=lmax, spin=[0, 2], niter=niter); out["map2alm_niter%d" % niter] = np.array(a)
	a = cs.map2alm_healpix(m.copy(), lmax=lmax, spin=[0, 2], niter=3); out["map2alm_roundtrip"] = np.array(a)
	ma = cs.map2alm_healpix(np.zeros((3, 12*nside**2)), alm=alm.copy(), spin=[0, 2], adjoint=True, niter=1); out["map2alm_adjoint"] = np.array(ma)
	# 1-D profile transforms (m = 0 on one-pixel rings): a Gaussian beam and a two-row stack with extrapolation values
	sig = np.deg2rad(1.5); rr = np.linspace(0, 12*sig, 400)
	br = np.stack([np.exp(-0.5*rr**2/sig**2), (1+rr)**-3.0])
	out["prof_r"] = rr; out["prof_br"] = br
	out["prof_bl"] = cs.profile2harm(br, rr, lmax=150)
	out["prof_bl_auto"] = cs.profile2harm(br[0], rr)
	out["prof_bl_lr"] = cs.profile2harm(br[1, 20:], rr[20:], lmax=90, left=2.0, right=0.0)
	r2 = np.array([0.0, 0.01, 0.05, 0.3, 1.0, 2.5, np.pi]); out["prof_r2"] = r2
	out["prof_back"] = cs.harm2profile(out["prof_bl"], r2)
	from pixell import uharm
	shape, wcs = ns.enmap.fullsky_geometry(shape=(46, 90))
	uht = uharm.UHT(shape, wcs, mode="curved", lmax=40)
	hp = np.exp(-0.5*np.arange(41.0)**2*sig**2*9)
	out["uht_hprof"] = hp; out["uht_rpow"] = uht.hprof_rpow(hp, 2.0)
	out["uht_rprof2hprof"] = uht.rprof2hprof(br[0], rr); out["uht_hprof2rprof"] = uht.hprof2rprof(hp, r2)
	# helper entry points by name: buffers, raw transforms on them, alm real <-> complex
	enmap = ns.enmap
	shape, wcs = enmap.fullsky_geometry(shape=(14, 24))
	bshape, bwcs = enmap.band_geometry(np.deg2rad(50), shape=(10, 24))
	mb = enmap.enmap(rng.standard_normal((3,)+tuple(bshape)), bwcs)
	info = cs.analyse_geometry(mb.shape, mb.wcs)
	pad = np.array([info.ypad, info.xpad]).T
	buf = cs.map2buffer(mb, info.flip, pad)
	out["buf_in"] = np.array(mb); out["buf_flip"] = np.array(info.flip); out["buf_pad"] = pad; out["buf_out"] = np.array(buf)
	out["buf_wcs"] = np.array([buf.wcs.wcs.cdelt, buf.wcs.wcs.crval, buf.wcs.wcs.crpix]); out["buf_in_wcs"] = np.array([bwcs.wcs.cdelt, bwcs.wcs.crval, bwcs.wcs.crpix])
	out["buf_back"] = np.array(cs.buffer2map(buf, info.flip, pad))
	a12 = so.rand_alm_simple(12, 3, 9, spin=(0, 2)); out["raw_alm"] = a12
	raw = cs.alm2map_raw_2d(a12.copy(), enmap.zeros((3,)+buf.shape[-2:], buf.wcs), spin=[0, 2]); out["raw_alm2map_2d"] = np.array(raw)
	out["raw_map2alm_2d"] = np.array(cs.map2alm_raw_2d(buf.copy(), alm=np.zeros_like(a12), spin=[0, 2]))      # (without alm= the reference returns None)
	cyl = cs.map2buffer(mb, info.flip, np.zeros((2, 2), int))
	out["raw_cyl_in"] = np.array(cyl); out["raw_cyl_wcs"] = np.array([cyl.wcs.wcs.cdelt, cyl.wcs.wcs.crval, cyl.wcs.wcs.crpix])
	out["raw_alm2map_cyl"] = np.array(cs.alm2map_raw_cyl(a12.copy(), enmap.zeros(cyl.shape, cyl.wcs), spin=[0, 2]))
	wq = 4*np.pi/24*np.sin(cs.get_ring_info(cyl.shape, cyl.wcs).theta)*np.pi/15      # explicit weights: pixell's quad_weights reverses the rows of EVERY map
	out["raw_cyl_weights"] = wq                                                     # (`if minfo.flip:` on a list, curvedsky.py:503), wrong for an asymmetric band stored north to south
	out["raw_map2alm_cyl"] = np.array(cs.map2alm_raw_cyl(cyl.copy(), alm=np.zeros_like(a12), spin=[0, 2], niter=1, weights=wq))
	out["c2r"] = cs.alm_complex2real(a12); out["r2c"] = cs.alm_real2complex(out["c2r"][0])
	out["maxlmax"] = np.array([[cs.get_ducc_maxlmax(n, k) for k in (8, 9, 30)] for n in ("CC", "F1", "MW", "MWflip", "DH", "F2")])
	from pixell import fft as rfft
	rfft.set_engine("numpy")          # (the harness stubs ducc0: its FFT engine must not be picked)
	xx = rng.standard_normal((3, 17)); out["cheb_in"] = xx; out["cheb"] = np.array([rfft.chebt(v.copy()) for v in xx])
	out["icheb"] = np.array([rfft.ichebt(v.copy()) for v in out["cheb"]])
	out["meta"] = np.array([nside, lmax])
	np.savez_compressed(os.path.join(HERE, "healpix.npz"), **out)
	print("healpix.npz: %d arrays; round trip error after 3 Jacobi steps %.2e" % (len(out), np.max(np.abs(a-alm))/np.max(np.abs(alm))))

if __name__ == "__main__": main()
