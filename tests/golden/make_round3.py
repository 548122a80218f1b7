"""Round-3 fixtures from the reference itself (THIS container only; needs /root/reference; never run on the GPU box).

Run:  python tests/golden/make_round3.py      -> tests/golden/round3.npz

  slice_*    enmap.slice_geometry (enmap.py:264-285) for steps +-1, +-2, +-3 with and without offsets: shape, crpix, cdelt
  qw_*       curvedsky.quad_weights (curvedsky.py:492-505) on rows of a Fejer-1 grid stored south-to-north (the standard order)
             and north-to-south, symmetric and asymmetric bands
  cyl_*      curvedsky.map2alm through the cyl path (curvedsky.py:843-873, 1050-1086) with the reference's own weight handling,
             run on top of the CPU oracle mounted as ducc0.sht.experimental:
               default weights on named-grid bands in both row orders; default (pixel-area) weights on a band shifted off the
               grid, both row orders; caller-supplied asymmetric weights, both row orders; niter 0 and 1
Only arrays (inputs, geometry numbers, outputs) are stored; no reference code.
"""
import sys, os
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
from oracle import sht_oracle as so
import _ref_harness as H

def wcsd(w, pre): return {pre+"cdelt": np.array(w.wcs.cdelt), pre+"crval": np.array(w.wcs.crval), pre+"crpix": np.array(w.wcs.crpix)}

def main():
	ns = H.load_reference(so)
	enmap, curvedsky = ns.enmap, ns.curvedsky
	out = {}
	# ---- slice_geometry --------------------------------------------------------------------------------------
	shape, wcs = enmap.fullsky_geometry(shape=(12, 24))
	sels = [(slice(None), slice(None)), (slice(2, 9), slice(4, 20)), (slice(None, None, -1), slice(None)), (slice(None, None, 2), slice(None, None, 2)),
		(slice(1, None, 2), slice(3, None, 3)), (slice(None, None, -2), slice(None, None, -3)), (slice(10, 2, -2), slice(20, 3, -3)), (slice(3, 11, 4), slice(None, None, -1))]
	out["slice_n"] = np.array(len(sels)); out.update(wcsd(wcs, "slice_in_")); out["slice_in_shape"] = np.array(shape)
	for i, sel in enumerate(sels):
		s2, w2 = enmap.slice_geometry(shape, wcs, sel)
		out["slice_%d_sel" % i] = np.array([[-9999 if v is None else v for v in (s.start, s.stop, s.step)] for s in sel])
		out["slice_%d_shape" % i] = np.array(s2); out.update(wcsd(w2, "slice_%d_" % i))
	# ---- quad_weights and the cyl path -----------------------------------------------------------------------
	lmax = 16
	rng = np.random.default_rng(5)
	def flipped(shape, wcs): return enmap.slice_geometry(shape, wcs, (slice(None, None, -1), slice(None)))
	geos = {}
	geos["band_sym"] = enmap.band_geometry(np.deg2rad(40), shape=(36, 72))
	geos["band_asym"] = enmap.band_geometry((np.deg2rad(-62), np.deg2rad(25)), shape=(36, 72))
	s, w = geos["band_asym"]; w = w.deepcopy(); w.wcs.crpix[1] += 0.3          # rows shifted off the Fejer-1 grid: no named grid, pixel-area weights
	geos["shift_asym"] = (s, w)
	for k in list(geos): geos[k+"_n2s"] = flipped(*geos[k])
	names = sorted(geos); out["geo_names"] = np.array(",".join(names))
	for k in names:
		shape, wcs = geos[k]; shape = tuple(int(v) for v in shape[-2:])
		out["geo_%s_shape" % k] = np.array(shape); out.update(wcsd(wcs, "geo_%s_" % k))
		mi = curvedsky.analyse_geometry(shape, wcs)
		out["geo_%s_flip" % k] = np.array([bool(f) for f in mi.flip]); out["geo_%s_case" % k] = np.array(str(mi.case))
		if mi.ducc_geo is not None and mi.ducc_geo.name is not None:
			out["qw_%s" % k] = np.array(curvedsky.quad_weights(shape, wcs))
		pix = enmap.ndmap(rng.standard_normal((3,)+shape), wcs)
		out["cyl_%s_map" % k] = np.array(pix)
		for niter in (0, 1):
			out["cyl_%s_default_niter%d" % (k, niter)] = np.array(curvedsky.map2alm(pix.copy(), lmax=lmax, spin=[0, 2], niter=niter, method="cyl"))
		wts = (0.5+rng.random(shape[0]))*4*np.pi/(shape[0]*shape[1]*2)                     # asymmetric caller-supplied row weights
		out["cyl_%s_weights" % k] = wts
		out["cyl_%s_explicit" % k] = np.array(curvedsky.map2alm(pix.copy(), lmax=lmax, spin=[0, 2], weights=wts.copy(), method="cyl"))
	out["lmax"] = np.array(lmax)
	np.savez_compressed(os.path.join(HERE, "round3.npz"), **out)
	print("round3.npz: %d arrays; geometries %s" % (len(out), names))
	for k in names: print("  ", k, out["geo_%s_shape" % k], "flip", out["geo_%s_flip" % k], "case", out["geo_%s_case" % k], "named grid:", "qw_%s" % k in out)

if __name__ == "__main__":
	main()
