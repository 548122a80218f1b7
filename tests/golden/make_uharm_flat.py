"""Flat-mode profile helpers of uharm.UHT and the radial binning behind them, from the REFERENCE (this container only): pixell.uharm.UHT(mode="flat")
.rprof2hprof / .hprof2rprof / .hprof_rpow, enmap.rbin, enmap.lbin(lop=...), enmap.modrmap, enmap.shift on the reference's numpy FFT engine
(tests/golden/_ref_harness.py).  Saves inputs and the reference's outputs to uharm_flat.npz; tests/test_uharm.py drives pixell_amd through the same calls.
Run:  python tests/golden/make_uharm_flat.py"""
import sys, os
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..")); sys.path.insert(0, HERE)
from oracle import sht_oracle as so
import _ref_harness as H

def main():
	ns = H.load_reference(so)
	ns.fft.set_engine("numpy")
	from pixell import uharm
	enmap = ns.enmap
	rng = np.random.default_rng(11)
	out = {}
	shape, wcs = enmap.band_geometry(np.deg2rad(12), res=np.deg2rad(1.5)); shape = tuple(int(v) for v in shape[-2:])
	out["cdelt"] = np.array(wcs.wcs.cdelt); out["crval"] = np.array(wcs.wcs.crval); out["crpix"] = np.array(wcs.wcs.crpix); out["shape"] = np.array(shape)
	u = uharm.UHT(shape, wcs, mode="flat")
	r = np.linspace(0, np.deg2rad(20), 200)
	br = np.array([np.exp(-0.5*(r/np.deg2rad(4))**2), 1/(1+(r/np.deg2rad(3))**2)**2])
	out["r"] = r; out["br"] = br
	hp = u.rprof2hprof(br, r); out["hprof"] = np.array(hp)
	hp1 = u.rprof2hprof(br[0], r); out["hprof_1d"] = np.array(hp1)
	rq = np.linspace(0, np.deg2rad(15), 50); out["rq"] = rq
	out["rprof"] = np.array(u.hprof2rprof(hp1, rq))
	out["rpow"] = np.array(u.hprof_rpow(hp1, 2))
	try: u.hrand(hp1); out["hrand_raises"] = np.array(0)
	except Exception as e: out["hrand_raises"] = np.array(1); out["hrand_error"] = np.array(type(e).__name__)
	m = enmap.ndmap(rng.standard_normal((2,)+shape), wcs); out["map"] = np.array(m)
	b, l = enmap.lbin(m, lop=np.log1p); out["lbin_lop_b"] = b; out["lbin_lop_l"] = l
	b, l, nh = enmap.lbin(m[0], bsize=0.75, lop=np.sqrt, return_nhit=True, return_bins=True); out["lbin_lop2_b"] = b; out["lbin_lop2_l"] = l; out["lbin_lop2_nhit"] = nh
	cen = np.array([0.01, 0.02]); out["center"] = cen
	b, rr = enmap.rbin(m, center=cen); out["rbin_b"] = b; out["rbin_r"] = rr
	b, rr, nh = enmap.rbin(m[1], center=cen, bsize=0.05, brel=1.5, return_nhit=True); out["rbin2_b"] = b; out["rbin2_r"] = rr; out["rbin2_nhit"] = nh
	out["modrmap_center"] = np.array(enmap.modrmap(shape, wcs)); out["modrmap_ref"] = np.array(enmap.modrmap(shape, wcs, cen))
	sh = enmap.shift(m, [3, -5]); out["shift"] = np.array(sh); out["shift_crpix"] = np.array(sh.wcs.wcs.crpix)
	np.savez_compressed(os.path.join(HERE, "uharm_flat.npz"), **out)
	print("uharm_flat.npz written:", len(out), "arrays; reference hrand(flat) raised:", int(out["hrand_raises"]), str(out.get("hrand_error")))

if __name__ == "__main__":
	main()
