"""Minimal WCS stand-in for CAR maps (astropy is not required by the hot path).

Any object exposing `.wcs.cdelt`, `.wcs.crval`, `.wcs.crpix`, `.wcs.ctype` (e.g. an
astropy.wcs.WCS as used by pixell/wcsutils.py) is accepted by pixell_amd; CarWCS is the
dependency-free equivalent.  Linear relation, exact for CAR with crval[1] == 0 (the
`is_separable` case of wcsutils.py:174-175 that the 2d/cyl SHT paths require):
   world = crval + (pix + 1 - crpix) * cdelt        (degrees, FITS 1-based crpix)"""
import numpy as np

class _Prm:
	def __init__(self):
		self.cdelt = np.ones(2); self.crval = np.zeros(2); self.crpix = np.zeros(2)
		self.ctype = ["RA---CAR", "DEC--CAR"]

class CarWCS:
	def __init__(self, cdelt=None, crval=None, crpix=None, ctype=None):
		self.wcs = _Prm()
		if cdelt is not None: self.wcs.cdelt = np.array(cdelt, float)
		if crval is not None: self.wcs.crval = np.array(crval, float)
		if crpix is not None: self.wcs.crpix = np.array(crpix, float)
		if ctype is not None: self.wcs.ctype = list(ctype)
	def deepcopy(self):
		return CarWCS(self.wcs.cdelt, self.wcs.crval, self.wcs.crpix, self.wcs.ctype)
	def wcs_pix2world(self, x, y, origin):
		x, y = np.asarray(x, float), np.asarray(y, float)
		return [self.wcs.crval[0]+(x+1-origin-self.wcs.crpix[0])*self.wcs.cdelt[0],
			self.wcs.crval[1]+(y+1-origin-self.wcs.crpix[1])*self.wcs.cdelt[1]]
	def wcs_world2pix(self, ra, dec, origin):
		ra, dec = np.asarray(ra, float), np.asarray(dec, float)
		return [(ra-self.wcs.crval[0])/self.wcs.cdelt[0]+self.wcs.crpix[0]-1+origin,
			(dec-self.wcs.crval[1])/self.wcs.cdelt[1]+self.wcs.crpix[1]-1+origin]
	def __repr__(self):
		return "car:{cdelt:%s,crval:%s,crpix:%s}" % (list(self.wcs.cdelt), list(self.wcs.crval), list(self.wcs.crpix))

def get_proj(wcs):
	ct = wcs.wcs.ctype[0]
	return ct[-3:].lower() if len(ct) >= 3 else ""

def is_cyl(wcs): return get_proj(wcs) in ["cyp", "cea", "car", "mer"]
def is_separable(wcs):
	"""wcsutils.is_separable (wcsutils.py:174-175)"""
	return is_cyl(wcs) and wcs.wcs.crval[1] == 0
def is_car(wcs): return get_proj(wcs) == "car"

def pix2world(wcs, x, y):
	"""0-based pixel -> (ra, dec) in degrees (linear CAR)"""
	w = wcs.wcs
	return w.crval[0]+(np.asarray(x, float)+1-w.crpix[0])*w.cdelt[0], w.crval[1]+(np.asarray(y, float)+1-w.crpix[1])*w.cdelt[1]

def world2pix(wcs, ra, dec):
	w = wcs.wcs
	return (np.asarray(ra, float)-w.crval[0])/w.cdelt[0]+w.crpix[0]-1, (np.asarray(dec, float)-w.crval[1])/w.cdelt[1]+w.crpix[1]-1

def flipped(shape, wcs, flip):
	"""geometry of map[..., ::-1 if flip[0], ::-1 if flip[1]] (enmap.slice_geometry, enmap.py:264-285)"""
	w = wcs.deepcopy()
	ny, nx = shape[-2:]
	if flip[0]: w.wcs.cdelt[1] = -w.wcs.cdelt[1]; w.wcs.crpix[1] = ny+1-w.wcs.crpix[1]
	if flip[1]: w.wcs.cdelt[0] = -w.wcs.cdelt[0]; w.wcs.crpix[0] = nx+1-w.wcs.crpix[0]
	return tuple(shape), w

def slice_geometry(shape, wcs, sel):
	"""geometry of map[..., sel_y, sel_x] for two slice objects (any start / stop / step, negative steps included), with the
	convention of enmap.slice_geometry (enmap.py:264-285): the new pixels tile the area the selected pixels covered, i.e. the
	EDGE of the first selected pixel (its lower edge for a positive step, its upper edge for a negative one) stays where it was
	and the pixel size is multiplied by the step.  For |step| = 1 that also keeps the pixel centres; for larger steps the centre
	of new pixel p lies (|step| - 1)/2 old pixels beyond old pixel start + p*step."""
	w = wcs.deepcopy()
	oshape = list(shape)
	for axis, sl in zip((-2, -1), sel):                   # y is the second WCS axis, x the first
		n = shape[axis]; start, stop, step = sl.indices(n)
		count = len(range(start, stop, step))
		k = 1 if axis == -2 else 0
		edge = start-0.5 if step > 0 else start+0.5          # 0-based position of the outer edge of the first selected pixel
		w.wcs.crpix[k] = (w.wcs.crpix[k]-1-edge)/step+0.5    # (1-based crpix: pixel centre c sits at crpix c+1)
		w.wcs.cdelt[k] = w.wcs.cdelt[k]*step
		oshape[axis] = count
	return tuple(oshape), w
