"""The few pieces of pixell.enmap the harmonic-transform path touches (host bookkeeping),
plus enmap.fft / enmap.ifft running on the HIP FFT engine.

Mirrors: ndmap (enmap.py:33-163), fullsky_geometry (enmap.py:1713-1740), spin_helper
(enmap.py:3378-3388), area_cyl / pixsize (enmap.py:1032-1036, 1097-1099), fft / ifft
(enmap.py:1307-1337)."""
import numpy as np
from . import wcs as wcsutils, fft as enfft
from .wcs import CarWCS

degree = np.pi/180

class ndmap(np.ndarray):
	"""numpy array + wcs (enmap.ndmap, enmap.py:33)"""
	def __new__(cls, arr, wcs):
		obj = np.asarray(arr).view(cls)
		obj.wcs = wcs
		return obj
	def __array_finalize__(self, obj):
		if obj is None: return
		self.wcs = getattr(obj, "wcs", None)
	def copy(self, order="C"): return ndmap(np.copy(self, order), self.wcs)
	@property
	def geometry(self): return self.shape, self.wcs
	def pixsize(self): return pixsize(self.shape, self.wcs)
	def area(self): return area(self.shape, self.wcs)

class dmap:
	"""device-resident map: a torch CUDA tensor + wcs, accepted wherever an ndmap is"""
	def __init__(self, tensor, wcs): self.tensor = tensor; self.wcs = wcs
	@property
	def shape(self): return tuple(self.tensor.shape)
	@property
	def ndim(self): return self.tensor.ndim
	@property
	def dtype(self):
		from .sht import _np_dtype
		return _np_dtype(self.tensor)
	def __getitem__(self, sel): return dmap(self.tensor[sel], self.wcs)
	def copy(self): return dmap(self.tensor.clone(), self.wcs)
	def pixsize(self): return pixsize(self.shape, self.wcs)

def enmap(arr, wcs=None, dtype=None, copy=True):
	if wcs is None: wcs = getattr(arr, "wcs", None)
	arr = np.array(arr, dtype=dtype, copy=copy) if copy else np.asarray(arr, dtype=dtype)
	return ndmap(arr, wcs)
def samewcs(arr, *args):
	for m in args:
		if hasattr(m, "wcs"): return ndmap(arr, m.wcs)
	return arr
def zeros(shape, wcs=None, dtype=None): return ndmap(np.zeros(shape, dtype=dtype), wcs)
def empty(shape, wcs=None, dtype=None): return ndmap(np.empty(shape, dtype=dtype), wcs)
def ones(shape, wcs=None, dtype=None): return ndmap(np.ones(shape, dtype=dtype), wcs)

def fullsky_geometry(res=None, shape=None, dims=(), proj="car", variant="fejer1"):
	"""enmap.fullsky_geometry (enmap.py:1713-1740)"""
	assert proj == "car", "Only CAR fullsky geometry implemented"
	if   variant.lower() == "cc":     yo = 1
	elif variant.lower() == "fejer1": yo = 0
	else: raise ValueError("Unrecognized CAR variant '%s'" % str(variant))
	if shape is None:
		res   = np.zeros(2)+res
		shape = np.round(([1*np.pi, 2*np.pi]/res)+(yo, 0)).astype(int)
	else:
		res = np.array([1*np.pi, 2*np.pi])/(np.array(shape)-(yo, 0))
	ny, nx = int(shape[0]), int(shape[1])
	assert abs(res[0]*(ny-yo)-np.pi) < 1e-8, "Vertical resolution does not evenly divide the sky; this is required for SHTs."
	assert abs(res[1]*nx-2*np.pi) < 1e-8, "Horizontal resolution does not evenly divide the sky; this is required for SHTs."
	wcs = CarWCS(cdelt=[-360./nx, 180./(ny-yo)], crval=[res[1]/2/degree, 0], crpix=[nx//2+0.5, (ny+1)/2])
	return tuple(dims)+(ny, nx), wcs

def band_geometry(dec_cut, res=None, shape=None, dims=(), proj="car", variant="fejer1"):
	"""rows of the full-sky geometry whose centres lie within the declination cut (enmap.py:1742-1772)"""
	dec_cut = np.atleast_1d(dec_cut)
	dmin, dmax = (-dec_cut[0], dec_cut[0]) if dec_cut.size == 1 else dec_cut
	fshape, fwcs = fullsky_geometry(res=res, shape=shape, dims=dims, proj=proj, variant=variant)
	y1 = wcsutils.world2pix(fwcs, 0, dmin/degree)[1]; y2 = wcsutils.world2pix(fwcs, 0, dmax/degree)[1]
	start = max(int(np.round(min(y1, y2))), 0); stop = min(int(np.round(max(y1, y2))), fshape[-2])
	w = fwcs.deepcopy(); w.wcs.crpix[1] -= start
	return tuple(dims)+(stop-start, fshape[-1]), w

def spin_helper(spin, n):
	"""enmap.spin_helper (enmap.py:3378-3388)"""
	spin  = np.array(spin).reshape(-1)
	scomp = 1+(spin != 0)
	ci, i1 = 0, 0
	while True:
		i2 = min(i1+scomp[ci], n)
		if i2-i1 != scomp[ci]: raise IndexError("Unpaired component in spin transform")
		yield spin[ci], i1, i2
		if i2 == n: break
		i1 = i2
		ci = (ci+1) % len(spin)

def pix2sky(shape, wcs, pix):
	"""[{y,x},...] -> [{dec,ra},...] in radians (enmap.pix2sky, enmap.py:483-494, linear CAR)"""
	pix = np.asarray(pix, float)
	ra, dec = wcsutils.pix2world(wcs, pix[1], pix[0])
	return np.array([dec*degree, ra*degree])

def area(shape, wcs):
	"""enmap.area_cyl (enmap.py:1032-1036)"""
	if not wcsutils.is_separable(wcs): raise NotImplementedError("area: only separable cylindrical geometries")
	d = pix2sky(shape, wcs, [[-0.5, shape[-2]-1+0.5], [0, 0]])[0]
	dec1, dec2 = np.sort(d)
	dec1, dec2 = max(-np.pi/2, dec1), min(np.pi/2, dec2)
	return (np.sin(dec2)-np.sin(dec1))*abs(wcs.wcs.cdelt[0])*shape[-1]*degree
def pixsize(shape, wcs): return area(shape, wcs)/np.prod(shape[-2:])

def _norm(emap, normalize, sign):
	norm = 1.0
	if normalize: norm /= np.prod(emap.shape[-2:])**0.5
	if normalize in ["phy", "phys", "physical"]: norm *= emap.pixsize()**(0.5*sign)
	return norm

def fft(emap, omap=None, nthread=0, normalize=True, adjoint_ifft=False, dct=False):
	"""enmap.fft (enmap.py:1307-1323): 2-D FFT over the last two axes, scaling fused into the
	last kernel pass instead of a separate `res *= norm` sweep."""
	if dct: raise NotImplementedError("dct is outside the accelerated path")
	norm = _norm(emap, normalize, -1 if adjoint_ifft else +1)
	res = enfft.fft(_data(emap), _data(omap) if omap is not None else None, axes=[-2, -1], nthread=nthread, _scale=norm)
	return _wrap(res, emap)

def ifft(emap, omap=None, nthread=0, normalize=True, adjoint_fft=False, dct=False):
	"""enmap.ifft (enmap.py:1325-1337)"""
	if dct: raise NotImplementedError("dct is outside the accelerated path")
	norm = _norm(emap, normalize, +1 if adjoint_fft else -1)
	res = enfft.ifft(_data(emap), _data(omap) if omap is not None else None, axes=[-2, -1], nthread=nthread, normalize=False, _scale=norm)
	return _wrap(res, emap)

def _data(m):
	if m is None: return None
	if isinstance(m, dmap): return m.tensor
	return np.asarray(m)
def _wrap(res, like):
	if isinstance(like, dmap): return dmap(res, like.wcs)
	return ndmap(res, getattr(like, "wcs", None))
