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

def _norm(emap, normalize, sign, dct=False):
	norm = 1.0
	if normalize: norm /= (np.prod(2*np.array(emap.shape[-2:])-1)**0.5 if dct else np.prod(emap.shape[-2:])**0.5)   # (enmap.py:1318,1331)
	if normalize in ["phy", "phys", "physical"]: norm *= emap.pixsize()**(0.5*sign)
	return norm

def fft(emap, omap=None, nthread=0, normalize=True, adjoint_ifft=False, dct=False):
	"""enmap.fft (enmap.py:1307-1323): 2-D FFT over the last two axes, scaling fused into the
	last kernel pass instead of a separate `res *= norm` sweep."""
	norm = _norm(emap, normalize, -1 if adjoint_ifft else +1, dct=dct)
	if dct: return _wrap(enfft.dct(_data(emap), _data(omap) if omap is not None else None, axes=[-2, -1], nthread=nthread, _scale=norm), emap)
	res = enfft.fft(_data(emap), _data(omap) if omap is not None else None, axes=[-2, -1], nthread=nthread, _scale=norm)
	return _wrap(res, emap)

def ifft(emap, omap=None, nthread=0, normalize=True, adjoint_fft=False, dct=False):
	"""enmap.ifft (enmap.py:1325-1337)"""
	norm = _norm(emap, normalize, +1 if adjoint_fft else -1, dct=dct)
	if dct: return _wrap(enfft.idct(_data(emap), _data(omap) if omap is not None else None, axes=[-2, -1], nthread=nthread, normalize=False, _scale=norm), emap)
	res = enfft.ifft(_data(emap), _data(omap) if omap is not None else None, axes=[-2, -1], nthread=nthread, normalize=False, _scale=norm)
	return _wrap(res, emap)

def dct(emap, omap=None, nthread=0, normalize=True): return fft(emap, omap=omap, nthread=nthread, normalize=normalize, dct=True)
def idct(emap, omap=None, nthread=0, normalize=True): return ifft(emap, omap=omap, nthread=nthread, normalize=normalize, dct=True)
def fft_adjoint(emap, omap=None, nthread=0, normalize=True): return ifft(emap, omap=omap, nthread=nthread, normalize=normalize, adjoint_fft=True)
def ifft_adjoint(emap, omap=None, nthread=0, normalize=True): return fft(emap, omap=omap, nthread=nthread, normalize=normalize, adjoint_ifft=True)

def _data(m):
	if m is None: return None
	if isinstance(m, dmap): return m.tensor
	if hasattr(m, "data_ptr"): return m          # a bare torch tensor
	return np.asarray(m)
def _wrap(res, like):
	if isinstance(like, dmap): return dmap(res, like.wcs)
	return ndmap(res, getattr(like, "wcs", None))

# ---------------------------------------------------------------------------------------
# flat-sky harmonic helpers around fft/ifft.  Geometry arithmetic is host numpy; everything
# that touches an [ny,nx] array runs on the GPU (include/pxsht.h pxm_*).
# ---------------------------------------------------------------------------------------
def extent(shape, wcs, signed=False, method="auto"):
	"""[height, width] of the patch in radians (enmap.extent / extent_cyl / extent_intermediate, enmap.py:917-1014)"""
	if method == "auto": method = "cylindrical" if wcsutils.is_separable(wcs) else "intermediate"
	if method in ("inter", "intermediate"):
		res = np.array(wcs.wcs.cdelt[::-1], float)*np.array(shape[-2:], float)*degree
		return res if signed else np.abs(res)
	if method not in ("cyl", "cylindrical"): raise NotImplementedError("extent: only the cylindrical and intermediate methods")
	dec1, dec2 = pix2sky(shape, wcs, [[-0.5, shape[-2]-1+0.5], [0, 0]])[0]
	if dec1 <= dec2: ysign = 1
	else: dec1, dec2, ysign = dec2, dec1, -1
	dec1, dec2 = max(-np.pi/2, dec1), min(np.pi/2, dec2)
	mean_cos = (np.sin(dec2)-np.sin(dec1))/(dec2-dec1)
	ext = np.array([(dec2-dec1)*ysign, shape[-1]*wcs.wcs.cdelt[0]*mean_cos*degree])
	return ext if signed else np.abs(ext)

def laxes(shape, wcs, oversample=1, method="auto", broadcastable=False):
	"""wavenumber axes ly[ny], lx[nx] of the 2-D FFT of a map (enmap.laxes, enmap.py:1275-1294)"""
	oversample = int(oversample)
	step = extent(shape, wcs, signed=True, method=method)/np.array(shape[-2:], float)
	ly = np.fft.fftfreq(shape[-2]*oversample, step[0])*2*np.pi
	lx = np.fft.fftfreq(shape[-1]*oversample, step[1])*2*np.pi
	if oversample > 1:
		def shift(l, a, n): return l+a/2*(-1+1./n)
		ly = shift(ly, ly[oversample], oversample)
		lx = shift(lx, lx[oversample], oversample)
	if broadcastable: ly, lx = ly[:, None], lx[None, :]
	return ly, lx

def lmap(shape, wcs, oversample=1, method="auto"):
	ly, lx = laxes(shape, wcs, oversample=oversample, method=method)
	data = np.empty((2, ly.size, lx.size))
	data[0] = ly[:, None]; data[1] = lx[None, :]
	return ndmap(data, wcs)

def modlmap(shape, wcs, oversample=1, method="auto", min=0):
	slmap = lmap(shape, wcs, oversample=oversample, method=method)
	l = np.sum(np.asarray(slmap)**2, 0)**0.5
	if min > 0: l = np.maximum(l, min)
	return ndmap(l, wcs)

def lpixshape(shape, wcs, signed=False, method="auto"): return 2*np.pi/extent(shape, wcs, signed=signed, method=method)
def lpixsize(shape, wcs, signed=False, method="auto"): return np.prod(lpixshape(shape, wcs, signed=signed, method=method))

def queb_rotmat(lmap, inverse=False, iau=False, spin=2, wcs=None):
	"""host version of the rotation matrix (enmap.py:1391-1400); the transforms below never build it"""
	sign = 1
	if iau: sign = -sign
	if inverse: sign = -sign
	a = spin*np.arctan2(sign*np.asarray(lmap[1]), np.asarray(lmap[0]))
	c, s = np.cos(a), np.sin(a)
	return samewcs(np.array([[c, -s], [s, c]]), lmap)

def _torch():
	import torch
	return torch
def _to_device(emap, dtype=None):
	"""(dmap on the GPU, True if the input was a host array)"""
	from . import sht
	if isinstance(emap, dmap): return emap, False
	if sht._lib.is_hostsim(): return ndmap(np.ascontiguousarray(emap, dtype=dtype), getattr(emap, "wcs", None)), False
	sht.device_index()
	t = _torch().from_numpy(np.ascontiguousarray(emap, dtype=dtype)).cuda()
	return dmap(t, getattr(emap, "wcs", None)), True
def _to_host(m):
	return ndmap(m.tensor.cpu().numpy(), m.wcs) if isinstance(m, dmap) else m
def _ptr(x):
	x = _data(x)
	return x.data_ptr() if hasattr(x, "data_ptr") else x.ctypes.data
def _dev_axes(shape, wcs, like):
	"""ly, lx as device f64 arrays living next to `like`"""
	from . import sht
	ly, lx = laxes(shape, wcs)
	if sht._lib.is_hostsim(): return np.ascontiguousarray(ly), np.ascontiguousarray(lx)
	torch = _torch()
	return torch.from_numpy(np.ascontiguousarray(ly)).to(like.device), torch.from_numpy(np.ascontiguousarray(lx)).to(like.device)

def _rotate_pairs(hmap, spin, iau, inverse):
	"""rotate every spin-s pair of a contiguous complex harmonic map [...,ncomp,ny,nx] in place"""
	from . import sht
	data = _data(hmap)
	if data.ndim <= 2: return
	ny, nx = data.shape[-2:]; nc = data.shape[-3]
	lib = sht._lib.load(); dev = sht.device_index(); st = sht.current_stream()
	dt = sht._DT[sht._np_dtype(data)]; esz = sht._np_dtype(data).itemsize
	ly, lx = _dev_axes(hmap.shape, hmap.wcs, data)
	npre = int(np.prod(data.shape[:-3], dtype=int))
	base = _ptr(data)
	for s, i1, i2 in spin_helper(spin, nc):
		if s == 0: continue
		sign = (-1 if iau else 1)*(-1 if inverse else 1)
		for p in range(npre):
			a = int(base+((p*nc+int(i1))*ny*nx)*esz); b = int(a+ny*nx*esz)
			sht._lib.check(lib.pxm_rotate_queb(ny, nx, _ptr(ly), _ptr(lx), int(s), 1 if sign < 0 else 0, a, b, dt, dev, st))

def map2harm(emap, nthread=0, normalize=True, iau=False, spin=[0, 2], adjoint_harm2map=False):
	"""2-D FFT + Q/U -> E/B rotation (enmap.map2harm, enmap.py:1358-1375); the rotation runs in place on the GPU"""
	dev, was_host = _to_device(emap)
	res = fft(dev, nthread=nthread, normalize=normalize, adjoint_ifft=adjoint_harm2map)
	if res.ndim > 2: _rotate_pairs(res, spin, iau, inverse=False)
	return _to_host(res) if was_host else res

def harm2map(emap, nthread=0, normalize=True, iau=False, spin=[0, 2], keep_imag=False, adjoint_map2harm=False):
	"""E/B -> Q/U rotation + inverse 2-D FFT (enmap.harm2map, enmap.py:1376-1389)"""
	dev, was_host = _to_device(emap)
	if dev.ndim > 2:
		dev = dev.copy()
		_rotate_pairs(dev, spin, iau, inverse=True)
	res = ifft(dev, nthread=nthread, normalize=normalize, adjoint_fft=adjoint_map2harm)
	if not keep_imag:
		r = _data(res).real
		res = _wrap(r.contiguous() if hasattr(r, "contiguous") else np.ascontiguousarray(r), res)
	return _to_host(res) if was_host else res

def map2harm_adjoint(emap, nthread=0, normalize=True, iau=False, spin=[0, 2], keep_imag=False):
	return harm2map(emap, nthread=nthread, normalize=normalize, iau=iau, spin=spin, keep_imag=keep_imag, adjoint_map2harm=True)
def harm2map_adjoint(emap, nthread=0, normalize=True, iau=False, spin=[0, 2]):
	return map2harm(emap, nthread=nthread, normalize=normalize, iau=iau, spin=spin, adjoint_harm2map=True)

def calc_ps2d(harm, harm2=None):
	"""2-D (cross) power spectrum Re(harm conj(harm2)) with numpy broadcasting of the leading axes
	(enmap.calc_ps2d, enmap.py:1959-2011); each distinct pair of 2-D maps is computed once."""
	from . import sht
	same = harm2 is None or harm2 is harm
	h1, host1 = _to_device(harm); h2, host2 = (h1, host1) if same else _to_device(harm2)
	d1, d2 = _data(h1), _data(h2)
	ct = np.result_type(sht._np_dtype(d1), sht._np_dtype(d2))
	if ct not in (np.dtype(np.complex64), np.dtype(np.complex128)): raise ValueError("calc_ps2d needs complex harmonic maps")
	def cast(d):
		if sht._np_dtype(d) == ct: return d if not hasattr(d, "contiguous") else d.contiguous()
		return d.to(getattr(_torch(), np.dtype(ct).name)) if hasattr(d, "data_ptr") else d.astype(ct)
	d1 = cast(d1); d2 = d1 if same else cast(d2)
	if not hasattr(d1, "data_ptr"): d1 = np.ascontiguousarray(d1); d2 = d1 if same else np.ascontiguousarray(d2)
	ny, nx = d1.shape[-2:]
	pshape = np.broadcast_shapes(tuple(d1.shape[:-2]), tuple(d2.shape[:-2]))
	i1 = np.broadcast_to(np.arange(int(np.prod(d1.shape[:-2], dtype=int))).reshape(d1.shape[:-2]), pshape).reshape(-1)
	i2 = np.broadcast_to(np.arange(int(np.prod(d2.shape[:-2], dtype=int))).reshape(d2.shape[:-2]), pshape).reshape(-1)
	rt = np.dtype(np.float32) if ct == np.dtype(np.complex64) else np.dtype(np.float64)
	if hasattr(d1, "data_ptr"): out = _torch().empty(tuple(pshape)+(ny, nx), dtype=getattr(_torch(), rt.name), device=d1.device)
	else: out = np.empty(tuple(pshape)+(ny, nx), rt)
	lib = sht._lib.load(); dev = sht.device_index(); st = sht.current_stream()
	n = ny*nx; done = {}
	flat = out.reshape(-1, ny, nx)
	for i in range(len(i1)):
		key = tuple(sorted((int(i1[i]), int(i2[i])))) if same else (int(i1[i]), int(i2[i]))
		if key in done: flat[i] = flat[done[key]]; continue
		done[key] = i
		sht._lib.check(lib.pxm_ps2d(n, _ptr(d1)+int(i1[i])*n*ct.itemsize, _ptr(d2)+int(i2[i])*n*ct.itemsize, sht._DT[ct],
			_ptr(out)+i*n*rt.itemsize, sht._DT[rt], dev, st))
	res = _wrap(out, h1)
	return _to_host(res) if host1 else res

def lbin(map, bsize=None, brel=1.0, return_nhit=False, return_bins=False, lop=None):
	"""radial binning of a real fourier-space map in |l| (enmap.lbin / _bin_helper, enmap.py:2526-2556): returns b(l), l"""
	from . import sht
	if lop is not None: raise NotImplementedError("lbin: lop is not supported by the accelerated path")
	ly, lx = laxes(map.shape, map.wcs)
	if bsize is None: bsize = min(abs(lx[1]), abs(ly[1]))
	bsize = float(bsize*brel)
	lmax = float(np.sqrt(np.max(ly**2)+np.max(lx**2)))
	n = int(lmax/bsize)
	dev_map, was_host = _to_device(map)
	d = _data(dev_map)
	if sht._np_dtype(d) not in (np.dtype(np.float32), np.dtype(np.float64)): raise ValueError("lbin needs a real map")
	if hasattr(d, "contiguous"): d = d.contiguous()
	ny, nx = d.shape[-2:]; npre = int(np.prod(d.shape[:-2], dtype=int))
	dly, dlx = _dev_axes(map.shape, map.wcs, d)
	if hasattr(d, "data_ptr"):
		torch = _torch()
		acc = torch.zeros((npre+2, max(n, 1)), dtype=torch.float64, device=d.device)
	else: acc = np.zeros((npre+2, max(n, 1)))
	lib = sht._lib.load(); dev = sht.device_index(); st = sht.current_stream()
	esz = sht._np_dtype(d).itemsize
	for i in range(npre):
		first = i == 0
		sht._lib.check(lib.pxm_lbin(ny, nx, _ptr(dly), _ptr(dlx), bsize, n, _ptr(d)+i*ny*nx*esz, sht._DT[sht._np_dtype(d)],
			_ptr(acc)+i*max(n, 1)*8, (_ptr(acc)+npre*max(n, 1)*8) if first else None, (_ptr(acc)+(npre+1)*max(n, 1)*8) if first else None, dev, st))
	acc = acc.cpu().numpy() if hasattr(acc, "data_ptr") else acc
	nhit = acc[npre+1, :n]
	with np.errstate(invalid="ignore", divide="ignore"):
		mout = (acc[:npre, :n]/nhit).reshape(tuple(d.shape[:-2])+(n,))
		orads = acc[npre, :n]/nhit
	if return_bins:
		edges = np.arange(len(orads)+1)*bsize
		orads = np.array([orads, edges[:-1], edges[1:]])
	if return_nhit: return mout, orads, nhit.astype(int)
	return mout, orads

for _cls in (ndmap, dmap):
	_cls.extent   = lambda self, **kw: extent(self.shape, self.wcs, **kw)
	_cls.laxes    = lambda self, **kw: laxes(self.shape, self.wcs, **kw)
	_cls.lmap     = lambda self, **kw: lmap(self.shape, self.wcs, **kw)
	_cls.modlmap  = lambda self, **kw: modlmap(self.shape, self.wcs, **kw)
	_cls.lbin     = lambda self, *a, **kw: lbin(self, *a, **kw)
