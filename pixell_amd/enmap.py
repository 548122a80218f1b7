"""The few pieces of pixell.enmap the harmonic-transform path touches (host bookkeeping),
plus enmap.fft / enmap.ifft running on the HIP FFT engine.

Mirrors: ndmap (enmap.py:33-163), fullsky_geometry (enmap.py:1713-1740), spin_helper
(enmap.py:3378-3388), area_cyl / pixsize (enmap.py:1032-1036, 1097-1099), fft / ifft
(enmap.py:1307-1337)."""
import threading
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
	def __getitem__(self, sel):
		"""slicing the two pixel axes moves the wcs with the data (enmap.ndmap.__getitem__, enmap.py:140-163); anything that
		removes or fancy-indexes a pixel axis returns a plain array, which has no geometry"""
		res = np.ndarray.__getitem__(self, sel)
		return _sliced(self, res, sel)
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
	def __getitem__(self, sel):
		res = _sliced(self, self.tensor[sel], sel)
		if not isinstance(res, dmap): raise IndexError("indexing a pixel axis of a dmap leaves no map geometry: slice the tensor itself")
		return res
	def copy(self): return dmap(self.tensor.clone(), self.wcs)
	def pixsize(self): return pixsize(self.shape, self.wcs)

def _sliced(parent, res, sel):
	"""wrap the result of parent[sel] with the geometry it now has (None axes / Ellipsis / leading-axis indexing allowed)"""
	nd = parent.ndim
	if isinstance(parent, ndmap) and not isinstance(res, np.ndarray): return res     # every axis indexed: a numpy scalar stays one
	sel = sel if isinstance(sel, tuple) else (sel,)
	if any(isinstance(x, (list, np.ndarray)) or hasattr(x, "data_ptr") for x in sel):
		return np.asarray(res) if isinstance(parent, ndmap) else res          # fancy indexing: no geometry
	nreal = sum(1 for x in sel if x is not None and x is not Ellipsis)
	full = []
	for x in sel:
		if x is Ellipsis: full += [slice(None)]*(nd-nreal)
		elif x is not None: full.append(x)
	full += [slice(None)]*(nd-len(full))
	ysel, xsel = full[nd-2], full[nd-1]
	if not (isinstance(ysel, slice) and isinstance(xsel, slice)):
		return np.asarray(res) if isinstance(parent, ndmap) else res          # a pixel axis was indexed away
	if ysel == slice(None) and xsel == slice(None): wcs = parent.wcs
	else: _, wcs = wcsutils.slice_geometry(parent.shape, parent.wcs, (ysel, xsel))
	return dmap(res, wcs) if isinstance(parent, dmap) else ndmap(res, wcs)

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

# rows a full-sky CAR grid spends ON the poles: Clenshaw-Curtis has a ring on each pole (ny - 1 intervals span pi),
# Fejer-1 keeps half a pixel away from both (ny intervals)
_POLE_ROWS = {"cc": 1, "fejer1": 0}

def fullsky_geometry(res=None, shape=None, dims=(), proj="car", variant="fejer1"):
	"""(shape, wcs) of a CAR map covering the whole sky, from a resolution in radians or a pixel shape (ny, nx): the
	geometry contract of enmap.fullsky_geometry (enmap.py:1713-1740).  Columns run east to west (cdelt_ra < 0) with pixel
	centres half a pixel off RA = 0; rows run south to north with the equator on the row lattice."""
	assert proj == "car", "Only CAR fullsky geometry implemented"
	key = str(variant).lower()
	if key not in _POLE_ROWS: raise ValueError("Unrecognized CAR variant '%s'" % str(variant))
	extra = _POLE_ROWS[key]
	if shape is None:
		dy, dx = np.broadcast_to(np.asarray(res, float), (2,)) if np.ndim(res) else (float(res), float(res))
		ny, nx = int(round(np.pi/dy))+extra, int(round(2*np.pi/dx))
	else:
		ny, nx = int(shape[-2]), int(shape[-1])
		dy, dx = np.pi/(ny-extra), 2*np.pi/nx
	assert abs(dy*(ny-extra)-np.pi) < 1e-8, "Vertical resolution does not evenly divide the sky; this is required for SHTs."
	assert abs(dx*nx-2*np.pi) < 1e-8, "Horizontal resolution does not evenly divide the sky; this is required for SHTs."
	wcs = CarWCS(cdelt=[-360.0/nx, 180.0/(ny-extra)], crval=[0.5*dx/degree, 0.0], crpix=[nx//2+0.5, 0.5*(ny+1)])
	return tuple(dims)+(ny, nx), wcs

def band_geometry(dec_cut, res=None, shape=None, dims=(), proj="car", variant="fejer1"):
	"""the rows of the full-sky geometry whose centres lie within |dec| <= dec_cut (or within [dec_cut[0], dec_cut[1]]), all
	columns kept (enmap.band_geometry, enmap.py:1742-1772)"""
	cut = np.atleast_1d(np.asarray(dec_cut, float))
	lo, hi = (-cut[0], cut[0]) if cut.size == 1 else (cut[0], cut[1])
	fshape, fwcs = fullsky_geometry(res=res, shape=shape, dims=dims, proj=proj, variant=variant)
	rows = [float(wcsutils.world2pix(fwcs, 0, d/degree)[1]) for d in (lo, hi)]
	first, last = max(int(np.round(min(rows))), 0), min(int(np.round(max(rows))), fshape[-2])
	w = fwcs.deepcopy(); w.wcs.crpix[1] -= first
	return tuple(dims)+(last-first, fshape[-1]), w

def spin_helper(spin, n):
	"""Walk the n components of a map / alm stack in spin groups: yields (spin, first, last+1); spin 0 takes one component,
	any other spin a pair; the spin list is cycled (enmap.spin_helper, enmap.py:3378-3388).  IndexError if a pair is cut."""
	import itertools
	first = 0
	for s in itertools.cycle(np.atleast_1d(spin).reshape(-1)):
		if first >= n: return
		width = 1 if s == 0 else 2
		if first+width > n: raise IndexError("Unpaired component in spin transform")
		yield s, first, first+width
		first += width

def pix2sky(shape, wcs, pix):
	"""[{y,x},...] -> [{dec,ra},...] in radians (enmap.pix2sky, enmap.py:483-494, linear CAR)"""
	pix = np.asarray(pix, float)
	ra, dec = wcsutils.pix2world(wcs, pix[1], pix[0])
	return np.array([dec*degree, ra*degree])

def area(shape, wcs):
	"""solid angle of a separable cylindrical map: (sin dec_hi - sin dec_lo) * RA range (enmap.area_cyl, enmap.py:1032-1036)"""
	if not wcsutils.is_separable(wcs): raise NotImplementedError("area: only separable cylindrical geometries")
	lo, hi, _ = _dec_span(shape, wcs)
	return (np.sin(hi)-np.sin(lo))*abs(wcs.wcs.cdelt[0])*degree*shape[-1]
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
def _dec_span(shape, wcs):
	"""declinations of the lower and upper map edge (pixel edges, not centres), clipped to the sphere, and the row direction"""
	edges = pix2sky(shape, wcs, [[-0.5, shape[-2]-0.5], [0, 0]])[0]
	sign = 1 if edges[0] <= edges[1] else -1
	lo, hi = np.clip(np.sort(edges), -np.pi/2, np.pi/2)
	return lo, hi, sign

def extent(shape, wcs, signed=False, method="auto"):
	"""[height, width] of the patch in radians (enmap.extent, enmap.py:917-1014).  "cylindrical" (separable geometries): the
	width is the RA range times the mean of cos(dec) over the patch, so that area = height * width holds exactly;
	"intermediate": pixel counts times the WCS increments."""
	if method == "auto": method = "cylindrical" if wcsutils.is_separable(wcs) else "intermediate"
	if method in ("inter", "intermediate"):
		ext = np.array([wcs.wcs.cdelt[1]*shape[-2], wcs.wcs.cdelt[0]*shape[-1]], float)*degree
	elif method in ("cyl", "cylindrical"):
		lo, hi, sign = _dec_span(shape, wcs)
		ext = np.array([sign*(hi-lo), shape[-1]*wcs.wcs.cdelt[0]*degree*(np.sin(hi)-np.sin(lo))/(hi-lo)])
	else: raise NotImplementedError("extent: only the cylindrical and intermediate methods")
	return ext if signed else np.abs(ext)

def laxes(shape, wcs, oversample=1, method="auto", broadcastable=False):
	"""the multipoles (ly[ny], lx[nx]) of the bins of the map's 2-D FFT: 2 pi times the FFT frequencies for the pixel pitch
	extent/shape (enmap.laxes, enmap.py:1275-1294); oversample > 1 describes the finer lattice of a zero-padded transform"""
	os_ = int(oversample)
	pitch = extent(shape, wcs, signed=True, method=method)/np.array(shape[-2:], float)
	axes = [2*np.pi*np.fft.fftfreq(n*os_, d) for n, d in zip(shape[-2:], pitch)]
	if os_ > 1: axes = [l+0.5*l[os_]*(1.0/os_-1) for l in axes]
	ly, lx = axes
	return (ly[:, None], lx[None, :]) if broadcastable else (ly, lx)

def lmap(shape, wcs, oversample=1, method="auto"):
	"""[{ly,lx},ny,nx] multipole of every 2-D FFT bin"""
	ly, lx = laxes(shape, wcs, oversample=oversample, method=method, broadcastable=True)
	return ndmap(np.stack(np.broadcast_arrays(ly, lx)).astype(float), wcs)

def modlmap(shape, wcs, oversample=1, method="auto", min=0):
	"""|l| of every 2-D FFT bin (floored at `min`)"""
	ly, lx = laxes(shape, wcs, oversample=oversample, method=method, broadcastable=True)
	l = np.hypot(ly, lx)
	return ndmap(np.maximum(l, min) if min > 0 else l, wcs)

def lpixshape(shape, wcs, signed=False, method="auto"): return 2*np.pi/extent(shape, wcs, signed=signed, method=method)
def lpixsize(shape, wcs, signed=False, method="auto"): return float(np.prod(lpixshape(shape, wcs, signed=signed, method=method)))

def queb_rotmat(lmap, inverse=False, iau=False, spin=2, wcs=None):
	"""host version of the [2,2,ny,nx] rotation between (Q,U) and (E,B) in flat-sky harmonic space: angle = spin * atan2(+-lx, ly),
	sign flipped by `iau` and by `inverse` (enmap.queb_rotmat, enmap.py:1391-1400).  The transforms below never build it: they rotate
	in place on the GPU (pxm_rotate_queb)."""
	sign = (-1 if iau else 1)*(-1 if inverse else 1)
	ang = spin*np.arctan2(sign*np.asarray(lmap[1]), np.asarray(lmap[0]))
	c, s_ = np.cos(ang), np.sin(ang)
	return samewcs(np.array([[c, -s_], [s_, c]]), lmap)

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
def _geo_key(shape, wcs):
	"""what the l axes of a map depend on: its pixel shape and the WCS numbers (None: no usable key, do not cache)"""
	try: return (tuple(int(v) for v in shape[-2:]), tuple(np.asarray(wcs.wcs.cdelt, float)), tuple(np.asarray(wcs.wcs.crval, float)), tuple(np.asarray(wcs.wcs.crpix, float)), tuple(wcs.wcs.ctype))
	except Exception: return None
_axes_cache = {}
_cache_lock = threading.Lock()      # (the geometry caches below are shared between threads)
def _dev_axes(shape, wcs, like):
	"""ly, lx as device f64 arrays living next to `like` (kept per geometry and device: a Monte-Carlo loop calls lbin / map2harm on the
	same geometry every realisation, and the upload from pageable memory is a host synchronisation each time)"""
	from . import sht
	if sht._lib.is_hostsim():
		ly, lx = laxes(shape, wcs)
		return np.ascontiguousarray(ly), np.ascontiguousarray(lx)
	key = _geo_key(shape, wcs)
	if key is not None: key = key+(str(like.device),)
	with _cache_lock:      # lookup, fill and eviction as one step; the cached tensors are shared and must not be modified by callers
		hit = _axes_cache.get(key) if key is not None else None
		if hit is None:
			torch = _torch()
			ly, lx = laxes(shape, wcs)
			hit = (torch.from_numpy(np.ascontiguousarray(ly)).to(like.device), torch.from_numpy(np.ascontiguousarray(lx)).to(like.device))
			torch.cuda.current_stream().synchronize()      # (complete before another thread's stream may read them)
			if key is not None:
				if len(_axes_cache) >= 8: _axes_cache.pop(next(iter(_axes_cache)))
				_axes_cache[key] = hit
	return hit

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

_lbin_cache = {}
def lbin(map, bsize=None, brel=1.0, return_nhit=False, return_bins=False, lop=None):
	"""radial binning of a real fourier-space map in |l| (enmap.lbin / _bin_helper, enmap.py:2526-2556): returns b(l), l"""
	from . import sht
	if lop is not None:
		# a transform of |l| (e.g. a logarithm): the bin of a pixel is no longer floor(|l| / bsize) of the kernel above but a table of the geometry,
		# made here as the reference makes it (the default bin width comes from the TRANSFORMED |l|, enmap.py:2528-2530)
		l = np.asarray(lop(np.asarray(modlmap(map.shape, map.wcs))))
		if bsize is None: bsize = min(abs(l[0, 1]), abs(l[1, 0]))
		return _bin_helper(map, l, bsize*brel, return_nhit=return_nhit, return_bins=return_bins)
	gkey = _geo_key(map.shape, map.wcs)
	with _cache_lock:
		geo = _lbin_cache.get((gkey, bsize, brel)) if gkey is not None else None
		if geo is None:
			ly, lx = laxes(map.shape, map.wcs)
			bs = min(abs(lx[1]), abs(ly[1])) if bsize is None else bsize
			bs = float(bs*brel)
			lmax = float(np.sqrt(np.max(ly**2)+np.max(lx**2)))
			geo = dict(bsize=bs, n=int(lmax/bs), lsum=None, nhit=None)
			if gkey is not None:
				if len(_lbin_cache) >= 8: _lbin_cache.pop(next(iter(_lbin_cache)))
				_lbin_cache[(gkey, bsize, brel)] = geo
	bsize, n = geo["bsize"], geo["n"]
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
	# the sums of |l| and the pixel counts per bin depend on the geometry alone: taken with the first map of the first call on a geometry
	# and kept (two of the kernel's three atomic adds per pixel)
	for i in range(npre):
		first = i == 0 and geo["nhit"] is None
		sht._lib.check(lib.pxm_lbin(ny, nx, _ptr(dly), _ptr(dlx), bsize, n, _ptr(d)+i*ny*nx*esz, sht._DT[sht._np_dtype(d)],
			_ptr(acc)+i*max(n, 1)*8, (_ptr(acc)+npre*max(n, 1)*8) if first else None, (_ptr(acc)+(npre+1)*max(n, 1)*8) if first else None, dev, st))
	acc = acc.cpu().numpy() if hasattr(acc, "data_ptr") else acc
	if geo["nhit"] is None and npre > 0:
		with _cache_lock: geo["lsum"], geo["nhit"] = acc[npre, :n].copy(), acc[npre+1, :n].copy()      # (the geometry-only sums: set once, under the lock)
	nhit = geo["nhit"] if geo["nhit"] is not None else acc[npre+1, :n]
	lsum = geo["lsum"] if geo["lsum"] is not None else acc[npre, :n]
	with np.errstate(invalid="ignore", divide="ignore"):
		mout = (acc[:npre, :n]/nhit).reshape(tuple(d.shape[:-2])+(n,))
		orads = lsum/nhit
	if return_bins:
		edges = np.arange(len(orads)+1)*bsize
		orads = np.array([orads, edges[:-1], edges[1:]])
	if return_nhit: return mout, orads, nhit.astype(int)
	return mout, orads

def _bin_helper(map, r, bsize, return_nhit=False, return_bins=False):
	"""means of the map over the bins floor(r / bsize) of a per-pixel coordinate r[ny,nx] (enmap._bin_helper, enmap.py:2533-2556): the bin table,
	the pixel counts and the mean coordinate of every bin are functions of the geometry (host); the sums over the map run on the GPU (pxm_bin_index)"""
	from . import sht
	r = np.asarray(r, float)
	n = int(np.max(r/bsize))
	rinds = np.floor(r/bsize).reshape(-1).astype(np.int64)
	nhit = np.bincount(rinds, minlength=n)[:n]
	with np.errstate(invalid="ignore", divide="ignore"): orads = np.bincount(rinds, weights=r.reshape(-1), minlength=n)[:n]/nhit
	dev_map, was_host = _to_device(map)
	d = _data(dev_map)
	if sht._np_dtype(d) not in (np.dtype(np.float32), np.dtype(np.float64)): raise ValueError("binning needs a real map")
	if hasattr(d, "contiguous"): d = d.contiguous()
	ny, nx = d.shape[-2:]; npre = int(np.prod(d.shape[:-2], dtype=int))
	if rinds.size != ny*nx: raise ValueError("binning: the coordinate map does not have the map's pixel shape")
	tab = np.where((rinds >= 0) & (rinds < n), rinds, -1).astype(np.int32)
	if hasattr(d, "data_ptr"):
		torch = _torch()
		dtab = torch.as_tensor(tab, device=d.device); acc = torch.zeros((npre, max(n, 1)), dtype=torch.float64, device=d.device)
	else: dtab = tab; acc = np.zeros((npre, max(n, 1)))
	lib = sht._lib.load(); dev = sht.device_index(); st = sht.current_stream()
	esz = sht._np_dtype(d).itemsize
	for i in range(npre):
		sht._lib.check(lib.pxm_bin_index(ny*nx, _ptr(dtab), n, _ptr(d)+i*ny*nx*esz, sht._DT[sht._np_dtype(d)], _ptr(acc)+i*max(n, 1)*8, dev, st))
	acc = acc.cpu().numpy() if hasattr(acc, "data_ptr") else acc
	with np.errstate(invalid="ignore", divide="ignore"): mout = (acc[:, :n]/nhit).reshape(tuple(d.shape[:-2])+(n,))
	if return_bins:
		edges = np.arange(len(orads)+1)*bsize
		orads = np.array([orads, edges[:-1], edges[1:]])
	if return_nhit: return mout, orads, nhit
	return mout, orads

def posaxes(shape, wcs):
	"""(dec[ny], ra[nx]) of the pixel centres of a separable geometry, radians (enmap.posaxes)"""
	if not wcsutils.is_separable(wcs): raise NotImplementedError("posaxes: only separable cylindrical geometries")
	dec = pix2sky(shape, wcs, [np.arange(shape[-2]), np.zeros(shape[-2])])[0]
	ra  = pix2sky(shape, wcs, [np.zeros(shape[-1]), np.arange(shape[-1])])[1]
	return dec, ra

def center(shape, wcs):
	"""[dec, ra] of the middle of the pixel grid (enmap.center, enmap.py:1254-1256)"""
	return pix2sky(shape, wcs, (np.array(shape[-2:])-1)/2.0)

def modrmap(shape, wcs, ref="center"):
	"""angular distance of every pixel centre from ref = [dec, ra] (radians; "center": the middle of the map) as a host ndmap: geometry only
	(enmap.modrmap, enmap.py:1263-1273, with utils.angdist's Vincenty form, stable at small and at large separations)"""
	if isinstance(ref, str):
		if ref != "center": raise ValueError(ref)
		ref = center(shape, wcs)
	ref = np.asarray(ref, float)
	dec, ra = posaxes(shape, wcs)
	dra = (ra-ref[1])[None, :]
	sd, cd = np.sin(dec)[:, None], np.cos(dec)[:, None]
	sr, cr = np.sin(ref[0]), np.cos(ref[0])
	y = np.hypot(cr*np.sin(dra), cd*sr-sd*cr*np.cos(dra))
	x = sd*sr+cd*cr*np.cos(dra)
	return ndmap(np.arctan2(y, x), wcs)

def shift(map, off, inplace=False, keepwcs=False):
	"""cyclic shift of the pixels: (i, j) -> (i + off[0], j + off[1]) (enmap.shift, enmap.py:3277-3290); unless keepwcs the reference pixel moves along"""
	off = np.atleast_1d(off).astype(int)
	axes = tuple(range(-len(off), 0))
	if isinstance(map, dmap):
		t = _torch().roll(map.tensor, tuple(int(o) for o in off), axes)
		if inplace: map.tensor.copy_(t); res = map
		else: res = dmap(t, map.wcs)
	else:
		t = np.roll(np.asarray(map), tuple(int(o) for o in off), axes)
		if inplace: map[...] = t; res = map
		else: res = ndmap(t, map.wcs)
	if not keepwcs:
		w = res.wcs.deepcopy(); w.wcs.crpix = np.array(w.wcs.crpix, float); w.wcs.crpix[:len(off)] += off[::-1]; res.wcs = w
	return res

def rbin(map, center=[0, 0], bsize=None, brel=1.0, return_nhit=False, return_bins=False, rop=None):
	"""mean of the map in rings of width bsize (default: the smaller pixel pitch) around center = [dec, ra]: b(r)[..., nbin], r[nbin]
	(enmap.rbin, enmap.py:2512-2524)"""
	r = np.asarray(modrmap(map.shape, map.wcs, ref=center))
	if rop: r = np.asarray(rop(r))
	if bsize is None: bsize = np.min(extent(map.shape, map.wcs)/np.array(map.shape[-2:]))
	return _bin_helper(map, r, bsize*brel, return_nhit=return_nhit, return_bins=return_bins)

for _cls in (ndmap, dmap):
	_cls.extent   = lambda self, **kw: extent(self.shape, self.wcs, **kw)
	_cls.modrmap  = lambda self, **kw: modrmap(self.shape, self.wcs, **kw)
	_cls.rbin     = lambda self, *a, **kw: rbin(self, *a, **kw)
	_cls.pix2sky  = lambda self, pix: pix2sky(self.shape, self.wcs, pix)
	_cls.laxes    = lambda self, **kw: laxes(self.shape, self.wcs, **kw)
	_cls.lmap     = lambda self, **kw: lmap(self.shape, self.wcs, **kw)
	_cls.modlmap  = lambda self, **kw: modlmap(self.shape, self.wcs, **kw)
	_cls.lbin     = lambda self, *a, **kw: lbin(self, *a, **kw)
