"""Drop-in for the spherical-harmonic-transform API of pixell/curvedsky.py on MI355X.

Same function names, argument meaning and error behaviour as the reference for the `2d` and
`cyl` methods (curvedsky.py:83-302, 756-1086, 1170-1446); the `general` method, healpix rings and
rotate_alm are outside the accelerated path and raise NotImplementedError.

Differences that matter for speed, not for results:
  * the flipped / padded copies of map2buffer / buffer2map (curvedsky.py:1384-1411) are not
    made: flips are negative strides inside the kernels (sht.*(flip=...));
  * `map` may be an enmap.ndmap (numpy; staged through the GPU) or an enmap.dmap wrapping a torch
    CUDA tensor (transformed in place, nothing leaves HBM); `alm` likewise numpy or tensor.
"""
import numpy as np
from . import enmap, sht, wcs as wcsutils
from .sht import _is_tensor, _np_dtype, _torch

class Bunch(dict):
	def __getattr__(self, k):
		try: return self[k]
		except KeyError: raise AttributeError(k)
	def __setattr__(self, k, v): self[k] = v

degree = np.pi/180

def hasoff(val, off, tol=1e-6): return np.abs((val-off+0.5) % 1-0.5) < tol
def nint(a): return np.round(a).astype(int)
def complex_dtype(dtype): return np.result_type(dtype, 0j)
def real_dtype(dtype): return np.zeros([0], dtype).real.dtype
def nditer(shape):
	for I in np.ndindex(*shape): yield tuple(I)

def nalm2lmax(nalm):
	return int((-1+(1+8*nalm)**0.5)/2)-1

class alm_info:
	"""alm layout (curvedsky.alm_info, curvedsky.py:409-476)"""
	def __init__(self, lmax=None, mmax=None, nalm=None, stride=1, layout="triangular"):
		if lmax is not None: lmax = int(lmax)
		if mmax is not None: mmax = int(mmax)
		if nalm is not None: nalm = int(nalm)
		if isinstance(layout, str):
			if layout == "triangular" or layout == "tri":
				if lmax is None: lmax = nalm2lmax(nalm)
				if mmax is None: mmax = lmax
				m = np.arange(mmax+1)
				mstart = stride*(m*(2*lmax+1-m)//2)
			elif layout == "rectangular" or layout == "rect":
				if lmax is None: lmax = int(nalm**0.5)-1
				if mmax is None: mmax = lmax
				mstart = np.arange(mmax+1)*(lmax+1)*stride
			else:
				raise ValueError("unkonwn layout: %s" % layout)
		else:
			mstart = np.asarray(layout)
		self.lmax  = lmax
		self.mmax  = mmax
		self.stride= int(stride)
		self.nelem = int(np.max(mstart) + (lmax+1)*stride)
		self.nreal = lmax**2+2*lmax+2
		if nalm is not None:
			assert self.nelem == nalm, "lmax must be explicitly specified when lmax != mmax"
		self.mstart= mstart.astype(np.uint64, copy=False)
	@property
	def nl(self): return self.lmax+1
	@property
	def nm(self): return self.mmax+1
	def lm2ind(self, l, m):
		return (self.mstart[m].astype(int, copy=False)+l*self.stride).astype(int, copy=False)
	def get_map(self):
		raise NotImplementedError
	def alm2cl(self, alm, alm2=None, dtype=None):
		"""cross power spectrum of alm and alm2, which broadcast (curvedsky.py:451-462); e.g.
		cl[{T,E,B},{T,E,B},nl] = ainfo.alm2cl(alm[:,None,:], alm[None,:,:])"""
		from . import almops
		return almops.alm2cl(self, alm, alm2=alm2, cl_dtype=dtype)
	def lmul(self, alm, lmat, out=None):
		"""res[a,lm] = lmat[a,b,l]*alm[b,lm] (or the broadcasting product for lmat[...,l]) (curvedsky.py:463-466)"""
		from . import almops
		return almops.lmul(self, alm, lmat, out=out)
	def __repr__(self):
		return "alm_info(lmax=%s,mmax=%s,mstart=%s)" % (str(self.lmax), str(self.mmax), str(self.mstart))

# ---------------------------------------------------------------------------------------
# geometry analysis (curvedsky.py:1252-1353)
# ---------------------------------------------------------------------------------------
def get_ducc_maxlmax(name, ny):
	if   name == "CC": return ny-2
	elif name == "DH": return (ny-2)//2
	elif name == "F2": return (ny-1)//2
	else:              return ny-1

def get_ducc_geo(wcs, shape=None, tol=1e-6):
	"""curvedsky.get_ducc_geo (curvedsky.py:1308-1347): which named ducc grid the (flipped) geometry is"""
	def near(a, b): return np.abs(a-b) < tol
	flip = [wcs.wcs.cdelt[1] > 0, wcs.wcs.cdelt[0] < 0]
	_, w = wcsutils.flipped(shape or (1, 1), wcs, flip)
	nx = 360/w.wcs.cdelt[0]
	if not hasoff(nx, 0, tol): return None
	phi0 = wcsutils.pix2world(w, 0, 0)[0]*degree
	y1 = wcsutils.world2pix(w, 0,  90)[1]
	y2 = wcsutils.world2pix(w, 0, -90)[1]
	Ny = shape[-2] if shape is not None else nint(y2)+1
	if   hasoff(y1, 0.0, tol) and hasoff(y2, 0.0, tol):
		if   near(y1, -1) and near(y2, Ny): name, o1, o2 = "F2", 1, 1
		elif near(y1,  0) and near(y2, Ny): name, o1, o2 = "DH", 1, 0
		else: name, o1, o2 = "CC", 0, 0
	elif hasoff(y1, 0.5, tol) and hasoff(y2, 0.5, tol): name, o1, o2 = "F1", 0.5, 0.5
	elif hasoff(y1, 0.5, tol) and hasoff(y2, 0.0, tol): name, o1, o2 = "MW", 0.5, 0.0
	elif hasoff(y1, 0.0, tol) and hasoff(y2, 0.5, tol): name, o1, o2 = "MWflip", 0.0, 0.5
	else: return None
	ny   = nint(y2-y1+1-o1-o2)
	yoff = nint(-y1-o1)
	lmax = get_ducc_maxlmax(name, ny)
	return Bunch(name=name, nx=nint(nx), ny=ny, pole_offs=[o1, o2], phi0=phi0, yoff=yoff, lmax=lmax)

def analyse_geometry(shape, wcs, tol=1e-6):
	"""curvedsky.analyse_geometry (curvedsky.py:1252-1306)"""
	separable = wcsutils.is_separable(wcs)
	divides   = hasoff(360/np.abs(wcs.wcs.cdelt[0]), 0, tol=tol)
	if not separable or not divides:
		return Bunch(case="general", flip=[False, False], ducc_geo=None, ypad=(0, 0), xpad=(0, 0), phi0=0)
	flip = [bool(wcs.wcs.cdelt[1] > 0), bool(wcs.wcs.cdelt[0] < 0)]
	wshape, wwcs = wcsutils.flipped(shape, wcs, flip)
	phi0 = wcsutils.pix2world(wwcs, 0, wshape[-2]//2)[0]*degree
	ducc_geo = get_ducc_geo(wwcs, shape=wshape, tol=tol)
	if ducc_geo is not None and shape[-2] == ducc_geo.ny and shape[-1] == ducc_geo.nx and np.abs(ducc_geo.yoff) < tol:
		return Bunch(case="2d", flip=flip, ducc_geo=ducc_geo, ypad=(0, 0), xpad=(0, 0), phi0=phi0)
	else:
		if ducc_geo is not None: ypad = (ducc_geo.yoff, ducc_geo.ny-ducc_geo.yoff-shape[-2])
		else: ypad = (0, 0)
		nx = nint(360/wwcs.wcs.cdelt[0])
		if shape[-1] == nx:
			return Bunch(case="cyl", flip=flip, ducc_geo=ducc_geo, ypad=ypad, xpad=(0, 0), phi0=phi0)
		else:
			return Bunch(case="partial", flip=flip, ducc_geo=ducc_geo, ypad=ypad, xpad=(0, nx-shape[-1]), phi0=phi0)

def get_method(shape, wcs, minfo=None, pix_tol=1e-6):
	if minfo is None: minfo = analyse_geometry(shape, wcs, tol=pix_tol)
	if   minfo.case == "general": return "general"
	elif minfo.case == "2d":      return "2d"
	else:                         return "cyl"

def get_ring_info(shape, wcs, dtype=np.float64):
	"""curvedsky.get_ring_info (curvedsky.py:1170-1190)"""
	y = np.arange(shape[-2])
	dec, ra = enmap.pix2sky(shape, wcs, [y, y*0])
	theta = np.asarray(np.pi/2-dec, dtype=dtype)
	ntheta = len(theta)
	nphi = np.zeros(ntheta, dtype=np.uint64)+shape[-1]
	phi0 = np.asarray(ra, dtype=dtype)
	offsets = (np.arange(ntheta)*shape[-1]).astype(np.uint64)
	stride = np.zeros(ntheta, dtype=np.int32)+1
	return Bunch(theta=theta, nphi=nphi, phi0=phi0, offsets=offsets, stride=stride, npix=int(np.sum(nphi)), nrow=ntheta)

def quad_weights(shape, wcs, pix_tol=1e-6):
	"""curvedsky.quad_weights (curvedsky.py:492-505)"""
	minfo = analyse_geometry(shape, wcs, tol=pix_tol)
	if minfo.ducc_geo is None or minfo.ducc_geo.name is None:
		raise ValueError("Quadrature weights not available for geometry %s,%s" % (str(shape), str(wcs)))
	ny = shape[-2]+int(np.sum(minfo.ypad))
	weights = sht.get_gridweights(minfo.ducc_geo.name, ny)
	weights = weights[minfo.ypad[0]:len(weights)-minfo.ypad[1]]
	if minfo.flip[0]: weights = weights[::-1]
	return weights/minfo.ducc_geo.nx

# ---------------------------------------------------------------------------------------
# array plumbing
# ---------------------------------------------------------------------------------------
def _mdata(map):
	return map.tensor if isinstance(map, enmap.dmap) else np.asarray(map)

def _zeros_like_kind(shape, dtype, like):
	if _is_tensor(like):
		torch = _torch()
		return torch.zeros(tuple(shape), dtype=getattr(torch, np.dtype(dtype).name), device=like.device)
	return np.zeros(shape, dtype)

def prepare_alm(alm=None, ainfo=None, lmax=None, pre=(), dtype=np.float64, convert=False, like=None):
	"""curvedsky.prepare_alm (curvedsky.py:1413-1427)"""
	ctype = complex_dtype(dtype)
	if alm is None:
		if ainfo is None:
			if lmax is None:
				raise ValueError("prepare_alm needs either alm, ainfo or lmax to be specified")
			ainfo = alm_info(lmax)
		alm = _zeros_like_kind(tuple(pre)+(ainfo.nelem,), ctype, like)
	if ainfo is None:
		ainfo = alm_info(nalm=alm.shape[-1])
	if not convert and _np_dtype(alm) != ctype:
		raise ValueError("alm had dtype '%s', but expected '%s'" % (str(_np_dtype(alm)), str(ctype)))
	if _np_dtype(alm) != ctype:
		alm = alm.to(getattr(_torch(), np.dtype(ctype).name)) if _is_tensor(alm) else alm.astype(ctype)
	return alm, ainfo

def _atleast(x, n):
	while x.ndim < n: x = x[None]
	return x

def _contig(x):
	if _is_tensor(x): return x if x.is_contiguous() else x.contiguous()
	return x

# ---------------------------------------------------------------------------------------
# public API
# ---------------------------------------------------------------------------------------
def alm2map(alm, map, spin=[0, 2], deriv=False, adjoint=False, copy=False, method="auto", ainfo=None,
		verbose=False, nthread=None, epsilon=1e-6, pix_tol=1e-6, locinfo=None, tweak=False):
	"""Spherical harmonics synthesis (curvedsky.alm2map, curvedsky.py:83-164).  See the reference
	docstring for the argument meaning; `map` is overwritten unless copy=True and returned."""
	minfo = analyse_geometry(map.shape, map.wcs, tol=pix_tol)
	if method == "auto": method = get_method(map.shape, map.wcs, minfo=minfo)
	if   method == "2d":
		return alm2map_2d(alm, map, ainfo=ainfo, minfo=minfo, spin=spin, deriv=deriv, copy=copy, verbose=verbose, adjoint=adjoint, nthread=nthread, pix_tol=pix_tol)
	elif method == "cyl":
		return alm2map_cyl(alm, map, ainfo=ainfo, minfo=minfo, spin=spin, deriv=deriv, copy=copy, verbose=verbose, adjoint=adjoint, nthread=nthread, pix_tol=pix_tol)
	elif method == "general":
		raise NotImplementedError("method 'general' (non-cylindrical pixelisations, ducc synthesis_general) is outside the accelerated path")
	else:
		raise ValueError("Unrecognized alm2map method '%s'" % str(method))

def alm2map_adjoint(map, alm=None, spin=[0, 2], deriv=False, copy=False, method="auto", ainfo=None, verbose=False, nthread=None, epsilon=None, pix_tol=1e-6, locinfo=None):
	return alm2map(alm, map, spin=spin, deriv=deriv, adjoint=True, copy=copy, method=method, ainfo=ainfo, verbose=verbose, nthread=nthread, epsilon=epsilon, pix_tol=pix_tol, locinfo=locinfo)

def map2alm(map, alm=None, lmax=None, spin=[0, 2], deriv=False, adjoint=False, copy=False, method="auto", ainfo=None,
		verbose=False, nthread=None, niter=0, epsilon=None, pix_tol=1e-6, weights=None, locinfo=None, tweak=False):
	"""Spherical harmonics analysis (curvedsky.map2alm, curvedsky.py:209-302)."""
	minfo = analyse_geometry(map.shape, map.wcs, tol=pix_tol)
	if method == "auto": method = get_method(map.shape, map.wcs, minfo=minfo)
	if   method == "2d":
		return map2alm_2d(map, alm, ainfo=ainfo, minfo=minfo, lmax=lmax, spin=spin, deriv=deriv, copy=copy, verbose=verbose, adjoint=adjoint, nthread=nthread, pix_tol=pix_tol)
	elif method == "cyl":
		return map2alm_cyl(map, alm, ainfo=ainfo, minfo=minfo, lmax=lmax, spin=spin, deriv=deriv, copy=copy, verbose=verbose, adjoint=adjoint, nthread=nthread, niter=niter, pix_tol=pix_tol, weights=weights)
	elif method == "general":
		raise NotImplementedError("method 'general' (non-cylindrical pixelisations, ducc synthesis_general) is outside the accelerated path")
	else:
		raise ValueError("Unrecognized alm2map method '%s'" % str(method))

def map2alm_adjoint(alm, map, lmax=None, spin=[0, 2], deriv=False, copy=False, method="auto", ainfo=None, verbose=False, nthread=None, niter=0, epsilon=1e-6, pix_tol=1e-6, weights=None, locinfo=None):
	return map2alm(map=map, alm=alm, lmax=lmax, spin=spin, deriv=deriv, adjoint=True, copy=copy, method=method, ainfo=ainfo, verbose=verbose, nthread=nthread, niter=niter, epsilon=epsilon, pix_tol=pix_tol, weights=weights, locinfo=locinfo)

# ---- 2d ---------------------------------------------------------------------------------
def _check_shapes(alm_full, map_full, deriv):
	if deriv:
		assert map_full.ndim >= 3 and map_full.shape[-3] == 2, "map must have shape [...,2,ny,nx] when deriv is True"
		assert tuple(map_full.shape[:-3]) == tuple(alm_full.shape[:-1]), "map and alm must agree on pre-dimensions"
	else:
		assert tuple(map_full.shape[:-2]) == tuple(alm_full.shape[:-1]), "map and alm must agree on pre-dimensions"

def _native_pads(minfo, use_y):
	"""((y_before, y_after), (x_before, x_after)) in the map's own pixel order.  minfo.ypad/xpad are in
	ducc orientation, i.e. after the flips of map2buffer (curvedsky.py:1384-1403)."""
	yp = tuple(int(v) for v in minfo.ypad) if use_y else (0, 0)
	xp = tuple(int(v) for v in minfo.xpad)
	if min(yp+xp) < 0: raise ValueError("map extends beyond the matching full-sky grid")
	if minfo.flip[0]: yp = yp[::-1]
	if minfo.flip[1]: xp = xp[::-1]
	return yp, xp

def _padded_like(map, pads, fill):
	"""zero-padded copy of the map (ndmap, or dmap on the GPU) with the wcs moved accordingly; this is the
	buffer map2buffer builds in the reference (curvedsky.py:1384-1403), without the flipped copy"""
	(y0, y1), (x0, x1) = pads
	mdata = _mdata(map)
	shape = tuple(map.shape[:-2])+(map.shape[-2]+y0+y1, map.shape[-1]+x0+x1)
	w = map.wcs.deepcopy(); w.wcs.crpix[0] += x0; w.wcs.crpix[1] += y0
	buf = _zeros_like_kind(shape, _np_dtype(mdata), mdata)
	if fill: buf[..., y0:y0+map.shape[-2], x0:x0+map.shape[-1]] = mdata
	return (enmap.dmap(buf, w) if isinstance(map, enmap.dmap) else enmap.ndmap(buf, w))

def _crop_into(map, pmap, pads):
	(y0, y1), (x0, x1) = pads
	_mdata(map)[...] = _mdata(pmap)[..., y0:y0+map.shape[-2], x0:x0+map.shape[-1]]

def alm2map_2d(alm, map, ainfo=None, minfo=None, spin=[0, 2], deriv=False, copy=False, verbose=False, adjoint=False, nthread=None, pix_tol=1e-6):
	"""curvedsky.alm2map_2d + alm2map_raw_2d (curvedsky.py:756-774, 900-926)"""
	if copy:
		if adjoint and alm is not None: alm = alm.clone() if _is_tensor(alm) else alm.copy()
		elif not adjoint: map = map.copy()
	mdata = _mdata(map)
	if adjoint: alm, ainfo = prepare_alm(alm=alm, ainfo=ainfo, pre=map.shape[:-2], dtype=_np_dtype(mdata), convert=False, like=mdata)
	else:       alm, ainfo = prepare_alm(alm=alm, ainfo=ainfo, pre=map.shape[:-2], dtype=_np_dtype(mdata), convert=True, like=mdata)
	if minfo is None: minfo = analyse_geometry(map.shape, map.wcs, tol=pix_tol)
	pads = _native_pads(minfo, use_y=True)
	if pads != ((0, 0), (0, 0)):
		# pad to the full grid as the reference does (curvedsky.py:766-772); the padded geometry is case "2d"
		pmap = _padded_like(map, pads, fill=adjoint)
		res = alm2map_2d(alm, pmap, ainfo=ainfo, spin=spin, deriv=deriv, copy=False, verbose=verbose, adjoint=adjoint, nthread=nthread, pix_tol=pix_tol)
		if adjoint: return res
		_crop_into(map, pmap, pads)
		return map
	alm_full = _atleast(alm, 2 if deriv else 3)
	map_full = _atleast(mdata, 4)
	_check_shapes(alm_full, map_full, deriv)
	func = sht.adjoint_synthesis_2d if adjoint else sht.synthesis_2d
	kwargs = dict(phi0=minfo.phi0, lmax=ainfo.lmax, mmax=ainfo.mmax, geometry=minfo.ducc_geo.name, mstart=ainfo.mstart, lstride=ainfo.stride, flip=minfo.flip)
	for I in nditer(map_full.shape[:-3]):
		if deriv:
			a = _contig(alm_full[I][None]); m = map_full[I]
			func(alm=a, map=m, mode="DERIV1", spin=1, **kwargs)
			if adjoint: alm_full[I] = a[0]
			else: map_full[I+(0,)] *= -1       # theta derivative -> dec derivative (curvedsky.py:919)
		else:
			for s, j1, j2 in enmap.spin_helper(spin, alm_full.shape[-2]):
				Ij = I+(slice(j1, j2),)
				v = alm_full[Ij]; a = _contig(v)
				func(alm=a, map=map_full[Ij], spin=int(s), **kwargs)
				if adjoint and a is not v: v[...] = a           # (only when the view was not contiguous)
	if adjoint: return alm
	else:       return map

def map2alm_2d(map, alm=None, ainfo=None, minfo=None, lmax=None, spin=[0, 2], deriv=False, copy=False, verbose=False, adjoint=False, nthread=None, pix_tol=1e-6):
	"""curvedsky.map2alm_2d + map2alm_raw_2d (curvedsky.py:822-841, 1018-1048)"""
	if adjoint:
		if copy and map is not None: map = map.copy()
	else:
		if copy and alm is not None: alm = alm.clone() if _is_tensor(alm) else alm.copy()
	mdata = _mdata(map)
	alm, ainfo = prepare_alm(alm=alm, ainfo=ainfo, lmax=lmax, pre=map.shape[:-2], dtype=_np_dtype(mdata), convert=adjoint, like=mdata)
	if minfo is None: minfo = analyse_geometry(map.shape, map.wcs, tol=pix_tol)
	if deriv:
		raise NotImplementedError("ducc does not support derivatives for map2alm operations. Can be worked around if necessary.")
	pads = _native_pads(minfo, use_y=True)
	if pads != ((0, 0), (0, 0)):
		pmap = _padded_like(map, pads, fill=not adjoint)
		res = map2alm_2d(pmap, alm=alm, ainfo=ainfo, lmax=lmax, spin=spin, deriv=deriv, copy=False, verbose=verbose, adjoint=adjoint, nthread=nthread, pix_tol=pix_tol)
		if not adjoint: return res
		_crop_into(map, pmap, pads)
		return map
	alm_full = _atleast(alm, 3)
	map_full = _atleast(mdata, 4)
	_check_shapes(alm_full, map_full, False)
	# Restrict to lmax and mmax that the grid allows. Higher ones are ignored (curvedsky.py:1027-1028)
	l = min(ainfo.lmax, minfo.ducc_geo.lmax)
	m = min(ainfo.mmax, l)
	func = sht.adjoint_analysis_2d if adjoint else sht.analysis_2d
	kwargs = dict(phi0=minfo.phi0, lmax=l, mmax=m, geometry=minfo.ducc_geo.name, mstart=ainfo.mstart[:m+1], lstride=ainfo.stride, flip=minfo.flip)
	for I in nditer(map_full.shape[:-3]):
		for s, j1, j2 in enmap.spin_helper(spin, alm_full.shape[-2]):
			Ij = I+(slice(j1, j2),)
			v = alm_full[Ij]; a = _contig(v)
			func(alm=a, map=map_full[Ij], spin=int(s), **kwargs)
			if not adjoint and a is not v: v[...] = a
	if adjoint: return map
	else:       return alm

# ---- cyl --------------------------------------------------------------------------------
def _ring_kwargs(map, minfo, ainfo):
	"""ring tables of the map in ducc orientation, with the flips of map2buffer expressed as
	a descending ringstart / negative pixel stride (no copy)."""
	assert not np.any(np.array(minfo.xpad) != 0), "partial-width maps are padded by the callers"
	shape = map.shape; ny, nx = shape[-2:]
	fshape, fwcs = wcsutils.flipped(shape, map.wcs, minfo.flip)
	rinfo = get_ring_info(fshape, fwcs)
	rows = np.arange(ny)[::-1] if minfo.flip[0] else np.arange(ny)
	start = rows*nx + (nx-1 if minfo.flip[1] else 0)
	return dict(theta=rinfo.theta, nphi=rinfo.nphi, phi0=rinfo.phi0, ringstart=start.astype(np.uint64),
		pixstride=-1 if minfo.flip[1] else 1, lmax=ainfo.lmax, mmax=ainfo.mmax, mstart=ainfo.mstart, lstride=ainfo.stride)

def _flat(m):
	return m.reshape(m.shape[:-2]+(m.shape[-2]*m.shape[-1],))

def alm2map_cyl(alm, map, ainfo=None, minfo=None, spin=[0, 2], deriv=False, copy=False, verbose=False, adjoint=False, nthread=None, pix_tol=1e-6):
	"""curvedsky.alm2map_cyl + alm2map_raw_cyl (curvedsky.py:776-794, 928-962)"""
	if copy:
		if adjoint and alm is not None: alm = alm.clone() if _is_tensor(alm) else alm.copy()
		elif not adjoint: map = map.copy()
	mdata = _mdata(map)
	alm, ainfo = prepare_alm(alm=alm, ainfo=ainfo, pre=map.shape[:-2], dtype=_np_dtype(mdata), convert=not adjoint, like=mdata)
	if minfo is None: minfo = analyse_geometry(map.shape, map.wcs, tol=pix_tol)
	pads = _native_pads(minfo, use_y=False)
	if pads != ((0, 0), (0, 0)):
		# partial-width map: extend the rings to the full circle (curvedsky.py:786-792)
		pmap = _padded_like(map, pads, fill=adjoint)
		res = alm2map_cyl(alm, pmap, ainfo=ainfo, spin=spin, deriv=deriv, copy=False, verbose=verbose, adjoint=adjoint, nthread=nthread, pix_tol=pix_tol)
		if adjoint: return res
		_crop_into(map, pmap, pads)
		return map
	kwargs = _ring_kwargs(map, minfo, ainfo)
	alm_full = _atleast(alm, 2 if deriv else 3)
	map_full = _atleast(mdata, 4)
	_check_shapes(alm_full, map_full, deriv)
	func = sht.adjoint_synthesis if adjoint else sht.synthesis
	for I in nditer(map_full.shape[:-3]):
		if deriv:
			a = _contig(alm_full[I][None]); m = _flat(map_full[I])
			func(alm=a, map=m, mode="DERIV1", spin=1, **kwargs)
			if adjoint: alm_full[I] = a[0]
			else: map_full[I+(0,)] *= -1
		else:
			for s, j1, j2 in enmap.spin_helper(spin, alm_full.shape[-2]):
				Ij = I+(slice(j1, j2),)
				v = alm_full[Ij]; a = _contig(v)
				func(alm=a, map=_flat(map_full[Ij]), spin=int(s), **kwargs)
				if adjoint and a is not v: v[...] = a
	if adjoint: return alm
	else:       return map

def jacobi_inverse(forward, approx_backward, y, niter=0):
	"""curvedsky.jacobi_inverse (curvedsky.py:1122-1136)"""
	x = approx_backward(y)
	for i in range(niter):
		x -= approx_backward(forward(x)-y)
	return x

def map2alm_cyl(map, alm=None, ainfo=None, minfo=None, lmax=None, spin=[0, 2], weights=None, deriv=False, copy=False, verbose=False, adjoint=False, nthread=None, pix_tol=1e-6, niter=0):
	"""curvedsky.map2alm_cyl + map2alm_raw_cyl (curvedsky.py:843-873, 1050-1086): quadrature
	weights (exact grid weights where available) + Jacobi refinement around the HIP transforms."""
	if adjoint:
		if copy and map is not None: map = map.copy()
	else:
		if copy and alm is not None: alm = alm.clone() if _is_tensor(alm) else alm.copy()
	mdata = _mdata(map)
	# (with deriv the alm has no component axis; the reference allocates [2,nelem] here and then fails its shape check)
	alm, ainfo = prepare_alm(alm=alm, ainfo=ainfo, lmax=lmax, pre=map.shape[:-3] if deriv else map.shape[:-2], dtype=_np_dtype(mdata), convert=adjoint, like=mdata)
	if minfo is None: minfo = analyse_geometry(map.shape, map.wcs, tol=pix_tol)
	if weights is None:
		if minfo.ducc_geo is not None and minfo.ducc_geo.name in ("CC", "F1", "MW", "MWflip"):
			weights = quad_weights(map.shape, map.wcs, pix_tol=pix_tol)
		else:
			# pixel area of each row (enmap.pixsizemap separable; curvedsky.py:858-860)
			ny, nx = map.shape[-2:]
			dec = enmap.pix2sky(map.shape, map.wcs, [np.concatenate([np.arange(ny)-0.5, [ny-0.5]]), np.zeros(ny+1)])[0]
			dec = np.clip(dec, -np.pi/2, np.pi/2)
			weights = np.abs(np.sin(dec[1:])-np.sin(dec[:-1]))*abs(map.wcs.wcs.cdelt[0])*degree
	weights = np.asarray(weights, dtype=_np_dtype(mdata))
	pads = _native_pads(minfo, use_y=False)
	if pads != ((0, 0), (0, 0)):
		# partial-width map: zero-extend the rings to the full circle (curvedsky.py:866-871); weights are per row
		pmap = _padded_like(map, pads, fill=not adjoint)
		res = map2alm_cyl(pmap, alm=alm, ainfo=ainfo, lmax=lmax, spin=spin, weights=weights, deriv=deriv, copy=False, verbose=verbose, adjoint=adjoint, nthread=nthread, pix_tol=pix_tol, niter=niter)
		if not adjoint: return res
		_crop_into(map, pmap, pads)
		return map
	kwargs = _ring_kwargs(map, minfo, ainfo)
	alm_full = _atleast(alm, 2 if deriv else 3)
	map_full = _atleast(mdata, 4)
	_check_shapes(alm_full, map_full, deriv)
	if _is_tensor(mdata): w = _torch().as_tensor(weights, device=mdata.device)[:, None]
	else: w = weights[:, None]
	def wmul(m): return m*w
	if deriv:
		# gradient maps [ddec, dra] <-> alm through the DERIV1 transforms (curvedsky.py:1067-1076)
		decflip = (_torch().as_tensor([-1.0, 1.0], device=mdata.device, dtype=mdata.dtype) if _is_tensor(mdata) else np.array([-1.0, 1.0], _np_dtype(mdata)))[:, None, None]
		for I in nditer(map_full.shape[:-3]):
			shp = map_full[I].shape
			def Y(a):   return sht.synthesis(alm=_contig(a), spin=1, mode="DERIV1", **kwargs).reshape(shp)
			def YT(m):  return sht.adjoint_synthesis(map=_flat(_contig(m)), spin=1, mode="DERIV1", **kwargs)
			def YTW(m): return YT(wmul(m))
			def WY(a):  return wmul(Y(a))
			if adjoint: map_full[I] = jacobi_inverse(YT, WY, _contig(alm_full[I][None]), niter=niter)*decflip
			else:       alm_full[I] = jacobi_inverse(Y, YTW, map_full[I]*decflip, niter=niter)[0]
		return map if adjoint else alm
	for I in nditer(map_full.shape[:-3]):
		for s, j1, j2 in enmap.spin_helper(spin, alm_full.shape[-2]):
			Ij = I+(slice(j1, j2),)
			shp = map_full[Ij].shape
			def Y(a):   return sht.synthesis(alm=_contig(a), spin=int(s), **kwargs).reshape(shp)
			def YT(m):  return sht.adjoint_synthesis(map=_flat(_contig(m)), spin=int(s), **kwargs)
			def YTW(m): return YT(wmul(m))
			def WY(a):  return wmul(Y(a))
			if adjoint: map_full[Ij] = jacobi_inverse(YT, WY, _contig(alm_full[Ij]), niter=niter)
			else:       alm_full[Ij] = jacobi_inverse(Y, YTW, map_full[Ij], niter=niter)
	if adjoint: return map
	else:       return alm

# ---------------------------------------------------------------------------------------
# alm post-processing either side of the transforms: almxfl, alm2cl, rand_alm.
# The per-element arithmetic (lmul, alm2cl) runs on the GPU (almops.py -> pxa_*); the random
# numbers come from numpy's legacy global RNG exactly as in the reference, so that a seed gives the
# same alm as pixell (curvedsky.py:61-79, 600-628).
# ---------------------------------------------------------------------------------------
def almxfl(alm, lfilter=None, ainfo=None, out=None):
	"""a_lm * lfilter(l); lfilter is an array starting at l=0 or a function of l (curvedsky.py:630-652)"""
	if not _is_tensor(alm): alm = np.asarray(alm)
	ainfo = alm_info(nalm=alm.shape[-1]) if ainfo is None else ainfo
	if callable(lfilter):
		l = np.arange(ainfo.lmax+1.0)
		lfilter = lfilter(l)
	return ainfo.lmul(alm, lfilter, out=out)

def alm2cl(alm, alm2=None, ainfo=None, dtype=None):
	"""(cross) power spectrum of alm (and alm2, which must broadcast) (curvedsky.py:674-712)"""
	if not _is_tensor(alm): alm = np.asarray(alm)
	ainfo = alm_info(nalm=alm.shape[-1]) if ainfo is None else ainfo
	return ainfo.alm2cl(alm, alm2=alm2, dtype=dtype)

def pad_spectrum(ps, lmax):
	ps = np.asarray(ps)
	ops = np.zeros(ps.shape[:-1]+(lmax+1,), ps.dtype)
	ops[..., :ps.shape[-1]] = ps[..., :ps.shape[-1]]
	return ops

def _sym_expand_diag(ps):
	"""powspec.sym_expand(ps, scheme="diag") (powspec.py:22-36, 53-100): [nspec,nl] -> [ncomp,ncomp,nl],
	healpy order: main diagonal first, then successive off-diagonals"""
	n = ps.shape[0]
	ncomp = int(np.ceil((np.sqrt(8*n+1)-1)/2))   # a truncated list: the smallest ncomp whose full list covers it
	which = [(i, i+d) for d in range(ncomp) for i in range(ncomp-d)][:n]
	res = np.zeros((ncomp, ncomp)+ps.shape[1:], ps.dtype)
	for v, (i, j) in zip(ps, which):
		res[i, j] = v; res[j, i] = v
	return res

def prepare_ps(ps, ainfo=None, lmax=None):
	ps = np.asarray(ps)
	if ainfo is None:
		if lmax is None: lmax = ps.shape[-1]-1
		if lmax > ps.shape[-1]-1: ps = pad_spectrum(ps, lmax)
		ainfo = alm_info(lmax)
	if   ps.ndim == 1: wps = ps[None, None]
	elif ps.ndim == 2: wps = _sym_expand_diag(ps)
	elif ps.ndim == 3: wps = ps
	else: raise ValueError("power spectrum must be [nl], [nspec,nl] or [ncomp,ncomp,nl]")
	return wps, ainfo

def _multi_sqrt(wps):
	"""enmap.multi_pow(wps, 0.5) (enmap.py:2021-2024 -> utils.eigpow): matrix square root of each
	[ncomp,ncomp] slice through its eigen-decomposition, negative eigenvalues set to zero"""
	A = np.moveaxis(np.asarray(wps, dtype=np.float64), (0, 1), (-2, -1))
	E, V = np.linalg.eigh(A)
	E = np.where(E < 0, 0, np.sqrt(np.abs(E)))
	res = np.einsum("...ij,...j,...kj->...ik", V, E, V)
	return np.moveaxis(res, (-2, -1), (0, 1))

def fill_gauss(arr, bsize=0x10000):
	rtype = real_dtype(arr.dtype)
	arr = arr.reshape(-1).view(rtype)
	for i in range(0, arr.size, bsize):
		arr[i:i+bsize] = np.random.standard_normal(min(bsize, arr.size-i))

def _transpose_index(ainfo):
	"""Source and destination positions of alm_info.transpose_alm (cmisc_core.c:116-135): the k-th
	element in storage order (m-major) moves to the k-th (l,m) pair in l-major order."""
	lmax, mmax = ainfo.lmax, ainfo.mmax
	m_src = np.repeat(np.arange(mmax+1), lmax+1-np.arange(mmax+1))
	first = np.concatenate([[0], np.cumsum(lmax+1-np.arange(mmax+1))[:-1]])
	l_src = np.arange(len(m_src))-np.repeat(first, lmax+1-np.arange(mmax+1))+m_src
	cnt = np.minimum(np.arange(lmax+1), mmax)+1
	l_dst = np.repeat(np.arange(lmax+1), cnt)
	firstl = np.concatenate([[0], np.cumsum(cnt)[:-1]])
	m_dst = np.arange(len(l_dst))-np.repeat(firstl, cnt)
	ms = ainfo.mstart.astype(np.int64)
	return ms[m_src]+l_src*ainfo.stride, ms[m_dst]+l_dst*ainfo.stride

def transpose_alm(ainfo, alm, out=None):
	"""alm_info.transpose_alm (curvedsky.py:443-451): reorder numbers generated in l-major order into
	the m-major layout.  alm is out is allowed."""
	src, dst = _transpose_index(ainfo)
	if out is None: out = alm.copy()
	vals = alm[..., src]
	out[..., dst] = vals
	return out
alm_info.transpose_alm = lambda self, alm, out=None: transpose_alm(self, alm, out=out)

def rand_alm_white(ainfo, pre=None, alm=None, seed=None, dtype=np.complex128, m_major=True):
	if seed is not None: np.random.seed(seed)
	if alm is None:
		if pre is None: alm = np.empty(ainfo.nelem, dtype)
		else:           alm = np.empty(tuple(pre)+(ainfo.nelem,), dtype)
	fill_gauss(alm)
	if m_major: ainfo.transpose_alm(alm, alm)
	return alm

def rand_alm(ps, ainfo=None, lmax=None, seed=None, dtype=np.complex128, m_major=True, return_ainfo=False):
	"""Gaussian alm with (cross) spectrum ps (curvedsky.py:61-79): white numbers drawn in l-major order
	from numpy's legacy RNG, then coloured with sqrt(ps) on the GPU (alm_info.lmul)."""
	ps = np.asarray(ps)
	rtype = real_dtype(dtype)
	wps, ainfo = prepare_ps(ps, ainfo=ainfo, lmax=lmax)
	alm = rand_alm_white(ainfo, pre=[wps.shape[0]], seed=seed, dtype=dtype, m_major=m_major)
	ps12 = _multi_sqrt(wps)
	alm = ainfo.lmul(alm, (ps12/2**0.5).astype(rtype, copy=False))
	alm[:, :ainfo.lmax+1].imag  = 0
	alm[:, :ainfo.lmax+1].real *= 2**0.5
	if ps.ndim == 1: alm = alm[0]
	if return_ainfo: return alm, ainfo
	else: return alm

def transfer_alm(iainfo, ialm, oainfo, oalm=None, op=lambda a, b: b):
	"""Copy alm between layouts / band limits (curvedsky.py:744-750 -> cmisc.pyx:131-151): for every (l,m) both
	layouts hold, oalm = op(oalm, ialm).  Works on numpy arrays and on torch CUDA tensors (one gather/scatter)."""
	tens = _is_tensor(ialm)
	if oalm is None:
		oalm = _zeros_like_kind(tuple(ialm.shape[:-1])+(oainfo.nelem,), _np_dtype(ialm), ialm)
	if tuple(ialm.shape[:-1]) != tuple(oalm.shape[:-1]):
		raise ValueError("ialm and oalm must agree on pre-dimensions")
	lmax = min(iainfo.lmax, oainfo.lmax); mmax = min(iainfo.mmax, oainfo.mmax)
	m = np.repeat(np.arange(mmax+1), lmax+1-np.arange(mmax+1))
	first = np.concatenate([[0], np.cumsum(lmax+1-np.arange(mmax+1))[:-1]])
	l = np.arange(len(m))-np.repeat(first, lmax+1-np.arange(mmax+1))+m
	src = iainfo.mstart.astype(np.int64)[m]+l*iainfo.stride
	dst = oainfo.mstart.astype(np.int64)[m]+l*oainfo.stride
	if tens:
		torch = _torch()
		src = torch.as_tensor(src, device=ialm.device); dst = torch.as_tensor(dst, device=oalm.device)
	oalm[..., dst] = op(oalm[..., dst], ialm[..., src])
	return oalm

def rand_map(shape, wcs, ps, lmax=None, dtype=np.float64, seed=None, spin=[0, 2], method="auto", verbose=False):
	"""Gaussian realisation of the (cross) spectrum ps on the given geometry (curvedsky.rand_map, curvedsky.py:17-37).
	The reference draws its alm with healpy.synalm (healpy is absent here); this uses rand_alm, i.e. the same
	distribution but pixell's own random-number order (SURVEY: 'replace by rand_alm semantics')."""
	ps = np.asarray(ps)
	while ps.ndim < 3: ps = ps[None]
	if ps.shape[0] != ps.shape[1]: raise ValueError("ps must be [ncomp,ncomp,nl] or [nl]")
	if len(shape) not in (2, 3): raise ValueError("shape must be (ncomp,ny,nx) or (ny,nx)")
	ncomp = 1 if len(shape) == 2 else shape[-3]
	ps = ps[:ncomp, :ncomp]
	ctype = np.result_type(dtype, 0j)
	alm = rand_alm(ps, lmax=lmax, seed=seed, dtype=ctype)
	map = enmap.empty((ncomp,)+tuple(shape[-2:]), wcs, dtype=dtype)
	alm2map(alm, map, spin=spin, method=method, verbose=verbose)
	if len(shape) == 2: map = map[0]
	return map

def filter(imap, lfilter, ainfo=None, lmax=None):
	"""alm2map(almxfl(map2alm(imap), lfilter)): isotropic filtering of a map (curvedsky.filter, curvedsky.py:654-671)"""
	alm = map2alm(imap, ainfo=ainfo, lmax=lmax, spin=0)
	alm = almxfl(alm, lfilter=lfilter, ainfo=ainfo)
	out = enmap.dmap(_torch().empty_like(imap.tensor), imap.wcs) if isinstance(imap, enmap.dmap) else enmap.empty(imap.shape, imap.wcs, dtype=imap.dtype)
	return alm2map(alm, out, spin=0, ainfo=ainfo)
