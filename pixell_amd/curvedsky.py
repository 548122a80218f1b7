"""Drop-in for the spherical-harmonic-transform API of pixell/curvedsky.py on MI355X.

Same function names, argument meaning and error behaviour as the reference for the `2d` and
`cyl` methods and the healpix / profile helpers (curvedsky.py:83-302, 312-403, 510-580, 756-1086, 1170-1446); the `general` method
(non-cylindrical pixelisations) and rotate_alm are outside the accelerated path and raise NotImplementedError (prof2alm: without the rotation).

Differences that matter for speed, not for results:
  * the flipped / padded copies of map2buffer / buffer2map (curvedsky.py:1384-1411) are not
    made: flips are negative strides inside the kernels (sht.*(flip=...));
  * `map` may be an enmap.ndmap (numpy; staged through the GPU) or an enmap.dmap wrapping a torch
    CUDA tensor (transformed in place, nothing leaves HBM); `alm` likewise numpy or tensor.
"""
import os
import numpy as np
from . import enmap, sht, wcs as wcsutils
from .sht import _is_tensor, _np_dtype, _torch

from .geometry import Bunch, degree

def nint(a): return np.round(a).astype(int)
def complex_dtype(dtype): return np.result_type(dtype, 0j)
def real_dtype(dtype): return np.zeros([0], dtype).real.dtype
def nditer(shape):
	for I in np.ndindex(*shape): yield tuple(I)

def nalm2lmax(nalm):
	"""band limit of a full triangular layout with nalm = (lmax+1)(lmax+2)/2 elements"""
	return int((np.sqrt(8.0*nalm+1)-1)/2)-1

# ---------------------------------------------------------------------------------------
# alm layout
# ---------------------------------------------------------------------------------------
def _mstart_triangular(lmax, mmax, stride):
	m = np.arange(mmax+1, dtype=np.int64)
	return stride*((m*(2*lmax+1-m))//2)           # column m starts where l = 0 WOULD be: element (l, m) sits at mstart[m] + l stride
def _mstart_rectangular(lmax, mmax, stride):
	return stride*(lmax+1)*np.arange(mmax+1, dtype=np.int64)
_LAYOUTS = {"triangular": (_mstart_triangular, nalm2lmax), "tri": (_mstart_triangular, nalm2lmax),
	"rectangular": (_mstart_rectangular, lambda nalm: int(np.sqrt(nalm))-1), "rect": (_mstart_rectangular, lambda nalm: int(np.sqrt(nalm))-1)}

class alm_info:
	"""Where element (l, m) of a spherical-harmonic coefficient array lives: index = mstart[m] + l*stride (the contract of
	pixell's curvedsky.alm_info, curvedsky.py:409-476, which the C ABI takes over as `mstart` / `lstride`).
	layout: "triangular" (m-major, no padding; the default), "rectangular" (every m has lmax+1 slots) or an explicit mstart array.
	Either lmax or the element count nalm fixes the size; mmax defaults to lmax."""
	def __init__(self, lmax=None, mmax=None, nalm=None, stride=1, layout="triangular"):
		lmax, mmax, nalm = (None if v is None else int(v) for v in (lmax, mmax, nalm))
		if isinstance(layout, str):
			if layout not in _LAYOUTS: raise ValueError("unknown alm layout: %s" % layout)
			build, lmax_from_count = _LAYOUTS[layout]
			if lmax is None:
				if nalm is None: raise ValueError("alm_info needs lmax or nalm")
				lmax = lmax_from_count(nalm)
			if mmax is None: mmax = lmax
			mstart = build(lmax, mmax, int(stride))
		else:
			mstart = np.asarray(layout)
			if lmax is None or mmax is None: raise ValueError("an explicit mstart needs lmax and mmax")
		self.lmax, self.mmax, self.stride = lmax, mmax, int(stride)
		self.nelem = int(np.max(mstart))+(lmax+1)*self.stride
		self.nreal = lmax*lmax+2*lmax+2
		assert nalm is None or self.nelem == nalm, "lmax must be explicitly specified when lmax != mmax"
		self.mstart = mstart.astype(np.uint64, copy=False)
	nl = property(lambda self: self.lmax+1)
	nm = property(lambda self: self.mmax+1)
	def lm2ind(self, l, m):
		return (self.mstart[m].astype(int, copy=False)+np.asarray(l)*self.stride).astype(int, copy=False)
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
# geometry analysis: thin names over pixell_amd/geometry.py (the reference's function names, curvedsky.py:478-505, 1170-1353)
# ---------------------------------------------------------------------------------------
from . import geometry as _geo
get_ducc_maxlmax = _geo.grid_maxlmax
def get_ducc_geo(wcs, shape=None, tol=1e-6): return _geo.classify_grid(wcs, shape=shape, tol=tol)
def analyse_geometry(shape, wcs, tol=1e-6): return _geo.analyse(shape, wcs, tol=tol)
def get_method(shape, wcs, minfo=None, pix_tol=1e-6):
	return _geo.method_of((minfo if minfo is not None else _geo.analyse(shape, wcs, tol=pix_tol)).case)
def get_ring_info(shape, wcs, dtype=np.float64): return _geo.ring_tables(shape, wcs, dtype=dtype)
def quad_weights(shape, wcs, pix_tol=1e-6, row_order="reference"):
	"""curvedsky.quad_weights (curvedsky.py:492-505), result for result: the grid's ring weights / nx, SOUTH-first whatever the
	map's row order (the reference reverses the north-first weights for every map).  row_order="map" returns them in the order of
	the map's rows instead -- the two differ only for maps stored north to south."""
	return _geo.ring_weights(shape, wcs, sht.get_gridweights, tol=pix_tol, row_order=row_order)

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
	"""(alm, ainfo) ready for a transform of maps of real type `dtype` (the contract of curvedsky.prepare_alm, curvedsky.py:1413-1427):
	a missing alm is allocated (zeros, next to `like`) for ainfo or lmax; an alm of the wrong precision is converted when
	`convert`, refused with ValueError otherwise; a missing ainfo is the triangular layout of the alm's length."""
	want = complex_dtype(dtype)
	if alm is None:
		if ainfo is None and lmax is None: raise ValueError("prepare_alm needs either alm, ainfo or lmax to be specified")
		ainfo = ainfo if ainfo is not None else alm_info(lmax)
		return _zeros_like_kind(tuple(pre)+(ainfo.nelem,), want, like), ainfo
	ainfo = ainfo if ainfo is not None else alm_info(nalm=alm.shape[-1])
	have = _np_dtype(alm)
	if have != want:
		if not convert: raise ValueError("alm had dtype '%s', but expected '%s'" % (str(have), str(want)))
		alm = alm.to(getattr(_torch(), np.dtype(want).name)) if _is_tensor(alm) else alm.astype(want)
	return alm, ainfo

def _atleast(x, n):
	while x.ndim < n: x = x[None]
	return x

def _contig(x):
	if _is_tensor(x): return x if x.is_contiguous() else x.contiguous()
	return x

def _batched_jobs(spin, alm_full, map_full):
	"""[(spin, alm [nb, nca, nelem], map [nb, ncm, ny, nx])]: the reference loops over the pre-dimensions and the spin groups
	(curvedsky.py:910-924, 1038-1046) with one ducc call each; here every spin group is ONE library call over all pre-dimension
	entries, and a stack of scalar maps (every component a spin-0 group) is one call over its components."""
	nelem = alm_full.shape[-1]; ncomp = alm_full.shape[-2]; ny, nx = map_full.shape[-2:]
	groups = [(int(s), j1, j2) for s, j1, j2 in enmap.spin_helper(spin, ncomp)]
	if int(np.prod(map_full.shape[:-3], dtype=int)) == 0: return []       # empty pre-dimension: nothing to transform
	a3 = _as_view(alm_full, (-1, ncomp, nelem)); m4 = _as_view(map_full, (-1, ncomp, ny, nx))
	if a3 is None or m4 is None:
		# pre-dimensions that cannot be merged without a copy (exotic strides): one call per entry, as the reference does
		return [(s, alm_full[I+(slice(j1, j2),)][None], map_full[I+(slice(j1, j2),)][None]) for I in nditer(map_full.shape[:-3]) for s, j1, j2 in groups]
	if all(s == 0 for s, _, _ in groups) and ncomp > 1:
		return [(0, a3.reshape((-1, 1, nelem)), m4.reshape((-1, 1, ny, nx)))]
	return [(s, a3[:, j1:j2], m4[:, j1:j2]) for s, j1, j2 in groups]

def _as_view(x, shape):
	"""x reshaped WITHOUT copying (results are written through these views), or None"""
	try:
		if _is_tensor(x): return x.view(shape)
		v = x.view(); v.shape = tuple(int(np.prod(x.shape[:x.ndim-len(shape)+1], dtype=int)) if n == -1 else n for n in shape); return v
	except (RuntimeError, AttributeError): return None

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
		verbose=False, nthread=None, niter=0, epsilon=None, pix_tol=1e-6, weights=None, locinfo=None, tweak=False, analysis=None, weights_order="reference"):
	"""Spherical harmonics analysis (curvedsky.map2alm, curvedsky.py:209-302).
	analysis (ours, method "2d" only): None / "ducc0" = the route ducc0's analysis_2d takes as published; "interpolant" = exact
	quadrature of the full theta-interpolant; "weights" = ring quadrature weights + adjoint synthesis (the reference's cyl route,
	curvedsky.py:852-861) on full grids with ny >= 2 lmax + 2: identical alm for band-limited maps (see pixell_amd.sht.analysis_2d).
	weights_order (ours, method "cyl" only): "reference" (default) applies weights to rows exactly as the reference does, including its
	mirrored pixel areas for maps stored north-to-south off a named grid; "map": weight i belongs to map row i (see map2alm_cyl)."""
	minfo = analyse_geometry(map.shape, map.wcs, tol=pix_tol)
	if method == "auto": method = get_method(map.shape, map.wcs, minfo=minfo)
	if   method == "2d":
		return map2alm_2d(map, alm, ainfo=ainfo, minfo=minfo, lmax=lmax, spin=spin, deriv=deriv, copy=copy, verbose=verbose, adjoint=adjoint, nthread=nthread, pix_tol=pix_tol, analysis=analysis)
	elif method == "cyl":
		return map2alm_cyl(map, alm, ainfo=ainfo, minfo=minfo, lmax=lmax, spin=spin, deriv=deriv, copy=copy, verbose=verbose, adjoint=adjoint, nthread=nthread, niter=niter, pix_tol=pix_tol, weights=weights, weights_order=weights_order)
	elif method == "general":
		raise NotImplementedError("method 'general' (non-cylindrical pixelisations, ducc synthesis_general) is outside the accelerated path")
	else:
		raise ValueError("Unrecognized alm2map method '%s'" % str(method))

def map2alm_adjoint(alm, map, lmax=None, spin=[0, 2], deriv=False, copy=False, method="auto", ainfo=None, verbose=False, nthread=None, niter=0, epsilon=1e-6, pix_tol=1e-6, weights=None, locinfo=None, analysis=None):
	"""curvedsky.map2alm_adjoint (curvedsky.py:304-310); analysis (ours): the form of map2alm whose transpose this is"""
	return map2alm(map=map, alm=alm, lmax=lmax, spin=spin, deriv=deriv, adjoint=True, copy=copy, method=method, ainfo=ainfo, verbose=verbose, nthread=nthread, niter=niter, epsilon=epsilon, pix_tol=pix_tol, weights=weights, locinfo=locinfo, analysis=analysis)

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
	if deriv:
		for I in nditer(map_full.shape[:-3]):
			a = _contig(alm_full[I][None]); m = map_full[I]
			func(alm=a, map=m, mode="DERIV1", spin=1, **kwargs)
			if adjoint: alm_full[I] = a[0]
			else: map_full[I+(0,)] *= -1       # theta derivative -> dec derivative (curvedsky.py:919)
	else:
		jobs = _batched_jobs(spin, alm_full, map_full)
		# numpy arrays: the inputs of all spin groups start uploading now, outputs come back while the next group runs (pixell_amd/hostio.py).
		# Map -> alm: the group that is cheapest to transform goes LAST (its transform is the only one no upload hides); alm -> map: first.
		if adjoint: jobs = sorted(jobs, key=lambda j: -j[2].shape[-3])
		with sht.host_pipeline([m if adjoint else a for s, a, m in jobs], [a if adjoint else m for s, a, m in jobs]):
			for s, a, m in jobs: func(alm=a, map=m, spin=s, **kwargs)
	if adjoint: return alm
	else:       return map

def map2alm_2d(map, alm=None, ainfo=None, minfo=None, lmax=None, spin=[0, 2], deriv=False, copy=False, verbose=False, adjoint=False, nthread=None, pix_tol=1e-6, analysis=None):
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
		res = map2alm_2d(pmap, alm=alm, ainfo=ainfo, lmax=lmax, spin=spin, deriv=deriv, copy=False, verbose=verbose, adjoint=adjoint, nthread=nthread, pix_tol=pix_tol, analysis=analysis)
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
	kwargs = dict(phi0=minfo.phi0, lmax=l, mmax=m, geometry=minfo.ducc_geo.name, mstart=ainfo.mstart[:m+1], lstride=ainfo.stride, flip=minfo.flip, analysis=analysis)
	jobs = _batched_jobs(spin, alm_full, map_full)
	if not adjoint: jobs = sorted(jobs, key=lambda j: -j[2].shape[-3])      # (see alm2map_2d: map -> alm takes the largest group first)
	with sht.host_pipeline([a if adjoint else m for s, a, m in jobs], [m if adjoint else a for s, a, m in jobs]):
		for s, a, m in jobs: func(alm=a, map=m, spin=s, **kwargs)
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
	"""Solve forward(x) = y given an approximate inverse B: x_0 = B y, x_{k+1} = x_k + B (y - forward(x_k))
	(Jacobi / Richardson refinement; what curvedsky.jacobi_inverse, curvedsky.py:1122-1136, does around the ring transforms)"""
	x = approx_backward(y)
	for _ in range(int(niter)):
		x = x+approx_backward(y-forward(x))
	return x

def map2alm_cyl(map, alm=None, ainfo=None, minfo=None, lmax=None, spin=[0, 2], weights=None, deriv=False, copy=False, verbose=False, adjoint=False, nthread=None, pix_tol=1e-6, niter=0, weights_order="reference"):
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
	# Row weights.  The reference applies `weights` to the rows of its flipped buffer (north first: map2buffer, curvedsky.py:866-868),
	# here the rings are the map's rows as stored.  weights_order="reference" (default) reproduces the reference result for result:
	#   caller-supplied weights are taken north-first; default weights of a named grid are the grid's (physically right in both);
	#   default weights off the grid are the pixel areas, which the reference reverses for every map (`if minfo.flip:` on a list,
	#   curvedsky.py:860) -- right for maps stored south to north, mirrored for maps stored north to south.
	# weights_order="map": weight i belongs to map row i, pixel areas unmirrored.
	if weights_order not in ("reference", "map"): raise ValueError("weights_order must be 'reference' or 'map'")
	ref_order = weights_order == "reference"
	if weights is None:
		if minfo.ducc_geo is not None and minfo.ducc_geo.name in _geo.WEIGHTED_GRIDS:
			weights = quad_weights(map.shape, map.wcs, pix_tol=pix_tol, row_order="map")
		else:
			# pixel area of each row (enmap.pixsizemap separable; curvedsky.py:858-860)
			ny, nx = map.shape[-2:]
			dec = enmap.pix2sky(map.shape, map.wcs, [np.concatenate([np.arange(ny)-0.5, [ny-0.5]]), np.zeros(ny+1)])[0]
			dec = np.clip(dec, -np.pi/2, np.pi/2)
			weights = np.abs(np.sin(dec[1:])-np.sin(dec[:-1]))*abs(map.wcs.wcs.cdelt[0])*degree
			if ref_order and not minfo.flip[0]: weights = weights[::-1]
	else:
		weights = np.asarray(weights.cpu() if _is_tensor(weights) else weights)
		if ref_order and minfo.flip[0]: weights = weights[::-1]
	weights = np.ascontiguousarray(weights, dtype=_np_dtype(mdata))
	pads = _native_pads(minfo, use_y=False)
	if pads != ((0, 0), (0, 0)):
		# partial-width map: zero-extend the rings to the full circle (curvedsky.py:866-871); weights are per row
		pmap = _padded_like(map, pads, fill=not adjoint)
		res = map2alm_cyl(pmap, alm=alm, ainfo=ainfo, lmax=lmax, spin=spin, weights=weights, deriv=deriv, copy=False, verbose=verbose, adjoint=adjoint, nthread=nthread, pix_tol=pix_tol, niter=niter, weights_order="map")
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
# The reference's helper entry points around the transforms, by name (curvedsky.py:900-1086, 1236-1250, 1349-1353, 1384-1473).
# pixell stages every map through a flipped, padded copy (map2buffer) and runs the `raw` functions on that buffer; here flips are
# strides inside the kernels, so the `raw` functions are the plain ones restricted to what a buffer is (a complete grid / complete
# rings), and map2buffer / buffer2map exist for callers that want the staged form.
# ---------------------------------------------------------------------------------------
class ShapeError(Exception): pass

def get_ducc_maxlmax(name, ny): return _geo.grid_maxlmax(name, ny)
def dangerous_dtype(dtype): return np.dtype(dtype).byteorder not in ("=", "|")

def flip2slice(flips):
	"""index expression that reverses the trailing axes whose flag is set"""
	return (Ellipsis,)+tuple(slice(None, None, -1 if f else 1) for f in flips)
def flip_array(arr, flips): return arr[flip2slice(flips)]
def flip_geometry(shape, wcs, flips): return wcsutils.flipped(shape, wcs, flips)
def pad_geometry(shape, wcs, pad):
	"""geometry with pad[0] = (rows, columns) added before and pad[1] after"""
	pad = np.asarray(pad)
	w = wcs.deepcopy(); w.wcs.crpix[0] += pad[0, 1]; w.wcs.crpix[1] += pad[0, 0]
	return tuple(shape[:-2])+(int(pad[0, 0]+shape[-2]+pad[1, 0]), int(pad[0, 1]+shape[-1]+pad[1, 1])), w

def map2buffer(map, flip, pad, obuf=False):
	"""zero-filled map of the flipped, padded geometry holding the flipped map (obuf: left empty, an output buffer)"""
	pad = np.asarray(pad)
	shape, wcs = pad_geometry(*flip_geometry(map.shape, map.wcs, flip), pad)
	buf = enmap.zeros(shape, wcs, np.dtype(map.dtype).newbyteorder("="))
	if not obuf: buf[..., pad[0, 0]:shape[-2]-pad[1, 0], pad[0, 1]:shape[-1]-pad[1, 1]] = flip_array(np.asarray(map), flip)
	return buf
def buffer2map(map, flip, pad):
	"""undo map2buffer: crop the padding, flip back (a view)"""
	pad = np.asarray(pad)
	return flip_array(map[..., pad[0, 0]:map.shape[-2]-pad[1, 0], pad[0, 1]:map.shape[-1]-pad[1, 1]], flip)

def prepare_raw(alm, map, ainfo=None, lmax=None, deriv=False, verbose=False, nthread=None, pixdims=2, convert_alm=False):
	"""(alm_full, map_full, ainfo, nthread): alm [..., ncomp, nelem] and map [..., ncomp, pixels] with agreeing leading dimensions
	(deriv: alm [..., nelem], map [..., 2, pixels]); AssertionError otherwise (curvedsky.py:1429-1446)"""
	mdata = _mdata(map)
	alm, ainfo = prepare_alm(alm, ainfo, lmax=lmax, pre=map.shape[:-pixdims], dtype=_np_dtype(mdata), convert=convert_alm, like=mdata)
	alm_full = _atleast(alm, 2 if deriv else 3); map_full = _atleast(mdata, pixdims+2)
	if deriv:
		assert map_full.shape[-pixdims-1] == 2, "map must have shape [...,2,%s] when deriv is True" % ("nloc" if pixdims == 1 else "ny,nx")
		assert tuple(map_full.shape[:-1-pixdims]) == tuple(alm_full.shape[:-1]), "map and alm must agree on pre-dimensions"
	else:
		assert tuple(map_full.shape[:-pixdims]) == tuple(alm_full.shape[:-1]), "map and alm must agree on pre-dimensions"
	env = os.environ.get("OMP_NUM_THREADS")
	return alm_full, map_full, ainfo, int(env) if env else int(nthread or 0)

def _require_case(map, minfo, cases, what):
	minfo = analyse_geometry(map.shape, map.wcs) if minfo is None else minfo
	if minfo.case not in cases or (minfo.case == "cyl" and cases == ("2d",)): raise ValueError("%s needs %s (got case '%s'): use the function without _raw, which pads" % (what, " or ".join(cases), minfo.case))
	return minfo

def alm2map_raw_2d(alm, map, ainfo=None, spin=[0, 2], deriv=False, copy=False, verbose=False, adjoint=False, nthread=None):
	"""synthesis_2d / adjoint_synthesis_2d on a map that IS a complete named grid (curvedsky.py:900-926)"""
	return alm2map_2d(alm, map, ainfo=ainfo, minfo=_require_case(map, None, ("2d",), "alm2map_raw_2d"), spin=spin, deriv=deriv, copy=copy, verbose=verbose, adjoint=adjoint, nthread=nthread)
def map2alm_raw_2d(map, alm=None, ainfo=None, lmax=None, spin=[0, 2], deriv=False, copy=False, verbose=False, adjoint=False, nthread=None):
	"""analysis_2d / adjoint_analysis_2d on a map that IS a complete named grid (curvedsky.py:1018-1048)"""
	return map2alm_2d(map, alm=alm, ainfo=ainfo, minfo=_require_case(map, None, ("2d",), "map2alm_raw_2d"), lmax=lmax, spin=spin, deriv=deriv, copy=copy, verbose=verbose, adjoint=adjoint, nthread=nthread)
def alm2map_raw_cyl(alm, map, ainfo=None, minfo=None, spin=[0, 2], deriv=False, copy=False, verbose=False, adjoint=False, nthread=None):
	"""synthesis / adjoint_synthesis on the rows of a map whose rings close the circle (curvedsky.py:928-962)"""
	return alm2map_cyl(alm, map, ainfo=ainfo, minfo=_require_case(map, minfo, ("2d", "cyl"), "alm2map_raw_cyl"), spin=spin, deriv=deriv, copy=copy, verbose=verbose, adjoint=adjoint, nthread=nthread)
def map2alm_raw_cyl(map, alm=None, ainfo=None, lmax=None, spin=[0, 2], weights=None, deriv=False, copy=False, verbose=False, adjoint=False, niter=0, nthread=None):
	"""weighted adjoint_synthesis + Jacobi refinement on the rows of a map whose rings close the circle (curvedsky.py:1050-1086)"""
	return map2alm_cyl(map, alm=alm, ainfo=ainfo, minfo=_require_case(map, None, ("2d", "cyl"), "map2alm_raw_cyl"), lmax=lmax, spin=spin, weights=weights, deriv=deriv, copy=copy, verbose=verbose, adjoint=adjoint, nthread=nthread, niter=niter, weights_order="map")      # (raw: weight i multiplies row i of the array handed in, curvedsky.py:1064-1065)

def alm_complex2real(alm, ainfo=None):
	"""complex alm (m >= 0 storage) -> real vector of the same information with unit Jacobian: the m = 0 block keeps its real
	parts, every m > 0 coefficient becomes sqrt(2) (Re, Im) (curvedsky.py:1451-1456)"""
	alm = np.asarray(alm)
	ainfo = alm_info(nalm=alm.shape[-1]) if ainfo is None else ainfo
	n0 = int(ainfo.mstart[1])+1
	return np.concatenate([alm[..., :n0].real, np.sqrt(2.0)*np.ascontiguousarray(alm[..., n0:]).view(real_dtype(alm.dtype))], -1)
def alm_real2complex(ralm, ainfo=None):
	"""inverse of alm_complex2real; without ainfo the triangular layout with (lmax+1)^2 = len is assumed"""
	ralm = np.asarray(ralm)
	if ainfo is None: ainfo = alm_info(lmax=nint((ralm.shape[-1]-1)**0.5)-1)       # length = lmax^2 + 2 lmax + 2 (a_00's zero slot included)
	n0 = int(ainfo.mstart[1])+1
	out = np.zeros(ralm.shape[:-1]+(ainfo.nelem,), complex_dtype(ralm.dtype))
	out[..., :n0] = ralm[..., :n0]
	out[..., n0:] = np.ascontiguousarray(ralm[..., n0:]).view(out.dtype)/np.sqrt(2.0)
	return out

# ---------------------------------------------------------------------------------------
# healpix maps: the same ring transforms on ring tables with per-ring nphi / phi0 (the general ring path of sht.hip)
# ---------------------------------------------------------------------------------------
def npix2nside(npix): return nint((npix/12)**0.5)
def get_ring_info_healpix(nside, rings=None): return _geo.healpix_rings(nside, rings)
def get_ring_info_radial(r): return _geo.radial_rings(r)
def apply_minfo_theta_lim(minfo, theta_min=None, theta_max=None): return _geo.theta_window(minfo, theta_min, theta_max)
def prepare_healmap(healmap, nside=None, pre=(), dtype=np.float64, like=None):
	if healmap is not None: return healmap
	return _zeros_like_kind(tuple(pre)+(12*int(nside)**2,), dtype, like)

def _healpix_kwargs(npix, ainfo, theta_min, theta_max):
	rinfo = apply_minfo_theta_lim(get_ring_info_healpix(npix2nside(npix)), theta_min, theta_max)
	return rinfo, dict(theta=rinfo.theta, nphi=rinfo.nphi, phi0=rinfo.phi0, ringstart=rinfo.offsets, lmax=ainfo.lmax, mmax=ainfo.mmax,
		mstart=ainfo.mstart, lstride=ainfo.stride)

def alm2map_healpix(alm, healmap=None, spin=[0, 2], deriv=False, adjoint=False, copy=False, ainfo=None, nside=None, theta_min=None, theta_max=None, nthread=None):
	"""alm[..., ncomp, nelem] -> healpix map[..., ncomp, 12 nside^2] in RING order (curvedsky.alm2map_healpix, curvedsky.py:312-351);
	adjoint: the transpose, map -> alm.  With deriv the map is [..., 2, npix] = (d/ddec, d/dra / cos dec)."""
	if copy:
		if adjoint and alm is not None: alm = alm.clone() if _is_tensor(alm) else alm.copy()
		elif not adjoint and healmap is not None: healmap = healmap.clone() if _is_tensor(healmap) else healmap.copy()
	rdt = real_dtype(_np_dtype(alm))
	alm, ainfo = prepare_alm(alm, ainfo, dtype=rdt, convert=not adjoint, like=alm)
	healmap = prepare_healmap(healmap, nside, alm.shape[:-2]+(2,) if deriv else alm.shape[:-1], rdt, like=alm)
	alm_full = _atleast(alm, 2 if deriv else 3); map_full = _atleast(healmap, 3)
	if deriv and (tuple(alm_full.shape[:-1]) != tuple(map_full.shape[:-2]) or map_full.shape[-2] != 2):
		raise ValueError("When deriv is True, alm must have shape [...,nelem] and map shape [...,2,npix]")
	if not deriv and tuple(alm_full.shape[:-1]) != tuple(map_full.shape[:-1]):
		raise ValueError("alm must have shape [...,[ncomp,]nelem] and map shape [...,[ncomp,]npix]")
	rinfo, kwargs = _healpix_kwargs(map_full.shape[-1], ainfo, theta_min, theta_max)
	if (theta_min is not None or theta_max is not None) and not adjoint: map_full[...] = 0     # rings outside the window are not written
	func = sht.adjoint_synthesis if adjoint else sht.synthesis
	for I in nditer(map_full.shape[:-2]):
		if deriv:
			a = _contig(alm_full[I][None])
			func(alm=a, map=map_full[I], mode="DERIV1", spin=1, **kwargs)
			if adjoint: alm_full[I] = a[0]
			else: map_full[I+(0,)] *= -1                   # d/dtheta -> d/ddec
		else:
			for s, j1, j2 in enmap.spin_helper(spin, alm_full.shape[-2]):
				Ij = I+(slice(j1, j2),)
				v = alm_full[Ij]; a = _contig(v)
				func(alm=a, map=map_full[Ij], spin=int(s), **kwargs)
				if adjoint and a is not v: v[...] = a
	return alm if adjoint else healmap

def map2alm_healpix(healmap, alm=None, ainfo=None, lmax=None, spin=[0, 2], weights=None, deriv=False, copy=False, verbose=False, adjoint=False, niter=0, theta_min=None, theta_max=None, nthread=None):
	"""healpix map -> alm with pixel-area weights 4 pi / npix (or `weights`) and `niter` Jacobi refinements
	(curvedsky.map2alm_healpix, curvedsky.py:353-403; like healpy.map2alm).  adjoint: the transpose of that operator, alm -> map."""
	if copy:
		if adjoint and healmap is not None: healmap = healmap.clone() if _is_tensor(healmap) else healmap.copy()
		elif not adjoint and alm is not None: alm = alm.clone() if _is_tensor(alm) else alm.copy()
	if deriv: raise NotImplementedError("map2alm_healpix with deriv=True is broken")          # (as in the reference, curvedsky.py:378)
	alm, ainfo = prepare_alm(alm=alm, ainfo=ainfo, lmax=lmax, pre=healmap.shape[:-1], dtype=_np_dtype(healmap), convert=adjoint, like=healmap)
	alm_full = _atleast(alm, 3); map_full = _atleast(healmap, 3)
	rinfo, kwargs = _healpix_kwargs(map_full.shape[-1], ainfo, theta_min, theta_max)
	if weights is None: weights = 4*np.pi/rinfo.npix
	if _is_tensor(healmap) and not np.isscalar(weights): weights = _torch().as_tensor(np.asarray(weights), device=healmap.device)
	for I in nditer(map_full.shape[:-2]):
		for s, j1, j2 in enmap.spin_helper(spin, alm_full.shape[-2]):
			Ij = I+(slice(j1, j2),)
			like = map_full[Ij]
			def Y(a):   return sht.synthesis(alm=_contig(a), map=_zeros_like_kind(tuple(like.shape), _np_dtype(like), like), spin=int(s), **kwargs)
			def YT(m):  return sht.adjoint_synthesis(map=_contig(m), spin=int(s), **kwargs)
			def YTW(m): return YT(m*weights)
			def WY(a):  return Y(a)*weights
			if adjoint: map_full[Ij] = jacobi_inverse(YT, WY, _contig(alm_full[Ij]), niter=niter)
			else:       alm_full[Ij] = jacobi_inverse(Y, YTW, map_full[Ij], niter=niter)
	return healmap if adjoint else alm

# ---------------------------------------------------------------------------------------
# 1-D transforms: radial profiles <-> functions of l, as m = 0 transforms on rings of one pixel (curvedsky.py:510-554)
# ---------------------------------------------------------------------------------------
def _m0_kwargs(theta, lmax):
	rinfo = get_ring_info_radial(theta)
	return dict(theta=rinfo.theta, nphi=rinfo.nphi, phi0=rinfo.phi0, ringstart=rinfo.offsets, spin=0, lmax=int(lmax), mmax=0, mstart=np.zeros(1, np.uint64))

def profile2harm(br, r, lmax=None, oversample=1, left=None, right=None):
	"""br[..., nr] sampled at ascending radii r (radians) -> bl[..., lmax+1] with b(r) = sum_l (2l+1)/(4 pi) b_l P_l(cos r): the
	profile is interpolated linearly onto the Clenshaw-Curtis nodes of spacing ~dr that cover [0, r_max], weighted with the CC ring
	weights and analysed with one m = 0 adjoint synthesis (the contract of curvedsky.profile2harm, curvedsky.py:510-541)"""
	br = np.asarray(br); r = np.asarray(r)
	step = (r[-1]-r[0])/(len(r)-1)
	nfull = nint(np.pi/step)+1; step = np.pi/(nfull-1)          # nodes k pi/(nfull-1) of the full circle, of which the first ncut reach r_max
	ncut = int(np.ceil(r[-1]/step))
	if lmax is None: lmax = int(nfull//2-1)
	theta = np.arange(ncut)*step
	kw = _m0_kwargs(theta, lmax)
	w = sht.get_gridweights("CC", nfull)[:ncut]
	norm = np.sqrt(4*np.pi/(2*np.arange(lmax+1)+1))
	out = np.zeros(br.shape[:-1]+(lmax+1,), br.dtype)
	for I in nditer(br.shape[:-1]):
		ring = np.interp(theta, r, br[I], left=left, right=right)[None]*w
		out[I] = sht.adjoint_synthesis(map=np.ascontiguousarray(ring, dtype=np.float64), **kw)[0].real*norm
	return out

def harm2profile(bl, r):
	"""bl[..., nl] -> b(r)[..., nr] = sum_l (2l+1)/(4 pi) b_l P_l(cos r): one m = 0 synthesis on rings of one pixel at
	colatitudes r (curvedsky.harm2profile, curvedsky.py:543-554)"""
	bl = np.asarray(bl); r = np.asarray(r)
	nl = bl.shape[-1]
	kw = _m0_kwargs(r.reshape(-1), nl-1)
	alm = bl*np.sqrt((2*np.arange(nl)+1)/(4*np.pi))+0j
	out = np.zeros(bl.shape[:-1]+(r.size,), bl.dtype)
	for I in nditer(bl.shape[:-1]):
		out[I] = sht.synthesis(alm=np.ascontiguousarray(alm[I][None], dtype=np.complex128), **kw)[0]
	return out

def prof2alm(profile, dir=[0, np.pi/2], spin=0, geometry="CC", nthread=None, norot=False):
	"""alm of a 1-D profile[..., n] sampled on the rings of `geometry` (an azimuthally symmetric field around the pole): one m = 0 analysis_2d of the
	profile as a map one pixel wide, band limit get_ducc_maxlmax(geometry, n) (curvedsky.prof2alm, curvedsky.py:556-580).  norot: the m = 0 layout
	as it is; otherwise expanded to the full triangular layout.  The reference then rotates the pole to dir = [ra, dec] with ducc0.sht.rotate_alm
	(out of this path's scope): only the default direction -- the pole itself, the identity rotation -- is accepted."""
	profile = np.asarray(profile)
	lmax = get_ducc_maxlmax(geometry, profile.shape[-1])
	iainfo = alm_info(lmax=lmax, mmax=0)
	oainfo = alm_info(lmax=lmax, mmax=lmax if not norot else 0)
	if not norot and not (float(dir[0]) == 0.0 and float(dir[1]) == np.pi/2):
		raise NotImplementedError("prof2alm: rotating the profile to dir needs rotate_alm, which is outside the accelerated path")
	ctype = complex_dtype(profile.dtype)
	oalm = np.zeros(profile.shape[:-1]+(oainfo.nelem,), ctype)
	pre = profile.shape[:-1]
	jobs = [(0, (None,))] if len(pre) == 0 else [(s, Ipre+(slice(i1, i2),)) for Ipre in nditer(pre[:-1]) for s, i1, i2 in enmap.spin_helper(spin, pre[-1])]
	for s, I in jobs:
		prof = np.ascontiguousarray(profile[I][..., None])
		alm = np.zeros(prof.shape[:-2]+(iainfo.nelem,), ctype)
		sht.analysis_2d(alm=alm, map=prof, spin=int(s), lmax=lmax, mmax=0, geometry=geometry)
		if not norot: alm = transfer_alm(iainfo, alm, oainfo)
		oalm[I] = alm
	return oalm

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

def _transpose_index(ainfo, device=None):
	"""Source and destination positions of alm_info.transpose_alm (cmisc_core.c:116-135): the k-th
	element in storage order (m-major) moves to the k-th (l,m) pair in l-major order.  With `device` the index arrays are built
	there with torch (50 M entries at lmax 10^4: 0.6 s with numpy on the host)."""
	lmax, mmax = ainfo.lmax, ainfo.mmax
	if device is None:
		arange, repeat, cumsum, cat, minimum = np.arange, np.repeat, np.cumsum, np.concatenate, np.minimum
		ms = ainfo.mstart.astype(np.int64); zero = np.zeros(1, np.int64)
	else:
		torch = _torch()
		arange = lambda n: torch.arange(n, device=device, dtype=torch.int64)
		repeat, cumsum, cat = torch.repeat_interleave, (lambda x: torch.cumsum(x, 0)), torch.cat
		minimum = lambda a, b: torch.clamp(a, max=b)
		ms = torch.as_tensor(ainfo.mstart.astype(np.int64), device=device); zero = torch.zeros(1, device=device, dtype=torch.int64)
	per_m = lmax+1-arange(mmax+1)                                   # number of l per m
	m_src = repeat(arange(mmax+1), per_m)
	l_src = arange(len(m_src))-repeat(cat([zero, cumsum(per_m)[:-1]]), per_m)+m_src
	per_l = minimum(arange(lmax+1), mmax)+1                         # number of m per l
	l_dst = repeat(arange(lmax+1), per_l)
	m_dst = arange(len(l_dst))-repeat(cat([zero, cumsum(per_l)[:-1]]), per_l)
	return ms[m_src]+l_src*ainfo.stride, ms[m_dst]+l_dst*ainfo.stride

def transpose_alm(ainfo, alm, out=None):
	"""alm_info.transpose_alm (curvedsky.py:443-451): reorder numbers generated in l-major order into
	the m-major layout.  alm is out is allowed."""
	if _is_tensor(alm):
		src, dst = _transpose_index(ainfo, alm.device)
		if out is None: out = alm.clone()
	else:
		src, dst = _transpose_index(ainfo)
		if out is None: out = alm.copy()
	vals = alm[..., src]
	out[..., dst] = vals
	return out
alm_info.transpose_alm = lambda self, alm, out=None: transpose_alm(self, alm, out=out)

def rand_alm_white(ainfo, pre=None, alm=None, seed=None, dtype=np.complex128, m_major=True, rng="numpy"):
	"""unit-variance complex Gaussian numbers for every alm slot.  To give the same alm as pixell for a seed (curvedsky.py:602-628)
	the numbers come from numpy's legacy global RNG, are drawn in l-major order ((l,m) = (0,0), (1,0), (1,1), ...) and are
	then moved to the m-major layout unless m_major is False.
	rng="device" (ours, for throughput: Monte-Carlo loops spend 1.4 s per lmax-10^4 realisation in the host generator): the numbers
	are drawn on the GPU (torch's Philox generator seeded with `seed`) straight into the m-major layout and stay there (a CUDA
	tensor is returned) -- same distribution, NOT the reference's numbers for a seed."""
	if rng == "device":
		torch = _torch()
		if alm is not None: raise ValueError("rng='device' allocates its own (device) alm")
		g = torch.Generator(device="cuda")
		if seed is not None: g.manual_seed(int(seed))
		else: g.seed()
		shape = (tuple(pre) if pre is not None else ())+(ainfo.nelem, 2)
		rt = torch.float32 if np.dtype(dtype) == np.dtype(np.complex64) else torch.float64
		return torch.view_as_complex(torch.randn(shape, generator=g, dtype=rt, device="cuda"))
	if rng != "numpy": raise ValueError("rng must be 'numpy' or 'device'")
	if seed is not None: np.random.seed(seed)
	if alm is None: alm = np.empty((tuple(pre) if pre is not None else ())+(ainfo.nelem,), dtype)
	fill_gauss(alm)
	if not m_major: return alm
	from . import _lib
	if not _lib.is_hostsim() and alm.nbytes >= (1 << 24):     # large alm: the reordering as one gather on the GPU (0.6 s of host indexing at lmax 10^4)
		torch = _torch()
		alm[...] = ainfo.transpose_alm(torch.from_numpy(alm).cuda()).cpu().numpy()
		return alm
	return ainfo.transpose_alm(alm, alm)

def rand_alm(ps, ainfo=None, lmax=None, seed=None, dtype=np.complex128, m_major=True, return_ainfo=False, rng="numpy"):
	"""Gaussian alm with (cross) spectrum ps [nl], [nspec,nl] or [ncomp,ncomp,nl] (curvedsky.rand_alm, curvedsky.py:61-79):
	white numbers (rand_alm_white), coloured by the matrix square root of the spectrum on the GPU (alm_info.lmul); the factor
	1/sqrt(2) shares the variance between real and imaginary parts, m = 0 is made real with the full variance.
	rng="device": see rand_alm_white (device-resident result, device generator)."""
	ps = np.asarray(ps)
	wps, ainfo = prepare_ps(ps, ainfo=ainfo, lmax=lmax)
	white = rand_alm_white(ainfo, pre=[wps.shape[0]], seed=seed, dtype=dtype, m_major=m_major, rng=rng)
	colour = (_multi_sqrt(wps)/np.sqrt(2.0)).astype(real_dtype(dtype), copy=False)
	alm = ainfo.lmul(white, colour)
	if _is_tensor(alm):
		alm[:, :ainfo.lmax+1] = (alm[:, :ainfo.lmax+1].real*np.sqrt(2.0)).to(alm.dtype)
	else:
		m0 = alm[:, :ainfo.lmax+1]                        # the m = 0 column comes first in the m-major layout
		m0.imag = 0; m0.real *= np.sqrt(2.0)
	alm = alm[0] if ps.ndim == 1 else alm
	return (alm, ainfo) if return_ainfo else alm

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
