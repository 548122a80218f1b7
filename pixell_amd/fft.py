"""pixell.fft-shaped front end of the HIP FFT engine (pxf_fft_nd).

Mirrors pixell/fft.py: fft (fft.py:133-156), ifft (:158-184), rfft (:186-195), irfft (:197-209)
and the engine protocol `engines[name].FFTW(a, b, axes, direction, threads, flags)()` /
`.empty_aligned` (fft.py:8-31, 78-82).  The transform kind is inferred from the shapes and
dtypes of input and output exactly as the reference's numpy engine does.  Arrays may be numpy
(staged through the GPU) or torch CUDA tensors (in place, no copies).
`register()` installs the engine into an imported `pixell.fft` as engines["hip"]."""
import ctypes
import numpy as np
from . import _lib
from .sht import _Buf, _np_dtype, _is_tensor, _torch, _DT, current_stream, device_index

def astuple(x):
	try: return tuple(x)
	except TypeError: return (x,)

def _empty_like(shape, dtype, like):
	if _is_tensor(like):
		torch = _torch()
		return torch.empty(tuple(shape), dtype=getattr(torch, np.dtype(dtype).name), device=like.device)
	return np.empty(shape, dtype)

def _strides(x):
	if _is_tensor(x): return list(x.stride())
	return [s//x.itemsize for s in x.strides]

def _exec(a, b, axes, forward, scale, r2r=0):
	"""a -> b along axes; kind from shapes/dtypes (fft.py:14-31)"""
	ad, bd = _np_dtype(a), _np_dtype(b)
	nd = a.ndim
	axes = [ax % nd for ax in astuple(axes)]
	ashape, bshape = tuple(a.shape), tuple(b.shape)
	half = lambda shp: tuple(n//2+1 if i == axes[-1] else n for i, n in enumerate(shp))
	if r2r:
		if ad.kind == "c" or bd.kind == "c" or ashape != bshape: raise ValueError("dct: real arrays of equal shape are needed")
		kind, shape = r2r, ashape
	elif bd.kind == "c" and ashape == bshape:
		kind, shape = 0, ashape
	elif bd.kind == "c":
		if ad.kind == "c" or bshape != half(ashape): raise ValueError("r2c: output shape %s does not match input %s" % (str(bshape), str(ashape)))
		kind, shape = 1, ashape
	else:
		if ad.kind != "c" or ashape != half(bshape): raise ValueError("c2r: input shape %s does not match output %s" % (str(ashape), str(bshape)))
		kind, shape = 2, bshape
	# numpy inputs are staged contiguously; tensors are used with their own strides
	ab = _Buf(a) if not _is_tensor(a) else None
	bb = _Buf(b, writeback=True) if not _is_tensor(b) else None
	def cstr(shp):
		st = [1]*len(shp)
		for i in range(len(shp)-2, -1, -1): st[i] = st[i+1]*shp[i+1]
		return st
	ist = cstr(ashape) if ab is not None else _strides(a)
	ost = cstr(bshape) if bb is not None else _strides(b)
	aptr = ab.ptr if ab is not None else a.data_ptr()
	bptr = bb.ptr if bb is not None else b.data_ptr()
	I64 = ctypes.c_int64
	sh = (I64*nd)(*shape); isa = (I64*nd)(*ist); osa = (I64*nd)(*ost); ax = (ctypes.c_int*len(axes))(*axes)
	_lib.check(_lib.load().pxf_fft_nd(nd, sh, isa, osa, len(axes), ax, kind, int(bool(forward)), float(scale),
		_DT[ad], _DT[bd], aptr, bptr, device_index(), current_stream()))
	if bb is not None: bb.finish()
	return b

def fft(tod, ft=None, nthread=0, axes=[-1], flags=None, _direction="FFTW_FORWARD", engine="auto", _scale=1.0):
	"""pixell.fft.fft: unnormalised forward transform of tod into ft (complex of tod.shape if omitted)"""
	axes = astuple(-1 if axes is None else axes)
	if int(np.prod(tod.shape)) == 0: return
	if ft is None:
		ft = _empty_like(tod.shape, np.result_type(_np_dtype(tod), 0j), tod)
	return _exec(tod, ft, axes, _direction == "FFTW_FORWARD", _scale)

def ifft(ft, tod=None, nthread=0, normalize=False, axes=[-1], flags=None, engine="auto", _scale=1.0):
	"""pixell.fft.ifft: unnormalised backward transform unless normalize=True"""
	axes = astuple(-1 if axes is None else axes)
	if int(np.prod(ft.shape)) == 0: return
	if tod is None: tod = _empty_like(ft.shape, _np_dtype(ft), ft)
	scale = _scale
	if normalize: scale = scale/np.prod([tod.shape[i] for i in axes])
	return _exec(ft, tod, axes, False, scale)

def rfft_shape(shape, axes=[-1]):
	s = list(shape); s[astuple(axes)[-1]] = s[astuple(axes)[-1]]//2+1
	return tuple(s)
def irfft_shape(shape, axes=[-1], n=None):
	s = list(shape); ax = astuple(axes)[-1]
	s[ax] = n if n is not None else (s[ax]-1)*2
	return tuple(s)

def rfft(tod, ft=None, nthread=0, axes=[-1], flags=None, engine="auto"):
	axes = astuple(-1 if axes is None else axes)
	if ft is None: ft = _empty_like(rfft_shape(tod.shape, axes), np.result_type(_np_dtype(tod), 0j), tod)
	return fft(tod, ft, nthread, axes, flags=flags)

def irfft(ft, tod=None, n=None, nthread=0, normalize=False, axes=[-1], flags=None, engine="auto"):
	axes = astuple(-1 if axes is None else axes)
	if tod is None: tod = _empty_like(irfft_shape(ft.shape, axes, n), np.zeros([], _np_dtype(ft)).real.dtype, ft)
	return ifft(ft, tod, nthread, normalize, axes, flags=flags)

# names, inverses and normalisation offsets as in pixell/fft.py:269-290
_dct_names = {
	"DCT-I": "FFTW_REDFT00", "DCT-II": "FFTW_REDFT10", "DCT-III": "FFTW_REDFT01", "DCT-IV": "FFTW_REDFT11",
	"DST-I": "FFTW_RODFT00", "DST-II": "FFTW_RODFT10", "DST-III": "FFTW_RODFT01", "DST-IV": "FFTW_RODFT11"}
_dct_names.update({v: v for v in list(_dct_names.values())})
_dct_inverses = {"FFTW_REDFT00": "FFTW_REDFT00", "FFTW_REDFT10": "FFTW_REDFT01", "FFTW_REDFT01": "FFTW_REDFT10", "FFTW_REDFT11": "FFTW_REDFT11",
	"FFTW_RODFT00": "FFTW_RODFT00", "FFTW_RODFT10": "FFTW_RODFT01", "FFTW_RODFT01": "FFTW_RODFT10", "FFTW_RODFT11": "FFTW_RODFT11"}
_dct_sizes = {"FFTW_REDFT00": -1, "FFTW_REDFT10": 0, "FFTW_REDFT01": 0, "FFTW_REDFT11": 0, "FFTW_RODFT00": +1, "FFTW_RODFT10": 0, "FFTW_RODFT01": 0, "FFTW_RODFT11": 0}
_r2r_kind = {"FFTW_REDFT00": 3, "FFTW_REDFT10": 4, "FFTW_REDFT01": 5, "FFTW_REDFT11": 6, "FFTW_RODFT00": 7, "FFTW_RODFT10": 8, "FFTW_RODFT01": 9, "FFTW_RODFT11": 10}
def _dct_type(type):
	if type not in _dct_names: raise ValueError("unknown DCT/DST type %s" % str(type))
	return _dct_names[type]
def _asreal(a):
	if _is_tensor(a): return a
	a = np.asarray(a)
	return a if a.dtype.kind == "f" else a.astype(np.result_type(a.dtype, 0.0))

def dct(tod, dt=None, nthread=0, normalize=False, axes=[-1], flags=None, type="DCT-I", engine="auto", _scale=1.0):
	"""pixell.fft.dct (fft.py:211-231): unnormalised DCT / DST of the given type along axes (normalize is ignored there too)"""
	t = _dct_type(type)
	axes = astuple(-1 if axes is None else axes)
	tod = _asreal(tod)
	if dt is None: dt = _empty_like(tod.shape, _np_dtype(tod), tod)
	return _exec(tod, dt, axes, True, _scale, r2r=_r2r_kind[t])

def idct(dt, tod=None, nthread=0, normalize=False, axes=[-1], flags=None, type="DCT-I", engine="auto", _scale=1.0):
	"""pixell.fft.idct (fft.py:233-267): applies the transform that inverts `type` (e.g. DCT-III for DCT-II); normalize divides
	by prod 2(n+d), d = -1 for DCT-I, +1 for DST-I, 0 otherwise"""
	t = _dct_inverses[_dct_type(type)]
	off = _dct_sizes[t]
	axes = astuple(-1 if axes is None else axes)
	dt = _asreal(dt)
	if tod is None: tod = _empty_like(dt.shape, _np_dtype(dt), dt)
	scale = _scale
	if normalize: scale = scale/float(np.prod([2*(dt.shape[i]+off) for i in axes]))
	return _exec(dt, tod, axes, True, scale, r2r=_r2r_kind[t])

def redft00(a, b=None, nthread=0, normalize=False, flags=None, engine="auto"):
	"""pixell.fft.redft00 (fft.py:292-307): DCT-I along the last axis"""
	n = a.shape[-1]
	return dct(a, b, axes=[-1], _scale=1.0/(2*(n-1)) if normalize else 1.0)

def chebt(a, b=None, nthread=0, flags=None, engine="auto"):
	"""pixell.fft.chebt (fft.py:309-313): Chebyshev transform along the last axis"""
	b = redft00(a, b, nthread, normalize=True)
	b[..., 1:-1] *= 2
	return b

def fft_len(n, direction="below", factors=None):
	"""nearest length the engine handles well (2,3,5-smooth), cf. pixell.fft.fft_len (fft.py:319)"""
	n = int(n)
	if direction == "above": return int(_lib.load().pxf_fft_good_size(n))
	m = n
	while m > 1 and int(_lib.load().pxf_fft_good_size(m)) != m: m -= 1
	return m

class HipFFTW:
	"""engine protocol object: plan = engines["hip"].FFTW(a, b, axes=..., direction=...); plan()"""
	def __init__(self, a, b, axes=(-1,), direction="FFTW_FORWARD", threads=1, flags=None, *args, **kwargs):
		self.a, self.b, self.axes, self.direction = a, b, astuple(axes), direction
		self.r2r = 0
		if not isinstance(direction, str):
			# FFTW r2r: one kind per transformed axis (pixell/fft.py:211-267); like the reference's ducc engine only homogeneous lists
			kinds = [_dct_type(d) for d in direction]
			if any(k != kinds[0] for k in kinds): raise ValueError("only homogeneous r2r transforms are supported")
			self.r2r = _r2r_kind[kinds[0]]
		elif direction not in ("FFTW_FORWARD", "FFTW_BACKWARD"): raise ValueError("unknown direction %s" % str(direction))
	def __call__(self, normalise_idft=False):
		if self.r2r:
			_exec(self.a, self.b, self.axes, True, 1.0, r2r=self.r2r)
			return self.b
		fwd = self.direction == "FFTW_FORWARD"
		scale = 1.0
		if not fwd and normalise_idft:
			shp = self.a.shape if tuple(self.a.shape) == tuple(self.b.shape) or _np_dtype(self.b).kind == "c" else self.b.shape
			scale = 1.0/np.prod([shp[i] for i in self.axes])
		_exec(self.a, self.b, self.axes, fwd, scale)
		return self.b

def empty_aligned(shape, dtype, n=None):
	return np.empty(shape, dtype)

class HipEngine: pass
hip_engine = HipEngine()
hip_engine.FFTW = HipFFTW
hip_engine.empty_aligned = empty_aligned

def register(pixell_fft_module, make_default=True):
	"""pixell.fft.engines["hip"] = this engine (pixell/fft.py:5, 78-131)"""
	pixell_fft_module.engines["hip"] = hip_engine
	if make_default: pixell_fft_module.set_engine("hip")
	return hip_engine

# ---------------------------------------------------------------------------------------
# shift / resample (pixell/fft.py:347-420): Fourier shifting and resizing on top of the transforms above
# ---------------------------------------------------------------------------------------
def fftfreq(n, d=1.0, dtype=np.float64): return np.fft.fftfreq(n, d=d).astype(dtype, copy=False)
def rfftfreq(n, d=1.0, dtype=np.float64): return np.arange(n//2+1, dtype=dtype)/(n*d)

def _to_dev(a):
	"""contiguous device copy of a numpy array (tensors pass through); second value: was it a host array"""
	if _is_tensor(a): return a, False
	a = np.ascontiguousarray(a)
	if _lib.is_hostsim(): return a, False
	device_index()
	import torch
	return torch.from_numpy(a).cuda(), True
def _to_host(a, was_host): return a.cpu().numpy() if was_host else a

def _mul_axis(ca, ax, vec):
	"""ca *= vec along axis ax (ca: contiguous complex array on the device / in the simulator)"""
	n = ca.shape[ax]; inner = int(np.prod(ca.shape[ax+1:], dtype=np.int64)); total = int(np.prod(ca.shape, dtype=np.int64))
	v = _Buf(np.ascontiguousarray(vec, dtype=np.complex128))
	ptr = ca.data_ptr() if _is_tensor(ca) else ca.ctypes.data
	_lib.check(_lib.load().pxm_mul_axis(total, n, inner, ptr, _DT[_np_dtype(ca)], v.ptr, device_index(), current_stream()))

def shift(a, shift, axes=None, nofft=False, deriv=None, engine="auto"):
	"""pixell.fft.shift (fft.py:347-368): shift a by a (fractional) number of samples along the given axes through phase
	ramps in Fourier space; deriv = i differentiates along the i-th listed axis as well"""
	host_in = not _is_tensor(a)
	if host_in: a = np.asanyarray(a)
	iscomplex = _np_dtype(a).kind == "c"
	ctype = np.result_type(_np_dtype(a), np.complex64)
	shift_ = np.atleast_1d(shift)
	if axes is None: axes = range(-len(shift_), 0)
	axes = astuple(axes)
	if host_in: ca = np.ascontiguousarray(a, dtype=ctype)+0
	else:
		import torch
		ca = a.to(getattr(torch, np.dtype(ctype).name)).contiguous().clone()
	ca, was_host = _to_dev(ca)
	fa = fft(ca, axes=axes) if not nofft else ca
	for i, ax in enumerate(axes):
		ax %= fa.ndim
		freqs = fftfreq(fa.shape[ax])
		phase = np.exp(-2j*np.pi*freqs*shift_[i])
		if deriv == i: phase = phase*(-2j*np.pi*freqs)
		_mul_axis(fa, ax, phase)
	res = ifft(fa, axes=axes, normalize=True) if not nofft else fa
	res = _to_host(res, was_host)
	return res if iscomplex else res.real

def resample_fft(fa, n, out=None, axes=-1, norm=1, op=lambda a, b: b):
	"""pixell.fft.resample_fft (fft.py:389-434): pad or truncate the Fourier array fa so that it is the transform of the
	same signal on n samples (pure block copies; works on numpy arrays and on CUDA tensors)"""
	axes = astuple(axes)
	n = np.zeros(len(axes), int)+n
	oshape = list(fa.shape)
	for i, ax in enumerate(axes): oshape[ax] = int(n[i])
	oshape = tuple(oshape)
	if out is None: out = _empty_like(oshape, _np_dtype(fa), fa); out[...] = 0
	elif tuple(out.shape) != oshape:
		raise ValueError("out argument has wrong shape in resample. Expected %s but got %s" % (str(oshape), str(tuple(out.shape))))
	for I in np.ndindex(*([2]*len(axes))):
		sel = [slice(None) for _ in oshape]
		for ai, ax in enumerate(axes):
			c = min(fa.shape[ax], oshape[ax])
			sel[ax] = slice(0, c//2) if I[ai] == 0 else slice(-(c-c//2), None)
		sel = tuple(sel)
		src = fa[sel]*norm if norm != 1 else fa[sel]
		out[sel] = op(out[sel], src)
	return out

def resample(a, n, axes=None, nthread=0, engine="auto"):
	"""pixell.fft.resample (fft.py:370-387): Fourier-resize the given axes of a to length n"""
	host_in = not _is_tensor(a)
	if host_in: a = np.asarray(a)
	n = astuple(n)
	if axes is None: axes = [-len(n)+i for i in range(len(n))]
	axes = astuple(axes)
	if len(n) != len(axes): raise ValueError("Resize size n = %s does not match axes = %s" % (str(n), str(axes)))
	iscomplex = _np_dtype(a).kind == "c"
	da, was_host = _to_dev(a)
	fa = fft(da, axes=axes)
	norm = 1/np.prod([a.shape[ax] for ax in axes])
	fa = resample_fft(fa, n, axes=axes, norm=norm)
	out = ifft(fa, axes=axes, normalize=False)
	out = _to_host(out, was_host)
	return out if iscomplex else out.real

# ---- the registry face of pixell.fft (fft.py:78-131, 313-338): on this backend there is one engine ---------------------------
engines = {"hip": hip_engine}
engine = "hip"
alignment = 32
def set_engine(eng):
	"""select the default engine by name (only "hip" exists here; pixell.fft.register() installs it next to the reference's own)"""
	global engine
	if eng not in engines: raise KeyError("no FFT engine '%s' (available: %s)" % (eng, ", ".join(sorted(engines))))
	engine = eng
def get_engine(eng): return engine if eng == "auto" else eng
def empty(shape, dtype): return engines[engine].empty_aligned(shape, dtype=dtype, n=alignment)
def asfcarray(a):
	"""a as an array of at least float precision (integers become float64)"""
	a = np.asarray(a)
	return np.asarray(a, np.result_type(a, 0.0))
def ichebt(a, b=None, nthread=0, engine="auto"):
	"""inverse of chebt along the last axis: halve the interior coefficients, DCT-I (fft.py:313-317)"""
	a = asfcarray(a).copy()
	a[..., 1:-1] *= 0.5
	return redft00(a, b, nthread)
# bin index <-> frequency of an n-point transform with sample spacing d
def ind2freq(n, i, d=1.0):  return np.where(np.asarray(i) < n/2, i, np.asarray(i)-n)/(d*n)
def int2rfreq(n, i, d=1.0): return np.asarray(i)/(n*d)
def freq2ind(n, f, d=1.0):
	j = np.asarray(f)*(d*n)
	return np.where(j >= 0, j, n+j)
def rfreq2ind(n, f, d=1.0): return np.asarray(f)*(n*d)
def fft_flat(tod, ft, nthread=1, axes=[-1], flags=None, _direction="FFTW_FORWARD"):
	"""pixell's work-around entry for engines that cannot take many leading dimensions (fft.py:669-683); this engine can"""
	return fft(tod, ft, nthread=nthread, axes=axes, flags=flags, _direction=_direction)
def ifft_flat(ft, tod, nthread=1, axes=[-1], flags=None):
	return ifft(ft, tod, nthread=nthread, normalize=False, axes=axes, flags=flags)
