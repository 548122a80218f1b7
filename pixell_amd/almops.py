"""alm post-processing on the GPU: alm2cl, lmul (almxfl), l-dependent matrix products.

Host mirror of cython/cmisc.pyx:8-110 (alm2cl) and :159-274 (lmul / lmatmul) of the reference, as
reached through alm_info.alm2cl / alm_info.lmul (pixell/curvedsky.py:451-474).  Same argument
meaning, broadcasting and error behaviour; the arithmetic runs in include/pxsht.h pxa_alm2cl /
pxa_lmatmul.  numpy inputs are staged through device memory, torch CUDA tensors are used in place
and the result is then a CUDA tensor.  No CPU fallback.
"""
import ctypes
import numpy as np
from . import _lib
from .sht import _Buf, _is_tensor, _np_dtype, _torch, _DT, device_index, current_stream

def _alloc_like(ref, shape, dtype):
	"""uninitialised array of the kind `ref` is (torch CUDA tensor or numpy)"""
	if _is_tensor(ref):
		torch = _torch()
		tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
			np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128}[np.dtype(dtype)]
		return torch.empty(tuple(shape), dtype=tdt, device=ref.device)
	return np.empty(tuple(shape), dtype)

def _astype(x, dtype):
	if _np_dtype(x) == np.dtype(dtype): return x
	if _is_tensor(x):
		torch = _torch()
		return x.to({np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128,
			np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}[np.dtype(dtype)])
	return x.astype(dtype)

def _flat2(x):
	"""[..., n] -> contiguous [npre, n] (view when possible)"""
	n = x.shape[-1]
	if _is_tensor(x): return x.contiguous().reshape(-1, n)
	return np.ascontiguousarray(x).reshape(-1, n)

def _mstart_buf(ainfo):
	return _Buf(np.ascontiguousarray(ainfo.mstart[:ainfo.mmax+1], dtype=np.uint64))

def alm2cl(ainfo, alm, alm2=None, cl_dtype=None):
	"""cmisc.alm2cl (cython/cmisc.pyx:8-110): cross spectrum of alm and alm2 (which broadcast); the
	result has the broadcast leading shape + (lmax+1,).  Each distinct pair of rows is computed once."""
	if not _is_tensor(alm): alm = np.asarray(alm)
	same = alm2 is None or alm2 is alm
	if same: alm2 = alm
	elif not _is_tensor(alm2): alm2 = np.asarray(alm2)
	if _is_tensor(alm) != _is_tensor(alm2): raise ValueError("alm and alm2 must both be numpy arrays or both torch tensors")
	ctype = np.result_type(_np_dtype(alm), _np_dtype(alm2))
	if ctype not in (np.dtype(np.complex64), np.dtype(np.complex128)):
		raise ValueError("alm2cl requires complex64 or complex128 arrays")
	rtype = np.dtype(np.float32) if ctype == np.dtype(np.complex64) else np.dtype(np.float64)
	cl_dtype = rtype if cl_dtype is None else np.dtype(cl_dtype)
	if cl_dtype not in (np.dtype(np.float32), np.dtype(np.float64)): raise ValueError("cl dtype must be float32 or float64")
	if ctype == np.dtype(np.complex128) and cl_dtype == np.dtype(np.float32):
		raise ValueError("float32 spectra of double precision alm are not supported")   # cmisc.pyx:84
	alm = _astype(alm, ctype); alm2 = alm if same else _astype(alm2, ctype)
	if alm.shape[-1] < ainfo.nelem or alm2.shape[-1] < ainfo.nelem: raise ValueError("alm too short for this alm_info")
	pshape = np.broadcast_shapes(tuple(alm.shape[:-1]), tuple(alm2.shape[:-1]))
	# row index of each broadcast element in the flattened inputs
	i1 = np.broadcast_to(np.arange(int(np.prod(alm.shape[:-1], dtype=int))).reshape(alm.shape[:-1]), pshape).reshape(-1)
	i2 = np.broadcast_to(np.arange(int(np.prod(alm2.shape[:-1], dtype=int))).reshape(alm2.shape[:-1]), pshape).reshape(-1)
	f1 = _flat2(alm); f2 = f1 if same else _flat2(alm2)
	b1 = _Buf(f1); b2 = b1 if same else _Buf(f2)
	npre = len(i1); nl = ainfo.lmax+1
	cl = _alloc_like(alm, (npre, nl), cl_dtype)
	bc = _Buf(cl, writeback=True)
	if bc.tmp is not None and not _lib.is_hostsim(): bc.tmp = _torch().empty((npre, nl), dtype=bc.tmp.dtype, device="cuda"); bc.ptr = bc.tmp.data_ptr()
	ms = _mstart_buf(ainfo)
	lib = _lib.load(); dev = device_index(); st = current_stream()
	csz = ctype.itemsize; rsz = cl_dtype.itemsize; n1 = f1.shape[-1]; n2 = f2.shape[-1]
	done = {}
	copies = []
	for i in range(npre):
		key = (int(i1[i]), int(i2[i])) if not same else tuple(sorted((int(i1[i]), int(i2[i]))))
		if key in done: copies.append((i, done[key])); continue
		done[key] = i
		_lib.check(lib.pxa_alm2cl(ainfo.lmax, ainfo.mmax, ms.ptr, ainfo.stride, b1.ptr + int(i1[i])*n1*csz, b2.ptr + int(i2[i])*n2*csz,
			_DT[ctype], bc.ptr + i*nl*rsz, _DT[cl_dtype], dev, st))
	tgt = bc.tmp if bc.tmp is not None else cl
	for i, j in copies: tgt[i] = tgt[j]
	bc.finish()
	return cl.reshape(tuple(pshape)+(nl,))

def lmul(ainfo, alm, lfun, out=None):
	"""cmisc.lmul (cython/cmisc.pyx:159-197): res[...,lm] = lfun[...,l] alm[...,lm] with broadcasting of the
	leading axes, or, for lfun[a,b,l] with alm[b,lm], the matrix product res[a,lm] = sum_b lfun[a,b,l] alm[b,lm].
	lfun shorter than lmax+1 counts as zero beyond its end."""
	tens = _is_tensor(alm)
	if not tens: alm = np.asarray(alm)
	ctype = np.result_type(_np_dtype(alm), np.complex64)
	if ctype not in (np.dtype(np.complex64), np.dtype(np.complex128)): raise ValueError("lmul requires complex64 or complex128 arrays")
	alm = _astype(alm, ctype)
	if alm.shape[-1] < ainfo.nelem: raise ValueError("alm too short for this alm_info")
	# the filter is small: keep it on the host in f64 and upload once per call
	if _is_tensor(lfun): lfun = lfun.detach().cpu().numpy()
	lfun = np.asarray(lfun, dtype=np.float64)
	if ctype == np.dtype(np.complex64): lfun = lfun.astype(np.float32).astype(np.float64)   # the reference casts the filter to the alm's real type
	if out is not None and (_np_dtype(out) != ctype or _is_tensor(out) != tens):
		raise ValueError("lmul's out argument must be contiguous along last axis, and have the same dtype as alm")
	lib = _lib.load(); dev = device_index(); st = current_stream(); ms = _mstart_buf(ainfo)
	nalm = alm.shape[-1]; csz = ctype.itemsize
	if lfun.ndim == 3 and alm.ndim == 2:
		N, M, nl = lfun.shape
		if M != alm.shape[0]: raise ValueError("lmul: matrix shape %s does not match alm shape %s" % (str(lfun.shape), str(alm.shape)))
		if M > 8: raise ValueError("lmul: at most 8 input components")
		if out is None: out = _alloc_like(alm, (N, nalm), ctype); _zero(out)
		bi = _Buf(_flat2(alm)); bo = _Buf(out, writeback=True); bl = _Buf(np.ascontiguousarray(lfun))
		_lib.check(lib.pxa_lmatmul(N, M, ainfo.lmax, ainfo.mmax, ms.ptr, ainfo.stride, bi.ptr, nalm, bo.ptr, nalm, _DT[ctype], bl.ptr, nl, dev, st))
		bo.finish()
		return out
	try:
		pre = np.broadcast_shapes(tuple(alm.shape[:-1]), lfun.shape[:-1])
	except ValueError:
		raise ValueError("lmul's alm and lfun's dimensions must either broadcast (when ignoring the last dimension), or have shape compatible with a matrix product (again ignoring the last dimension)")
	npre = int(np.prod(pre, dtype=int)); nl = lfun.shape[-1]
	ia = np.broadcast_to(np.arange(int(np.prod(alm.shape[:-1], dtype=int))).reshape(alm.shape[:-1]), pre).reshape(-1)
	il = np.broadcast_to(np.arange(int(np.prod(lfun.shape[:-1], dtype=int))).reshape(lfun.shape[:-1]), pre).reshape(-1)
	fa = _flat2(alm); fl = np.ascontiguousarray(lfun).reshape(-1, nl)
	if out is not None and tuple(out.shape) != tuple(pre)+(nalm,): raise ValueError("lmul: out has the wrong shape")
	bl = _Buf(fl)
	# work[npre, nalm] starts as the (broadcast) input rows and is scaled in place, like the reference's
	# `out[:] = aflat` followed by lmul_dp on each row
	if _lib.is_hostsim():
		work = np.ascontiguousarray((fa.cpu().numpy() if tens else fa)[ia]); ptr = work.ctypes.data
	else:
		torch = _torch()
		dev_in = fa if tens else torch.from_numpy(fa if fa.flags.writeable else fa.copy()).cuda()
		work = dev_in[torch.as_tensor(np.array(ia), device=dev_in.device)]          # gather = fresh contiguous copy (np.array: broadcast index views are read-only)
		ptr = work.data_ptr()
	for i in range(npre):
		row = ptr + i*nalm*csz
		_lib.check(lib.pxa_lmatmul(1, 1, ainfo.lmax, ainfo.mmax, ms.ptr, ainfo.stride, row, nalm, row, nalm, _DT[ctype], bl.ptr + int(il[i])*nl*8, nl, dev, st))
	shape = tuple(pre)+(nalm,)
	if tens:
		if _lib.is_hostsim(): work = _torch().from_numpy(work)
		if out is None: return work.reshape(shape)
		out.copy_(work.reshape(shape)); return out
	if not _lib.is_hostsim(): work = work.cpu().numpy()
	if out is None: return work.reshape(shape)
	out[...] = work.reshape(shape); return out

def _zero(x):
	if _is_tensor(x): x.zero_()
	else: x[...] = 0
