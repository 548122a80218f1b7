"""Host-array route: numpy (pageable) arrays <-> device memory through pinned double-buffered slabs.

The reference's callers hold numpy arrays (pixell/curvedsky.py:1429-1446 strips every input to np.asarray before ducc sees it).
A 3 x 21600 x 43200 float64 map is 22.4 GB: staged in one synchronous pageable copy it costs more than the transform.  Here a
large C-contiguous array moves in slabs: a host thread copies slab k+1 into one of two pinned buffers while the DMA engine sends
slab k from the other (uploads), or drains the pinned buffer of slab k-1 into the array while slab k arrives (downloads).  All
DMA goes on ONE private copy stream; callers order it against their compute stream with the returned event.

`Pipeline` (used by pixell_amd.curvedsky for its 2-d transforms) adds the overlap between the spin groups of one call: every
numpy input of the call is uploaded by a background thread in the order the groups will need them, and every output is
downloaded in the background as soon as its group is done -- T's transform runs under the upload of Q and U, the download of the
T map under the synthesis of Q and U.  Results are the bytes the device produced: nothing is recomputed on the host.
"""
import os, threading, queue
import numpy as np

SLAB_BYTES = int(float(os.environ.get("PIXELL_AMD_SLAB_MB", "256"))*(1 << 20))
MIN_BYTES = int(float(os.environ.get("PIXELL_AMD_PIPE_MIN_MB", "64"))*(1 << 20))      # smaller arrays take the plain one-shot copy

def _torch():
	import torch
	return torch

_lock = threading.Lock()
_state = {}
class _State:
	def __init__(self):
		torch = _torch()
		self.slab = SLAB_BYTES
		self.stream = torch.cuda.Stream()
		self.pinned = [torch.empty(SLAB_BYTES, dtype=torch.uint8).pin_memory() for _ in range(2)]
		self.pinned_np = [p.numpy() for p in self.pinned]
		self.free = [torch.cuda.Event(), torch.cuda.Event()]      # recorded after the DMA that last used the buffer
		self.used = [False, False]
		self.io_lock = threading.Lock()                            # one transfer at a time owns the two buffers

def _st():
	"""copy stream and pinned buffers of the calling thread's current device"""
	dev = _torch().cuda.current_device()
	with _lock:
		if dev not in _state or _state[dev].slab != SLAB_BYTES: _state[dev] = _State()
		return _state[dev]

def eligible(arr):
	"""large, C-contiguous, native byte order: what the slab route handles (everything else keeps the one-shot copy)"""
	return isinstance(arr, np.ndarray) and arr.nbytes >= MIN_BYTES and arr.flags.c_contiguous and arr.dtype.isnative and arr.dtype.kind in "fc"

def _bytes_view(arr):
	return arr.reshape(-1).view(np.uint8)

# host side of a slab: pageable <-> pinned, split over a few threads (one memcpy runs at ~10 GB/s, the link at ~55; numpy releases the GIL)
_NTHR = max(1, min(int(os.environ.get("PIXELL_AMD_COPY_THREADS", "12")), os.cpu_count() or 1))
_pool = None
def _par_copy(dst, src):
	"""dst[...] = src for 1-d uint8 numpy views of equal length"""
	global _pool
	n = dst.shape[0]
	if _NTHR == 1 or n < (8 << 20): np.copyto(dst, src); return
	if _pool is None:
		import concurrent.futures
		_pool = concurrent.futures.ThreadPoolExecutor(_NTHR)
	step = ((n + _NTHR - 1)//_NTHR + 4095) & ~4095
	list(_pool.map(lambda o: np.copyto(dst[o:o+step], src[o:o+step]), range(0, n, step)))

def upload(arr, dst=None, after=None):
	"""numpy array -> device tensor of the same shape and dtype (allocated if dst is None).  Returns (tensor, event): the event is
	recorded on the copy stream after the last slab; `after`: an event the first DMA waits for (e.g. the consumer of a reused dst)."""
	torch = _torch(); S = _st()
	tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64, np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128}[arr.dtype]
	if dst is None: dst = torch.empty(arr.shape, dtype=tdt, device="cuda")
	dst.record_stream(S.stream)            # (allocated on the caller's stream, written on the copy stream)
	src = _bytes_view(arr); out = dst.view(-1).view(torch.uint8)
	n = src.shape[0]
	with S.io_lock:
		if after is not None: S.stream.wait_event(after)
		k = 0
		for o in range(0, n, SLAB_BYTES):
			m = min(SLAB_BYTES, n - o); b = k & 1
			if S.used[b]: S.free[b].synchronize()
			_par_copy(S.pinned_np[b][:m], src[o:o+m])             # host memcpy; the other buffer's DMA runs meanwhile
			with torch.cuda.stream(S.stream):
				out[o:o+m].copy_(S.pinned[b][:m], non_blocking=True)
				S.free[b].record(S.stream)
			S.used[b] = True; k += 1
		ev = torch.cuda.Event(); ev.record(S.stream)
	return dst, ev

def download(src, arr, after=None):
	"""device tensor -> numpy array (same bytes).  Blocks until the array is complete.  `after`: event the first DMA waits for (the
	producer of src)."""
	torch = _torch(); S = _st()
	src.record_stream(S.stream)
	dstb = _bytes_view(arr); inb = src.contiguous().view(-1).view(torch.uint8)
	n = dstb.shape[0]
	with S.io_lock:
		if after is not None: S.stream.wait_event(after)
		pend = None; k = 0
		for o in range(0, n, SLAB_BYTES):
			m = min(SLAB_BYTES, n - o); b = k & 1
			if S.used[b]: S.free[b].synchronize()                 # (an earlier upload may still read the buffer)
			with torch.cuda.stream(S.stream):
				S.pinned[b][:m].copy_(inb[o:o+m], non_blocking=True)
				S.free[b].record(S.stream)
			S.used[b] = True
			if pend is not None:                                  # drain the previous slab while this one arrives
				pb, po, pm = pend; S.free[pb].synchronize(); _par_copy(dstb[po:po+pm], S.pinned_np[pb][:pm])
			pend = (b, o, m); k += 1
		if pend is not None:
			pb, po, pm = pend; S.free[pb].synchronize(); _par_copy(dstb[po:po+pm], S.pinned_np[pb][:pm])
	return arr

PREFETCH_DEPTH = 2      # inputs uploaded ahead of their transform

def _key(arr):
	return (arr.__array_interface__["data"][0], arr.shape, arr.strides, arr.dtype.str)

class Pipeline:
	"""Background transfers for the numpy arrays of ONE API call (see the module docstring).  prefetch(arrs): uploads start now, in
	order; take(arr) -> (device tensor, event) or None; writeback(tensor, arr, event): download in the background; close(): wait."""
	def __init__(self):
		self.up = {}; self.jobs = queue.Queue(); self.err = None; self.device = _torch().cuda.current_device()
		# at most PREFETCH_DEPTH uploaded inputs wait on the device for their transform (a call with many spin groups would otherwise hold all
		# its inputs AND outputs at once); an upload that does not fit is left to the caller's synchronous path
		self.slots = threading.Semaphore(PREFETCH_DEPTH)
		self.closing = False      # set by close(): queued uploads nobody will take are dropped instead of waiting for a slot
		self.thread = threading.Thread(target=self._run, daemon=True); self.thread.start()
	def _run(self):
		_torch().cuda.set_device(self.device)      # (the current device is per thread)
		while True:
			job = self.jobs.get()
			if job is None: return
			kind, a, b, c, done = job
			try:
				if kind == "up":
					# wait for a slot, but never for ever: if the call ended (an exception between prefetch and take), nobody frees one
					while not self.slots.acquire(timeout=0.05):
						if self.closing: break
					if self.closing: done.result = None; continue
					ok = False
					try: done.result = upload(a); ok = True
					except _torch().cuda.OutOfMemoryError:      # no room to run ahead: take() returns None and the caller uploads when it gets there
						done.result = None; _torch().cuda.empty_cache()
					finally:
						if not ok: self.slots.release()      # (any failure gives the slot back, not only running out of memory)
				else: download(a, b, after=c)
			except BaseException as e:      # noqa: surfaced by close() / take()
				self.err = e
			finally: done.set()
	def prefetch(self, arrs):
		for a in arrs:
			if not eligible(a) or _key(a) in self.up: continue
			done = threading.Event(); done.result = None
			self.up[_key(a)] = done; self.jobs.put(("up", a, None, None, done))
	def take(self, arr):
		done = self.up.pop(_key(arr), None) if isinstance(arr, np.ndarray) else None
		if done is None: return None
		done.wait()
		if self.err is not None: raise self.err
		if done.result is not None: self.slots.release()
		return done.result
	def writeback(self, tensor, arr, event):
		done = threading.Event(); self.jobs.put(("down", tensor, arr, event, done)); return done
	def close(self):
		self.closing = True
		self.jobs.put(None); self.thread.join()
		if self.err is not None: raise self.err
