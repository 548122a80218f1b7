"""Does a memory-bound chain (ring FFT + resample) overlap with an FP64-bound Legendre chain when they are issued on
two HIP streams?  T (spin 0) round trip on stream A and Q/U (spin 2) round trip on stream B, two plans (own scratch),
against the same work issued back to back on one stream.  Usage: python tools/overlap_probe.py [c2|c3]"""
import sys, time, ctypes, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixell_amd import sht, _lib
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
ny, nx, lmax = dict(c2=(5400, 10800, 4000), c3=(21600, 43200, 10000))[cfg]
dev = torch.device("cuda:0")
ms = sht.tri_mstart(lmax); nalm = int(ms[-1])+lmax+1
def mkplan():
	h = ctypes.c_void_p()
	_lib.check(_lib.load().pxs_plan_grid2d(ctypes.byref(h), b"F1", ny, nx, 0.0, 0, 0, lmax, lmax, ms.ctypes.data, 1, 0))
	return sht.Plan(h)
pa, pb = mkplan(), mkplan()
g = torch.Generator(device=dev); g.manual_seed(1)
almT = torch.randn(1, nalm, dtype=torch.complex128, device=dev); almP = torch.randn(2, nalm, dtype=torch.complex128, device=dev)
almT[:, :lmax+1].imag = 0; almP[:, :lmax+1].imag = 0
mapT = torch.empty(1, ny, nx, dtype=torch.float64, device=dev); mapP = torch.empty(2, ny, nx, dtype=torch.float64, device=dev)
lib = _lib.load()
def rt(plan, spin, alm, mp, stream):
	st = ctypes.c_void_p(stream.cuda_stream)
	_lib.check(lib.pxs_synthesis(plan.handle, spin, 0, 0, alm.data_ptr(), 3, nalm, mp.data_ptr(), 1, ny*nx, st))
	_lib.check(lib.pxs_analysis(plan.handle, spin, 0, mp.data_ptr(), 1, ny*nx, alm.data_ptr(), 3, nalm, st))
sA, sB = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
def serial():
	rt(pa, 0, almT, mapT, sA); rt(pb, 2, almP, mapP, sA)
def conc():
	rt(pa, 0, almT, mapT, sA); rt(pb, 2, almP, mapP, sB)
def only(which):
	if which == 0: rt(pa, 0, almT, mapT, sA)
	else: rt(pb, 2, almP, mapP, sB)
for name, f in [("serial", serial), ("concurrent", conc), ("T only", lambda: only(0)), ("QU only", lambda: only(1)), ("concurrent", conc), ("serial", serial)]:
	f(); torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(3): f()
	torch.cuda.synchronize()
	print("%-12s %8.1f ms per round trip" % (name, (time.perf_counter()-t0)/3*1e3), flush=True)
