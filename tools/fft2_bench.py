#!/usr/bin/env python
"""enmap.fft / ifft of one 21600x43200 component on device-resident data (the `fft` block of bench.py alone); PIXELL_AMD_LIB selects a variant build"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixell_amd import enmap, _lib
ny, nx = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (21600, 43200)
shape, wcs = enmap.fullsky_geometry(shape=(ny, nx))
m = enmap.dmap(torch.randn((ny, nx), dtype=torch.float64, device="cuda"), wcs)
def t(fn, reps=5):
	fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
	for _ in range(reps): r = fn()
	torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps*1e3, r
tr, f = t(lambda: enmap.fft(m, normalize=False))
tc, _ = t(lambda: enmap.ifft(f, normalize=False))
print("%s  %dx%d  real->complex %.2f ms  complex->complex %.2f ms" % (os.path.basename(_lib.lib_path()), ny, nx, tr, tc))
