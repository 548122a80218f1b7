"""enmap.fft-sized micro-benchmark of the HIP FFT engine (user bytes read+written per second)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pixell_amd import fft as pfft
def bench(shape, axes, dtype=torch.complex128, reps=3):
	a = torch.randn(shape, dtype=torch.float64, device="cuda").to(dtype) if dtype.is_complex else torch.randn(shape, dtype=dtype, device="cuda")
	b = torch.empty_like(a) if dtype.is_complex else torch.empty(a.shape, dtype=torch.complex128, device="cuda")
	pfft.fft(a, b, axes=axes); torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(reps): pfft.fft(a, b, axes=axes)
	torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/reps
	nbytes = a.numel()*a.element_size()+b.numel()*b.element_size()
	print("%-22s axes=%-8s %-10s %8.3f ms  %7.1f GB/s" % (str(shape), str(axes), str(dtype).split(".")[1], dt*1e3, nbytes/dt/1e9), flush=True)
for shp in [(5400, 10800), (10800, 21600), (21600, 43200)]:
	bench(shp, [-1]); bench(shp, [-2]); bench(shp, [-2, -1]); bench(shp, [-2, -1], torch.float64)
