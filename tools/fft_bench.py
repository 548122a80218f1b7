"""micro-benchmark of the HIP FFT engine on device-resident data: GB/s (read+write of the user arrays)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pixell_amd import fft as pfft
def bench(shape, axes, dtype=torch.complex128, reps=5):
	a = torch.randn(shape, dtype=torch.float64, device="cuda").to(dtype) if dtype.is_complex else torch.randn(shape, dtype=dtype, device="cuda")
	if dtype.is_complex: b = torch.empty_like(a)
	else: b = torch.empty(pfft.rfft_shape(shape, axes), dtype=torch.complex128, device="cuda")
	pfft.fft(a, b, axes=axes); torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(reps): pfft.fft(a, b, axes=axes)
	torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/reps
	nbytes = a.numel()*a.element_size()+b.numel()*b.element_size()
	print("%-22s axes=%-8s %-10s %8.3f ms  %7.1f GB/s" % (str(shape), str(axes), str(dtype).split(".")[1], dt*1e3, nbytes/dt/1e9), flush=True)
for n in [64, 200, 216, 256, 512, 1024, 2048]:
	bench((2**24//n, n), [-1])
for n in [4096, 8100, 10800, 19200, 43200, 64000]:
	bench((max(2**25//n, 64), n), [-1])
bench((4096, 4096), [-2])
bench((4096, 4096), [-2, -1])
bench((5400, 10800), [-2, -1])
bench((3, 5400, 10800), [-1], torch.float64)
bench((3, 5400, 10800), [-2, -1], torch.float64)
