# Row transforms of nrows x 10800 reals through the generic engine (lines in LDS, one pass) -- what a one-pass ring FFT at C2 / C4 would cost -- against
# the chain's two-pass ring stages (tools/chain_lab.py c4: map2leg / h2map of 8 maps 3.5 ms each)
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from pixell_amd import fft
dev = torch.device("cuda")
for nrows, nx in ((8*5400, 10800), (8*5400, 8192), (3*5400, 10800)):
	x = torch.randn((nrows, nx), dtype=torch.float64, device=dev)
	f = torch.empty((nrows, nx//2+1), dtype=torch.complex128, device=dev)
	y = torch.empty_like(x)
	def t(fn, reps=5):
		fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
		for _ in range(reps): fn()
		torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps*1e3
	a = t(lambda: fft.rfft(x, f)); b = t(lambda: fft.irfft(f, y))
	gb = (x.numel()*8+f.numel()*16)/1e9
	err = float((y/nx-x).abs().max())
	print("%d x %d: rfft %.2f ms (%.0f GB/s)  irfft %.2f ms (%.0f GB/s)  round-trip error %.1e" % (nrows, nx, a, gb/a*1e3, b, gb/b*1e3, err), flush=True)
