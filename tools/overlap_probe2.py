"""Co-execution probe: a Legendre-only workload (ring plan with 8-pixel rings: the ring FFT is negligible) on one stream
and an FFT-only workload (batched c2c) on another.  Prints each alone and both together."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixell_amd import sht, fft as pfft
SPIN = int(sys.argv[1]) if len(sys.argv) > 1 else 0; ADJ = len(sys.argv) > 2 and sys.argv[2] == "adj"
lmax = 10000; nr = 10082; nphi = 8
theta = (np.arange(nr)+0.5)*np.pi/nr
ms = sht.tri_mstart(lmax); nalm = int(ms[-1])+lmax+1
dev = torch.device("cuda:0")
nc = 1 if SPIN == 0 else 2
alm = torch.randn(nc, nalm, dtype=torch.complex128, device=dev); alm[:, :lmax+1].imag = 0
mp = torch.randn(nc, nr*nphi, dtype=torch.float64, device=dev)
kw = dict(theta=theta, nphi=np.full(nr, nphi, np.uint64), phi0=np.zeros(nr), ringstart=np.arange(nr, dtype=np.uint64)*nphi, lmax=lmax, spin=SPIN)
x = torch.randn(12000, 43200, dtype=torch.complex128, device=dev); y = torch.empty_like(x)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def leg(n=4):
	with torch.cuda.stream(sA):
		for _ in range(n):
			if ADJ: sht.adjoint_synthesis(alm=alm, map=mp, **kw)
			else: sht.synthesis(alm=alm, map=mp, **kw)
def ffts(n=8):
	with torch.cuda.stream(sB):
		for _ in range(n): pfft.fft(x, y, axes=[-1])
def both(): leg(); ffts()
for name, f in [("legendre", leg), ("fft", ffts), ("both", both), ("legendre", leg), ("fft", ffts), ("both", both)]:
	f(); torch.cuda.synchronize()
	t0 = time.perf_counter(); f(); torch.cuda.synchronize()
	print("%-10s %8.1f ms" % (name, (time.perf_counter()-t0)*1e3), flush=True)
