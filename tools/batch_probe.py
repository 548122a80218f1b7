"""BASELINE config 4 on one GPU's share: 8 scalar maps 5400x10800, lmax 4000, one map2alm + alm2map call on the batch"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pixell_amd import curvedsky, enmap
nb, ny, nx, lmax = 8, 5400, 10800, 4000
shape, wcs = enmap.fullsky_geometry(shape=(ny, nx))
ai = curvedsky.alm_info(lmax)
alm = torch.randn(nb, ai.nelem, dtype=torch.complex128, device="cuda"); alm[:, :lmax+1].imag = 0
m = enmap.dmap(torch.zeros((nb, ny, nx), dtype=torch.float64, device="cuda"), wcs)
out = torch.zeros_like(alm)
def step():
	curvedsky.alm2map(alm, m, spin=[0], ainfo=ai)
	curvedsky.map2alm(m, alm=out, spin=[0], ainfo=ai)
step(); torch.cuda.synchronize()
print("round-trip rms error %.2e" % float((out-alm).abs().pow(2).mean().sqrt()/alm.abs().pow(2).mean().sqrt()))
t0 = time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/3
print("lanes=%s: %.1f ms per batch of %d round trips, %.1f round trips/s" % (os.environ.get("PIXELL_AMD_LANES", "1"), dt*1e3, nb, nb/dt))
