"""timing of the general ring path on healpix maps (device-resident): alm2map_healpix + its adjoint"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from pixell_amd import curvedsky
for nside, lmax in [(256, 512), (1024, 2048), (2048, 4096)]:
	ainfo = curvedsky.alm_info(lmax)
	g = torch.Generator(device="cuda"); g.manual_seed(1)
	alm = torch.randn((3, ainfo.nelem), dtype=torch.complex128, device="cuda", generator=g)
	m = torch.zeros((3, 12*nside**2), dtype=torch.float64, device="cuda")
	for rep in range(3):
		torch.cuda.synchronize(); t0 = time.perf_counter()
		curvedsky.alm2map_healpix(alm, m, spin=[0, 2]); torch.cuda.synchronize(); t1 = time.perf_counter()
		curvedsky.alm2map_healpix(alm, m, spin=[0, 2], adjoint=True); torch.cuda.synchronize(); t2 = time.perf_counter()
	print("nside %5d lmax %5d: alm2map %.1f ms, adjoint %.1f ms (3 components)" % (nside, lmax, (t1-t0)*1e3, (t2-t1)*1e3), flush=True)
