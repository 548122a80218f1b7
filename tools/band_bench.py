"""map2alm / alm2map on a declination band (the cyl path: explicit rings, no theta resampling), device resident.
ACT-like band: dec -63 .. +23 deg at 0.5 arcmin (10320 x 43200), T/Q/U, lmax 10000; and the same at 2 arcmin / lmax 4000."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from pixell_amd import curvedsky, enmap
for res_am, lmax in [(2.0, 4000), (0.5, 10000)]:
	shape, wcs = enmap.band_geometry(np.deg2rad([-63.0, 23.0]), res=np.deg2rad(res_am/60))
	ainfo = curvedsky.alm_info(lmax)
	g = torch.Generator(device="cuda"); g.manual_seed(1)
	alm = torch.randn((3, ainfo.nelem), dtype=torch.complex128, device="cuda", generator=g)
	m = enmap.dmap(torch.zeros((3,)+tuple(shape), dtype=torch.float64, device="cuda"), wcs)
	for niter in (0, 1):
		for rep in range(2):
			torch.cuda.synchronize(); t0 = time.perf_counter()
			curvedsky.alm2map(alm, m, spin=[0, 2], ainfo=ainfo); torch.cuda.synchronize(); t1 = time.perf_counter()
			out = curvedsky.map2alm(m, lmax=lmax, spin=[0, 2], ainfo=ainfo, niter=niter); torch.cuda.synchronize(); t2 = time.perf_counter()
		err = float(torch.sqrt(torch.mean(torch.abs(out-alm)**2)/torch.mean(torch.abs(alm)**2)))
		print("band %s res %.1f' lmax %d niter %d: alm2map %.1f ms, map2alm %.1f ms (method %s; recovery rms %.2e: a band does not determine alm)" % (
			str(tuple(shape)), res_am, lmax, niter, (t1-t0)*1e3, (t2-t1)*1e3, curvedsky.get_method(m.shape, m.wcs), err), flush=True)
	del m, alm
	torch.cuda.empty_cache()
