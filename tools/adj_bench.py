"""alm2map_adjoint (adjoint_synthesis_2d) at C3 / C2: through the CC grid (default) or directly on the map's rings (PXS_ADJ_VIA_CC=0)"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from pixell_amd import curvedsky, enmap
for shp, lmax in [((5400, 10800), 4000), ((21600, 43200), 10000)]:
	shape, wcs = enmap.fullsky_geometry(shape=shp); ainfo = curvedsky.alm_info(lmax)
	g = torch.Generator(device="cuda"); g.manual_seed(1)
	m = enmap.dmap(torch.randn((3,)+tuple(shape), dtype=torch.float64, device="cuda", generator=g), wcs)
	for rep in range(3):
		torch.cuda.synchronize(); t0 = time.perf_counter()
		a = curvedsky.alm2map_adjoint(m, spin=[0, 2], ainfo=ainfo); torch.cuda.synchronize(); dt = time.perf_counter()-t0
	print("alm2map_adjoint 3x%dx%d lmax %d: %.1f ms" % (shp[0], shp[1], lmax, dt*1e3), flush=True)
	del m; torch.cuda.empty_cache()
for shp, lmax in [((5400, 10800), 4000), ((21600, 43200), 10000)]:
	shape, wcs = enmap.fullsky_geometry(shape=shp); ainfo = curvedsky.alm_info(lmax)
	g = torch.Generator(device="cuda"); g.manual_seed(1)
	alm = torch.randn((3, ainfo.nelem), dtype=torch.complex128, device="cuda", generator=g)
	m = enmap.dmap(torch.zeros((3,)+tuple(shape), dtype=torch.float64, device="cuda"), wcs)
	for form in ("ducc0", "interpolant"):      # the default form and the full-interpolant option, each through its fused transposed chain
		for rep in range(2):
			torch.cuda.synchronize(); t0 = time.perf_counter()
			curvedsky.map2alm_adjoint(alm, m, spin=[0, 2], ainfo=ainfo, analysis=form); torch.cuda.synchronize(); dt = time.perf_counter()-t0
		print("map2alm_adjoint (adjoint_analysis_2d) 3x%dx%d lmax %d, analysis=%s: %.1f ms" % (shp[0], shp[1], lmax, form, dt*1e3), flush=True)
	del m; torch.cuda.empty_cache()
