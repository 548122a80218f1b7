# batched scalar / T,Q,U maps at small and medium sizes: the FP64-MFMA Legendre kernels (default from 4 maps) against the VALU kernels (PXS_ANA_MM_MIN=1000 PXS_SYN_MM_MIN=1000)
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from pixell_amd import curvedsky, enmap, sht
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for (ny, nx), lmax, nb, ncomp in [((256, 512), 250, 16, 1), ((900, 1800), 750, 16, 1), ((900, 1800), 750, 8, 3), ((1600, 3200), 1500, 16, 1), ((1600, 3200), 1500, 8, 3), ((2700, 5400), 2500, 16, 1)]:
	shape, wcs = enmap.fullsky_geometry(shape=(ny, nx)); ainfo = curvedsky.alm_info(lmax)
	g = torch.Generator(device="cuda"); g.manual_seed(1)
	alm = torch.randn((nb, ncomp, ainfo.nelem), dtype=torch.complex128, device="cuda", generator=g)
	m = enmap.dmap(torch.zeros((nb, ncomp, ny, nx), dtype=torch.float64, device="cuda"), wcs)
	out = torch.zeros_like(alm)
	def rt(): curvedsky.map2alm(m, alm=out, spin=[0, 2], ainfo=ainfo); curvedsky.alm2map(out, m, spin=[0, 2], ainfo=ainfo)
	curvedsky.alm2map(alm, m, spin=[0, 2], ainfo=ainfo)
	for _ in range(2): rt()
	plan = list(sht._plans.d.values())[-1]; plan.profile(True); torch.cuda.synchronize()
	n = 10; t0 = time.perf_counter()
	for _ in range(n): rt()
	torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/n*1e3
	st = plan.profile_read(reset=True); plan.profile(False)
	print("%s %dx%dx%dx%d lmax %d: %.3f ms per batch round trip; leg_syn %.3f leg_ana %.3f ring %.3f resample %.3f" % (tag, nb, ncomp, ny, nx, lmax, dt, st["leg_syn"][0]/n, st["leg_ana"][0]/n, st["ring_fft"][0]/n, st["resample"][0]/n), flush=True)
	sht.clear_plans()
