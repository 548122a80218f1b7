# Legendre stage times of mid-size single-map T/Q/U transforms for K sweeps (lab build)
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from pixell_amd import curvedsky, enmap, sht
tag = sys.argv[1] if len(sys.argv) > 1 else ""
sizes = [((1100, 2200), 1050), ((1350, 2700), 1300), ((1600, 3200), 1500), ((2160, 4320), 2000), ((2700, 5400), 2500)] if os.environ.get("KMID_SIZES", "") == "" else [((3000, 6000), 2900), ((3200, 6400), 3100), ((3600, 7200), 3500), ((4050, 8100), 3900)]
for (ny, nx), lmax in sizes:
	shape, wcs = enmap.fullsky_geometry(shape=(ny, nx)); ainfo = curvedsky.alm_info(lmax)
	g = torch.Generator(device="cuda"); g.manual_seed(1)
	alm = torch.randn((3, ainfo.nelem), dtype=torch.complex128, device="cuda", generator=g)
	m = enmap.dmap(torch.zeros((3, ny, nx), dtype=torch.float64, device="cuda"), wcs)
	out = torch.zeros_like(alm)
	def rt(): curvedsky.map2alm(m, alm=out, spin=[0, 2], ainfo=ainfo); curvedsky.alm2map(out, m, spin=[0, 2], ainfo=ainfo)
	curvedsky.alm2map(alm, m, spin=[0, 2], ainfo=ainfo)
	for _ in range(3): rt()
	plan = list(sht._plans.d.values())[-1]
	res = []
	for rep in range(3):
		plan.profile(True); torch.cuda.synchronize(); n = 20
		for _ in range(n): rt()
		torch.cuda.synchronize(); st = plan.profile_read(reset=True); plan.profile(False)
		res.append((st["leg_syn"][0]/n, st["leg_ana"][0]/n))
	print("%s lmax %d: leg_syn %s  leg_ana %s" % (tag, lmax, " ".join("%.3f" % r[0] for r in res), " ".join("%.3f" % r[1] for r in res)), flush=True)
	sht.clear_plans()
