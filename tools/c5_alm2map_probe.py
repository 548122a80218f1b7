# Where does alm2map of a batch of 8 maps at C5 (10800x21600, lmax 6000) spend its time?  wall / GPU-event time of the call against the library's stage timers
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from pixell_amd import curvedsky, enmap, sht
nb, ny, nx, lmax = 8, 10800, 21600, 6000
dev = torch.device("cuda")
shape, wcs = enmap.fullsky_geometry(shape=(ny, nx))
ainfo = curvedsky.alm_info(lmax)
nalm = (lmax+1)*(lmax+2)//2
alm = torch.complex(torch.randn((nb, nalm), device=dev, dtype=torch.float64), torch.randn((nb, nalm), device=dev, dtype=torch.float64))
alm[:, :lmax+1] = alm[:, :lmax+1].real+0j
m = enmap.dmap(torch.zeros((nb, ny, nx), dtype=torch.float64, device=dev), wcs)
back = torch.zeros_like(alm)
def run(fn, name, reps=3):
	fn(); torch.cuda.synchronize()
	plan = list(sht._plans.d.values())[-1]
	plan.profile(True)
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	t0 = time.perf_counter(); e0.record()
	for _ in range(reps): fn()
	th = time.perf_counter()-t0
	e1.record(); torch.cuda.synchronize(); tw = time.perf_counter()-t0
	st = plan.profile_read(reset=True); plan.profile(False)
	print("%-10s wall %.1f ms  GPU events %.1f ms  host enqueue %.1f ms  stages %s  sum %.1f ms" % (name, tw/reps*1e3, e0.elapsed_time(e1)/reps, th/reps*1e3,
		{k: round(v[0]/reps, 1) for k, v in st.items()}, sum(v[0] for v in st.values())/reps), flush=True)
for rep in range(2):
	run(lambda: curvedsky.alm2map(alm, m, spin=[0], ainfo=ainfo), "alm2map")
	run(lambda: curvedsky.map2alm(m, alm=back, spin=[0], ainfo=ainfo), "map2alm")
