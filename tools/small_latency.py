"""Host vs device cost of small transforms (the reference's own benchmark shape 900x1800, lmax 750; C1 1024x2048, lmax 512):
wall time per call with a synchronize after each call (latency), without (throughput: host and device pipelined), and the host time
of a call alone (time until the call returns)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from pixell_amd import curvedsky, enmap
for (ny, nx), lmax in [((900, 1800), 750), ((1024, 2048), 512), ((64, 128), 48)]:
	shape, wcs = enmap.fullsky_geometry(shape=(ny, nx)); ainfo = curvedsky.alm_info(lmax)
	g = torch.Generator(device="cuda"); g.manual_seed(1)
	alm = torch.randn((1, ainfo.nelem), dtype=torch.complex128, device="cuda", generator=g)
	m = enmap.dmap(torch.zeros((1, ny, nx), dtype=torch.float64, device="cuda"), wcs)
	out = torch.zeros_like(alm)
	def rt(): curvedsky.map2alm(m, alm=out, spin=[0], ainfo=ainfo); curvedsky.alm2map(out, m, spin=[0], ainfo=ainfo)
	curvedsky.alm2map(alm, m, spin=[0], ainfo=ainfo)
	for _ in range(5): rt()
	torch.cuda.synchronize(); n = 200
	t0 = time.perf_counter()
	for _ in range(n): rt(); torch.cuda.synchronize()
	lat = (time.perf_counter()-t0)/n
	t0 = time.perf_counter()
	for _ in range(n): rt()
	host = (time.perf_counter()-t0)/n
	torch.cuda.synchronize(); thr = (time.perf_counter()-t0)/n
	print("%dx%d lmax %d: round trip latency %.3f ms, throughput %.3f ms, host side of the two calls %.3f ms" % (ny, nx, lmax, lat*1e3, thr*1e3, host*1e3), flush=True)
