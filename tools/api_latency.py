"""wall time of the calls either side of the transforms at C3 / C2 scale (device resident where the API allows): a check for
host-side pathologies, not a benchmark"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from pixell_amd import curvedsky, enmap, uharm
def T(label, f, n=2):
	for _ in range(n):
		torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); dt = time.perf_counter()-t0
	print("%-58s %9.1f ms" % (label, dt*1e3), flush=True); return r
lmax = 10000; ainfo = curvedsky.alm_info(lmax)
shape, wcs = enmap.fullsky_geometry(shape=(21600, 43200))
g = torch.Generator(device="cuda"); g.manual_seed(1)
alm = torch.randn((3, ainfo.nelem), dtype=torch.complex128, device="cuda", generator=g)
T("analyse_geometry (21600x43200)", lambda: curvedsky.analyse_geometry((3,)+tuple(shape), wcs))
T("get_ring_info", lambda: curvedsky.get_ring_info(shape, wcs))
T("quad_weights", lambda: curvedsky.quad_weights(shape, wcs))
T("alm2cl 3x3 lmax 1e4 (device)", lambda: curvedsky.alm2cl(alm[:, None], alm[None, :]))
T("almxfl (device)", lambda: curvedsky.almxfl(alm, lambda l: 1/(1+l)**2))
T("alm_info.lmul 3x3 matrix filter (device)", lambda: ainfo.lmul(alm, np.ones((3, 3, lmax+1))))
T("transpose_alm (host index build + device gather)", lambda: curvedsky.transpose_alm(ainfo, alm[0]), n=1)
ps = np.ones((1, 1, lmax+1))
T("rand_alm lmax 1e4 1 comp (numpy legacy RNG on the host, as the reference)", lambda: curvedsky.rand_alm(ps, lmax=lmax, seed=1), n=1)
m = enmap.dmap(torch.zeros((3,)+tuple(shape), dtype=torch.float64, device="cuda"), wcs)
T("alm2map C3 (dmap)", lambda: curvedsky.alm2map(alm, m, spin=[0, 2], ainfo=ainfo))
T("map2alm C3 (dmap)", lambda: curvedsky.map2alm(m, lmax=lmax, spin=[0, 2], ainfo=ainfo))
T("enmap.fft C3 one comp (dmap)", lambda: enmap.fft(m[0]))
del m; torch.cuda.empty_cache()
s2, w2 = enmap.fullsky_geometry(shape=(5400, 10800))
m2 = enmap.dmap(torch.randn((3,)+tuple(s2), dtype=torch.float64, device="cuda", generator=g), w2)
h = T("enmap.map2harm C2 size (dmap, phys)", lambda: enmap.map2harm(m2, normalize="phys"))
p2 = T("calc_ps2d", lambda: enmap.calc_ps2d(h))
T("lbin", lambda: enmap.lbin(p2[0, 0] if p2.ndim == 4 else p2[0]))
T("modlmap C2 size", lambda: enmap.modlmap(s2, w2))
mh = np.zeros((3,)+tuple(s2)); mh = enmap.ndmap(mh, w2)
a2 = np.zeros((3, curvedsky.alm_info(4000).nelem), complex)
T("alm2map C2 with host ndmap / numpy alm (staged over PCIe)", lambda: curvedsky.alm2map(a2, mh, spin=[0, 2]))
T("map2alm C2 with host ndmap", lambda: curvedsky.map2alm(mh, lmax=4000, spin=[0, 2]))
u = uharm.UHT(s2, w2, mode="curved", lmax=4000)
T("UHT.map2harm curved C2 (dmap)", lambda: u.map2harm(m2))
