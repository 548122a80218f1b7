#!/usr/bin/env python
"""Host-array route: round trip of numpy maps / alm (pixell_amd/hostio.py) against the device-resident one and against the former
one-shot staging (PIXELL_AMD_PIPE_MIN_MB=1e9 turns the slab route off).  usage: tools/host_bench.py [c2|c3] [reps]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pixell_amd import curvedsky, enmap, hostio
cfg = {"c2": ((5400, 10800), 4000), "c3": ((21600, 43200), 10000)}[sys.argv[1] if len(sys.argv) > 1 else "c3"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
(ny, nx), lmax = cfg
shape, wcs = enmap.fullsky_geometry(shape=(ny, nx)); ainfo = curvedsky.alm_info(lmax)
g = torch.Generator(device="cuda"); g.manual_seed(1)
alm = torch.complex(torch.randn((3, ainfo.nelem), generator=g, device="cuda", dtype=torch.float64), torch.randn((3, ainfo.nelem), generator=g, device="cuda", dtype=torch.float64))
alm[:, :lmax+1] = alm[:, :lmax+1].real+0j
m_of = torch.repeat_interleave(torch.arange(lmax+1, device="cuda"), torch.arange(lmax+1, 0, -1, device="cuda")); l_of = torch.arange(ainfo.nelem, device="cuda")-(m_of*(2*lmax+1-m_of))//2
alm = alm/(l_of+1.0); alm[1:, l_of < 2] = 0
dm = enmap.dmap(torch.zeros((3, ny, nx), dtype=torch.float64, device="cuda"), wcs)
curvedsky.alm2map(alm, dm, spin=[0, 2], ainfo=ainfo); back = torch.zeros_like(alm); curvedsky.map2alm(dm, alm=back, spin=[0, 2], ainfo=ainfo)
def timed(fn):
	fn(); torch.cuda.synchronize(); t = time.perf_counter()
	for _ in range(reps): fn()
	torch.cuda.synchronize(); return (time.perf_counter()-t)/reps*1e3
res = dict(config=sys.argv[1] if len(sys.argv) > 1 else "c3", slab_MB=hostio.SLAB_BYTES >> 20)
res["resident_ms"] = round(timed(lambda: (curvedsky.map2alm(dm, alm=back, spin=[0, 2], ainfo=ainfo), curvedsky.alm2map(back, dm, spin=[0, 2], ainfo=ainfo))), 1)
hm = enmap.ndmap(dm.tensor.cpu().numpy(), wcs); ha = back.cpu().numpy()
res["host_map2alm_ms"] = round(timed(lambda: curvedsky.map2alm(hm, alm=ha, spin=[0, 2], ainfo=ainfo)), 1)
res["host_alm2map_ms"] = round(timed(lambda: curvedsky.alm2map(ha, hm, spin=[0, 2], ainfo=ainfo)), 1)
res["host_roundtrip_ms"] = round(res["host_map2alm_ms"]+res["host_alm2map_ms"], 1)
res["bytes_each_way_GB"] = round((hm.nbytes+ha.nbytes)/1e9, 2)
res["alm_rms_error"] = float(np.sqrt(np.mean(np.abs(ha-alm.cpu().numpy())**2)/np.mean(np.abs(ha)**2)))
print(json.dumps(res))
