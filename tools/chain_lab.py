#!/usr/bin/env python
"""Times the fused FFT chain stages of a grid plan on scratch data (pxs_debug_chain), without Legendre tables or maps:
a kernel experiment takes seconds.  usage: tools/chain_lab.py [c3|c4|c2] [reps]   (PIXELL_AMD_LIB selects a variant build)"""
import sys, os, ctypes, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pixell_amd import sht, enmap, curvedsky, _lib
cfgs = {"c3": ((21600, 43200), 10000, [(1, 0), (2, 2)]), "c2": ((5400, 10800), 4000, [(1, 0), (2, 2)]), "c4": ((5400, 10800), 4000, [(8, 0)]), "c5": ((10800, 21600), 6000, [(1, 0)]), "tiny": ((90, 180), 60, [(1, 0), (2, 2)]),
	"p2": ((4096, 8192), 4095, [(8, 0)]),
	# off-BASELINE grids for the planner sweeps (tools/r06_planner_sweep.sh): 4', 1.32', 0.67', 0.9' Fejer-1 grids at band limits a user would pick
	"o1": ((2700, 5400), 2000, [(1, 0), (2, 2)]), "o2": ((8192, 16384), 6000, [(1, 0), (2, 2)]), "o3": ((16200, 32400), 8000, [(1, 0), (2, 2)]), "o4": ((12000, 24000), 7000, [(1, 0), (2, 2)]),
	"o5": ((6480, 12960), 5000, [(4, 0)])}      # power-of-two circles (N = 8192, N_cc = 8192, M = 16384): the line engine with radix 16 only
name = sys.argv[1] if len(sys.argv) > 1 else "c3"; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
(ny, nx), lmax, groups = cfgs[name]
shape, wcs = enmap.fullsky_geometry(shape=(ny, nx))
mi = curvedsky.analyse_geometry(shape, wcs); ai = curvedsky.alm_info(lmax)
plan = sht.grid_plan(mi.ducc_geo.name, ny, nx, mi.phi0, mi.flip, lmax, lmax, ai.mstart, 1)
lib = _lib.load(); res = {}
for nc, spin in groups:
	for kind, label in enumerate(["map2leg", "h2map", "to_cc", "from_cc", "from_cc_adj"]):
		ms = ctypes.c_double()
		_lib.check(lib.pxs_debug_chain(plan.handle, kind, nc, spin, reps, ctypes.byref(ms)))
		res["%s_nc%d" % (label, nc)] = round(ms.value, 3)
tot = sum(v for k, v in res.items() if not k.startswith("from_cc_adj"))
print(json.dumps(dict(config=name, lib=os.path.basename(_lib.lib_path()), ms=res, round_trip_chain_ms=round(tot, 3))))
