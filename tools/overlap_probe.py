# Do the spin groups of one map (T: spin 0, Q/U: spin 2) overlap when they run on two HIP streams with a plan (scratch) each?
# The ring-FFT / theta-resampling chains are memory-bound, the Legendre kernels FP64-bound: a round trip is their SUM today.
#   python tools/overlap_probe.py [ny nx lmax]      (default: BASELINE config 3)
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pixell_amd import sht

ny, nx, lmax = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (21600, 43200, 10000)
dev = torch.device("cuda")
nalm = (lmax+1)*(lmax+2)//2
g = torch.Generator(device=dev); g.manual_seed(1)
def rand_alm(nc):
	a = torch.complex(torch.randn((nc, nalm), generator=g, device=dev, dtype=torch.float64), torch.randn((nc, nalm), generator=g, device=dev, dtype=torch.float64))
	a[:, :lmax+1] = a[:, :lmax+1].real+0j
	if nc == 2:
		m_of = torch.repeat_interleave(torch.arange(lmax+1, device=dev), torch.arange(lmax+1, 0, -1, device=dev))
		l_of = torch.arange(nalm, device=dev)-(m_of*(2*lmax+1-m_of))//2
		a[:, l_of < 2] = 0
	return a
ms = sht.tri_mstart(lmax)
def new_plan():
	p = sht.grid_plan("F1", ny, nx, -np.pi, (True, True), lmax, lmax, ms)
	sht.clear_plans()        # the next grid_plan call makes another pxs_plan (own scratch)
	return p
pT, pP = new_plan(), new_plan()
almT, almP = rand_alm(1), rand_alm(2)
mapT = torch.zeros((1, ny, nx), dtype=torch.float64, device=dev); mapP = torch.zeros((2, ny, nx), dtype=torch.float64, device=dev)
sht._run_syn(pT, almT, mapT, 0, "STANDARD", False, map_overwrite=True); sht._run_syn(pP, almP, mapP, 2, "STANDARD", False, map_overwrite=True)
outT, outP = torch.zeros_like(almT), torch.zeros_like(almP)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
hi = torch.cuda.Stream(priority=-1)

def rt_T(plan):
	sht._run_ana(plan, mapT, outT, 0, False); sht._run_syn(plan, outT, mapT, 0, "STANDARD", False, map_overwrite=True)
def rt_P(plan):
	sht._run_ana(plan, mapP, outP, 2, False); sht._run_syn(plan, outP, mapP, 2, "STANDARD", False, map_overwrite=True)

def serial():
	sht._run_ana(pT, mapT, outT, 0, False); sht._run_ana(pP, mapP, outP, 2, False)
	sht._run_syn(pT, outT, mapT, 0, "STANDARD", False, map_overwrite=True); sht._run_syn(pP, outP, mapP, 2, "STANDARD", False, map_overwrite=True)
def two_streams(sa, sb):
	cur = torch.cuda.current_stream()
	sa.wait_stream(cur); sb.wait_stream(cur)
	with torch.cuda.stream(sb): rt_P(pP)
	with torch.cuda.stream(sa): rt_T(pT)
	cur.wait_stream(sa); cur.wait_stream(sb)
def two_streams_joined(sa, sb):
	# as two library calls would: both groups of map2alm, join, both groups of alm2map, join
	cur = torch.cuda.current_stream()
	sa.wait_stream(cur); sb.wait_stream(cur)
	with torch.cuda.stream(sb): sht._run_ana(pP, mapP, outP, 2, False)
	with torch.cuda.stream(sa): sht._run_ana(pT, mapT, outT, 0, False)
	cur.wait_stream(sa); cur.wait_stream(sb)
	sa.wait_stream(cur); sb.wait_stream(cur)
	with torch.cuda.stream(sb): sht._run_syn(pP, outP, mapP, 2, "STANDARD", False, map_overwrite=True)
	with torch.cuda.stream(sa): sht._run_syn(pT, outT, mapT, 0, "STANDARD", False, map_overwrite=True)
	cur.wait_stream(sa); cur.wait_stream(sb)

def timed(name, fn, reps=4):
	fn(); torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(reps): fn()
	torch.cuda.synchronize()
	ms_ = (time.perf_counter()-t0)/reps*1e3
	print("%-46s %8.2f ms per round trip" % (name, ms_), flush=True)
	return ms_

for rep in range(2):
	timed("one stream (T then Q/U, as the product)", serial)
	timed("T alone", lambda: rt_T(pT)); timed("Q/U alone", lambda: rt_P(pP))
	timed("two streams, independent round trips", lambda: two_streams(s1, s2))
	timed("two streams, joined after each direction", lambda: two_streams_joined(s1, s2))
	timed("two streams, T on a high-priority stream", lambda: two_streams(hi, s2))
	timed("two streams, Q/U on a high-priority stream", lambda: two_streams(s1, hi))
eT = float(((outT-almT).abs().pow(2).mean().sqrt()/almT.abs().pow(2).mean().sqrt()).item()); eP = float(((outP-almP).abs().pow(2).mean().sqrt()/almP.abs().pow(2).mean().sqrt()).item())
print("round-trip rms error T %.2e  Q/U %.2e" % (eT, eP))
