"""curvedsky.prof2alm(norot=True) of the REFERENCE (this container only; pixell/curvedsky.py:556-580 over the long-double oracle mounted as
ducc0.sht.experimental, tests/golden/_ref_harness.py) -> prof2alm.npz: inputs and outputs for tests/test_curvedsky_api.py.
Run:  python tests/golden/make_prof2alm.py"""
import sys, os
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..")); sys.path.insert(0, HERE)
from oracle import sht_oracle as so
import _ref_harness as H

def main():
	ns = H.load_reference(so)
	cs = ns.curvedsky
	out = {}
	n = 97; th = np.arange(n)*np.pi/(n-1)
	prof = np.exp(-0.5*(th/0.25)**2)
	out["prof_cc"] = prof; out["alm_cc"] = cs.prof2alm(prof, norot=True)
	# a stack [T, Q, U] of profiles: spin 0 and a spin-2 pair, on the Fejer-1 rings
	n = 80; th = (np.arange(n)+0.5)*np.pi/n
	stack = np.array([np.cos(th)**2, np.sin(th)**2*np.exp(-th), np.sin(th)**2*np.cos(3*th)])
	out["prof_f1"] = stack; out["alm_f1"] = cs.prof2alm(stack, spin=[0, 2], geometry="F1", norot=True)
	np.savez_compressed(os.path.join(HERE, "prof2alm.npz"), **out)
	print("prof2alm.npz written:", {k: v.shape for k, v in out.items()})

if __name__ == "__main__":
	main()
