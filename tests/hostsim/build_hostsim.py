"""TEST-ONLY: compile pixell_amd/csrc/*.hip with g++ -DPXS_HOST_SIM into libpxsht_hostsim.so so the
kernels' index logic can be exercised without a GPU (every lane its own fiber, or OS thread with PXS_SIM_THREADS=1).  Never used by the
product path; see pixell_amd/csrc/hostsim.hpp."""
import os, subprocess, glob
HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "..", "pixell_amd", "csrc")
LIB  = os.path.join(HERE, "libpxsht_hostsim.so")

def build(force=False):
	srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))+[os.path.join(CSRC, "hostsim.cpp")]
	deps = srcs+glob.glob(os.path.join(CSRC, "*.hpp"))
	if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
		return LIB
	bdir = os.path.join(HERE, "build"); os.makedirs(bdir, exist_ok=True)
	objs, procs = [], []
	hpp_t = max(os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.hpp")))
	for s in srcs:
		o = os.path.join(bdir, os.path.basename(s)+".o"); objs.append(o)
		if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hpp_t):
			cmd = ["g++", "-x", "c++", "-DPXS_HOST_SIM", "-O2", "-g", "-std=c++17", "-fPIC", "-pthread", "-w", "-c", s, "-o", o]
			procs.append((cmd, subprocess.Popen(cmd)))
	for cmd, p in procs:
		if p.wait() != 0: raise RuntimeError("hostsim build failed: "+" ".join(cmd))
	subprocess.check_call(["g++", "-shared", "-pthread", "-o", LIB]+objs)
	return LIB

if __name__ == "__main__":
	print(build())
