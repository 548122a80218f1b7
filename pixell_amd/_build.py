"""Build libpxsht.so (HIP, gfx950) in-tree.  Called by __graft_entry__.build()."""
import os, subprocess, glob, shutil

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB  = os.path.join(HERE, "libpxsht.so")

def sources():
	return sorted(glob.glob(os.path.join(CSRC, "*.hip")))

def needs_build(lib=LIB):
	if not os.path.exists(lib): return True
	t = os.path.getmtime(lib)
	deps = sources()+glob.glob(os.path.join(CSRC, "*.hpp"))+[os.path.join(HERE, "..", "include", "pxsht.h")]
	return any(os.path.getmtime(d) > t for d in deps)

# per-file flags.  thetaline.hip, ringline.hip (the register-resident line FFT): SimplifyCFG's sinking of common instructions merges the register-array accesses of the radix
# switch into pointer phis before the array is split into scalars, and the whole line then lives in scratch memory (regfft_dev.hpp).
FILE_FLAGS = {"thetaline.hip": ["-mllvm", "-simplifycfg-sink-common=false"], "ringline.hip": ["-mllvm", "-simplifycfg-sink-common=false"]}

def flags_for(src):
	return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]+FILE_FLAGS.get(os.path.basename(src), [])+os.environ.get("PXS_EXTRA_HIPCC_FLAGS", "").split()

def hipcc_path():
	return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

def compile_one(src, obj, verbose=False):
	"""one translation unit -> gfx950 object (the build check of tests/test_capi.py compiles one with this, always for real)"""
	cmd = [hipcc_path()]+flags_for(src)+["-c", src, "-o", obj]
	if verbose: print(" ".join(cmd))
	subprocess.check_call(cmd)
	return obj

def build(force=False, verbose=False):
	"""hipcc cross-compiles for gfx950 without a GPU present.
	force (or PXS_FORCE_BUILD=1, or no object directory yet -- a fresh checkout): every translation unit is recompiled and the
	library relinked, whatever the timestamps say; otherwise only what is older than its sources."""
	force = force or os.environ.get("PXS_FORCE_BUILD", "0") == "1" or not os.path.isdir(os.path.join(HERE, "build"))
	if not force and not needs_build(): return LIB
	hipcc = hipcc_path()
	objs = []
	bdir = os.path.join(HERE, "build"); os.makedirs(bdir, exist_ok=True)
	procs = []
	for s in sources():
		o = os.path.join(bdir, os.path.basename(s)+".o")
		if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s),
				max(os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.hpp")))):
			cmd = [hipcc]+flags_for(s)+["-c", s, "-o", o]
			if verbose: print(" ".join(cmd))
			procs.append((cmd, subprocess.Popen(cmd)))
		objs.append(o)
	for cmd, p in procs:
		if p.wait() != 0: raise RuntimeError("build failed: "+" ".join(cmd))
	cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB]+objs
	if verbose: print(" ".join(cmd))
	subprocess.check_call(cmd)
	return LIB

if __name__ == "__main__":
	print(build(force=False, verbose=True))
