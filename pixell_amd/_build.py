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

def build(force=False, verbose=False):
	"""hipcc cross-compiles for gfx950 without a GPU present."""
	if not force and not needs_build(): return LIB
	hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
	objs = []
	bdir = os.path.join(HERE, "build"); os.makedirs(bdir, exist_ok=True)
	procs = []
	for s in sources():
		o = os.path.join(bdir, os.path.basename(s)+".o")
		if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s),
				max(os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.hpp")))):
			cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]+os.environ.get("PXS_EXTRA_HIPCC_FLAGS", "").split()+["-c", s, "-o", o]
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
