"""ctypes loader of libpxsht.so (the HIP kernels behind include/pxsht.h).

The product path has NO CPU fallback: if the library is missing or cannot be loaded this
module raises.  There is no switch in here that selects anything but pixell_amd/libpxsht.so: the
CPU test-suite points `lib_path` at its simulator build itself (tests/conftest.py), and such a
build says what it is in pxs_version() (`is_hostsim`)."""
import ctypes, os

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_is_hostsim = False

class PxsError(RuntimeError):
	def __init__(self, code, msg):
		RuntimeError.__init__(self, "libpxsht error %d: %s" % (code, msg))
		self.code = code

def _declare(lib):
	c = ctypes
	vp, i32, i64, dbl = c.c_void_p, c.c_int, c.c_int64, c.c_double
	lib.pxs_last_error.restype = c.c_char_p
	lib.pxs_version.restype = c.c_char_p
	lib.pxs_plan_rings.argtypes = [c.POINTER(vp), i32, vp, vp, vp, vp, i64, i32, i32, vp, i64, i32]
	lib.pxs_plan_grid2d.argtypes = [c.POINTER(vp), c.c_char_p, i32, i32, dbl, i32, i32, i32, i32, vp, i64, i32]
	lib.pxs_plan_destroy.argtypes = [vp]; lib.pxs_plan_destroy.restype = None
	lib.pxs_synthesis.argtypes = [vp, i32, i32, i32, i32, vp, i32, i64, i64, vp, i32, i64, i64, vp]
	lib.pxs_analysis.argtypes = [vp, i32, i32, i32, vp, i32, i64, i64, vp, i32, i64, i64, vp]
	lib.pxs_gridweights.argtypes = [c.c_char_p, i32, vp]
	lib.pxs_grid_maxlmax.argtypes = [c.c_char_p, i32]
	lib.pxs_plan_info.argtypes = [vp, c.POINTER(i32), c.POINTER(i32), c.POINTER(i64)]
	lib.pxs_plan_option.argtypes = [vp, c.c_char_p, i64]
	lib.pxs_plan_query.argtypes = [vp, c.c_char_p, c.POINTER(i64)]
	lib.pxs_profile.argtypes = [vp, i32]; lib.pxs_profile_read.argtypes = [vp, vp, vp, i32]; lib.pxs_profile_flops.argtypes = [vp, vp, i32]
	lib.pxs_debug_theta_plan.argtypes = [i64, i32, vp]; lib.pxs_debug_chain.argtypes = [vp, i32, i32, i32, i32, vp]
	lib.pxs_memory.argtypes = [i32, vp]
	lib.pxa_alm2cl.argtypes = [i32, i32, vp, i64, vp, vp, i32, vp, i32, i32, vp]
	lib.pxa_lmatmul.argtypes = [i32, i32, i32, i32, vp, i64, vp, i64, vp, i64, i32, vp, i32, i32, vp]
	f64 = ctypes.c_double
	lib.pxm_rotate_queb.argtypes = [i32, i32, vp, vp, i32, i32, vp, vp, i32, i32, vp]
	lib.pxm_ps2d.argtypes = [i64, vp, vp, i32, vp, i32, i32, vp]
	lib.pxm_lbin.argtypes = [i32, i32, vp, vp, f64, i32, vp, i32, vp, vp, vp, i32, vp]
	lib.pxm_bin_index.argtypes = [i64, vp, i32, vp, i32, vp, i32, vp]
	lib.pxm_mul_axis.argtypes = [i64, i64, i64, vp, i32, vp, i32, vp]
	lib.pxf_fft_nd.argtypes = [i32, vp, vp, vp, i32, vp, i32, i32, dbl, i32, i32, vp, vp, i32, vp]
	lib.pxf_fft_supported.argtypes = [i64]
	lib.pxf_fft_good_size.argtypes = [i64]; lib.pxf_fft_good_size.restype = i64
	for name in ["pxs_plan_rings", "pxs_plan_grid2d", "pxs_synthesis", "pxs_analysis", "pxs_gridweights",
			"pxs_grid_maxlmax", "pxs_plan_info", "pxs_plan_option", "pxs_plan_query", "pxf_fft_nd", "pxf_fft_supported", "pxs_profile", "pxs_profile_read", "pxs_profile_flops", "pxs_debug_theta_plan", "pxs_debug_chain", "pxs_memory", "pxa_alm2cl", "pxa_lmatmul", "pxm_rotate_queb", "pxm_ps2d", "pxm_lbin", "pxm_bin_index", "pxm_mul_axis"]:
		getattr(lib, name).restype = i32
	return lib

EXPORTS = ["pxs_plan_rings", "pxs_plan_grid2d", "pxs_plan_destroy", "pxs_synthesis", "pxs_analysis",
	"pxs_gridweights", "pxs_grid_maxlmax", "pxs_plan_info", "pxs_plan_option", "pxs_plan_query", "pxf_fft_nd", "pxf_fft_supported",
	"pxf_fft_good_size", "pxs_last_error", "pxs_version", "pxs_profile", "pxs_profile_read", "pxs_profile_flops", "pxs_debug_theta_plan", "pxs_debug_chain", "pxs_memory", "pxa_alm2cl", "pxa_lmatmul", "pxm_rotate_queb", "pxm_ps2d", "pxm_lbin", "pxm_bin_index", "pxm_mul_axis"]

def lib_path():
	# PIXELL_AMD_LIB: another build of the same library (kernel A/B experiments, tools/build_variants.sh, tools/gpu_v2lab.sh, tools/gpu_leg_ab.sh)
	return os.environ.get("PIXELL_AMD_LIB") or os.path.join(HERE, "libpxsht.so")

def load():
	"""Return the loaded library, loading it on first use.  Raises if it is absent."""
	global _lib, _is_hostsim
	if _lib is not None: return _lib
	path = lib_path()
	if not os.path.exists(path):
		raise ImportError("pixell_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
			"(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
	# torch ships its own libamdhip64: let it be the HIP runtime of the process.  Loading libpxsht.so (which pulls in
	# /opt/rocm's runtime) before torch left two runtimes in one process and hipSetDevice then reported no device.
	try: import torch  # noqa: F401
	except ImportError: pass
	lib = _declare(ctypes.CDLL(path))
	_is_hostsim = b"HOSTSIM" in lib.pxs_version()      # a g++ build of the kernel sources (tests/hostsim): host pointers, no streams
	_lib = lib
	return _lib

def is_hostsim():
	load(); return _is_hostsim

def check(code):
	if code != 0:
		raise PxsError(code, load().pxs_last_error().decode("utf-8", "replace"))
