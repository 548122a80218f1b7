import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path: sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

def _have_gpu():
	try:
		import torch
		return torch.cuda.is_available()
	except Exception:
		return False

HAVE_GPU = _have_gpu()
HOSTSIM_LIB = os.path.join(ROOT, "tests", "hostsim", "libpxsht_hostsim.so")

def use_hostsim():
	"""GPU-less container: the kernels' index logic is exercised through the TEST-ONLY host simulator (same .hip sources compiled
	with g++, see pixell_amd/csrc/hostsim.hpp).  The switch lives here, not in the product loader: the test process points the
	loader's `lib_path` at the simulator build before anything loads the library."""
	from pixell_amd import _lib
	assert _lib._lib is None, "the library was loaded before the test configuration could select the simulator"
	_lib.lib_path = lambda: HOSTSIM_LIB

def pytest_configure(config):
	config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
	config.addinivalue_line("markers", "hostsim: runs the kernels in the test-only CPU simulator (GPU-less environments only)")
	if not HAVE_GPU:
		sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
		import build_hostsim
		build_hostsim.build()
		use_hostsim()

def pytest_collection_modifyitems(config, items):
	skip_gpu = pytest.mark.skip(reason="no GPU in this environment")
	skip_sim = pytest.mark.skip(reason="host-simulator tests only run where there is no GPU")
	for item in items:
		if "gpu" in item.keywords and not HAVE_GPU: item.add_marker(skip_gpu)
		if "hostsim" in item.keywords and HAVE_GPU: item.add_marker(skip_sim)

@pytest.fixture(scope="session")
def golden_dir():
	return GOLDEN

@pytest.fixture(scope="session")
def have_gpu():
	return HAVE_GPU
