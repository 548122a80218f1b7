"""Stub-import harness for the reference (THIS container only; never runs on the GPU box).

Imports /root/reference's pixell.{enmap,curvedsky,powspec,lensing,fft,utils} with
  * a linear-CAR stand-in for astropy.wcs.WCS (astropy is absent here),
  * the reference's own cython/cmisc compiled out-of-tree into a temp dir,
  * `ducc0.sht.experimental` provided by a caller-supplied module (recorder or oracle).
Used only by tests/golden/make_golden.py to generate fixtures.  Nothing from the
reference is copied into the repo: only input/output arrays are saved.
"""
import sys, os, types, subprocess, tempfile, importlib
import numpy as np

REF = "/root/reference"

class _FITSFixedWarning(Warning): pass

class _Wcsprm:
	def __init__(self, naxis=2):
		self.cdelt = np.ones(naxis); self.crval = np.zeros(naxis)
		self.crpix = np.zeros(naxis); self.ctype = [""]*naxis
		self.lonpole = 180.0; self.latpole = 0.0; self.cunit = ["deg"]*naxis
		self.naxis = naxis
	def bounds_check(self, a, b): pass
	def get_pv(self): return []
	def set_pv(self, pv): pass

class StubWCS:
	"""Linear CAR: world = crval + (pix + 1 - crpix) * cdelt  (exact for CAR with crval[1]==0)"""
	def __init__(self, header=None, naxis=2, **kw):
		self.naxis = naxis
		self.wcs = _Wcsprm(naxis)
	def deepcopy(self):
		o = StubWCS(naxis=self.naxis)
		o.wcs.cdelt = np.array(self.wcs.cdelt, float); o.wcs.crval = np.array(self.wcs.crval, float)
		o.wcs.crpix = np.array(self.wcs.crpix, float); o.wcs.ctype = list(self.wcs.ctype)
		return o
	def _p2w(self, pix, origin):
		pix = np.asarray(pix, float)
		return np.asarray(self.wcs.crval)+(pix+1-origin-np.asarray(self.wcs.crpix))*np.asarray(self.wcs.cdelt)
	def _w2p(self, w, origin):
		w = np.asarray(w, float)
		return (w-np.asarray(self.wcs.crval))/np.asarray(self.wcs.cdelt)+np.asarray(self.wcs.crpix)-1+origin
	def wcs_pix2world(self, *args):
		if len(args) == 2:
			return self._p2w(args[0], args[1])
		x, y, origin = args
		x, y = np.asarray(x, float), np.asarray(y, float)
		return [self.wcs.crval[0]+(x+1-origin-self.wcs.crpix[0])*self.wcs.cdelt[0],
			self.wcs.crval[1]+(y+1-origin-self.wcs.crpix[1])*self.wcs.cdelt[1]]
	def wcs_world2pix(self, *args):
		if len(args) == 2:
			return self._w2p(args[0], args[1])
		x, y, origin = args
		x, y = np.asarray(x, float), np.asarray(y, float)
		return [(x-self.wcs.crval[0])/self.wcs.cdelt[0]+self.wcs.crpix[0]-1+origin,
			(y-self.wcs.crval[1])/self.wcs.cdelt[1]+self.wcs.crpix[1]-1+origin]
	all_pix2world = wcs_pix2world
	all_world2pix = wcs_world2pix
	def to_header(self, *a, **k): return {}
	def __repr__(self): return "StubWCS(cdelt=%s,crval=%s,crpix=%s)" % (self.wcs.cdelt, self.wcs.crval, self.wcs.crpix)

def _build_cmisc(tmpdir):
	"""compile the reference's cython/cmisc where it lies; outputs only into tmpdir"""
	src = os.path.join(REF, "cython")
	subprocess.check_call([sys.executable, "-m", "cython", "-3", os.path.join(src, "cmisc.pyx"),
		"-o", os.path.join(tmpdir, "cmisc.c")])
	import sysconfig
	inc = [sysconfig.get_paths()["include"], np.get_include(), src]
	ext = sysconfig.get_config_var("EXT_SUFFIX")
	out = os.path.join(tmpdir, "cmisc"+ext)
	cmd = ["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-w"]+["-I"+i for i in inc]+[
		os.path.join(tmpdir, "cmisc.c"), os.path.join(src, "cmisc_core.c"), "-o", out, "-lm"]
	subprocess.check_call(cmd)
	return tmpdir

def load_reference(sht_module):
	"""returns a namespace with the reference modules; sht_module provides the
	ducc0.sht.experimental functions."""
	tmp = tempfile.mkdtemp(prefix="pixell_ref_harness_")
	_build_cmisc(tmp)
	def mod(name, **attrs):
		m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m; return m
	astropy = mod("astropy"); awcs = mod("astropy.wcs", WCS=StubWCS, FITSFixedWarning=_FITSFixedWarning)
	aio = mod("astropy.io"); fits = mod("astropy.io.fits")
	astropy.wcs = awcs; astropy.io = aio; aio.fits = fits
	ducc0 = mod("ducc0", __version__="oracle-stub"); sht = mod("ducc0.sht"); fftm = mod("ducc0.fft"); nufft = mod("ducc0.nufft")
	ducc0.sht = sht; sht.experimental = sht_module; ducc0.fft = None; ducc0.nufft = nufft
	sys.modules["ducc0.sht.experimental"] = sht_module
	sys.path.insert(0, tmp)
	import cmisc
	sys.path.insert(0, REF)
	sys.modules["pixell.cmisc"] = cmisc
	import pixell
	pixell.cmisc = cmisc
	# ducc0.fft stub must not register as an FFT engine: make attribute access fail softly
	del sys.modules["ducc0.fft"]
	from pixell import utils, powspec, fft as pfft, enmap, curvedsky, lensing
	ns = types.SimpleNamespace(utils=utils, powspec=powspec, fft=pfft, enmap=enmap,
		curvedsky=curvedsky, lensing=lensing, cmisc=cmisc, WCS=StubWCS)
	return ns
