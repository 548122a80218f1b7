"""Unified harmonic transform (flat-sky 2-D FFTs or curved-sky SHTs behind one interface) on the HIP backend.

The method set and semantics of pixell's uharm.UHT (pixell/uharm.py:8-182) for the part of it that sits on the accelerated
path: map2harm / harm2map and their adjoints, quad_weights, lprof2hprof, hmul, harm2powspec, sum_hprof / mean_hprof.  In "flat"
mode the harmonic representation is a complex map of the 2-D FFT bins (enmap.map2harm, normalize="phys"); in "curved" mode it is
an alm array.  The profile transforms (rprof2hprof, hprof2rprof: 1-D Legendre transforms of beams) run in curved mode as m = 0
transforms on one-pixel rings, in flat mode through the 2-D FFT of the painted profile and ring means of its inverse (enmap.rbin); hrand in flat
mode raises, as the reference's does for every input.
Maps may be numpy ndmaps (staged) or enmap.dmap (device resident, nothing leaves HBM).
The stock pixell.uharm.UHT itself also runs on this backend unchanged through the integration routes of INTEGRATION.md
(pixell.curvedsky over pixell_amd.sht, pixell.fft.engines["hip"]); this class is for callers that keep their data on the GPU."""
import numpy as np
from . import enmap, curvedsky
from .sht import _is_tensor, _torch

def res2lmax(res):
	"""band limit that resolves the angular scale res (radians)"""
	return int(np.round(np.pi/res))

def estimate_distortion(shape, wcs):
	"""largest relative change of the east-west pixel scale across a cylindrical-projection patch: cos(dec) at the row nearest
	the equator over cos(dec) at the row farthest from it, minus one"""
	dec = enmap.pix2sky(shape, wcs, [[0, shape[-2]-1], [0, 0]])[0]
	c = np.cos(dec)
	widest = 1.0 if dec[0]*dec[1] <= 0 else float(np.max(c))          # the patch straddles the equator
	return widest/float(np.min(c))-1

class UHT:
	def __init__(self, shape, wcs, mode="auto", lmax=None, max_distortion=0.1, niter=0):
		self.shape, self.wcs = tuple(shape[-2:]), wcs
		self.area = enmap.area(self.shape, self.wcs)
		self.fsky = self.area/(4*np.pi)
		if mode == "auto": mode = "flat" if estimate_distortion(shape, wcs) <= max_distortion else "curved"
		if mode not in ("flat", "curved"): raise ValueError("Unrecognized mode in UHT: '%s'" % str(mode))
		self.mode, self.niter, self.quad = mode, niter, None
		if mode == "flat":
			self.l = enmap.modlmap(shape, wcs)
			self.lmax = int(np.round(np.max(np.asarray(self.l))))
			self.nper = 1/self.fsky                    # modes per 2-D bin, in the units sum_hprof uses for the curved case
			self.ntot = self.nper*self.shape[-2]*self.shape[-1]
		else:
			if lmax is None: lmax = res2lmax(np.min(np.abs(wcs.wcs.cdelt))*np.pi/180)
			self.lmax = int(lmax)
			self.l = np.arange(self.lmax+1)
			self.ainfo = curvedsky.alm_info(lmax=self.lmax)
			self.nper = 2*self.l+1
			self.ntot = np.sum(self.nper)
	@property
	def npix(self): return self.shape[-2]*self.shape[-1]
	def _zeros_map(self, harm):
		"""a real map for every alm of harm[..., nelem], next to it"""
		pre = tuple(harm.shape[:-1])
		if _is_tensor(harm):
			torch = _torch()
			rt = torch.float32 if harm.dtype == torch.complex64 else torch.float64
			return enmap.dmap(torch.zeros(pre+self.shape, dtype=rt, device=harm.device), self.wcs)
		return enmap.zeros(pre+self.shape, self.wcs, np.zeros(1, harm.dtype).real.dtype)
	def map2harm(self, map, spin=0):
		if self.mode == "flat": return enmap.map2harm(map, spin=spin, normalize="phys")
		return curvedsky.map2alm(map, ainfo=self.ainfo, spin=spin, niter=self.niter)
	def harm2map(self, harm, spin=0):
		if self.mode == "flat": return enmap.harm2map(harm, spin=spin, normalize="phys")
		return curvedsky.alm2map(harm, self._zeros_map(harm), ainfo=self.ainfo, spin=spin)
	def harm2map_adjoint(self, map, spin=0):
		if self.mode == "flat": return enmap.harm2map_adjoint(map, spin=spin, normalize="phys")
		return curvedsky.alm2map_adjoint(map, ainfo=self.ainfo)            # (the reference does not pass spin here either)
	def map2harm_adjoint(self, harm, spin=0):
		if self.mode == "flat": return enmap.map2harm_adjoint(harm, spin=spin, normalize="phys")
		return curvedsky.map2alm_adjoint(harm, self._zeros_map(harm), ainfo=self.ainfo, spin=spin, niter=self.niter)
	def quad_weights(self):
		"""W with map2harm = harm2map_adjoint * W; broadcasts against maps ([ny,1])"""
		if self.quad is None:
			if self.mode == "flat":
				edges = enmap.pix2sky(self.shape, self.wcs, [np.arange(self.shape[-2]+1)-0.5, np.zeros(self.shape[-2]+1)])[0]
				edges = np.clip(edges, -np.pi/2, np.pi/2)
				self.quad = enmap.ndmap((np.abs(np.sin(edges[1:])-np.sin(edges[:-1]))*abs(self.wcs.wcs.cdelt[0])*np.pi/180)[:, None], self.wcs)
			else: self.quad = curvedsky.quad_weights(self.shape, self.wcs)[:, None]
		return self.quad
	def lprof2hprof(self, lprof):
		"""1-D function of l -> harmonic profile: flat: its value at |l| of every 2-D bin (linear interpolation, ZERO beyond the
		last sample, as the reference's interpol(order=1, border="constant") gives); curved: the first lmax+1 samples, zero padded"""
		lprof = np.asarray(lprof)
		if self.mode == "flat":
			l = np.asarray(self.l); x = np.arange(lprof.shape[-1])
			flat = lprof.reshape(-1, lprof.shape[-1])
			out = np.stack([np.interp(l.reshape(-1), x, row, right=0.0) for row in flat]).reshape(lprof.shape[:-1]+l.shape)
			return enmap.ndmap(out, self.wcs)
		n = self.lmax+1
		if lprof.shape[-1] >= n: return lprof[..., :n]
		return np.concatenate([lprof, np.zeros(lprof.shape[:-1]+(n-lprof.shape[-1],), lprof.dtype)], -1)
	def hprof2harm(self, hprof):
		if self.mode == "flat": return hprof.copy()
		raise NotImplementedError("hprof2harm in curved mode needs alm_info.get_map, which pixell does not implement either")
	def hmul(self, hprof, harm, inplace=False):
		"""hprof * harm -> harm.  flat: hprof [ny,nx], [ncomp,ny,nx] (elementwise) or [ncomp,ncomp,ny,nx] (matrix product over the
		component axis); curved: [nl], [ncomp,nl] or [ncomp,ncomp,nl] through alm_info.lmul on the GPU"""
		if self.mode == "curved":
			if not _is_tensor(harm):
				harm = np.asanyarray(harm); harm = harm.astype(np.result_type(harm, 0j), copy=False)
			return self.ainfo.lmul(harm, hprof, out=harm if inplace else None)
		dev = isinstance(harm, enmap.dmap)
		h = harm.tensor if dev else np.asanyarray(harm)
		p = np.asarray(hprof) if not isinstance(hprof, enmap.dmap) else hprof.tensor
		if dev and not _is_tensor(p): p = _torch().as_tensor(np.ascontiguousarray(p), device=h.device)
		if p.ndim <= 3: res = p*h
		else: res = (_torch() if dev else np).einsum("...abyx,...byx->...ayx", p.to(h.dtype) if dev else p, h)
		if inplace: h[...] = res; return harm
		return enmap.dmap(res, harm.wcs) if dev else enmap.ndmap(res, getattr(harm, "wcs", self.wcs))
	def hrand(self, hprof):
		if self.mode == "curved": return curvedsky.rand_alm(hprof, lmax=self.lmax)
		# (the reference's flat branch -- map_mul(multi_pow(hprof / pixsize, 0.5), rand_gauss_harm(shape, wcs)), uharm.py:166-170 -- raises for EVERY hprof: multi_pow
		# needs [ncomp, ncomp, ny, nx], map_mul's matrix product then needs a noise map with a component axis, and the noise is drawn as [ny, nx])
		raise NotImplementedError("hrand in flat mode: the reference's own flat-mode hrand raises for every input (uharm.py:166-170)")
	def harm2powspec(self, harm, harm2=None, patch=False):
		"""pseudo (cross) power spectrum as a harmonic profile; patch: divide the curved-sky spectrum by fsky"""
		if self.mode == "flat": return enmap.calc_ps2d(harm, harm2)
		ps = curvedsky.alm2cl(harm, harm2)
		return ps/self.fsky if patch else ps
	def sum_hprof(self, hprof):
		hprof = hprof.cpu().numpy() if _is_tensor(hprof) else np.asanyarray(hprof.tensor.cpu().numpy() if isinstance(hprof, enmap.dmap) else hprof)
		return np.sum(hprof*self.nper, (-2, -1) if self.mode == "flat" else -1)
	def mean_hprof(self, hprof): return self.sum_hprof(hprof)/self.ntot
	def rprof2hprof(self, br, r):
		"""radial profile br[..., nr] at radii r -> harmonic profile (uharm.py:127-132).  curved: a function of l up to lmax; flat: the real part of
		the 2-D FFT of the profile painted around pixel (0, 0) of the patch, times the pixel area, so that l = 0 holds the integral of the profile
		(profile2harm_flat_2d, uharm.py:230-245)"""
		if self.mode == "curved": return curvedsky.profile2harm(br, r, lmax=self.lmax)
		br, r = np.asarray(br), np.asarray(r)
		cpix = np.array(self.shape)//2-1
		cpos = enmap.pix2sky(self.shape, self.wcs, cpix)
		rmap = np.roll(np.asarray(enmap.modrmap(self.shape, self.wcs, cpos)), tuple(-cpix), (-2, -1))           # distance from the pixel that ends up at (0, 0)
		flat = br.reshape(-1, br.shape[-1])
		bmap = np.stack([np.interp(rmap, r, row, right=0) for row in flat]).reshape(br.shape[:-1]+self.shape)
		harm = enmap.fft(enmap.ndmap(bmap, self.wcs), normalize=False)
		return enmap.ndmap(np.asarray(harm).real*enmap.pixsize(self.shape, self.wcs), self.wcs)
	def hprof2rprof(self, harm, r):
		"""the inverse: harmonic profile -> radial profile at radii r (flat: harm2profile_flat_2d, uharm.py:247-258: inverse FFT, the centre moved to
		the middle of the patch, means over rings of one pixel pitch around it, interpolated to r; zero beyond the last ring)"""
		if self.mode == "curved": return curvedsky.harm2profile(harm, r)
		h = enmap.ndmap(np.asarray(harm)+0j, self.wcs)
		bmap = np.asarray(enmap.ifft(h, normalize=False)).real/(enmap.pixsize(self.shape, self.wcs)*self.npix)
		cpix = np.array(self.shape)//2-1
		cpos = enmap.pix2sky(self.shape, self.wcs, cpix)
		bmap = enmap.shift(enmap.ndmap(bmap, self.wcs), cpix, keepwcs=True)
		wbr, wr = enmap.rbin(bmap, center=cpos)
		if r is None: return wbr, r
		flat = wbr.reshape(-1, wbr.shape[-1])
		return np.stack([np.interp(r, wr, row, right=0) for row in flat]).reshape(wbr.shape[:-1]+np.shape(r))
	def hprof_rpow(self, hprof, power):
		"""the harmonic profile of (the real-space profile of hprof)**power: map2harm(harm2map(hprof)**power) for profiles
		(uharm.py:191-207).  curved: the profile is sampled at a tenth of the beam's 1/e^(1/2) scale out to 20 of them; flat: through the maps"""
		if self.mode != "curved":
			norm = self.area**0.5
			m = self.harm2map(enmap.ndmap(np.asarray(hprof)/norm+0j, self.wcs))
			return self.map2harm(m**power)*norm
		hprof = np.asarray(hprof)
		scale = 1/max(1, np.where(hprof > np.max(hprof)*np.exp(-0.5))[0][-1])
		r = np.arange(0, 20*scale, scale/10)
		return self.rprof2hprof(self.hprof2rprof(hprof, r)**power, r)
