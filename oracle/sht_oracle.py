"""CPU oracle for the SHT hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product path (pixell_amd/) never does: it fails loudly when the HIP
library is missing.

What is restated here
---------------------
The reference's hot path ends in calls to the third-party `ducc0.sht.experimental`
module (ducc0>=0.36.0, pyproject.toml:28 of the reference; source NOT present under
/root/reference), at these call sites:

  pixell/curvedsky.py:907-924   synthesis_2d / adjoint_synthesis_2d
  pixell/curvedsky.py:1032-1046 analysis_2d / adjoint_analysis_2d
  pixell/curvedsky.py:936-960, 1068-1084  synthesis / adjoint_synthesis (explicit rings)
  pixell/curvedsky.py:501, 855  get_gridweights

This file restates the *published mathematics* those functions implement (healpix /
libsharp conventions, which docs/usage.rst:401-402 of the reference asserts):

  spin 0 :  map(theta,phi) = sum_l a_l0 Y_l0 + 2 Re sum_{m>0} a_lm Y_lm,
            Y_lm = lambda_lm(theta) e^{i m phi}, Condon-Shortley phase.
  spin s>0: with E = alm[0], B = alm[1],  +s a = -(E + iB),  -s a = -(-1)^s (E - iB),
            map[0] +- i map[1] = sum_lm (+-s a_lm) (+-s Y_lm)   (Goldberg sYlm).
  DERIV1  : spin-1 transform of E = sqrt(l(l+1)) a_lm, B = 0  -> (d_theta f, d_phi f / sin theta).
  adjoint_synthesis  = exact transpose of synthesis (a_lm = sum_pix map Y*_lm, no weights).
  analysis_2d        = quadrature of the trigonometric interpolant (in theta) of the parity-extended ring-FFT of the map
                       against Y*_lm: the left inverse of synthesis_2d for band-limited maps whenever the grid carries
                       enough rings (curvedsky.py:1349-1353 get_ducc_maxlmax).  Default: ducc0's own route as its published
                       source has it (Clenshaw-Curtis weights on the grid of 2 good_size_complex(lmax+1) + 1 rings, see
                       analysis_2d below); fine_cc=False: the exact integral of the full interpolant against |sin theta|;
                       weights=True: the grid's own quadrature weights.  The three agree on band-limited maps (pinned by the
                       reference's round-trip tests, tests/golden/make_golden.py); on other maps they differ and which one
                       ducc0 computes is NOT pinned by anything in the reference tree: parity unpinned there.
  adjoint_analysis_2d= exact transpose of analysis_2d.

It is deliberately written differently from the HIP implementation so that agreement is
meaningful: 80-bit long double arithmetic, the plain three-term recurrences in l (no
Ishioka two-step recurrence, no exponent scaling), Gauss-Legendre quadrature of the
interpolant (no |sin| convolution, no Clenshaw-Curtis resampling), numpy's pocketfft.

Parity pinning (see tests/test_oracle_golden.py): alm2map of the reference's own
`rand_alm(seed=1)` reproduces tests/data/MM_unlensed_071123.fits of the reference
(spin [0,2], CC 181x360, lmax=400); spin-0 functions equal scipy.special.sph_harm_y;
spin-s functions equal Goldberg's explicit sum; the reference's round-trip and
adjointness invariants (tests/test_pixell.py:870-965, 1051-1085) hold.
Against ducc0's own numerics on non-band-limited maps: parity unpinned (ducc0 absent).
"""
import numpy as np

LD = np.longdouble
CLD = np.clongdouble
PI = LD(np.pi) + LD(1.2246467991473532e-16)  # pi to long double precision


# ----------------------------------------------------------------------------------
# Grids (names as in curvedsky.get_ducc_geo, curvedsky.py:1308-1347)
# ----------------------------------------------------------------------------------
def grid_info(name, n):
	"""Full-circle description of a named equiangular grid with n rings.
	theta_j = theta0 + 2 pi j / N for j = 0..n-1; c = theta0*N/pi (integer) gives the
	mirror map j' -> (-j'-c) mod N of the 2pi-periodic extension."""
	n = int(n)
	if   name == "CC":     N, c = 2*n-2, 0
	elif name == "F1":     N, c = 2*n,   1
	elif name == "MW":     N, c = 2*n-1, 1
	elif name == "MWflip": N, c = 2*n-1, 0
	elif name == "DH":     N, c = 2*n,   0
	elif name == "F2":     N, c = 2*n+2, 2
	else: raise ValueError("unknown grid '%s'" % str(name))
	theta = (LD(c)*PI/N + 2*PI*np.arange(n, dtype=LD)/N)
	return dict(N=N, c=c, theta=theta, theta0=LD(c)*PI/N)

def grid_theta(name, n):
	return np.asarray(grid_info(name, n)["theta"], dtype=np.float64)

def grid_maxlmax(name, n):
	"""curvedsky.get_ducc_maxlmax (curvedsky.py:1349-1353)"""
	if   name == "CC": return n-2
	elif name == "DH": return (n-2)//2
	elif name == "F2": return (n-1)//2
	else:              return n-1

def _abs_sin_series(K):
	"""Fourier coefficients s_k, k=0..K of |sin theta| = sum_k s_k e^{ik theta}."""
	k = np.arange(K+1)
	s = np.zeros(K+1)
	ev = (k % 2) == 0
	s[ev] = -(2/np.pi)/(k[ev].astype(float)**2-1)
	return s

def get_gridweights(name, n):
	"""Quadrature weights per ring, sum = 4 pi (phi integral included), exact for the
	trigonometric interpolant on the grid.  Mirrors ducc0.sht.experimental.get_gridweights
	as used at curvedsky.py:501,855."""
	if name in ("DH", "F2"):
		# Fejer's second rule on the interior nodes theta_j = j pi/(K+1), j = 1..K (classical closed form, exact for polynomials
		# in cos(theta) of degree <= K-1, which is where get_ducc_maxlmax's (n-1)/2 and (n-2)/2 come from):
		#   w_j = 4 sin(theta_j)/(K+1) * sum_{k=1}^{floor((K+1)/2)} sin((2k-1) theta_j)/(2k-1)      (sum over the ring = 2)
		# F2 is exactly that grid (K = n); DH is the north pole (weight 0) followed by the K = n-1 interior nodes of spacing pi/n.
		K = n if name == "F2" else n-1
		j = np.arange(1, K+1, dtype=LD); th = j*PI/(K+1)
		acc = np.zeros(K, LD)
		for k in range(1, (K+1)//2+1): acc += np.sin((2*k-1)*th)/(2*k-1)
		w = np.asarray(4*np.sin(th)/(K+1)*acc, np.float64)*2*np.pi
		return w if name == "F2" else np.concatenate([[0.0], w])
	g = grid_info(name, n); N, c = g["N"], g["c"]
	th = np.asarray(g["theta0"] + 2*PI*np.arange(N, dtype=LD)/N, dtype=np.float64)
	K = (N-1)//2
	s = _abs_sin_series(N//2)
	v = np.full(N, s[0])
	for k in range(1, K+1):
		if s[k] != 0: v += 2*s[k]*np.cos(k*th)
	if N % 2 == 0 and s[N//2] != 0:
		v += s[N//2]*np.cos((N//2)*(th-float(g["theta0"])))*np.cos((N//2)*float(g["theta0"]))
	v *= np.pi/N            # integral of g(theta)|sin|/2 over the circle
	w = np.zeros(n)
	for jp in range(N):
		w[_ring_of(jp, N, c, n)] += v[jp]
	return w*2*np.pi

def _ring_of(jp, N, c, n):
	"""ring index carrying full-circle sample jp"""
	if jp < n: return jp
	return (-jp-c) % N


# ----------------------------------------------------------------------------------
# Spin-weighted lambda_lm(theta) generators (long double, three-term recurrence in l)
# ----------------------------------------------------------------------------------
class LamGen:
	"""Iterates l = 0..lmax and exposes cur[m, ring] = (s)lambda_lm(theta_ring) for all
	m <= min(l, mmax) with l >= max(m,|s|) (rows not yet started hold zeros)."""
	def __init__(self, s, lmax, mmax, theta):
		self.s, self.lmax, self.mmax = int(s), int(lmax), int(mmax)
		th = np.asarray(theta, dtype=LD)
		self.x  = np.cos(th)
		self.sh = np.sin(th/2)
		self.ch = np.cos(th/2)
		nm, nr = self.mmax+1, len(th)
		self.prev = np.zeros((nm, nr), LD)
		self.cur  = np.zeros((nm, nr), LD)
		self.m    = np.arange(nm, dtype=LD)
		self.l    = -1
		self._norms()
	def _norms(self):
		"""sqrt of the normalisation of the starting functions, per m (long double)."""
		s, nm = abs(self.s), self.mmax+1
		g = np.zeros(nm, LD)
		# m < s : h(m) = (2s+1) (2s)! / ((s+m)!(s-m)!)
		if s > 0:
			h = LD(2*s+1)
			for i in range(1, s+1): h = h*LD(s+i)/LD(i)   # (2s)!/(s!)^2
			for m in range(0, min(s, nm)):
				if m > 0: h = h*LD(s-m+1)/LD(s+m)
				g[m] = h
		# m >= s : g(m) = (2m+1)(2m)!/((m+s)!(m-s)!)
		if s < nm:
			v = LD(2*s+1)
			g[s] = v
			for m in range(s+1, nm):
				v = v*LD(2*m+1)*LD(2*m)/(LD(m+s)*LD(m-s))
				g[m] = v
		self.norm = np.sqrt(g/(4*PI))
	def _start(self, m):
		"""closed form at l0 = max(m,|s|) (Goldberg's sum has a single term there)"""
		s, sa = self.s, abs(self.s)
		if m >= sa:
			val = self.norm[m]*self.sh**(m+s)*self.ch**(m-s)
			if m % 2: val = -val
		else:
			if s > 0:
				val = self.norm[m]*self.sh**(sa+m)*self.ch**(sa-m)
				if m % 2: val = -val
			else:
				val = self.norm[m]*self.sh**(sa-m)*self.ch**(sa+m)
				if sa % 2: val = -val
		return val
	def step(self):
		"""advance to l+1; afterwards self.cur holds degree self.l"""
		l = self.l; s = self.s; sa = abs(s)
		lnew = l+1
		nact = min(lnew, self.mmax)+1       # rows m <= lnew
		if lnew > max(0, sa):
			# rows with l0 < lnew advance: those are m < lnew (and lnew > sa)
			hi = min(lnew, self.mmax+1)     # rows m = 0..lnew-1
			if hi > 0 and l >= sa:
				m = self.m[:hi]
				L = LD(l)
				if sa == 0:
					e1 = np.sqrt((LD(lnew)**2-m*m)/(4*LD(lnew)**2-1))
					e0 = np.sqrt((L*L-m*m)/(4*L*L-1)) if l > 0 else np.zeros(hi, LD)
					new = (self.x[None,:]*self.cur[:hi]-e0[:,None]*self.prev[:hi])/e1[:,None]
				else:
					S1 = np.sqrt((LD(lnew)**2-m*m)*(LD(lnew)**2-LD(sa)**2))
					S0 = np.sqrt(np.maximum((L*L-m*m)*(L*L-LD(sa)**2), 0))
					f1 = np.sqrt(LD(2*l+3)/LD(2*l+1))*LD(2*l+1)/(L*S1)
					f2 = np.sqrt(LD(2*l+3)/LD(max(2*l-1,1)))*LD(l+1)*S0/(L*S1)
					mid = L*LD(l+1)*self.x[None,:] + (m*LD(s))[:,None]
					new = f1[:,None]*mid*self.cur[:hi] - f2[:,None]*self.prev[:hi]
				# rows that have not started yet (m < sa and l < sa can't happen here since l>=sa)
				self.prev[:hi] = self.cur[:hi]
				self.cur[:hi]  = new
		# rows starting at lnew
		if lnew == sa:
			for m in range(0, min(sa, self.mmax)+1):
				self.cur[m] = self._start(m); self.prev[m] = 0
		elif lnew > sa and lnew <= self.mmax:
			self.cur[lnew] = self._start(lnew); self.prev[lnew] = 0
		self.l = lnew
		return nact


def goldberg_sYlm_theta(s, l, m, theta):
	"""Explicit (slow, exact-binomial) Goldberg sum for s lambda_lm(theta), l small.
	Used only to validate LamGen."""
	from math import comb, factorial, pi, sqrt
	theta = np.asarray(theta, dtype=np.float64)
	if l < max(abs(m), abs(s)): return np.zeros_like(theta)
	pref = (-1)**m*sqrt(factorial(l+m)*factorial(l-m)*(2*l+1)/(4*pi*factorial(l+s)*factorial(l-s)))
	sh, ch = np.sin(theta/2), np.cos(theta/2)
	out = np.zeros_like(theta)
	for r in range(0, l-s+1):
		k2 = r+s-m
		if k2 < 0 or k2 > l+s: continue
		# sin^{2l}(t/2) cot^{2r+s-m}(t/2) = sin^{2l-2r-s+m} cos^{2r+s-m}
		a, b = 2*l-2*r-s+m, 2*r+s-m
		if a < 0 or b < 0: continue
		out = out + comb(l-s, r)*comb(l+s, k2)*(-1)**(l-r-s)*sh**a*ch**b
	return pref*out


# ----------------------------------------------------------------------------------
# alm <-> leg (Legendre stage)
# ----------------------------------------------------------------------------------
def _alm_rows(alm, l, nact, mstart, lstride):
	idx = (mstart[:nact].astype(np.int64) + l*lstride)
	return idx

def alm2leg(alm, spin, lmax, mmax, mstart, theta, lstride=1, mode="STANDARD"):
	"""alm[ncomp, nelem] -> leg[ncomp_map, nring, mmax+1] complex128.
	spin 0: leg = F_m(theta) = sum_l a_lm lambda_lm.  spin s: (Q_m, U_m)."""
	alm = np.atleast_2d(np.asarray(alm))
	mstart = np.asarray(mstart)
	nr = len(theta); nm = mmax+1
	if mode == "DERIV1":
		assert spin == 1 and alm.shape[0] == 1
		l = np.arange(lmax+1, dtype=np.float64)
		fac = np.sqrt(l*(l+1))
		a2 = np.zeros((2, alm.shape[1]), np.complex128)
		for m in range(nm):
			i = int(mstart[m])+np.arange(m, lmax+1)*lstride
			a2[0, i] = alm[0, i]*fac[m:]
		alm = a2
	if spin == 0:
		assert alm.shape[0] == 1
		leg = np.zeros((1, nm, nr), CLD)
		gen = LamGen(0, lmax, mmax, theta)
		for l in range(lmax+1):
			nact = gen.step()
			a = alm[0, _alm_rows(alm, l, nact, mstart, lstride)].astype(CLD)
			leg[0, :nact] += a[:, None]*gen.cur[:nact]
	else:
		assert alm.shape[0] == 2
		s = int(spin); sg = LD(-1 if s % 2 else 1)
		leg = np.zeros((2, nm, nr), CLD)
		gp = LamGen(+s, lmax, mmax, theta); gm = LamGen(-s, lmax, mmax, theta)
		I = CLD(1j)
		for l in range(lmax+1):
			nact = gp.step(); gm.step()
			if l < s: continue
			rows = _alm_rows(alm, l, nact, mstart, lstride)
			E = alm[0, rows].astype(CLD)[:, None]; B = alm[1, rows].astype(CLD)[:, None]
			W = (gp.cur[:nact] + sg*gm.cur[:nact])/2
			X = (gp.cur[:nact] - sg*gm.cur[:nact])/2
			leg[0, :nact] -= E*W + I*B*X
			leg[1, :nact] -= B*W - I*E*X
	return np.ascontiguousarray(np.transpose(leg, (0, 2, 1)).astype(np.complex128))

def leg2alm(leg, spin, lmax, mmax, mstart, theta, nelem, lstride=1, mode="STANDARD"):
	"""leg[ncomp_map, nring, mmax+1] -> alm[ncomp, nelem]; exact transpose of alm2leg
	(a_lm = sum_r lambda_lm(theta_r) leg[r,m]; weights, if any, are applied by the caller)."""
	leg = np.asarray(leg)
	mstart = np.asarray(mstart)
	nm = mmax+1
	L = np.transpose(leg, (0, 2, 1)).astype(CLD)    # [c, m, r]
	if spin == 0:
		alm = np.zeros((1, nelem), np.complex128)
		gen = LamGen(0, lmax, mmax, theta)
		for l in range(lmax+1):
			nact = gen.step()
			val = np.sum(gen.cur[:nact]*L[0, :nact], axis=1)
			alm[0, _alm_rows(alm, l, nact, mstart, lstride)] = val.astype(np.complex128)
	else:
		s = int(spin); sg = LD(-1 if s % 2 else 1)
		alm = np.zeros((2, nelem), np.complex128)
		gp = LamGen(+s, lmax, mmax, theta); gm = LamGen(-s, lmax, mmax, theta)
		I = CLD(1j)
		for l in range(lmax+1):
			nact = gp.step(); gm.step()
			if l < s: continue
			rows = _alm_rows(alm, l, nact, mstart, lstride)
			W = (gp.cur[:nact] + sg*gm.cur[:nact])/2
			X = (gp.cur[:nact] - sg*gm.cur[:nact])/2
			Q, U = L[0, :nact], L[1, :nact]
			alm[0, rows] = (-np.sum(W*Q + I*X*U, axis=1)).astype(np.complex128)
			alm[1, rows] = (-np.sum(W*U - I*X*Q, axis=1)).astype(np.complex128)
		if mode == "DERIV1":
			l = np.arange(lmax+1, dtype=np.float64)
			fac = np.sqrt(l*(l+1))
			out = np.zeros((1, nelem), np.complex128)
			for m in range(nm):
				i = int(mstart[m])+np.arange(m, lmax+1)*lstride
				out[0, i] = alm[0, i]*fac[m:]
			alm = out
	return alm


# ----------------------------------------------------------------------------------
# leg <-> map (ring FFT stage; numpy pocketfft)
# ----------------------------------------------------------------------------------
def leg2map_ring(F, nphi, phi0):
	"""F[mmax+1] complex -> ring[nphi] real: F_0.re + 2 Re sum_{m>0} F_m e^{i m (phi0+2 pi x/nphi)}"""
	mmax = len(F)-1
	m = np.arange(mmax+1)
	c = F*np.exp(1j*m*phi0)
	spec = np.zeros(nphi, np.complex128)
	np.add.at(spec, m % nphi, c)
	np.add.at(spec, (-m[1:]) % nphi, np.conj(c[1:]))
	spec[0] = spec[0] - 1j*c[0].imag   # only Re F_0 contributes once
	return np.fft.ifft(spec).real*nphi

def map2leg_ring(ring, phi0, mmax):
	"""ring[nphi] real -> F[mmax+1] = sum_x ring[x] e^{-i m phi_x}"""
	nphi = len(ring)
	m = np.arange(mmax+1)
	return np.fft.fft(ring)[m % nphi]*np.exp(-1j*m*phi0)

def leg2map(leg, nphi, phi0, ringstart, npix, pixstride=1, dtype=np.float64):
	"""leg[nc, nring, nm] -> map[nc, npix]"""
	nc, nr, nm = leg.shape
	out = np.zeros((nc, npix), dtype)
	for c in range(nc):
		for r in range(nr):
			n = int(nphi[r]); o = int(ringstart[r])
			out[c, o+pixstride*np.arange(n)] = leg2map_ring(leg[c, r], n, float(phi0[r]))       # (index array: negative strides too)
	return out

def map2leg(map, nphi, phi0, ringstart, mmax, pixstride=1):
	map = np.atleast_2d(map)
	nc, nr = map.shape[0], len(nphi)
	leg = np.zeros((nc, nr, mmax+1), np.complex128)
	for c in range(nc):
		for r in range(nr):
			n = int(nphi[r]); o = int(ringstart[r])
			leg[c, r] = map2leg_ring(np.asarray(map[c, o+pixstride*np.arange(n)], np.float64), float(phi0[r]), mmax)
	return leg


# ----------------------------------------------------------------------------------
# ducc0.sht.experimental-shaped entry points (keyword-only, as called by curvedsky.py)
# ----------------------------------------------------------------------------------
def _ncomp(spin, mode):
	if mode == "DERIV1": return 1, 2
	return (1, 1) if spin == 0 else (2, 2)

def _nelem_default(lmax, mmax, mstart, lstride):
	return int(np.max(np.asarray(mstart).astype(np.int64)) + lmax*lstride + 1)

def synthesis(*, alm, theta, nphi, phi0, ringstart, lmax, mmax=None, mstart=None, spin=0,
		map=None, lstride=1, pixstride=1, nthreads=0, mode="STANDARD"):
	"""ducc0.sht.experimental.synthesis (curvedsky.py:936-960)"""
	alm = np.atleast_2d(np.asarray(alm))
	if mmax is None: mmax = lmax
	if mstart is None: mstart = _tri_mstart(lmax, mmax)
	nca, ncm = _ncomp(spin, mode)
	assert alm.shape[0] == nca
	rs_ = np.asarray(ringstart).astype(np.int64); np_ = np.asarray(nphi).astype(np.int64)
	npix = int(max(np.max(rs_), np.max(rs_+(np_-1)*pixstride))+1)
	if map is not None: npix = max(npix, map.shape[-1])
	leg = alm2leg(alm, spin, lmax, mmax, mstart, np.asarray(theta, LD), lstride, mode)
	res = leg2map(leg, nphi, phi0, ringstart, npix, pixstride)
	if map is None: return res
	# only the pixels of the rings are written (a ring subset of a larger map leaves the rest alone, curvedsky.py:333-335)
	hit = np.zeros(npix, bool)
	for r in range(len(rs_)): hit[rs_[r]+pixstride*np.arange(np_[r])] = True
	map[..., hit] = res[..., hit].astype(map.dtype)
	return map

def adjoint_synthesis(*, map, theta, nphi, phi0, ringstart, lmax, mmax=None, mstart=None, spin=0,
		alm=None, lstride=1, pixstride=1, nthreads=0, mode="STANDARD"):
	"""ducc0.sht.experimental.adjoint_synthesis (curvedsky.py:936, 1068-1084)"""
	map = np.atleast_2d(np.asarray(map))
	if mmax is None: mmax = lmax
	if mstart is None: mstart = _tri_mstart(lmax, mmax)
	nelem = alm.shape[-1] if alm is not None else _nelem_default(lmax, mmax, mstart, lstride)
	leg = map2leg(map, nphi, phi0, ringstart, mmax, pixstride)
	res = leg2alm(leg, spin, lmax, mmax, mstart, np.asarray(theta, LD), nelem, lstride, mode)
	if alm is None: return res.astype(np.result_type(map.dtype, 1j))
	alm[...] = res.astype(alm.dtype)
	return alm

def _tri_mstart(lmax, mmax):
	m = np.arange(mmax+1)
	return (m*(2*lmax+1-m)//2).astype(np.uint64)

def _grid_rings(geometry, ntheta, nphi, phi0):
	theta = grid_info(geometry, ntheta)["theta"]
	return (theta, np.full(ntheta, nphi, np.uint64), np.full(ntheta, phi0, np.float64),
		(np.arange(ntheta)*nphi).astype(np.uint64))

def synthesis_2d(*, alm, map, spin, lmax, geometry, mmax=None, mstart=None, phi0=0.0,
		nthreads=0, lstride=1, mode="STANDARD"):
	"""ducc0.sht.experimental.synthesis_2d (curvedsky.py:907-924).  map[nc, ntheta, nphi]."""
	nt, nph = map.shape[-2:]
	th, nphi, p0, rs = _grid_rings(geometry, nt, nph, phi0)
	flat = synthesis(alm=alm, theta=th, nphi=nphi, phi0=p0, ringstart=rs, lmax=lmax, mmax=mmax,
		mstart=mstart, spin=spin, lstride=lstride, mode=mode)
	map[...] = flat.reshape(map.shape).astype(map.dtype)
	return map

def adjoint_synthesis_2d(*, alm, map, spin, lmax, geometry, mmax=None, mstart=None, phi0=0.0,
		nthreads=0, lstride=1, mode="STANDARD"):
	nt, nph = map.shape[-2:]
	th, nphi, p0, rs = _grid_rings(geometry, nt, nph, phi0)
	m2 = np.asarray(map).reshape(map.shape[0], -1)
	return adjoint_synthesis(map=m2, alm=alm, theta=th, nphi=nphi, phi0=p0, ringstart=rs, lmax=lmax,
		mmax=mmax, mstart=mstart, spin=spin, lstride=lstride, mode=mode)

def _interp_matrix(geometry, ntheta, theta_out, kcut=None):
	"""Dense matrix M[g, j'] evaluating at theta_out the canonical trigonometric
	interpolant through N full-circle samples at theta0 + 2 pi j'/N.
	kcut (2 kcut <= N): the interpolant low-passed to |k| < kcut."""
	g = grid_info(geometry, ntheta); N = g["N"]; t0 = float(g["theta0"])
	d = np.asarray(theta_out, np.float64)[:, None] - (t0 + 2*np.pi*np.arange(N)/N)[None, :]
	K = (N-1)//2
	lowpass = kcut is not None and 2*kcut <= N
	if lowpass: K = min(K, kcut-1)
	M = np.ones_like(d)
	for k in range(1, K+1): M += 2*np.cos(k*d)
	if N % 2 == 0 and not lowpass:
		# Nyquist term: X_{N/2} cos((N/2)(theta-theta0)), X_{N/2} = sum_j g_j (-1)^j
		sgn = 1-2*(np.arange(N) % 2)
		M += np.cos((N//2)*(np.asarray(theta_out, np.float64)[:, None]-t0))*sgn[None, :]
	return M/N

def _extension(geometry, ntheta, m_plus_s_parity):
	"""index and sign arrays mapping ring data to the N full-circle samples.
	m_plus_s_parity: array over m of (m+spin) % 2."""
	g = grid_info(geometry, ntheta); N, c = g["N"], g["c"]
	ring = np.array([_ring_of(jp, N, c, ntheta) for jp in range(N)])
	mirrored = np.arange(N) >= ntheta
	return ring, mirrored

def _gl(nq):
	from scipy.special import roots_legendre
	x, w = roots_legendre(nq)
	return np.arccos(x)[::-1].copy(), w[::-1].copy()

def good_size_complex(n):
	"""smallest 2^a 3^b 5^c 7^d 11^e >= n (ducc0's good_size_complex; used for the ring count of its Legendre stage)"""
	n = int(n)
	while True:
		m = n
		for f in (2, 3, 5, 7, 11):
			while m % f == 0: m //= f
		if m == 1: return n
		n += 1

def _fine_cc_nodes(lmax, fine_cc):
	"""circle length N_cc of the Legendre-stage CC grid of the fine-CC form, nodes and weights of the CC grid of N_cc + 1 rings the
	weights are applied on.  fine_cc=True: ducc0's sizes (N_cc = 2 good_size_complex(lmax + 1)); an integer: that N_cc."""
	Ncc = 2*good_size_complex(lmax+1) if fine_cc is True else int(fine_cc)
	nf = Ncc+1
	return Ncc, np.pi*np.arange(nf)/Ncc, get_gridweights("CC", nf)/(2*np.pi)

def analysis_2d(*, map, spin, lmax, geometry, alm=None, mmax=None, mstart=None, phi0=0.0,
		nthreads=0, lstride=1, weights=False, fine_cc=True):
	"""ducc0.sht.experimental.analysis_2d (curvedsky.py:1032-1046).
	fine_cc=True (default) or an N_cc (the product's analysis="ducc0"): ducc0's own route for the CC / F1 / MW / MWflip grids as
	published (ducc0 >= 0.36 is a PyPI dependency of the reference and is absent here; src/ducc0/sht/sht.cc, analysis_2d ->
	resample_to_prepared_CC): the theta-interpolant of the rings -- low-passed to |k| < N_cc where the grid's circle has at least
	2 N_cc samples -- is evaluated on the Clenshaw-Curtis grid of N_cc + 1 rings (circle of 2 N_cc points) and integrated with that
	grid's quadrature weights; N_cc = 2 good_size_complex(lmax + 1) unless given.  (ducc0 then carries the weighted samples to the
	N_cc/2 + 1 rings of its Legendre stage with the transposed band-limited upsampling, which changes nothing in exact arithmetic:
	lambda_lm is band-limited to lmax < N_cc/2.)  Restated here as the direct sum over the N_cc + 1 rings.  A CC grid with
	nt >= 2 lmax + 2 is multiplied by its own quadrature weights directly (ducc0: need_first_resample = false), i.e. weights=True.
	On maps that are not band-limited the forms differ at the 1e-3 level, and no test or fixture of the reference pins which one
	ducc0 computes: parity unpinned against ducc0 itself for such maps; on band-limited maps all forms agree and are pinned
	(tests/golden/make_golden.py).
	fine_cc=False (the product's analysis="interpolant"): exact integration of the full theta-interpolant, evaluated on Gauss-Legendre nodes.
	weights=True (the product's analysis="weights"): ring quadrature weights + adjoint synthesis, the reference's cyl route
	(curvedsky.py:852-861, 1068-1084); exact for band-limited maps on grids with nt >= 2 lmax + 2."""
	map = np.asarray(map)
	nc, nt, nph = map.shape
	if mmax is None: mmax = lmax
	if mstart is None: mstart = _tri_mstart(lmax, mmax)
	if lmax > grid_maxlmax(geometry, nt):
		raise ValueError("too few rings for analysis up to requested lmax")
	if alm is None:      # (ducc0 allocates the output when it is not given: curvedsky.py:573 relies on it)
		alm = np.zeros((nc, _nelem_default(lmax, mmax, mstart, lstride)), np.complex64 if map.dtype == np.float32 else np.complex128)
	if fine_cc is True and geometry == "CC" and nt >= 2*lmax+2: weights = True
	if geometry in ("DH", "F2") or weights:
		w = get_gridweights(geometry, nt)/nph
		return adjoint_synthesis_2d(alm=alm, map=map*w[None, :, None], spin=spin, lmax=lmax,
			geometry=geometry, mmax=mmax, mstart=mstart, phi0=phi0, lstride=lstride)
	th, nphi, p0, rs = _grid_rings(geometry, nt, nph, phi0)
	leg = map2leg(map.reshape(nc, -1), nphi, p0, rs, mmax)          # [nc, nt, nm]
	N = grid_info(geometry, nt)["N"]
	if fine_cc:
		Ncc, thq, wq = _fine_cc_nodes(lmax, fine_cc)
		kcut = Ncc; nq = len(thq)
	else:
		nq = (N//2 + lmax)//2 + 2
		thq, wq = _gl(nq)
		kcut = None
	Ma = _interp_matrix(geometry, nt, thq, kcut)                    # [nq, N]  at theta_q
	Mb = _interp_matrix(geometry, nt, 2*np.pi-thq, kcut)            # at the mirror points 2pi - theta_q
	ring, mirrored = _extension(geometry, nt, None)
	par = ((np.arange(mmax+1)+spin) % 2)
	sign = np.where(mirrored[:, None] & (par[None, :] == 1), -1.0, 1.0)   # [N, nm]
	psgn = np.where(par == 1, -1.0, 1.0)
	legq = np.zeros((nc, nq, mmax+1), np.complex128)
	for c in range(nc):
		ext = leg[c, ring, :]*sign
		# the integral runs over the full circle against |sin|: only the (-1)^(m+s)-parity part of
		# the interpolant survives (pole samples of the wrong parity are projected out)
		legq[c] = 0.5*(Ma @ ext + (Mb @ ext)*psgn[None, :])
	legq *= (wq*2*np.pi/nph)[None, :, None]
	res = leg2alm(legq, spin, lmax, mmax, mstart, thq.astype(LD), alm.shape[-1], lstride)
	alm[...] = res.astype(alm.dtype)
	return alm

def adjoint_analysis_2d(*, alm, map, spin, lmax, geometry, mmax=None, mstart=None, phi0=0.0,
		nthreads=0, lstride=1, weights=False, fine_cc=True):
	"""Exact transpose of analysis_2d (curvedsky.py:1032), form by form (same keywords)."""
	nc, nt, nph = map.shape
	if mmax is None: mmax = lmax
	if mstart is None: mstart = _tri_mstart(lmax, mmax)
	if fine_cc is True and geometry == "CC" and nt >= 2*lmax+2: weights = True
	if geometry in ("DH", "F2") or weights:
		w = get_gridweights(geometry, nt)/nph
		synthesis_2d(alm=alm, map=map, spin=spin, lmax=lmax, geometry=geometry, mmax=mmax,
			mstart=mstart, phi0=phi0, lstride=lstride)
		map *= w[None, :, None].astype(map.dtype)
		return map
	th, nphi, p0, rs = _grid_rings(geometry, nt, nph, phi0)
	N = grid_info(geometry, nt)["N"]
	if fine_cc:
		Ncc, thq, wq = _fine_cc_nodes(lmax, fine_cc)
		kcut = Ncc; nq = len(thq)
	else:
		nq = (N//2 + lmax)//2 + 2
		thq, wq = _gl(nq)
		kcut = None
	Ma = _interp_matrix(geometry, nt, thq, kcut)
	Mb = _interp_matrix(geometry, nt, 2*np.pi-thq, kcut)
	ring, mirrored = _extension(geometry, nt, None)
	par = ((np.arange(mmax+1)+spin) % 2)
	sign = np.where(mirrored[:, None] & (par[None, :] == 1), -1.0, 1.0)
	psgn = np.where(par == 1, -1.0, 1.0)
	legq = alm2leg(np.atleast_2d(alm), spin, lmax, mmax, mstart, thq.astype(LD), lstride)
	legq = legq*(wq*2*np.pi/nph)[None, :, None]
	leg = np.zeros((nc, nt, mmax+1), np.complex128)
	for c in range(nc):
		ext = 0.5*(Ma.T @ legq[c] + Mb.T @ (legq[c]*psgn[None, :]))*sign
		np.add.at(leg[c], ring, ext)
	res = leg2map(leg, nphi, p0, rs, nt*nph)
	map[...] = res.reshape(map.shape).astype(map.dtype)
	return map


# ----------------------------------------------------------------------------------
# helpers used by tests / bench to make band-limited Gaussian inputs
# ----------------------------------------------------------------------------------
def nalm(lmax, mmax=None):
	if mmax is None: mmax = lmax
	return (mmax+1)*(2*lmax+2-mmax)//2

def rand_alm_simple(lmax, ncomp, seed, spin=(0, 2)):
	"""white-ish Gaussian alm with C_l = 1/(l+1)^2 (T) and 0.01 C_l (E,B), m=0 imaginary
	parts zero, l < |s| entries zero.  SURVEY 8(d) synthetic input recipe."""
	rng = np.random.default_rng(seed)
	n = nalm(lmax)
	alm = (rng.standard_normal((ncomp, n)) + 1j*rng.standard_normal((ncomp, n)))/np.sqrt(2)
	ms = _tri_mstart(lmax, lmax).astype(np.int64)
	l_of = np.zeros(n, np.int64)
	for m in range(lmax+1):
		l_of[ms[m]+m:ms[m]+lmax+1] = np.arange(m, lmax+1)
	amp = 1.0/(l_of+1.0)
	alm *= amp[None]
	alm[:, :lmax+1] = alm[:, :lmax+1].real*np.sqrt(2)
	# spin components
	ci = 0; si = 0; spin = list(np.atleast_1d(spin))
	while ci < ncomp:
		s = spin[si % len(spin)]
		k = 1 if s == 0 else 2
		if s != 0:
			alm[ci:ci+k] *= 0.1
			alm[ci:ci+k, l_of < s] = 0
		ci += k; si += 1
	return alm
