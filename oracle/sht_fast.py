"""Large-size CPU checkers built on oracle/sht_port.c -- TEST INFRASTRUCTURE ONLY.

The long-double oracle (sht_oracle.py) is O(lmax^3) Python and its analysis_2d builds dense
[nq, N] interpolation matrices: fine up to lmax ~ 500, useless at the BASELINE sizes
(lmax 4000 / 6000 / 10000).  This module restates the same mathematics in a form that runs in
seconds on SUBSETS of rings or of m at those sizes, so that the -m gpu tests can compare the HIP
path with a CPU computation at full size instead of relying on round trips alone:

  synth_rings()      alm -> leg[m][ring] for ALL m on a few rings (C port, OpenMP over m), then
  pixels_on_rings()  the ring values at chosen phi by direct summation over m
                     (= ducc0 synthesis_2d, pixell/curvedsky.py:907-924, evaluated pixel by pixel);
  theta_resample()   the exact integration of the theta-interpolant against |sin theta|
                     (= what analysis_2d must deliver, curvedsky.py:1032-1046) as an FFT-based
                     numpy computation per m column: interpolant spectrum -> convolution with the
                     |sin| series -> samples on a Clenshaw-Curtis grid that integrates degree
                     2 lmax exactly;
  analysis_columns() theta_resample + the C port's Legendre adjoint for a few m.

theta_resample is pinned to sht_oracle.analysis_2d (dense Gauss-Legendre evaluation, a different
algorithm) in tests/test_oracle_fast.py; the C port is pinned to the long-double oracle in
tests/test_oracle_port.py.  Nothing here is imported by the product path.
"""
import numpy as np
from . import sht_oracle as so, sht_port

def _good_even(n):
	"""smallest even 2^a 3^b 5^c >= n"""
	n = int(n)
	while True:
		r = n
		for p in (2, 3, 5):
			while r % p == 0: r //= p
		if r == 1 and n % 2 == 0: return n
		n += 1

def tri_columns(alm, lmax, msel, mstart=None):
	"""alm[nc, nelem] (triangular layout) -> A[len(msel), nc, lmax+1] with A[i, c, l] = a_{l, msel[i]} (l < m unused = 0)"""
	alm = np.atleast_2d(alm); nc = alm.shape[0]
	if mstart is None: mstart = so._tri_mstart(lmax, lmax)
	A = np.zeros((len(msel), nc, lmax+1), np.complex128)
	for i, m in enumerate(msel):
		o = int(mstart[m])
		A[i, :, m:] = alm[:, o+m:o+lmax+1]
	return A

def symmetric_subset(nring, idx):
	"""ascending ring subset closed under the north/south mirror i <-> nring-1-i"""
	idx = np.asarray(idx, int)
	return np.unique(np.concatenate([idx, nring-1-idx]))

def synth_rings(alm, spin, lmax, theta_sub, mchunk=512, mstart=None):
	"""leg[nm, nc, nsub] on the rings theta_sub (an ascending, mirror-symmetric list) for m = 0..lmax.
	(A list that is NOT closed under theta -> pi - theta is accepted, but a lone ring next to the south pole is then good to ~1e-11 of the map rms only:
	the port works from cos(theta).  tests/test_grid_fuzz.py::synth_rings_any evaluates the mirror-symmetric closure instead.)"""
	alm = np.atleast_2d(alm); nc = alm.shape[0]
	out = np.zeros((lmax+1, nc, len(theta_sub)), np.complex128)
	for m0 in range(0, lmax+1, mchunk):
		msel = np.arange(m0, min(lmax+1, m0+mchunk))
		out[msel] = sht_port.leg(spin, lmax, msel, theta_sub, alm=tri_columns(alm, lmax, msel, mstart))
	return out

def pixels_on_rings(leg, phi):
	"""leg[nm, nc, nsub], phi[nsub, npts] -> values[nc, nsub, npts] = Re F_0 + 2 Re sum_{m>0} F_m e^{i m phi}"""
	nm, nc, ns = leg.shape
	m = np.arange(nm, dtype=np.longdouble); tp = 2*np.pi*np.ones(1, np.longdouble)[0]
	out = np.zeros((nc, ns, phi.shape[1]))
	for r in range(ns):
		arg = np.outer(m, np.asarray(phi[r], np.longdouble)) % tp      # m phi reaches 1e5 rad: reduce in long double
		ph = np.exp(1j*arg.astype(np.float64))            # [nm, npts]
		ph[1:] *= 2
		for c in range(nc):
			out[c, r] = np.real(leg[:, c, r] @ ph)        # (F_0 enters with its real part only)
	return out

_TABLES = {}
def _resample_tables(geometry, ntheta, lmax):
	"""what theta_resample needs that does not depend on the data (cached: the ring table alone is a Python loop over the circle)"""
	key = (geometry, int(ntheta), int(lmax))
	if key in _TABLES: return _TABLES[key]
	g = so.grid_info(geometry, ntheta); N, c = g["N"], g["c"]; th0 = float(g["theta0"])
	ring = np.array([so._ring_of(jp, N, c, ntheta) for jp in range(N)])
	mirrored = np.arange(N) >= ntheta
	K2 = N//2
	k = np.arange(-K2, K2+1)
	Q = lmax+K2
	s_half = so._abs_sin_series(Q)
	s = np.concatenate([s_half[:0:-1], s_half])                     # q = -Q..Q
	nconv = len(k)+len(s)-1
	nf = _good_even(nconv)
	Ncc = _good_even(2*lmax+2)
	t = dict(N=N, th0=th0, ring=ring, mirrored=mirrored, K2=K2, k=k, kmod=k % N, phase=np.exp(-1j*k*th0), Q=Q, nconv=nconv, nf=nf,
		sfft=np.fft.fft(s, nf), hsel=np.arange(-lmax, lmax+1)+K2+Q, Ncc=Ncc, kk=np.arange(-lmax, lmax+1) % Ncc)
	_TABLES[key] = t
	return t

def theta_resample(L, par, geometry, ntheta, lmax, workers=None, fine_cc=None):
	"""Exact |sin|-weighted integration of the theta-interpolant, as samples on a CC grid.
	fine_cc = N_cc: the fine-CC form instead (sht_oracle.analysis_2d, ducc0's route): the interpolant, low-passed to |k| < N_cc when
	the circle has at least 2 N_cc samples, evaluated on the CC grid of N_cc + 1 rings and multiplied by that grid's quadrature
	weights; the returned grid is that fine grid.

	L[ncol, ntheta]: ring values of one m column each (any complex numbers: NOT required to be band limited);
	par[ncol]: parity (m+spin) % 2 of each column.  Returns (theta_cc[ncc], G[ncol, ncc]) such that
	   integral_0^pi f_par(theta) lambda(theta) sin(theta) dtheta = sum_j G[:, j] lambda(theta_cc[j])
	for every lambda that extends to the circle as a trigonometric polynomial of degree <= lmax with that parity,
	where f is the canonical trigonometric interpolant of the parity-extended samples (Nyquist term as a cosine
	about theta0, as sht_oracle._interp_matrix has it) and f_par its part of parity `par`."""
	L = np.atleast_2d(np.asarray(L, np.complex128)); ncol = L.shape[0]
	par = np.asarray(par, int).reshape(ncol)
	if workers:          # pocketfft through scipy (bench.py's cpu_baseline: one call per host thread, workers=1); numpy's otherwise
		import scipy.fft as sfft
		fft = lambda a, n=None, axis=-1: sfft.fft(a, n=n, axis=axis, workers=workers)
		ifft = lambda a, n=None, axis=-1: sfft.ifft(a, n=n, axis=axis, workers=workers)
	else: fft, ifft = np.fft.fft, np.fft.ifft
	T = _resample_tables(geometry, ntheta, lmax)
	N, th0, K2 = T["N"], T["th0"], T["K2"]
	# parity extension to the N full-circle samples
	sign = np.where(T["mirrored"][None, :] & (par[:, None] == 1), -1.0, 1.0)
	f = L[:, T["ring"]]*sign                                        # [ncol, N]
	# interpolant spectrum c_k, |k| <= N/2 (index k + K2)
	F = fft(f, axis=1)/N
	cspec = F[:, T["kmod"]]*T["phase"][None, :]
	if N % 2 == 0:
		X = F[:, K2]                                                  # (1/N) sum_j f_j (-1)^j
		cspec[:, 0]  = 0.5*X*np.exp(+1j*K2*th0)                       # k = -N/2
		cspec[:, -1] = 0.5*X*np.exp(-1j*K2*th0)                       # k = +N/2
	if fine_cc:
		Nf = int(fine_cc); M2 = 2*Nf
		k = T["k"]
		if M2 <= N: cspec = np.where((np.abs(k) < Nf)[None, :], cspec, 0)        # low pass (the Nyquist pair of N goes with it)
		keep = np.abs(k) <= min(K2, Nf)
		spec = np.zeros((ncol, M2), np.complex128)
		np.add.at(spec, (slice(None), k[keep] % M2), cspec[:, keep])               # (|k| = N/2 < N_cc: the two halves of the Nyquist term land on k and -k)
		fc = ifft(spec, axis=1)*M2                                                # f(2 pi j / M2)
		nfr = Nf+1
		psgn = np.where(par == 1, -1.0, 1.0)[:, None]
		fpar = 0.5*(fc[:, :nfr]+psgn*fc[:, (-np.arange(nfr)) % M2])
		w = so.get_gridweights("CC", nfr)/(2*np.pi)                               # sum = 2
		theta_f = np.pi*np.arange(nfr)/Nf
		return theta_f, fpar*w[None, :]
	# h = f |sin|: Fourier coefficients |k| <= lmax by convolution with the |sin| series (full convolution via zero-padded FFTs)
	H = ifft(fft(cspec, T["nf"], axis=1)*T["sfft"][None, :], axis=1)
	# index i of H <-> k = i - K2 - Q
	h = H[:, T["hsel"]]                                             # [ncol, 2 lmax + 1]
	# evaluate g(theta) = sum_{|k|<=lmax} h_k e^{ik theta} on the CC circle of Ncc > 2 lmax points
	Ncc = T["Ncc"]
	spec = np.zeros((ncol, Ncc), np.complex128)
	spec[:, T["kk"]] = h
	gcirc = ifft(spec, axis=1)*Ncc                           # g(2 pi j / Ncc)
	ncc = Ncc//2+1
	psgn = np.where(par == 1, -1.0, 1.0)[:, None]
	gpar = 0.5*(gcirc[:, :ncc]+psgn*gcirc[:, (-np.arange(ncc)) % Ncc])
	eps = np.full(ncc, 2.0); eps[0] = eps[-1] = 1.0
	# (1/2) integral over the circle = (1/2)(2 pi / Ncc) sum over the circle = (pi/Ncc) sum_rings eps_j ...
	G = gpar*(eps*np.pi/Ncc)[None, :]
	theta_cc = 2*np.pi*np.arange(ncc)/Ncc; theta_cc[-1] = np.pi
	return theta_cc, G

def analysis_columns(L, msel, spin, lmax, geometry, ntheta, nphi, fine_cc=None):
	"""analysis_2d restricted to the m columns msel.  L[len(msel), nc, ntheta] = sum_x ring[x] e^{-i m phi_x}
	(the ring-FFT output, phi0 included).  Returns alm columns [len(msel), nc, lmax+1].  fine_cc = N_cc: the fine-CC form (ducc0's route)."""
	L = np.asarray(L, np.complex128); nms, nc, nt = L.shape
	par = (np.asarray(msel)+spin) % 2
	th_cc, G = theta_resample(L.reshape(nms*nc, nt), np.repeat(par, nc), geometry, ntheta, lmax, fine_cc=fine_cc)
	G = G.reshape(nms, nc, -1)*(2*np.pi/nphi)
	return sht_port.leg(spin, lmax, np.asarray(msel), th_cc, leg=G)
