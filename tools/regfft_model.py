"""Host model of the register-resident line FFT of pixell_amd/csrc/thetaline.hip (TEST / DESIGN TOOL, not product code).

A line of n complex points lives in the registers of one workgroup of NT threads, PMAX points per thread at most.  A transform
is a Stockham decimation-in-frequency sequence of register radix passes; between two passes the line goes through the LDS one
component (re, im) at a time.  The model executes exactly the index arithmetic of the kernel -- thread t, register slot c,
LDS word pad(i) -- vectorised over t, and checks it against numpy.fft and against the plain statement of the theta chain
(FftChain::to_cc, fftchain.hip).  Run: python tools/regfft_model.py
"""
import numpy as np

RADICES = (16, 15, 12, 10, 9, 8, 7, 6, 5, 4, 3, 2)

def plan_radices(n, NT, PMAX, first=None, last=None):
	"""fewest passes with K*R <= PMAX for every pass (K = ceil(n/(R NT)) butterflies per thread); larger radices first"""
	best = None
	def rec(rem, seq):
		nonlocal best
		if best is not None and len(seq) >= len(best): return
		if rem == 1:
			best = list(seq); return
		for R in RADICES:
			if rem % R: continue
			K = -(-(n//R)//NT)
			if K*R > PMAX: continue
			rec(rem//R, seq+[R])
	rec(n, [])
	return best

class Pass:
	def __init__(self, n, NT, PMAX, R, s):
		self.n, self.R, self.s = n, R, s
		self.nb = n//R
		self.K = -(-self.nb//NT)
		assert self.K*R <= PMAX
		# read pattern: slot c = i*R + k holds x[t + NT i + nb k] when t + NT i < nb
		self.roff = np.full(PMAX, 0); self.rlim = np.zeros(PMAX, int)
		for i in range(self.K):
			for k in range(R):
				c = i*R + k
				self.roff[c] = NT*i + self.nb*k
				self.rlim[c] = self.nb - NT*i

def pad(i): return i      # (the kernel's layout: regfft_dev.hpp)

class Line:
	"""the registers of a workgroup: v[t, c]"""
	def __init__(self, NT, PMAX):
		self.NT, self.PMAX = NT, PMAX
		self.v = np.zeros((NT, PMAX), complex)
		self.t = np.arange(NT)
	def fill(self, ps, f):
		"""slot c of thread t <- f(idx) in the read pattern of pass ps"""
		for c in range(self.PMAX):
			ok = self.t < ps.rlim[c]
			idx = self.t + ps.roff[c]
			self.v[ok, c] = f(idx[ok])
	def butterflies(self, ps, twn, inverse=False):
		"""R-point DFTs of the K butterflies of every thread + the Stockham twiddles W_n^{(b - b mod s) j}"""
		R, s, n = ps.R, ps.s, ps.n
		w = np.exp(-2j*np.pi*np.arange(R)[:, None]*np.arange(R)[None, :]/R)
		for i in range(ps.K):
			b = self.t + self.NT*i
			ok = b < ps.nb
			a = self.v[:, i*R:(i+1)*R]
			o = a @ w.T
			e = b - b % s
			tw = np.exp(-2j*np.pi*(e[:, None]*np.arange(R)[None, :] % n)/n) if s < ps.nb else np.ones((self.NT, R))
			self.v[ok, i*R:(i+1)*R] = (o*tw)[ok]
	def write(self, ps, lds):
		"""outputs of pass ps into the LDS line (natural Stockham positions)"""
		R, s = ps.R, ps.s
		for i in range(ps.K):
			b = self.t + self.NT*i
			ok = b < ps.nb
			base = R*b - (R - 1)*(b % s)
			for j in range(R):
				idx = base + s*j
				lds[pad(idx[ok])] = self.v[ok, i*R + j]
	def read(self, ps, lds):
		self.fill(ps, lambda idx: lds[pad(idx)])
	def gather(self, ps):
		"""natural-order copy of the line when the registers hold the OUTPUT of the last pass ps (s = nb): slot (i, j) = X[b + nb j]"""
		out = np.zeros(ps.n, complex)
		for c in range(self.PMAX):
			ok = self.t < ps.rlim[c]
			out[(self.t + ps.roff[c])[ok]] = self.v[ok, c]
		return out

def make_passes(n, NT, PMAX, radices):
	ps, s = [], 1
	for R in radices:
		ps.append(Pass(n, NT, PMAX, R, s)); s *= R
	assert s == n
	return ps

def fft_line(line, passes, lds):
	"""registers hold the input in the read pattern of passes[0]; on return they hold the spectrum in the read pattern of a pass
	with the radix of passes[-1] (thread t, slot (i, j): X[t + NT i + nb j])"""
	for p, ps in enumerate(passes):
		if p > 0:
			line.write(passes[p-1], lds); line.read(ps, lds)
		line.butterflies(ps, None)

# ------------------------------------------------------------------------------------------------------------------------
# the theta chain of the analysis (FftChain::to_cc), plain statement and register form
# ------------------------------------------------------------------------------------------------------------------------
def pair_src(a, b, N, nr, mir_c, a_odd, wring=None):
	z = np.zeros(N, complex)
	for j in range(N):
		src, mir = j, False
		if j >= nr:
			src = N - j - mir_c
			if src < 0: src += N
			mir = True
		tj = 2*j + mir_c
		selfm = tj in (0, N, 2*N)
		va, vb = a[src], b[src]
		vo = va if a_odd else vb
		if selfm: vo = 0
		elif mir: vo = -vo
		if a_odd: va = vo
		else: vb = vo
		z[j] = (va + vb)*(wring[src] if wring is not None else 1.0)
	return z

def resize_rule(jp, X1, X2, kmax, nyq):
	"""source bin and factor of slot jp of the X2 spectrum (None: zero); the sign of kappa selects conj(ph)"""
	kap = jp if 2*jp <= X2 else jp - X2
	ak = abs(kap)
	if (kmax >= 0 and ak > kmax) or 2*ak > X1: return None
	k = kap if kap >= 0 else kap + X1
	return k, (0.5 if (nyq and 2*ak == X1) else 1.0), ak, kap < 0

def resize(Y, X1, X2, kmax, nyq, ph):
	Z = np.zeros(X2, complex)
	for jp in range(X2):
		r = resize_rule(jp, X1, X2, kmax, nyq)
		if r is None: continue
		k, f, ak, neg = r
		v = Y[k]*f
		if ph is not None: v = v*(np.conj(ph[ak]) if neg else ph[ak])
		Z[jp] = v
	return Z

def split(U, X, mir_c, nr_out, a_odd, w, self_half=False):
	oa, ob = np.zeros(nr_out, complex), np.zeros(nr_out, complex)
	for t in range(nr_out):
		tm = (X - t - mir_c) % X
		z = U[t]
		if tm == t: ev, od = (0.5*z if self_half else z), 0
		else: ev, od = 0.5*(z + U[tm]), 0.5*(z - U[tm])
		va, vb = (od, ev) if a_odd else (ev, od)
		oa[t], ob[t] = va*w[t], vb*w[t]
	return oa, ob

def to_cc_plain(a, b, N, M, Ncc, nr, mir_c, a_odd, lmax, ph, sigma, wcc):
	z = pair_src(a, b, N, nr, mir_c, a_odd)
	Y = np.fft.fft(z)
	Z = resize(Y, N, M, -1 if M > N else M//2 - 1, 1 if M > N else 0, ph)
	V = np.fft.fft(sigma*np.fft.ifft(Z)*M)
	U = np.fft.ifft(resize(V, M, Ncc, lmax, 0, None))*Ncc
	return split(U, Ncc, 0, Ncc//2 + 1, a_odd, wcc)

def to_cc_regs(a, b, N, M, Ncc, nr, mir_c, a_odd, lmax, ph, sigma, wcc, NT, PMAX):
	rN = plan_radices(N, NT, PMAX); rM = plan_radices(M, NT, PMAX); rC = plan_radices(Ncc, NT, PMAX)
	assert rN and rM and rC, (rN, rM, rC)
	pN = make_passes(N, NT, PMAX, rN)
	pMi = make_passes(M, NT, PMAX, rM[::-1])      # backward: ends with the radix the forward transform starts with
	pMf = make_passes(M, NT, PMAX, rM)
	pC = make_passes(Ncc, NT, PMAX, rC)
	lds = np.zeros(max(N, M, Ncc) + 16, complex)
	line = Line(NT, PMAX)
	z = pair_src(a, b, N, nr, mir_c, a_odd)
	line.fill(pN[0], lambda idx: z[idx])
	fft_line(line, pN, lds)
	# spectrum of N in natural order -> LDS; the first pass of the backward M transform reads it through the resize rule
	last = Pass(N, NT, PMAX, rN[-1], N//rN[-1])
	lds[pad(np.arange(N))] = line.gather(last)
	def rd(X1, X2, kmax, nyq, phs):
		def f(idx):
			out = np.zeros(len(idx), complex)
			for q, jp in enumerate(idx):
				r = resize_rule(int(jp), X1, X2, kmax, nyq)
				if r is None: continue
				k, fac, ak, neg = r
				v = lds[pad(k)]*fac
				if phs is not None: v = v*(np.conj(phs[ak]) if neg else phs[ak])
				out[q] = np.conj(v)      # backward transform = conj FFT conj
			return out
		return f
	line.v[:] = 0
	line.fill(pMi[0], rd(N, M, -1 if M > N else M//2 - 1, 1 if M > N else 0, ph))
	fft_line(line, pMi, lds)
	# pointwise in registers: slot (i, j) of the last pass = sample t + NT i + nb j, and the forward transform starts from the same slots
	lastM = pMf[0]
	for c in range(PMAX):
		ok = line.t < lastM.rlim[c]
		idx = (line.t + lastM.roff[c])[ok]
		line.v[ok, c] = np.conj(line.v[ok, c])*sigma[idx]
	fft_line(line, pMf, lds)
	lastF = Pass(M, NT, PMAX, rM[-1], M//rM[-1])
	lds[pad(np.arange(M))] = line.gather(lastF)
	line.v[:] = 0
	line.fill(pC[0], rd(M, Ncc, lmax, 0, None))
	fft_line(line, pC, lds)
	lastC = Pass(Ncc, NT, PMAX, rC[-1], Ncc//rC[-1])
	U = np.conj(line.gather(lastC))
	return split(U, Ncc, 0, Ncc//2 + 1, a_odd, wcc), (rN, rM, rC)

if __name__ == "__main__":
	rng = np.random.default_rng(1)
	for n, NT, PMAX in [(10800, 1024, 21), (16128, 1024, 21), (8064, 1024, 21), (360, 64, 21), (1008, 64, 21), (720, 64, 21), (252, 64, 21), (504, 64, 21)]:
		r = plan_radices(n, NT, PMAX)
		if r is None: print(n, NT, PMAX, "no plan"); continue
		ps = make_passes(n, NT, PMAX, r)
		x = rng.standard_normal(n) + 1j*rng.standard_normal(n)
		line = Line(NT, PMAX); lds = np.zeros(n + 16, complex)
		line.fill(ps[0], lambda idx: x[idx])
		fft_line(line, ps, lds)
		X = line.gather(Pass(n, NT, PMAX, r[-1], n//r[-1]))
		print(n, NT, PMAX, r, [p.K for p in ps], "err", np.abs(X - np.fft.fft(x)).max()/np.abs(X).max())
	# theta chain, small grid: F1 with ny = 90 rings (N = 180, mir_c = 1), lmax = 60 -> Ncc = 2*63 = 126, M = 252
	for (ny, lmax, Ncc, a_odd) in [(90, 60, 126, 0), (90, 60, 126, 1), (200, 60, 126, 0)]:
		N, M, nr, mir_c = 2*ny, 2*Ncc, ny, 1
		a = rng.standard_normal(nr) + 1j*rng.standard_normal(nr); b = rng.standard_normal(nr) + 1j*rng.standard_normal(nr)
		th0 = mir_c*np.pi/N
		ph = np.exp(-1j*np.arange(N//2 + 1)*th0)
		sigma = rng.standard_normal(M) + 0j; wcc = rng.standard_normal(Ncc//2 + 1)
		pa, pb = to_cc_plain(a, b, N, M, Ncc, nr, mir_c, a_odd, lmax, ph, sigma, wcc)
		(ra, rb), rad = to_cc_regs(a, b, N, M, Ncc, nr, mir_c, a_odd, lmax, ph, sigma, wcc, 64, 21)
		print("to_cc", ny, lmax, a_odd, rad, "err", max(np.abs(pa - ra).max(), np.abs(pb - rb).max())/np.abs(pa).max())
