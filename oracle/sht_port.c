/* CPU port of the Legendre stage -- TEST / BASELINE INFRASTRUCTURE ONLY (see oracle/sht_oracle.py).
 *
 * Plain C (float64, OpenMP over m) restatement of the same published algorithm the HIP kernels
 * use, so that bench.py's cpu_baseline leg times a reasonable CPU implementation on the GPU
 * box's host cores ("kind": "port"):
 *   spin 0 : Ishioka (2018) two-step recurrence in x^2, north/south ring pairs share a chain;
 *   spin s : scaled three-term recurrence for the +-s functions;
 *   extended exponent for the start value sin^m(theta), libsharp-style m-limit per ring.
 * It restates what ducc0.sht.experimental.{synthesis,adjoint_synthesis} do inside (ducc0>=0.36.0,
 * not in the reference tree) for the calls at pixell/curvedsky.py:907-960, 1032-1084.
 * Validated against oracle/sht_oracle.py (long double, different recurrences) in tests/test_oracle_port.py.
 * Only tests/ and bench.py's cpu_baseline may use it; the product path never does.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BIG   0x1p+400
#define SMALL 0x1p-800
#define STEP  800

static void pow_scaled(double x, int n, double* mant, int* e) {
	double rm = 0.5; int re = 1, be = 0, d; double bm = frexp(x, &be);
	while (n) {
		if (n & 1) { rm *= bm; re += be; rm = frexp(rm, &d); re += d; }
		bm *= bm; be *= 2; bm = frexp(bm, &d); be += d;
		n >>= 1;
	}
	*mant = rm; *e = re;
}
static void to_scaled(double mant, int e, double* v, int* scale) {
	if (mant == 0.0) { *v = 0; *scale = 0; return; }
	int s = (e >= 0) ? (e + STEP/2)/STEP : -((-e + STEP/2)/STEP);
	if (s > 0) s = 0;
	*v = ldexp(mant, e - STEP*s); *scale = s;
}
static long double epsl(int l, int m) { if (l <= m) return 0; long double L = l, M = m; return sqrtl((L*L-M*M)/(4*L*L-1)); }

/* threads of the OpenMP loops (0: leave as is); returns the previous maximum */
int sht_port_set_threads(int n) {
#ifdef _OPENMP
	const int old = omp_get_max_threads();
	if (n > 0) omp_set_num_threads(n);
	return old;
#else
	(void)n; return 1;
#endif
}
int sht_port_threads(void) {
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

/* spin-0 Legendre for the listed m values on `np` ring pairs.
 * cth/sth[np]: northern ring of each pair; has_s[np]: 1 if the pair has a southern member.
 * alm[nmsel][lmax+1] complex (index l, entries l<m unused), legn/legs[nmsel][np] complex.
 * dir = 0: synthesis (alm -> leg), 1: adjoint (leg -> alm, unweighted sums). */
void sht_port_leg_s0(int lmax, int nmsel, const int* msel, int np, const double* cth, const double* sth, const int* has_s,
                     double* alm, double* legn, double* legs, int dir)
{
	const double ofs = fmax(100.0, 0.01*lmax);
#pragma omp parallel
	{
		double* lam1 = (double*)malloc(sizeof(double)*np*9);
		double* lam2 = lam1+np; double* csq = lam2+np; double* a1r = csq+np; double* a1i = a1r+np; double* a2r = a1i+np; double* a2i = a2r+np;
		double* gate = a2i+np; double* pol = gate+np;   /* pol[p] = 1: polar ring, recurrence in -sin^2 (see below) */
		int* sc = (int*)malloc(sizeof(int)*np);
#pragma omp for schedule(dynamic,1)
		for (int im = 0; im < nmsel; im++) {
			const int m = msel[im];
			const int nk = (lmax-m)/2+1;
			double* A = alm + (size_t)im*(lmax+1)*2;
			/* coefficients */
			double* ca = (double*)malloc(sizeof(double)*nk*4); double* cb = ca+nk; double* al = cb+nk; double* cab = al+nk;
			{
				long double cm = 1/sqrtl(4*3.141592653589793238462643383279502884L);
				for (int q = 1; q <= m; q++) cm = -cm*sqrtl((long double)(2*q+1)/(long double)(2*q));
				long double ap = 0, ac = sqrtl((long double)(2*m+3))*cm;
				for (int k = 0; k < nk; k++) {
					int lp = m+2*k+1;
					long double e2 = epsl(lp+1, m)*epsl(lp+1, m)+epsl(lp, m)*epsl(lp, m), f = epsl(lp, m)*epsl(lp-1, m), d = epsl(lp+1, m)*epsl(lp+2, m);
					long double an = (k == 0) ? ac/d : -f*ap/d;
					long double ak = ac/(an*d);
					ca[k] = (double)ak; cb[k] = (double)(-ak*e2); al[k] = (double)ac; cab[k] = (double)(ak-ak*e2);
					ap = ac; ac = an;
				}
			}
			/* start values, first active pair */
			int p0 = np;
			for (int p = 0; p < np; p++) {
				/* Near the poles x = cos(theta) rounds away the information about theta (1 - x ~ theta^2/2): rings with cos^2 > 1/2
				 * run the recurrence in -sin^2(theta) with the constant a+b (rounded from long double) instead of b:
				 * a x^2 + b = (a+b) - a sin^2.  Same measure as the HIP kernels take (legendre.hip, leg_wave_polar). */
				pol[p] = cth[p]*cth[p] > 0.5 ? 1.0 : 0.0;
				csq[p] = pol[p] != 0.0 ? -sth[p]*sth[p] : cth[p]*cth[p]; lam1[p] = 0; lam2[p] = 0; sc[p] = 0;
				a1r[p] = a1i[p] = a2r[p] = a2i[p] = 0;
				if ((double)m <= lmax*sth[p]+ofs) { double mt; int e; pow_scaled(sth[p], m, &mt, &e); to_scaled(mt, e, &lam2[p], &sc[p]); if (p < p0) p0 = p; }
				if (dir == 1) {
					const double nr = legn[((size_t)im*np+p)*2], ni = legn[((size_t)im*np+p)*2+1];
					const double sr = has_s[p] ? legs[((size_t)im*np+p)*2] : 0, si = has_s[p] ? legs[((size_t)im*np+p)*2+1] : 0;
					a1r[p] = nr+sr; a1i[p] = ni+si; a2r[p] = (nr-sr)*cth[p]; a2i[p] = (ni-si)*cth[p];
				}
			}
			double* M = (double*)calloc((size_t)nk*4, sizeof(double));   /* moments (dir 1) or pre-scaled alm (dir 0) */
			if (dir == 0) for (int k = 0; k < nk; k++) {
				int l = m+2*k;
				double e1 = (double)epsl(l+1, m), e2 = (double)epsl(l+2, m);
				double a0r = A[2*l], a0i = A[2*l+1];
				double b1r = l+1 <= lmax ? A[2*(l+1)] : 0, b1i = l+1 <= lmax ? A[2*(l+1)+1] : 0;
				double c2r = l+2 <= lmax ? A[2*(l+2)] : 0, c2i = l+2 <= lmax ? A[2*(l+2)+1] : 0;
				M[4*k] = al[k]*(e1*a0r+e2*c2r); M[4*k+1] = al[k]*(e1*a0i+e2*c2i); M[4*k+2] = al[k]*b1r; M[4*k+3] = al[k]*b1i;
			}
			/* ring blocks (like the GPU waves): only blocks that still hold a chain below scale 0 take the
			 * gated / rescaling path; the others run the plain vectorised loops */
			const int BLK = 128;
			if (dir == 1) memset(M, 0, sizeof(double)*(size_t)nk*4);
			for (int pb = p0; pb < np; pb += BLK) {
				const int pe = pb+BLK < np ? pb+BLK : np;
				int nscaled = 0, anylive = 0;
				for (int p = pb; p < pe; p++) { if (sc[p] < 0) nscaled++; if (lam2[p] != 0.0) anylive = 1; }
				if (!anylive) continue;
				for (int k = 0; k < nk; k++) {
					const double a = ca[k], b = cb[k], ab = cab[k];
					if (nscaled > 0) for (int p = pb; p < pe; p++) gate[p] = sc[p] == 0 ? lam2[p] : 0.0;
					const double* g = nscaled > 0 ? gate : lam2;
					if (dir == 0) {
						const double er = M[4*k], ei = M[4*k+1], orr = M[4*k+2], oi = M[4*k+3];
#pragma omp simd
						for (int p = pb; p < pe; p++) { a1r[p] += g[p]*er; a1i[p] += g[p]*ei; a2r[p] += g[p]*orr; a2i[p] += g[p]*oi; }
					} else {
						double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma omp simd reduction(+:t0,t1,t2,t3)
						for (int p = pb; p < pe; p++) { t0 += g[p]*a1r[p]; t1 += g[p]*a1i[p]; t2 += g[p]*a2r[p]; t3 += g[p]*a2i[p]; }
						M[4*k] += t0; M[4*k+1] += t1; M[4*k+2] += t2; M[4*k+3] += t3;
					}
#pragma omp simd
					for (int p = pb; p < pe; p++) { double t = (a*csq[p]+(pol[p] != 0.0 ? ab : b))*lam2[p]+lam1[p]; lam1[p] = lam2[p]; lam2[p] = t; }
					if (nscaled > 0) {
						for (int p = pb; p < pe; p++) if (sc[p] < 0 && fabs(lam2[p]) > BIG) { lam1[p] *= SMALL; lam2[p] *= SMALL; sc[p]++; if (sc[p] == 0) nscaled--; }
					}
				}
			}
			if (dir == 0) {
				for (int p = 0; p < np; p++) {
					legn[((size_t)im*np+p)*2] = a1r[p]+cth[p]*a2r[p]; legn[((size_t)im*np+p)*2+1] = a1i[p]+cth[p]*a2i[p];
					if (has_s[p]) { legs[((size_t)im*np+p)*2] = a1r[p]-cth[p]*a2r[p]; legs[((size_t)im*np+p)*2+1] = a1i[p]-cth[p]*a2i[p]; }
				}
			} else {
				for (int k = 0; k < nk; k++) {
					int l = m+2*k;
					double e1 = (double)epsl(l+1, m), e0 = (double)epsl(l, m);
					double vr = e1*al[k]*M[4*k], vi = e1*al[k]*M[4*k+1];
					if (k > 0) { vr += e0*al[k-1]*M[4*k-4]; vi += e0*al[k-1]*M[4*k-3]; }
					A[2*l] = vr; A[2*l+1] = vi;
					if (l+1 <= lmax) { A[2*(l+1)] = al[k]*M[4*k+2]; A[2*(l+1)+1] = al[k]*M[4*k+3]; }
				}
			}
			free(M); free(ca);
		}
		free(lam1); free(sc);
	}
}

/* spin-s Legendre: alm[nmsel][2][lmax+1] complex (E,B), leg{n,s}[nmsel][2][np] complex (Q_m,U_m). */
void sht_port_leg_spin(int spin, int lmax, int nmsel, const int* msel, int np, const double* cth, const double* sth,
                       const double* sh2, const double* ch2, const int* has_s, double* alm, double* legn, double* legs, int dir)
{
	const double ofs = fmax(100.0, 0.01*lmax);
	const int s = spin;
	const double sg = (s & 1) ? -1.0 : 1.0;
	const long double PIl = 3.141592653589793238462643383279502884L;
#pragma omp parallel
	{
		double* buf = (double*)malloc(sizeof(double)*np*16);
		double *gp1 = buf, *gp2 = gp1+np, *gm1 = gp2+np, *gm2 = gm1+np, *pnr = gm2+np, *pni = pnr+np, *mnr = pni+np, *mni = mnr+np,
			*psr = mni+np, *psi = psr+np, *msr = psi+np, *msi = msr+np, *ggp = msi+np, *ggm = ggp+np, *xv = ggm+np, *pol = xv+np;
		int* scp = (int*)malloc(sizeof(int)*np*2); int* scm = scp+np;
#pragma omp for schedule(dynamic,1)
		for (int im = 0; im < nmsel; im++) {
			const int m = msel[im];
			const int l0 = m > s ? m : s;
			const int nl = lmax-l0+1;
			double* E = alm + (size_t)im*2*(lmax+1)*2; double* B = E + (size_t)(lmax+1)*2;
			if (nl <= 0) continue;
			double* ca = (double*)malloc(sizeof(double)*nl*5); double* cb = ca+nl; double* be = cb+nl; double* cpb = be+nl; double* cmb = cpb+nl;
			{
				long double nrm;
				if (m < s) { long double h = 2*s+1; for (int i = 1; i <= s; i++) h = h*(long double)(s+i)/(long double)i;
					for (int q = 1; q <= m; q++) h = h*(long double)(s-q+1)/(long double)(s+q); nrm = sqrtl(h/(4*PIl)); }
				else { long double c2 = (long double)(2*s+1)/(4*PIl*powl(4.0L, s));
					for (int q = s+1; q <= m; q++) c2 = c2*(long double)(2*q+1)*(long double)(2*q)/(4*(long double)(q+s)*(long double)(q-s)); nrm = sqrtl(c2); }
				long double bp = 0, bc = ((m & 1) ? -1 : 1)*nrm;
				for (int l = l0; l <= lmax; l++) {
					long double L = l, Mm = m, Sp = s;
					long double S1 = sqrtl(((L+1)*(L+1)-Mm*Mm)*((L+1)*(L+1)-Sp*Sp)), S0 = sqrtl((L*L-Mm*Mm)*(L*L-Sp*Sp));
					long double q = sqrtl((2*L+3)/(2*L+1))*(2*L+1);
					long double A_ = q*(L+1)/S1, B_ = q*Mm*Sp/(L*S1), C_ = l > l0 ? sqrtl((2*L+3)/(2*L-1))*(L+1)*S0/(L*S1) : 0;
					long double bn = (l == l0) ? A_*bc : C_*bp;
					ca[l-l0] = (double)(A_*bc/bn); cb[l-l0] = (double)(B_*bc/bn); be[l-l0] = (double)bc;
					cpb[l-l0] = (double)(A_*bc/bn+B_*bc/bn); cmb[l-l0] = (double)(A_*bc/bn-B_*bc/bn);
					bp = bc; bc = bn;
				}
			}
			int p0 = np;
			for (int p = 0; p < np; p++) {
				gp1[p] = gm1[p] = gp2[p] = gm2[p] = 0; scp[p] = scm[p] = 0;
				/* polar rings: a x +- b = a u + (a +- b) with u = cos(theta) - 1 = -2 sin^2(theta/2) (cf. the spin-0 routine) */
				pol[p] = cth[p]*cth[p] > 0.5 ? 1.0 : 0.0;
				xv[p] = pol[p] != 0.0 ? -2.0*sh2[p]*sh2[p] : cth[p];
				pnr[p] = pni[p] = mnr[p] = mni[p] = psr[p] = psi[p] = msr[p] = msi[p] = 0;
				double t1 = lmax*sth[p]+ofs, b = -2.0*s*fabs(cth[p]), c = (double)s*s-t1*t1, discr = b*b-4*c;
				double mlim = discr <= 0 ? lmax : fmin((double)lmax, 0.5*(-b+sqrt(discr)));
				if ((double)m <= mlim+0.5) {
					double m1, m2; int e1, e2, e; double mt;
					if (p < p0) p0 = p;
					if (m >= s) {
						pow_scaled(sh2[p], m+s, &m1, &e1); pow_scaled(ch2[p], m-s, &m2, &e2); mt = m1*m2; e = e1+e2+m; { int d; mt = frexp(mt, &d); e += d; } to_scaled(mt, e, &gp2[p], &scp[p]);
						pow_scaled(sh2[p], m-s, &m1, &e1); pow_scaled(ch2[p], m+s, &m2, &e2); mt = m1*m2; e = e1+e2+m; { int d; mt = frexp(mt, &d); e += d; } to_scaled(mt, e, &gm2[p], &scm[p]);
					} else {
						pow_scaled(sh2[p], s+m, &m1, &e1); pow_scaled(ch2[p], s-m, &m2, &e2); mt = m1*m2; e = e1+e2; { int d; mt = frexp(mt, &d); e += d; } to_scaled(mt, e, &gp2[p], &scp[p]);
						pow_scaled(sh2[p], s-m, &m1, &e1); pow_scaled(ch2[p], s+m, &m2, &e2); mt = m1*m2; e = e1+e2; { int d; mt = frexp(mt, &d); e += d; } to_scaled(mt, e, &gm2[p], &scm[p]);
						if ((s-m) & 1) gm2[p] = -gm2[p];
					}
				}
				if (dir == 1) {
					size_t iq = (((size_t)im*2+0)*np+p)*2, iu = (((size_t)im*2+1)*np+p)*2;
					double qr = legn[iq], qi = legn[iq+1], ur = legn[iu], ui = legn[iu+1];
					pnr[p] = qr-ui; pni[p] = qi+ur; mnr[p] = qr+ui; mni[p] = qi-ur;
					if (has_s[p]) { qr = legs[iq]; qi = legs[iq+1]; ur = legs[iu]; ui = legs[iu+1]; psr[p] = qr-ui; psi[p] = qi+ur; msr[p] = qr+ui; msi[p] = qi-ur; }
				}
			}
			const int BLK = 128;
			double* Mt = (dir == 1) ? (double*)calloc((size_t)nl*4, sizeof(double)) : NULL;
			for (int pb = p0; pb < np; pb += BLK) {
				const int pe = pb+BLK < np ? pb+BLK : np;
				int nscaled = 0, anylive = 0;
				for (int p = pb; p < pe; p++) { if (scp[p] < 0) nscaled++; if (scm[p] < 0) nscaled++; if (gp2[p] != 0.0 || gm2[p] != 0.0) anylive = 1; }
				if (!anylive) continue;
				double sgn = ((l0+m) & 1) ? -1.0 : 1.0;
				for (int j = 0; j < nl; j++, sgn = -sgn) {
					const int l = l0+j;
					const double a = ca[j], b = cb[j], apb = cpb[j], amb = cmb[j];
					const double* Gp = gp2; const double* Gm = gm2;
					if (nscaled > 0) { for (int p = pb; p < pe; p++) { ggp[p] = scp[p] == 0 ? gp2[p] : 0; ggm[p] = scm[p] == 0 ? gm2[p] : 0; } Gp = ggp; Gm = ggm; }
					if (dir == 0) {
						const double Er = E[2*l], Ei = E[2*l+1], Br = B[2*l], Bi = B[2*l+1], bb = be[j];
						const double apr = -bb*(Er-Bi), api = -bb*(Ei+Br), amr = -sg*bb*(Er+Bi), ami = -sg*bb*(Ei-Br);
						const double sapr = sgn*apr, sapi = sgn*api, samr = sgn*amr, sami = sgn*ami;
#pragma omp simd
						for (int p = pb; p < pe; p++) {
							pnr[p] += Gp[p]*apr; pni[p] += Gp[p]*api; mnr[p] += Gm[p]*amr; mni[p] += Gm[p]*ami;
							psr[p] += Gm[p]*sapr; psi[p] += Gm[p]*sapi; msr[p] += Gp[p]*samr; msi[p] += Gp[p]*sami;
						}
					} else {
						double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma omp simd reduction(+:t0,t1,t2,t3)
						for (int p = pb; p < pe; p++) {
							t0 += Gp[p]*pnr[p]+sgn*Gm[p]*psr[p]; t1 += Gp[p]*pni[p]+sgn*Gm[p]*psi[p];
							t2 += Gm[p]*mnr[p]+sgn*Gp[p]*msr[p]; t3 += Gm[p]*mni[p]+sgn*Gp[p]*msi[p];
						}
						Mt[4*j] += t0; Mt[4*j+1] += t1; Mt[4*j+2] += t2; Mt[4*j+3] += t3;
					}
#pragma omp simd
					for (int p = pb; p < pe; p++) {
						double tp = a*xv[p]+(pol[p] != 0.0 ? apb : b), tm = a*xv[p]+(pol[p] != 0.0 ? amb : -b);
						double n1 = tp*gp2[p]-gp1[p], n2 = tm*gm2[p]-gm1[p];
						gp1[p] = gp2[p]; gp2[p] = n1; gm1[p] = gm2[p]; gm2[p] = n2;
					}
					if (nscaled > 0) for (int p = pb; p < pe; p++) {
						if (scp[p] < 0 && fabs(gp2[p]) > BIG) { gp1[p] *= SMALL; gp2[p] *= SMALL; scp[p]++; if (scp[p] == 0) nscaled--; }
						if (scm[p] < 0 && fabs(gm2[p]) > BIG) { gm1[p] *= SMALL; gm2[p] *= SMALL; scm[p]++; if (scm[p] == 0) nscaled--; }
					}
				}
			}
			if (dir == 1) {
				for (int j = 0; j < nl; j++) {
					const int l = l0+j; const double bb = be[j];
					const double t0 = Mt[4*j], t1 = Mt[4*j+1], t2 = Mt[4*j+2], t3 = Mt[4*j+3];
					E[2*l] = -0.5*bb*(t0+sg*t2); E[2*l+1] = -0.5*bb*(t1+sg*t3);
					B[2*l] = -0.5*bb*(t1-sg*t3); B[2*l+1] = 0.5*bb*(t0-sg*t2);
				}
				free(Mt);
			}
			if (dir == 0) for (int p = 0; p < np; p++) {
				size_t iq = (((size_t)im*2+0)*np+p)*2, iu = (((size_t)im*2+1)*np+p)*2;
				legn[iq] = 0.5*(pnr[p]+mnr[p]); legn[iq+1] = 0.5*(pni[p]+mni[p]); legn[iu] = 0.5*(pni[p]-mni[p]); legn[iu+1] = -0.5*(pnr[p]-mnr[p]);
				if (has_s[p]) { legs[iq] = 0.5*(psr[p]+msr[p]); legs[iq+1] = 0.5*(psi[p]+msi[p]); legs[iu] = 0.5*(psi[p]-msi[p]); legs[iu+1] = -0.5*(psr[p]-msr[p]); }
			}
			free(ca);
		}
		free(buf); free(scp);
	}
}
