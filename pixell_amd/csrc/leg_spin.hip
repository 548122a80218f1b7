// Spin-s Legendre kernels for gfx950 (design notes: head of legendre.hip): leg_syn_spin / leg_ana_spin (one wave per workgroup, K ring pairs per lane) and the
// FP64-MFMA forms of batched calls, leg_syn_spin_mm / leg_ana_spin_mm.  Replaces ducc0's alm2leg / leg2alm for spin > 0 as reached from
// pixell/curvedsky.py:907-960, 1032-1084.
#include "legendre_dev.hpp"

namespace pxs {

// ---------------------------------------------------------------------------------
// spin-s kernels.  rows l = l0..lmax.  chains G+ (spin +s) and G- (spin -s) of the NORTH ring;
// south ring: F+_S = (-1)^(l+m) F-_N, F-_S = (-1)^(l+m) F+_N.
// G_{l+1} = (a x +- b) G_l - G_{l-1}; in polar waves x -> u = -2 sin^2(theta/2), +-b -> a +- b.
// ---------------------------------------------------------------------------------
template<int K> struct SpinState {
	double x[K], gp1[K], gp2[K], gm1[K], gm2[K];
	int scp[K], scm[K];
};

template<int K> __device__ __forceinline__ bool spin_init(const LegK& a, int wv, int lane, int m, SpinState<K>& S, int* rn, int* rs, bool polar) {
	const int s_ = a.spin;
	bool alive_any = false;
#pragma unroll
	for (int s = 0; s < K; s++) {
		const int p = (wv*K + s)*64 + lane;
		const bool valid = p < a.npairs;
		rn[s] = valid ? a.ring_n[p] : -1; rs[s] = valid ? a.ring_s[p] : -1;
		const double cth = valid ? a.cth[p] : 0.0;
		const double sth = valid ? a.sth[p] : 0.0;
		const double shh = valid ? a.sh2[p] : 0.0;
		S.x[s] = polar ? -2.0*shh*shh : cth;
		// libsharp's m-limit generalised to spin: rings with m beyond it carry nothing up to lmax
		const double t1 = a.lmax*sth + a.ofs;
		const double b = -2.0*s_*fabs(cth);
		const double c = (double)s_*s_ - t1*t1;
		const double discr = b*b - 4*c;
		const double mlim = discr <= 0 ? a.lmax : fmin((double)a.lmax, 0.5*(-b + sqrt(discr)));
		const bool alive = valid && ((double)m <= mlim + 0.5);
		S.gp1[s] = S.gm1[s] = 0; S.gp2[s] = S.gm2[s] = 0; S.scp[s] = S.scm[s] = 0;
		if (alive && a.seed_mode != 2) {
			const double sh = shh, ch = a.ch2[p];
			double m1, m2; int e1, e2;
			if (m >= s_) {
				pow_scaled(sh, m + s_, m1, e1); pow_scaled(ch, m - s_, m2, e2);
				double mt = m1*m2; int e = e1 + e2 + m; frexp_norm(mt, e); to_scaled(mt, e, S.gp2[s], S.scp[s]);
				pow_scaled(sh, m - s_, m1, e1); pow_scaled(ch, m + s_, m2, e2);
				mt = m1*m2; e = e1 + e2 + m; frexp_norm(mt, e); to_scaled(mt, e, S.gm2[s], S.scm[s]);
			} else {
				pow_scaled(sh, s_ + m, m1, e1); pow_scaled(ch, s_ - m, m2, e2);
				double mt = m1*m2; int e = e1 + e2; frexp_norm(mt, e); to_scaled(mt, e, S.gp2[s], S.scp[s]);
				pow_scaled(sh, s_ - m, m1, e1); pow_scaled(ch, s_ + m, m2, e2);
				mt = m1*m2; e = e1 + e2; frexp_norm(mt, e); to_scaled(mt, e, S.gm2[s], S.scm[s]);
				if ((s_ - m) & 1) S.gm2[s] = -S.gm2[s];
			}
		}
		alive_any |= alive;
	}
	return alive_any;
}

// phase A of the spin kernels (see S0_PHASE_A): 4 steps per rescale / activity test; sgn is unchanged by 4 steps
#define SPIN_PHASE_A \
	while (j + 4 <= nl) { \
		bool act = false; \
		_Pragma("unroll") for (int s = 0; s < K; s++) act |= (S.scp[s] == 0 && S.gp2[s] != 0.0) || (S.scm[s] == 0 && S.gm2[s] != 0.0); \
		if (__any(act)) break; \
		const double4_t q0 = LDC(coef, j), q1 = LDC(coef, j+1), q2 = LDC(coef, j+2), q3 = LDC(coef, j+3); \
		_Pragma("unroll") for (int s = 0; s < K; s++) { \
			double ax; \
			ax = q0.a*S.x[s]; S.gp1[s] = fma(ax + (polar ? q0.c : q0.b), S.gp2[s], -S.gp1[s]); S.gm1[s] = fma(ax + (polar ? q0.d : -q0.b), S.gm2[s], -S.gm1[s]); \
			ax = q1.a*S.x[s]; S.gp2[s] = fma(ax + (polar ? q1.c : q1.b), S.gp1[s], -S.gp2[s]); S.gm2[s] = fma(ax + (polar ? q1.d : -q1.b), S.gm1[s], -S.gm2[s]); \
			ax = q2.a*S.x[s]; S.gp1[s] = fma(ax + (polar ? q2.c : q2.b), S.gp2[s], -S.gp1[s]); S.gm1[s] = fma(ax + (polar ? q2.d : -q2.b), S.gm2[s], -S.gm1[s]); \
			ax = q3.a*S.x[s]; S.gp2[s] = fma(ax + (polar ? q3.c : q3.b), S.gp1[s], -S.gp2[s]); S.gm2[s] = fma(ax + (polar ? q3.d : -q3.b), S.gm1[s], -S.gm2[s]); \
			if (S.scp[s] < 0 && fabs(S.gp2[s]) > SC_BIG) { S.gp1[s] *= SC_SMALL; S.gp2[s] *= SC_SMALL; S.scp[s]++; } \
			if (S.scm[s] < 0 && fabs(S.gm2[s]) > SC_BIG) { S.gm1[s] *= SC_SMALL; S.gm2[s] *= SC_SMALL; S.scm[s]++; } \
		} \
		j += 4; \
	}

// (step coefficient a x +- b as one FMA with the additive constant copied to a VGPR once per step -- gfx950 allows one
// scalar source per VALU op -- instead of a multiply shared by two adds: 12 + 2/K instead of 13 VALU ops per ring pair and l)
// two fast steps of the spin synthesis (G1/G2 swap roles).  The south-ring sums take (-1)^(l+m) a: they are
// accumulated with sign +1 on even steps and -1 on odd steps and multiplied by the sign of the first step at the end.
#define SPIN_SYN_PAIR(f0, f1, a0, a1) { \
	{ \
		const double ca = f0.a, c1 = polar ? f0.c : f0.b, c2 = polar ? f0.d : -f0.b; \
		PXS_VCOPY(v1, c1); PXS_VCOPY(v2, c2); \
		_Pragma("unroll") for (int s = 0; s < K; s++) { \
			const double gp = S.gp2[s], gm = S.gm2[s]; \
			pnr[s] = fma(gp, a0.a, pnr[s]); pni[s] = fma(gp, a0.b, pni[s]); \
			mnr[s] = fma(gm, a0.c, mnr[s]); mni[s] = fma(gm, a0.d, mni[s]); \
			qsr[s] = fma(gm, a0.a, qsr[s]); qsi[s] = fma(gm, a0.b, qsi[s]); \
			nsr[s] = fma(gp, a0.c, nsr[s]); nsi[s] = fma(gp, a0.d, nsi[s]); \
			S.gp1[s] = fma(fma(ca, S.x[s], v1), gp, -S.gp1[s]); S.gm1[s] = fma(fma(ca, S.x[s], v2), gm, -S.gm1[s]); \
		} \
	} \
	{ \
		const double ca = f1.a, c1 = polar ? f1.c : f1.b, c2 = polar ? f1.d : -f1.b; \
		PXS_VCOPY(v1, c1); PXS_VCOPY(v2, c2); \
		_Pragma("unroll") for (int s = 0; s < K; s++) { \
			const double gp = S.gp1[s], gm = S.gm1[s]; \
			pnr[s] = fma(gp, a1.a, pnr[s]); pni[s] = fma(gp, a1.b, pni[s]); \
			mnr[s] = fma(gm, a1.c, mnr[s]); mni[s] = fma(gm, a1.d, mni[s]); \
			qsr[s] = fma(-gm, a1.a, qsr[s]); qsi[s] = fma(-gm, a1.b, qsi[s]); \
			nsr[s] = fma(-gp, a1.c, nsr[s]); nsi[s] = fma(-gp, a1.d, nsi[s]); \
			S.gp2[s] = fma(fma(ca, S.x[s], v1), gp, -S.gp2[s]); S.gm2[s] = fma(fma(ca, S.x[s], v2), gm, -S.gm2[s]); \
		} \
	} }

// (leg_syn_spin<3> must stay below 128 VGPRs = 4 waves per SIMD; computing the lane as threadIdx.x & 63 for multi-wave
// workgroups once pushed it to 132 = 3 waves and leg_syn from 120 to 151 ms at config 3 -- keep an eye on that cliff.)
// (leg_syn_spin<3> held to 96 VGPRs = 5 waves per SIMD by __launch_bounds__: 20 bytes of spills, 102.0 -> 100.1 ms at C3: inside the noise, not kept)
template<int K> __global__ __launch_bounds__(64) void leg_syn_spin(const LegK a)
{
	const int lane = threadIdx.x; int wv, m, bb;
	if (!leg_block(a, wv, m, bb)) return;
	const int l0 = max(m, a.spin);
	const int nl = a.lmax - l0 + 1;
	double2* __restrict__ outq = a.leg + (long)bb*a.leg_bs + (long)m*a.ld;
	double2* __restrict__ outu = a.leg + (long)bb*a.leg_bs + ((long)a.nm + m)*a.ld;
	SpinState<K> S; int rn[K], rs[K];
	// north: P = sum G+ a+, M = sum G- a-;  south (before the sign): qs = sum +-G- a+, ns = sum +-G+ a-
	double pnr[K], pni[K], mnr[K], mni[K], qsr[K], qsi[K], nsr[K], nsi[K];
#pragma unroll
	for (int s = 0; s < K; s++) pnr[s] = pni[s] = mnr[s] = mni[s] = qsr[s] = qsi[s] = nsr[s] = nsi[s] = 0;
	const bool polar = leg_wave_polar(a, wv, K);
	const bool alive_any = spin_init<K>(a, wv, lane, m, S, rn, rs, polar);
	double sg0 = 1.0;
	if (nl > 0 && __any(alive_any)) {
		const long row0 = PXS_UNIFORM_LONG(a.row[m]);
		const double4_t* __restrict__ coef = a.coef + row0;
		const double4_t* __restrict__ at = reinterpret_cast<const double4_t*>(a.almt + (long)bb*a.almt_bs) + row0;
		int j = 0;
		SPIN_SEEDED_PHASE_A
		j = PXS_UNIFORM_INT(j); coef = (const double4_t*)PXS_UNIFORM_LONG((long)coef);
		PXS_COUNT(0, (long)(nl - j)*K*12 + (a.seed_mode != 2 ? (long)j*K*4 : 0L));
		sg0 = ((l0 + j + m) & 1) ? -1.0 : 1.0;      // (-1)^(l+m) of the first accumulated step; pairs of steps keep the parity
		// phase B: plain fast steps; every 4 steps the chains below scale 0 are rescaled.  A lane's sums hold scaled-up
		// garbage until both of its chains are at scale 0, when they are reset (true terms before that: < 2^-340 of the result)
		while (j + 1 < nl) {
			bool pend = false;
#pragma unroll
			for (int s = 0; s < K; s++) pend |= (S.scp[s] < 0) || (S.scm[s] < 0);
			if (!__any(pend)) break;
			for (int it = 0; it < 2 && j + 1 < nl; it++, j += 2) {
				const double4_t f0 = LDC(coef, j), f1 = LDC(coef, j+1), a0 = LDC(at, j), a1 = LDC(at, j+1);
				SPIN_SYN_PAIR(f0, f1, a0, a1)
			}
#pragma unroll
			for (int s = 0; s < K; s++) {
				const bool was = (S.scp[s] < 0) || (S.scm[s] < 0);
				if (S.scp[s] < 0 && fabs(S.gp2[s]) > SC_BIG) { S.gp1[s] *= SC_SMALL; S.gp2[s] *= SC_SMALL; S.scp[s]++; }
				if (S.scm[s] < 0 && fabs(S.gm2[s]) > SC_BIG) { S.gm1[s] *= SC_SMALL; S.gm2[s] *= SC_SMALL; S.scm[s]++; }
				if (was && S.scp[s] == 0 && S.scm[s] == 0) pnr[s] = pni[s] = mnr[s] = mni[s] = qsr[s] = qsi[s] = nsr[s] = nsi[s] = 0;
			}
		}
#pragma unroll
		for (int s = 0; s < K; s++)
			if (S.scp[s] < 0 || S.scm[s] < 0) {      // never reached scale 0
				pnr[s] = pni[s] = mnr[s] = mni[s] = qsr[s] = qsi[s] = nsr[s] = nsi[s] = 0;
				S.gp1[s] = S.gp2[s] = S.gm1[s] = S.gm2[s] = 0;
			}
		// phase C: fast loop, next coefficients prefetched
		double4_t f0 = LDC(coef, j), f1 = LDC(coef, j+1), a0 = LDC(at, j), a1 = LDC(at, j+1);
#ifndef PXS_NO_PHASEC_UNROLL
		while (j + 3 < nl) {      // (two pairs per iteration on alternating row sets, see leg_syn_s0)
			double4_t n0 = LDC(coef, j+2), n1 = LDC(coef, j+3), m0 = LDC(at, j+2), m1 = LDC(at, j+3);
			SPIN_SYN_PAIR(f0, f1, a0, a1)
			j += 2;
			f0 = LDC(coef, j+2); f1 = LDC(coef, j+3); a0 = LDC(at, j+2); a1 = LDC(at, j+3);
			SPIN_SYN_PAIR(n0, n1, m0, m1)
			j += 2;
		}
#endif
		for (; j + 1 < nl; j += 2) {
			const double4_t n0 = LDC(coef, j+2), n1 = LDC(coef, j+3), m0 = LDC(at, j+2), m1 = LDC(at, j+3);
			SPIN_SYN_PAIR(f0, f1, a0, a1)
			f0 = n0; f1 = n1; a0 = m0; a1 = m1;
		}
		if (j < nl) {
#pragma unroll
			for (int s = 0; s < K; s++) {
				const double gp = S.gp2[s], gm = S.gm2[s];
				pnr[s] = fma(gp, a0.a, pnr[s]); pni[s] = fma(gp, a0.b, pni[s]);
				mnr[s] = fma(gm, a0.c, mnr[s]); mni[s] = fma(gm, a0.d, mni[s]);
				qsr[s] = fma(gm, a0.a, qsr[s]); qsi[s] = fma(gm, a0.b, qsi[s]);
				nsr[s] = fma(gp, a0.c, nsr[s]); nsi[s] = fma(gp, a0.d, nsi[s]);
			}
		}
	}
	// Q = (P+M)/2, U = -i (P-M)/2.  The ring indices are re-read here rather than kept in registers through the loops.
#pragma unroll
	for (int s = 0; s < K; s++) {
		const int p = (wv*K + s)*64 + lane;
		const int rn_ = p < a.npairs ? a.ring_n[p] : -1, rs_ = p < a.npairs ? a.ring_s[p] : -1;
		if (rn_ >= 0) {
			outq[rn_] = make_double2(0.5*(pnr[s] + mnr[s]), 0.5*(pni[s] + mni[s]));
			outu[rn_] = make_double2(0.5*(pni[s] - mni[s]), -0.5*(pnr[s] - mnr[s]));
		}
		if (rs_ >= 0) {
			const double psr = sg0*qsr[s], psi = sg0*qsi[s], msr = sg0*nsr[s], msi = sg0*nsi[s];
			outq[rs_] = make_double2(0.5*(psr + msr), 0.5*(psi + msi));
			outu[rs_] = make_double2(0.5*(psi - msi), -0.5*(psr - msr));
		}
	}
}

// two fast steps of the spin analysis; mu+ = G+ T+_N + sgn G- T+_S, mu- = G- T-_N + sgn G+ T-_S with the sign of the
// first step already folded into the south-ring data (even steps +, odd steps -)
#define SPIN_ANA_PAIR(f0, f1) { \
	double t0 = 0, t1 = 0, t2 = 0, t3 = 0, u0 = 0, u1 = 0, u2 = 0, u3 = 0; \
	{ \
		const double ca = f0.a, c1 = polar ? f0.c : f0.b, c2 = polar ? f0.d : -f0.b; \
		PXS_VCOPY(v1, c1); PXS_VCOPY(v2, c2); \
		_Pragma("unroll") for (int s = 0; s < K; s++) { \
			const double gp = S.gp2[s], gm = S.gm2[s]; \
			t0 = fma(gp, tpnr[s], t0); t0 = fma(gm, tpsr[s], t0); \
			t1 = fma(gp, tpni[s], t1); t1 = fma(gm, tpsi[s], t1); \
			t2 = fma(gm, tmnr[s], t2); t2 = fma(gp, tmsr[s], t2); \
			t3 = fma(gm, tmni[s], t3); t3 = fma(gp, tmsi[s], t3); \
			S.gp1[s] = fma(fma(ca, S.x[s], v1), gp, -S.gp1[s]); S.gm1[s] = fma(fma(ca, S.x[s], v2), gm, -S.gm1[s]); \
		} \
	} \
	{ \
		const double ca = f1.a, c1 = polar ? f1.c : f1.b, c2 = polar ? f1.d : -f1.b; \
		PXS_VCOPY(v1, c1); PXS_VCOPY(v2, c2); \
		_Pragma("unroll") for (int s = 0; s < K; s++) { \
			const double gp = S.gp1[s], gm = S.gm1[s]; \
			u0 = fma(gp, tpnr[s], u0); u0 = fma(-gm, tpsr[s], u0); \
			u1 = fma(gp, tpni[s], u1); u1 = fma(-gm, tpsi[s], u1); \
			u2 = fma(gm, tmnr[s], u2); u2 = fma(-gp, tmsr[s], u2); \
			u3 = fma(gm, tmni[s], u3); u3 = fma(-gp, tmsi[s], u3); \
			S.gp2[s] = fma(fma(ca, S.x[s], v1), gp, -S.gp2[s]); S.gm2[s] = fma(fma(ca, S.x[s], v2), gm, -S.gm2[s]); \
		} \
	} \
	/* steps come in aligned pairs: kk is even here */ \
	LEG_RED_PUT(kk, t0, t1, t2, t3) \
	LEG_RED_PUT(kk+1, u0, u1, u2, u3) \
	kk += 2; \
	if (kk == LEG_FSTEPS) { leg_flush(red, pout + 4*jbase, lane, LEG_FSTEPS, a.atomic); kk = 0; jbase = j+2; } }

template<int K> __global__ __launch_bounds__(64) void leg_ana_spin(const LegK a)
{
	PXS_SHARED(double, red);
	const int lane = threadIdx.x; int wv, m, bb;
	if (!leg_block(a, wv, m, bb)) return;
	const int l0 = max(m, a.spin);
	const int nl = a.lmax - l0 + 1;
	if (nl <= 0) return;
	const long row0 = PXS_UNIFORM_LONG(a.row[m]);
	const double4_t* __restrict__ coef = a.coef + row0;
	double* __restrict__ pout = a.part + (long)bb*a.mom_bs + ((long)wv*a.rows_chunk + (row0 - a.rowbase))*4;
	const double2* __restrict__ inq = a.leg + (long)bb*a.leg_bs + (long)m*a.ld;
	const double2* __restrict__ inu = a.leg + (long)bb*a.leg_bs + ((long)a.nm + m)*a.ld;
	SpinState<K> S; int rn[K], rs[K];
	const bool polar = leg_wave_polar(a, wv, K);
	const bool alive_any = spin_init<K>(a, wv, lane, m, S, rn, rs, polar);
	if (!__any(alive_any)) return;
	// T+ = Q + iU, T- = Q - iU for north and south rings; the south values carry (-1)^(l+m) of the first accumulated
	// step (phase A advances in multiples of 4, so that is the sign at l0).  A lane whose chains are still below scale 0
	// keeps zero data until both get there (phase B), so that it can run the ungated steps.
	const double sgn0 = ((l0 + m) & 1) ? -1.0 : 1.0;
	double tpnr[K], tpni[K], tmnr[K], tmni[K], tpsr[K], tpsi[K], tmsr[K], tmsi[K];
	auto load_data = [&](int s) {
		// ring indices are re-read here rather than kept in registers through the loops (the kernel sits at the 256-VGPR line)
		const int p = (wv*K + s)*64 + lane;
		const int rn_ = p < a.npairs ? a.ring_n[p] : -1, rs_ = p < a.npairs ? a.ring_s[p] : -1;
		double2 q = rn_ >= 0 ? inq[rn_] : make_double2(0, 0), u = rn_ >= 0 ? inu[rn_] : make_double2(0, 0);
		tpnr[s] = q.x - u.y; tpni[s] = q.y + u.x; tmnr[s] = q.x + u.y; tmni[s] = q.y - u.x;
		q = rs_ >= 0 ? inq[rs_] : make_double2(0, 0); u = rs_ >= 0 ? inu[rs_] : make_double2(0, 0);
		tpsr[s] = sgn0*(q.x - u.y); tpsi[s] = sgn0*(q.y + u.x); tmsr[s] = sgn0*(q.x + u.y); tmsi[s] = sgn0*(q.y - u.x);
	};
#pragma unroll
	for (int s = 0; s < K; s++) tpnr[s] = tpni[s] = tmnr[s] = tmni[s] = tpsr[s] = tpsi[s] = tmsr[s] = tmsi[s] = 0;
	int j = 0;
	SPIN_SEEDED_PHASE_A
	j = PXS_UNIFORM_INT(j); coef = (const double4_t*)PXS_UNIFORM_LONG((long)coef);
	PXS_COUNT(1, (long)(nl - j)*K*12 + (a.seed_mode != 2 ? (long)j*K*4 : 0L));
	if (lane == 0 && !a.atomic) a.first[wv*a.nmc + (m - a.m0)] = j + 1;      // rows before j are not written (reduce_partials skips them)
	// ring data of the lanes whose chains start at scale 0 or both reached it during phase A
#pragma unroll
	for (int s = 0; s < K; s++) if (S.scp[s] == 0 && S.scm[s] == 0) load_data(s);
	int kk = 0, jbase = j;
	// phase B: plain fast steps; every 4 steps the chains below scale 0 are rescaled, and a lane whose two chains
	// have both reached scale 0 fetches its ring data (true terms before that: < 2^-340 of the result)
	while (j + 1 < nl) {
		bool pend = false;
#pragma unroll
		for (int s = 0; s < K; s++) pend |= (S.scp[s] < 0) || (S.scm[s] < 0);
		if (!__any(pend)) break;
		for (int it = 0; it < 2 && j + 1 < nl; it++, j += 2) {
			const double4_t f0 = LDC(coef, j), f1 = LDC(coef, j+1);
			SPIN_ANA_PAIR(f0, f1)
		}
#pragma unroll
		for (int s = 0; s < K; s++) {
			const bool was = (S.scp[s] < 0) || (S.scm[s] < 0);
			if (S.scp[s] < 0 && fabs(S.gp2[s]) > SC_BIG) { S.gp1[s] *= SC_SMALL; S.gp2[s] *= SC_SMALL; S.scp[s]++; }
			if (S.scm[s] < 0 && fabs(S.gm2[s]) > SC_BIG) { S.gm1[s] *= SC_SMALL; S.gm2[s] *= SC_SMALL; S.scm[s]++; }
			if (was && S.scp[s] == 0 && S.scm[s] == 0) load_data(s);
		}
	}
	// phase C: next coefficients prefetched
	double4_t f0 = LDC(coef, j), f1 = LDC(coef, j+1);
#ifndef PXS_NO_PHASEC_UNROLL
	while (j + 3 < nl) {      // (two pairs per iteration on alternating row sets, see leg_syn_s0)
		double4_t n0 = LDC(coef, j+2), n1 = LDC(coef, j+3);
		SPIN_ANA_PAIR(f0, f1)
		j += 2;
		f0 = LDC(coef, j+2); f1 = LDC(coef, j+3);
		SPIN_ANA_PAIR(n0, n1)
		j += 2;
	}
#endif
	for (; j + 1 < nl; j += 2) {
		const double4_t n0 = LDC(coef, j+2), n1 = LDC(coef, j+3);
		SPIN_ANA_PAIR(f0, f1)
		f0 = n0; f1 = n1;
	}
	if (j < nl) {
		double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
		for (int s = 0; s < K; s++) {
			const double gp = S.gp2[s], gm = S.gm2[s];
			t0 = fma(gp, tpnr[s], t0); t0 = fma(gm, tpsr[s], t0);
			t1 = fma(gp, tpni[s], t1); t1 = fma(gm, tpsi[s], t1);
			t2 = fma(gm, tmnr[s], t2); t2 = fma(gp, tmsr[s], t2);
			t3 = fma(gm, tmni[s], t3); t3 = fma(gp, tmsi[s], t3);
		}
		LEG_RED_PUT(kk, t0, t1, t2, t3)
		kk++;
	}
	if (kk > 0) leg_flush(red, pout + 4*jbase, lane, kk, a.atomic);
}


// ---- batched spin-s analysis as an FP64-MFMA GEMM (round 5) -----------------------------------------------------------------------
// Stacks of T/Q/U maps (Monte-Carlo polarisation sims): the Q/U pairs of 4 or more maps in one call.  Per m
//   mu+[l][map] = sum_ring G+_l T+_N + sgn_l G-_l T+_S,   mu-[l][map] = sum_ring G-_l T-_N + sgn_l G+_l T-_S,   sgn_l = (-1)^(l + m)
// (leg_ana_spin) with the SAME G+ / G- for every map.  One chain per HALF-WAVE: lanes 0-31 of a wave run G+ of 32 ring pairs, lanes 32-63 G- of the
// same pairs, and the G- lanes park sgn_l G-.  With B = (T+_N re, im, T-_S re, im) on the G+ slots and (T+_S re, im, T-_N re, im) on the G- slots ONE
// GEMM over the 64 slots gives (mu+, sgn_l mu-): the sign of the last two columns is a function of the row and is applied at the flush.  That makes the
// kernel the shape of leg_ana_s0_mm -- one P tile per wave, 8 waves over 256 ring pairs, 8 maps per workgroup, four waves per SIMD.  (First form: both
// chains in every lane, two P tiles per wave, two accumulators: the LDS held two waves per SIMD and the f64 MFMA, which needs several issuing waves for
// its rate, ran the Q/U analysis of 16 maps in 111 ms against the VALU kernel's 123.)
// one chain of leg_ana_spin's pair of recurrences: G_{l+1} = (a x + c) G_l - G_{l-1}, c = +-b (polar waves: a +- b with x = -2 sin^2(theta / 2))
struct SpinChain { double x, g1, g2, sgl, pa; int sc; };
__device__ __forceinline__ bool spin_chain_init(const LegK& a, int p, int m, int half, bool polar, SpinChain& C) {
	const int s_ = a.spin;
	const bool valid = p < a.npairs;
	const double cth = valid ? a.cth[p] : 0.0, sth = valid ? a.sth[p] : 0.0, shh = valid ? a.sh2[p] : 0.0;
	C.x = polar ? -2.0*shh*shh : cth; C.sgl = half ? -1.0 : 1.0; C.pa = polar ? 1.0 : 0.0;
	const double t1 = a.lmax*sth + a.ofs, b = -2.0*s_*fabs(cth), c = (double)s_*s_ - t1*t1, discr = b*b - 4*c;      // (libsharp's m-limit generalised to spin, as spin_init)
	const double mlim = discr <= 0 ? a.lmax : fmin((double)a.lmax, 0.5*(-b + sqrt(discr)));
	const bool alive = valid && ((double)m <= mlim + 0.5);
	C.g1 = 0; C.g2 = 0; C.sc = 0;
	if (alive) {
		const double sh = shh, ch = a.ch2[p];
		double m1, m2; int e1, e2;
		// exponents of sin(theta/2), cos(theta/2) of the start value: G+ (m + s, m - s) / G- (m - s, m + s) for m >= s, (s + m, s - m) / (s - m, s + m) below
		const int es = m >= s_ ? (half ? m - s_ : m + s_) : (half ? s_ - m : s_ + m), ec = m >= s_ ? (half ? m + s_ : m - s_) : (half ? s_ + m : s_ - m);
		pow_scaled(sh, es, m1, e1); pow_scaled(ch, ec, m2, e2);
		double mt = m1*m2; int e = e1 + e2 + (m >= s_ ? m : 0); frexp_norm(mt, e); to_scaled(mt, e, C.g2, C.sc);
		if (m < s_ && half && ((s_ - m) & 1)) C.g2 = -C.g2;
	}
	return alive;
}
// coefficient of a step from the row (a, b): a x + (polar ? a : 0) +- b
__device__ __forceinline__ double spin_chain_coef(const SpinChain& C, double ca, double cb) { return fma(ca, C.x, fma(C.sgl, cb, C.pa*ca)); }

template<int NG, int W> __global__ __launch_bounds__(64*W, 4) void leg_ana_spin_mm(const LegK a)
{
	PXS_SHARED(double, sh);
	double* __restrict__ ptile = sh;                              // [W][16][MM_PSTRIDE]
	double* __restrict__ red = sh + W*16*MM_PSTRIDE;              // [2][4 NG][64]
	int* __restrict__ s_kmin = reinterpret_cast<int*>(sh + mm_lds_doubles(NG, W));
	const int tid = threadIdx.x, lane = tid & 63, w = PXS_UNIFORM_INT(tid >> 6), half = lane >> 5;
	int wv, m, bb;
	if (!leg_block(a, wv, m, bb)) return;
	const int l0 = max(m, a.spin);
	const int nl = a.lmax - l0 + 1;
	if (nl <= 0) return;
	const long row0 = PXS_UNIFORM_LONG(a.row[m]);
	const int pbase = wv*32*W;
	const bool polar = [&] { const double c = a.cth[min(pbase + 32*(w + 1), a.npairs) - 1]; return c*c > PXS_POLAR_COS2; }();      // (per wave)
	const int pmine_ = pbase + 32*w + (lane & 31);       // ring pair of this lane's chain
	SpinChain C;
	const bool alive = spin_chain_init(a, pmine_, m, half, polar, C);
	const double* __restrict__ tab = reinterpret_cast<const double*>(a.coef2) + 2*row0;      // (a, b) of step k at tab[2 k]
	if (tid == 0) *s_kmin = nl;
	__syncthreads();
	// phase A, per wave: recurrence only until the first lane of the wave is at scale 0
	int k = 0;
	const bool wave_alive = __any(alive);
	if (wave_alive) {
		while (k + 4 <= nl) {
			if (__any(C.sc == 0 && C.g2 != 0.0)) break;
			double cq[8];
#pragma unroll
			for (int i = 0; i < 8; i++) cq[i] = LDCD(tab, 2L*k + i);
			C.g1 = fma(spin_chain_coef(C, cq[0], cq[1]), C.g2, -C.g1);
			C.g2 = fma(spin_chain_coef(C, cq[2], cq[3]), C.g1, -C.g2);
			C.g1 = fma(spin_chain_coef(C, cq[4], cq[5]), C.g2, -C.g1);
			C.g2 = fma(spin_chain_coef(C, cq[6], cq[7]), C.g1, -C.g2);
			if (C.sc < 0 && fabs(C.g2) > SC_BIG) { C.g1 *= SC_SMALL; C.g2 *= SC_SMALL; C.sc++; }
			k += 4;
		}
	}
	const int kw = PXS_UNIFORM_INT(wave_alive ? k : nl + 16);
	if (lane == 0) atomicMin(s_kmin, kw);
	__syncthreads();
	const int kmin = PXS_UNIFORM_INT(*s_kmin);
	if (kmin >= nl) return;      // (workgroup-uniform) no ring of this chunk carries signal at this m
	// B operands through the LDS: thread = slot (chain of a ring pair), 4 maps per round: G+ slots (T+_N re, im, T-_S re, im), G- slots (T+_S re, im, T-_N re, im)
	double breg[NG][16];
	{
		const bool ok = pmine_ < a.npairs;
		const int rn = ok ? a.ring_n[pmine_] : -1, rs = ok ? a.ring_s[pmine_] : -1;
		double* __restrict__ ent = sh + tid*MM_ESTRIDE;
		const double* __restrict__ rd = sh + (64*w + 16*(lane >> 4))*MM_ESTRIDE + (lane & 15);
#pragma unroll
		for (int g = 0; g < NG; g++) {
			double2 qn[4], un[4], qs[4], us[4];
#pragma unroll
			for (int mm = 0; mm < 4; mm++) {
				const int map = (bb*NG + g)*4 + mm;
				const double2* __restrict__ inq = a.leg + (long)map*a.leg_bs + (long)m*a.ld;
				const double2* __restrict__ inu = a.leg + (long)map*a.leg_bs + ((long)a.nm + m)*a.ld;
				const bool okm = map < a.nmaps;
				qn[mm] = (okm && rn >= 0) ? inq[rn] : make_double2(0, 0); un[mm] = (okm && rn >= 0) ? inu[rn] : make_double2(0, 0);
				qs[mm] = (okm && rs >= 0) ? inq[rs] : make_double2(0, 0); us[mm] = (okm && rs >= 0) ? inu[rs] : make_double2(0, 0);
			}
			if (g > 0) __syncthreads();      // the reads of the previous round
#pragma unroll
			for (int mm = 0; mm < 4; mm++) {
				// T+ = Q + iU, T- = Q - iU
				const double tpn_r = qn[mm].x - un[mm].y, tpn_i = qn[mm].y + un[mm].x, tmn_r = qn[mm].x + un[mm].y, tmn_i = qn[mm].y - un[mm].x;
				const double tps_r = qs[mm].x - us[mm].y, tps_i = qs[mm].y + us[mm].x, tms_r = qs[mm].x + us[mm].y, tms_i = qs[mm].y - us[mm].x;
				ent[4*mm + 0] = half ? tps_r : tpn_r; ent[4*mm + 1] = half ? tps_i : tpn_i;
				ent[4*mm + 2] = half ? tmn_r : tms_r; ent[4*mm + 3] = half ? tmn_i : tms_i;
			}
			__syncthreads();
#pragma unroll
			for (int q = 0; q < 16; q++) breg[g][q] = rd[q*MM_ESTRIDE];
		}
		__syncthreads();
		for (int i = tid; i < 2*NG*4*64; i += 64*W) red[i] = 0.0;
		__syncthreads();
	}
	bool pend = __any(C.sc < 0);
	double* __restrict__ pmine = ptile + w*16*MM_PSTRIDE;
	const double* __restrict__ pread = pmine + (lane & 15)*MM_PSTRIDE + 16*(lane >> 4);
	double cf[32];      // (a, b) of the 16 steps of a tile, requested a tile ahead
	int cf_tile = -1;
	long ntile = 0;
	auto mm_flush = [&](int tf) {      // rows 4 r + lane / 16 of tile tf, column lane % 16 = 4 (map in the group) + c; c >= 2 (mu-): x sgn of the row
		double* __restrict__ redf = red + (tf & 1)*NG*4*64;
		for (int cidx = w; cidx < 4*NG; cidx += W) {
			const int g = cidx >> 2, r = cidx & 3;
			double* rp = redf + cidx*64 + lane;
			double v = *rp; *rp = 0.0;
			const int krow = 16*tf + 4*r + (lane >> 4), map = (bb*NG + g)*4 + ((lane & 15) >> 2), c = lane & 3;
			if (krow < nl && map < a.nmaps) {
				if (c >= 2 && ((l0 + krow + m) & 1)) v = -v;
				double* dst = a.mom + (long)map*a.mom_bs + 4*(row0 + krow) + c;
#ifdef PXS_HOST_SIM
				atomicAdd(dst, v);
#else
				unsafeAtomicAdd(dst, v);
#endif
			}
		}
	};
	int tlast = -1;
	for (int t = kmin >> 4; 16*t < nl; t++) {
		const int k0 = 16*t;
		double* __restrict__ redt = red + (t & 1)*NG*4*64;
		if (k0 + 16 > kw) {      // (wave-uniform) this wave has steps in the tile
			ntile++;
			if (cf_tile != t) {
#pragma unroll
				for (int i = 0; i < 32; i++) cf[i] = LDCD(tab, 2L*k0 + i);
			}
#pragma unroll
			for (int q4 = 0; q4 < 4; q4++) {
				const int kq = k0 + 4*q4;
				double p0 = 0, p1 = 0, p2 = 0, p3 = 0;
				if (kq >= kw && kq < nl) {
					p0 = C.g2; C.g1 = fma(spin_chain_coef(C, cf[8*q4 + 0], cf[8*q4 + 1]), C.g2, -C.g1);
					p1 = C.g1; C.g2 = fma(spin_chain_coef(C, cf[8*q4 + 2], cf[8*q4 + 3]), C.g1, -C.g2);
					p2 = C.g2; C.g1 = fma(spin_chain_coef(C, cf[8*q4 + 4], cf[8*q4 + 5]), C.g2, -C.g1);
					p3 = C.g1; C.g2 = fma(spin_chain_coef(C, cf[8*q4 + 6], cf[8*q4 + 7]), C.g1, -C.g2);
					if (pend) {      // phase B: a chain below scale 0 contributes nothing yet; rescale it every 4 steps
						if (C.sc < 0) { p0 = p1 = p2 = p3 = 0.0; if (fabs(C.g2) > SC_BIG) { C.g1 *= SC_SMALL; C.g2 *= SC_SMALL; C.sc++; } }
						pend = __any(C.sc < 0);
					}
					// the G- lanes park sgn_l G-: the sign of the first row of the group, alternating
					const double se = (half && ((l0 + kq + m) & 1)) ? -1.0 : 1.0, so = half ? -se : 1.0;
					p0 *= se; p1 *= so; p2 *= se; p3 *= so;
					if (kq + 1 >= nl) p1 = 0.0;
					if (kq + 2 >= nl) p2 = 0.0;
					if (kq + 3 >= nl) p3 = 0.0;
				}
				pmine[(4*q4 + 0)*MM_PSTRIDE + lane] = p0; pmine[(4*q4 + 1)*MM_PSTRIDE + lane] = p1;
				pmine[(4*q4 + 2)*MM_PSTRIDE + lane] = p2; pmine[(4*q4 + 3)*MM_PSTRIDE + lane] = p3;
			}
			MM_WAVE_SYNC();
			double av[4];
#pragma unroll
			for (int q = 0; q < 4; q++) av[q] = pread[q];
			MM_WAVE_SYNC();
			if (k0 + 16 < nl) {
#pragma unroll
				for (int i = 0; i < 32; i++) cf[i] = LDCD(tab, 2L*(k0 + 16) + i);
				cf_tile = t + 1;
			}
			mm_acc acc[NG];
#pragma unroll
			for (int g = 0; g < NG; g++) { acc[g][0] = 0; acc[g][1] = 0; acc[g][2] = 0; acc[g][3] = 0; }
#pragma unroll
			for (int q = 0; q < 16; q++) {
				const double aq = q < 4 ? av[q] : pread[q];
#pragma unroll
				for (int g = 0; g < NG; g++) acc[g] = mm_mfma(aq, breg[g][q], acc[g]);
			}
			if (tlast >= 0) { mm_flush(tlast); tlast = -1; }
#pragma unroll
			for (int g = 0; g < NG; g++)
#pragma unroll
				for (int r = 0; r < 4; r++) mm_lds_add(redt + (g*4 + r)*64 + lane, acc[g][r]);
		}
		if (tlast >= 0) mm_flush(tlast);
		tlast = t;
		__syncthreads();
	}
	if (tlast >= 0) mm_flush(tlast);
	PXS_COUNT(1, ntile*(NG*256L + 32L) + (wave_alive ? (long)kw*2 : 0L));
}

// ---- batched spin-s synthesis as an FP64-MFMA GEMM (round 5) ----------------------------------------------------------------------
// The transpose of leg_ana_spin_mm (cf. leg_syn_spin): north  P = sum_l G+ a+, M = sum_l G- a-;  south  P' = sum_l sgn_l G- a+, M' = sum_l sgn_l G+ a-.
// One chain per half-wave as in the analysis: the 64 rows of a wave's accumulators are the G+ slots of 32 ring pairs (row blocks 0, 1) and their
// G- slots (row blocks 2, 3); the G- lanes park sgn_l G-, and with B = (a+, sgn_l a-) -- ONE B for all rows -- the G+ rows come out as (P, M') and the G-
// rows as (P', M).  The shape of leg_syn_s0_mm: one P tile, 4 x 8 accumulator VGPRs per group of 4 maps, one wave per workgroup, no cross-wave step.
template<int NG> __global__ __launch_bounds__(64, 4) void leg_syn_spin_mm(const LegK a)
{
	PXS_SHARED(double, pmine);      // [16][MMS_PSTRIDE]
	const int lane = threadIdx.x, half = lane >> 5;
	int wv, m, bb;
	if (!leg_block(a, wv, m, bb)) return;
	const int l0 = max(m, a.spin);
	const int nl = a.lmax - l0 + 1;
	const long row0 = PXS_UNIFORM_LONG(a.row[m]);
	const int pbase = wv*32;
	const bool polar = [&] { const double c = a.cth[min(pbase + 32, a.npairs) - 1]; return c*c > PXS_POLAR_COS2; }();
	SpinChain C;
	const bool alive = spin_chain_init(a, pbase + (lane & 31), m, half, polar, C);
	mm_acc acc[NG][4];
#pragma unroll
	for (int g = 0; g < NG; g++)
#pragma unroll
		for (int rb = 0; rb < 4; rb++) { acc[g][rb][0] = 0; acc[g][rb][1] = 0; acc[g][rb][2] = 0; acc[g][rb][3] = 0; }
	long ntile = 0;
	const double* __restrict__ tab = reinterpret_cast<const double*>(a.coef2) + 2*row0;      // (a, b) of step k at tab[2 k]
	// phase A: recurrence only until the first lane of the wave is at scale 0 (a wave without a live ring skips the loop below)
	int k = 0;
	const bool wave_alive = nl > 0 && __any(alive);
	if (wave_alive) {
		while (k + 4 <= nl) {
			if (__any(C.sc == 0 && C.g2 != 0.0)) break;
			double cq[8];
#pragma unroll
			for (int i = 0; i < 8; i++) cq[i] = LDCD(tab, 2L*k + i);
			C.g1 = fma(spin_chain_coef(C, cq[0], cq[1]), C.g2, -C.g1);
			C.g2 = fma(spin_chain_coef(C, cq[2], cq[3]), C.g1, -C.g2);
			C.g1 = fma(spin_chain_coef(C, cq[4], cq[5]), C.g2, -C.g1);
			C.g2 = fma(spin_chain_coef(C, cq[6], cq[7]), C.g1, -C.g2);
			if (C.sc < 0 && fabs(C.g2) > SC_BIG) { C.g1 *= SC_SMALL; C.g2 *= SC_SMALL; C.sc++; }
			k += 4;
		}
	}
	const int kw = PXS_UNIFORM_INT(wave_alive ? k : max(nl, 0) + 16);
	{
		// B operand of MFMA step-quad q: lane (j, kk) holds column j & 3 of map 4 (bb NG + g) + (j >> 2) at step q + 4 kk of the tile: (a+ re, a+ im, sgn a- re, sgn a- im)
		const int jcol = lane & 15, kk4 = lane >> 4, cc = jcol & 3;
		const double* bsrc[NG]; bool bok[NG];
#pragma unroll
		for (int g = 0; g < NG; g++) {
			const int map = (bb*NG + g)*4 + (jcol >> 2);
			bok[g] = map < a.nmaps;
			bsrc[g] = a.almt + (long)(bok[g] ? map : 0)*a.almt_bs + 4*row0 + cc + 16*kk4;
		}
		auto load_b = [&](int k0, double (*b)[4]) {
#pragma unroll
			for (int g = 0; g < NG; g++)
#pragma unroll
				for (int q = 0; q < 4; q++) {
					const int row = k0 + q + 4*kk4;
					const double v = (bok[g] && row < nl) ? bsrc[g][4L*(k0 + q)] : 0.0;
					b[g][q] = (cc >= 2 && ((l0 + row + m) & 1)) ? -v : v;
				}
		};
		bool pend = __any(C.sc < 0);
		const double* __restrict__ pread = pmine + 4*(lane >> 4)*MMS_PSTRIDE + (lane & 15);
		double cf[32];      // (a, b) of the 16 steps of the tile, requested a tile ahead (every tile from the wave's first one on is run)
		double bcur[NG][4], bnxt[NG][4];
		load_b(16*(kw >> 4), bcur);
		if (kw < nl) {
#pragma unroll
			for (int i = 0; i < 32; i++) cf[i] = LDCD(tab, 32L*(kw >> 4) + i);
		}
		for (int t = kw >> 4; 16*t < nl; t++) {
			const int k0 = 16*t;
			ntile++;
#pragma unroll
			for (int q4 = 0; q4 < 4; q4++) {
				const int kq = k0 + 4*q4;
				double p0 = 0, p1 = 0, p2 = 0, p3 = 0;
				if (kq >= kw && kq < nl) {
					p0 = C.g2; C.g1 = fma(spin_chain_coef(C, cf[8*q4 + 0], cf[8*q4 + 1]), C.g2, -C.g1);
					p1 = C.g1; C.g2 = fma(spin_chain_coef(C, cf[8*q4 + 2], cf[8*q4 + 3]), C.g1, -C.g2);
					p2 = C.g2; C.g1 = fma(spin_chain_coef(C, cf[8*q4 + 4], cf[8*q4 + 5]), C.g2, -C.g1);
					p3 = C.g1; C.g2 = fma(spin_chain_coef(C, cf[8*q4 + 6], cf[8*q4 + 7]), C.g1, -C.g2);
					if (pend) {
						if (C.sc < 0) { p0 = p1 = p2 = p3 = 0.0; if (fabs(C.g2) > SC_BIG) { C.g1 *= SC_SMALL; C.g2 *= SC_SMALL; C.sc++; } }
						pend = __any(C.sc < 0);
					}
					const double se = (half && ((l0 + kq + m) & 1)) ? -1.0 : 1.0, so = half ? -se : 1.0;      // the G- lanes park sgn_l G-
					p0 *= se; p1 *= so; p2 *= se; p3 *= so;
					if (kq + 1 >= nl) p1 = 0.0;
					if (kq + 2 >= nl) p2 = 0.0;
					if (kq + 3 >= nl) p3 = 0.0;
				}
				pmine[(4*q4 + 0)*MMS_PSTRIDE + lane] = p0; pmine[(4*q4 + 1)*MMS_PSTRIDE + lane] = p1;
				pmine[(4*q4 + 2)*MMS_PSTRIDE + lane] = p2; pmine[(4*q4 + 3)*MMS_PSTRIDE + lane] = p3;
			}
			MM_WAVE_SYNC();
			double av[4];
#pragma unroll
			for (int rb = 0; rb < 4; rb++) av[rb] = pread[16*rb];
			MM_WAVE_SYNC();
			if (k0 + 16 < nl) {      // the rows of the next tile (coefficients and pre-scaled alm), on their way during the MFMAs
#pragma unroll
				for (int i = 0; i < 32; i++) cf[i] = LDCD(tab, 2L*(k0 + 16) + i);
				load_b(k0 + 16, bnxt);
			}
#pragma unroll
			for (int q = 0; q < 4; q++)
#pragma unroll
				for (int rb = 0; rb < 4; rb++) {
					const double aq = q == 0 ? av[rb] : pread[q*MMS_PSTRIDE + 16*rb];
#pragma unroll
					for (int g = 0; g < NG; g++) acc[g][rb] = mm_mfma(aq, bcur[g][q], acc[g][rb]);
				}
			MM_WAVE_SYNC();      // the A operands are out of the tile before the next one is written
#pragma unroll
			for (int g = 0; g < NG; g++)
#pragma unroll
				for (int q = 0; q < 4; q++) bcur[g][q] = bnxt[g][q];
		}
	}
	// register r of acc[g][rb] at lane (i4 = lane / 16, jc = lane % 16): slot 16 rb + 4 r + i4 (rb < 2: G+ of ring pair 16 rb + 4 r + i4, rb >= 2: G- of pair
	// 16 (rb - 2) + 4 r + i4), column 4 (map in the group) + c.  G+ rows: c = 0, 1: P re / im (north); 2, 3: M' re / im (south).  G- rows: c = 0, 1: P' re / im
	// (south); 2, 3: M re / im (north).  Q = (P + M) / 2, U = -i (P - M) / 2: lanes c < 2 write the north ring, lanes c >= 2 the south ring; even c the
	// real part of Q and the imaginary part of U, odd c the other two.
	const int c = lane & 3;
#pragma unroll
	for (int g = 0; g < NG; g++) {
		const int map = (bb*NG + g)*4 + ((lane & 15) >> 2);
		double* __restrict__ outq = reinterpret_cast<double*>(a.leg + (long)(map < a.nmaps ? map : 0)*a.leg_bs + (long)m*a.ld);
		double* __restrict__ outu = reinterpret_cast<double*>(a.leg + (long)(map < a.nmaps ? map : 0)*a.leg_bs + ((long)a.nm + m)*a.ld);
#pragma unroll
		for (int rb = 0; rb < 2; rb++)
#pragma unroll
			for (int r = 0; r < 4; r++) {
				const int p = pbase + 16*rb + 4*r + (lane >> 4);
				const bool valid = p < a.npairs && map < a.nmaps;
				const double own = acc[g][rb][r], oth = MMS_XOR2(acc[g][rb + 2][r]);
				const double P = c < 2 ? own : oth, M = c < 2 ? oth : own;
				const double sum = 0.5*(P + M), dif = 0.5*(P - M);
				const int ring = valid ? (c < 2 ? a.ring_n[p] : a.ring_s[p]) : -1;
				if (ring >= 0) {
					if (c & 1) { outq[2*ring + 1] = sum; outu[2*ring] = dif; }
					else       { outq[2*ring] = sum; outu[2*ring + 1] = -dif; }
				}
			}
	}
	PXS_COUNT(0, ntile*(NG*256L + 32L) + (wave_alive ? (long)kw*2 : 0L));
}

// ---- launchers ----
void launch_leg_syn_spin(int K, dim3 grid, hipStream_t st, const LegK& a) {
	// (ring pairs per lane the product's rules select: 3 and 2 for the synthesis, 4, 3 and 2 for the analysis; lab builds compile the others too)
#ifdef PXS_LAB
	if (K == 4) { hipLaunchKernelGGL(leg_syn_spin<4>, grid, dim3(64), 0, st, a); return; }
#endif
	if (K == 3) hipLaunchKernelGGL(leg_syn_spin<3>, grid, dim3(64), 0, st, a);
	else        hipLaunchKernelGGL(leg_syn_spin<2>, grid, dim3(64), 0, st, a);
}
void launch_leg_ana_spin(int K, dim3 grid, size_t lds, hipStream_t st, const LegK& a) {
#ifdef PXS_LAB
	if (K >= 6) { hipLaunchKernelGGL(leg_ana_spin<6>, grid, dim3(64), lds, st, a); return; }
	if (K == 5) { hipLaunchKernelGGL(leg_ana_spin<5>, grid, dim3(64), lds, st, a); return; }
#endif
	if (K >= 4)      hipLaunchKernelGGL(leg_ana_spin<4>, grid, dim3(64), lds, st, a);
	else if (K == 3) hipLaunchKernelGGL(leg_ana_spin<3>, grid, dim3(64), lds, st, a);
	else             hipLaunchKernelGGL(leg_ana_spin<2>, grid, dim3(64), lds, st, a);
}
void launch_leg_syn_spin_mm(int ng, dim3 grid, hipStream_t st, const LegK& a) {
	if (ng == 2) hipLaunchKernelGGL(leg_syn_spin_mm<2>, grid, dim3(64), mm_syn_lds(), st, a);
	else         hipLaunchKernelGGL(leg_syn_spin_mm<1>, grid, dim3(64), mm_syn_lds(), st, a);
}
void launch_leg_ana_spin_mm(int ng, dim3 grid, hipStream_t st, const LegK& a) {
	constexpr int W = MM_WAVES;
	static const bool once = [] { (void)hipFuncSetAttribute((const void*)leg_ana_spin_mm<2, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 256);
		(void)hipFuncSetAttribute((const void*)leg_ana_spin_mm<1, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 256); return true; }(); (void)once;
	if (ng == 2) hipLaunchKernelGGL((leg_ana_spin_mm<2, W>), grid, dim3(64*W), mm_ana_lds(2, W), st, a);
	else         hipLaunchKernelGGL((leg_ana_spin_mm<1, W>), grid, dim3(64*W), mm_ana_lds(1, W), st, a);
}

} // namespace pxs
