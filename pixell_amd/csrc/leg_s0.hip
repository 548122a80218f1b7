// Spin-0 Legendre kernels for gfx950 (design notes: head of legendre.hip): leg_syn_s0 / leg_ana_s0 (one wave per workgroup, K ring pairs per lane, plain
// v_fma_f64) and the FP64-MFMA forms of batched calls, leg_syn_s0_mm / leg_ana_s0_mm.  Replaces ducc0's alm2leg / leg2alm for spin 0 as reached from
// pixell/curvedsky.py:907-960, 1032-1084.
#include "legendre_dev.hpp"

namespace pxs {

// two fast steps of the spin-0 synthesis (lam1/lam2 swap roles)
#define S0_SYN_PAIR(c0, c1, a0, a1) { \
	PXS_VCOPY(vb0, polar ? c0.c : c0.b); \
	PXS_VCOPY(vb1, polar ? c1.c : c1.b); \
	_Pragma("unroll") for (int s = 0; s < K; s++) { \
		p1r[s] = fma(lam2[s], a0.a, p1r[s]); p1i[s] = fma(lam2[s], a0.b, p1i[s]); \
		p2r[s] = fma(lam2[s], a0.c, p2r[s]); p2i[s] = fma(lam2[s], a0.d, p2i[s]); \
		lam1[s] = fma(fma(c0.a, csq[s], vb0), lam2[s], lam1[s]); \
	} \
	_Pragma("unroll") for (int s = 0; s < K; s++) { \
		p1r[s] = fma(lam1[s], a1.a, p1r[s]); p1i[s] = fma(lam1[s], a1.b, p1i[s]); \
		p2r[s] = fma(lam1[s], a1.c, p2r[s]); p2i[s] = fma(lam1[s], a1.d, p2i[s]); \
		lam2[s] = fma(fma(c1.a, csq[s], vb1), lam1[s], lam2[s]); \
	} }

// K = 4 is asked to fit 7 waves per SIMD (72 instead of 74 VGPRs, no spills): the synthesis kernels are short of waves, not of
// registers per wave (pinned to 2 / 3 / 4 waves per SIMD leg_syn_spin<3> takes 1.51 / 1.29 / 1.0 of its time): C4 leg_syn 126.3 -> 121.9 ms
template<int K> __global__ __launch_bounds__(64, (K == 4 ? 7 : 1)) void leg_syn_s0(const LegK a)
{
	const int lane = threadIdx.x; int wv, m, bb;
	if (!leg_block(a, wv, m, bb)) return;
	const long row0 = PXS_UNIFORM_LONG(a.row[m]);
	const int nk = (a.lmax - m)/2 + 1;
	const double4_t* __restrict__ coef = a.coef + row0;
	const double4_t* __restrict__ at = reinterpret_cast<const double4_t*>(a.almt + (long)bb*a.almt_bs) + row0;
	double x[K], csq[K], lam1[K], lam2[K], p1r[K], p1i[K], p2r[K], p2i[K];
	int sc[K], rn[K], rs[K];
	bool alive_any = false;
	const bool polar = leg_wave_polar(a, wv, K);
#pragma unroll
	for (int s = 0; s < K; s++) {
		const int p = (wv*K + s)*64 + lane;
		const bool valid = p < a.npairs;
		rn[s] = valid ? a.ring_n[p] : -1; rs[s] = valid ? a.ring_s[p] : -1;
		x[s] = valid ? a.cth[p] : 0.0;
		const double sth = valid ? a.sth[p] : 0.0;
		csq[s] = polar ? -sth*sth : x[s]*x[s];
		const bool alive = valid && ((double)m <= a.lmax*sth + a.ofs);
		lam1[s] = 0; lam2[s] = 0; sc[s] = 0;
		if (alive && a.seed_mode != 2) { double mt; int e; pow_scaled(sth, m, mt, e); to_scaled(mt, e, lam2[s], sc[s]); }
		p1r[s] = p1i[s] = p2r[s] = p2i[s] = 0;
		alive_any |= alive;
	}
	int k = 0;
	if (__any(alive_any)) {
		// phase A: nobody at scale 0 yet -> recurrence only, 4 steps per check (S0_PHASE_A)
		S0_SEEDED_PHASE_A
		k = PXS_UNIFORM_INT(k); coef = (const double4_t*)PXS_UNIFORM_LONG((long)coef);
		PXS_COUNT(0, (long)(nk - k)*K*6 + (a.seed_mode != 2 ? (long)k*K*2 : 0L));
		// phase B: some lanes are still below scale 0.  The steps are the plain fast steps (no per-lane gating); every
		// 4 steps the lanes below scale 0 are rescaled.  Such a lane accumulates scaled-up garbage meanwhile; its sums are
		// reset when it reaches scale 0 (its true terms before that are < 2^-340 of the final value).
		while (k + 1 < nk) {
			bool pend = false;
#pragma unroll
			for (int s = 0; s < K; s++) pend |= (sc[s] < 0);
			if (!__any(pend)) break;
			for (int it = 0; it < 2 && k + 1 < nk; it++, k += 2) {
				const double4_t c0 = LDC(coef, k), c1 = LDC(coef, k+1), a0 = LDC(at, k), a1 = LDC(at, k+1);
				S0_SYN_PAIR(c0, c1, a0, a1)
			}
#pragma unroll
			for (int s = 0; s < K; s++)
				if (sc[s] < 0 && fabs(lam2[s]) > SC_BIG) {
					lam1[s] *= SC_SMALL; lam2[s] *= SC_SMALL;
					if (++sc[s] == 0) p1r[s] = p1i[s] = p2r[s] = p2i[s] = 0;
				}
		}
#pragma unroll
		for (int s = 0; s < K; s++) if (sc[s] < 0) { p1r[s] = p1i[s] = p2r[s] = p2i[s] = 0; lam1[s] = lam2[s] = 0; }   // never reached scale 0
		// phase C: fast loop, two steps per iteration (lam1/lam2 swap roles, no register moves),
		// coefficients of the next iteration prefetched with scalar loads (tables are padded by 2 rows)
		double4_t c0 = LDC(coef, k), c1 = LDC(coef, k+1), a0 = LDC(at, k), a1 = LDC(at, k+1);
#ifndef PXS_NO_PHASEC_UNROLL
		// two pairs per iteration on alternating row sets: the prefetched rows are consumed where they landed (the single-pair loop
		// rotated them with 8-16 s_mov_b64 per pair; SALU was 27-45 % of the VALU count, profiles/r04_leg_sq_counters_c3.txt)
		while (k + 3 < nk) {
			double4_t n0 = LDC(coef, k+2), n1 = LDC(coef, k+3), m0 = LDC(at, k+2), m1 = LDC(at, k+3);
			S0_SYN_PAIR(c0, c1, a0, a1)
			k += 2;
			c0 = LDC(coef, k+2); c1 = LDC(coef, k+3); a0 = LDC(at, k+2); a1 = LDC(at, k+3);
			S0_SYN_PAIR(n0, n1, m0, m1)
			k += 2;
		}
#endif
		for (; k + 1 < nk; k += 2) {
			const double4_t n0 = LDC(coef, k+2), n1 = LDC(coef, k+3), m0 = LDC(at, k+2), m1 = LDC(at, k+3);
			S0_SYN_PAIR(c0, c1, a0, a1)
			c0 = n0; c1 = n1; a0 = m0; a1 = m1;
		}
		if (k < nk) {
#pragma unroll
			for (int s = 0; s < K; s++) {
				p1r[s] = fma(lam2[s], a0.a, p1r[s]); p1i[s] = fma(lam2[s], a0.b, p1i[s]);
				p2r[s] = fma(lam2[s], a0.c, p2r[s]); p2i[s] = fma(lam2[s], a0.d, p2i[s]);
			}
		}
	}
	double2* __restrict__ out = a.leg + (long)bb*a.leg_bs + (long)m*a.ld;
#pragma unroll
	for (int s = 0; s < K; s++) {      // ring indices and cos(theta) are re-read here rather than kept in registers through the loops
		const int p = (wv*K + s)*64 + lane;
		const bool valid = p < a.npairs;
		const int rn_ = valid ? a.ring_n[p] : -1, rs_ = valid ? a.ring_s[p] : -1;
		const double x_ = valid ? a.cth[p] : 0.0;
		if (rn_ >= 0) out[rn_] = make_double2(p1r[s] + x_*p2r[s], p1i[s] + x_*p2i[s]);
		if (rs_ >= 0) out[rs_] = make_double2(p1r[s] - x_*p2r[s], p1i[s] - x_*p2i[s]);
	}
}

// two fast steps of the spin-0 analysis: 2 x 4 lane sums into the LDS reduction tile, flush every 4 steps
#define S0_ANA_PAIR(c0, c1) { \
	PXS_VCOPY(vb0, polar ? c0.c : c0.b); \
	PXS_VCOPY(vb1, polar ? c1.c : c1.b); \
	double t0 = 0, t1 = 0, t2 = 0, t3 = 0, u0 = 0, u1 = 0, u2 = 0, u3 = 0; \
	_Pragma("unroll") for (int s = 0; s < K; s++) { \
		t0 = fma(lam2[s], d1r[s], t0); t1 = fma(lam2[s], d1i[s], t1); t2 = fma(lam2[s], d2r[s], t2); t3 = fma(lam2[s], d2i[s], t3); \
		lam1[s] = fma(fma(c0.a, csq[s], vb0), lam2[s], lam1[s]); \
	} \
	_Pragma("unroll") for (int s = 0; s < K; s++) { \
		u0 = fma(lam1[s], d1r[s], u0); u1 = fma(lam1[s], d1i[s], u1); u2 = fma(lam1[s], d2r[s], u2); u3 = fma(lam1[s], d2i[s], u3); \
		lam2[s] = fma(fma(c1.a, csq[s], vb1), lam1[s], lam2[s]); \
	} \
	/* steps come in aligned pairs (phase A advances by 4, phases B and C by 2): kk is even here */ \
	LEG_RED_PUT(kk, t0, t1, t2, t3) \
	LEG_RED_PUT(kk+1, u0, u1, u2, u3) \
	kk += 2; \
	if (kk == LEG_FSTEPS) { leg_flush(red, pout + 4*kbase, lane, LEG_FSTEPS, a.atomic); kk = 0; kbase = k+2; } }

template<int K> __global__ __launch_bounds__(64) void leg_ana_s0(const LegK a)
{
	PXS_SHARED(double, red);
	const int lane = threadIdx.x; int wv, m, bb;
	if (!leg_block(a, wv, m, bb)) return;
	const long row0 = PXS_UNIFORM_LONG(a.row[m]);
	const int nk = (a.lmax - m)/2 + 1;
	const double4_t* __restrict__ coef = a.coef + row0;
	double* __restrict__ pout = a.part + (long)bb*a.mom_bs + ((long)wv*a.rows_chunk + (row0 - a.rowbase))*4;
	const double2* __restrict__ in = a.leg + (long)bb*a.leg_bs + (long)m*a.ld;
	double csq[K], lam1[K], lam2[K], d1r[K], d1i[K], d2r[K], d2i[K];
	int sc[K];
	bool alive_any = false;
	const bool polar = leg_wave_polar(a, wv, K);
	// ring data of slot s: sum and (difference x cos theta) of the north and south ring
	auto load_data = [&](int s) {
		const int p = (wv*K + s)*64 + lane;
		const bool valid = p < a.npairs;
		const int rn = valid ? a.ring_n[p] : -1, rs = valid ? a.ring_s[p] : -1;
		const double x = valid ? a.cth[p] : 0.0;
		const double2 vn = rn >= 0 ? in[rn] : make_double2(0, 0);
		const double2 vs = rs >= 0 ? in[rs] : make_double2(0, 0);
		d1r[s] = vn.x + vs.x; d1i[s] = vn.y + vs.y;
		d2r[s] = (vn.x - vs.x)*x; d2i[s] = (vn.y - vs.y)*x;
	};
#pragma unroll
	for (int s = 0; s < K; s++) {
		const int p = (wv*K + s)*64 + lane;
		const bool valid = p < a.npairs;
		const double x = valid ? a.cth[p] : 0.0;
		const double sth = valid ? a.sth[p] : 0.0;
		csq[s] = polar ? -sth*sth : x*x;
		const bool alive = valid && ((double)m <= a.lmax*sth + a.ofs);
		lam1[s] = 0; lam2[s] = 0; sc[s] = 0;
		if (alive && a.seed_mode != 2) { double mt; int e; pow_scaled(sth, m, mt, e); to_scaled(mt, e, lam2[s], sc[s]); }
		// a lane below scale 0 keeps zero data until it gets there, so that it can run the ungated steps
		d1r[s] = d1i[s] = d2r[s] = d2i[s] = 0;
		alive_any |= alive;
	}
	if (!__any(alive_any)) return;      // partial buffer is pre-zeroed
	int k = 0;
	S0_SEEDED_PHASE_A
	k = PXS_UNIFORM_INT(k); coef = (const double4_t*)PXS_UNIFORM_LONG((long)coef);
	PXS_COUNT(1, (long)(nk - k)*K*6 + (a.seed_mode != 2 ? (long)k*K*2 : 0L));
	if (lane == 0 && !a.atomic) a.first[wv*a.nmc + (m - a.m0)] = k + 1;      // rows before k are not written (reduce_partials skips them)
	// ring data of the lanes that start at scale 0 or reached it during phase A (rings without signal have lam = 0)
#pragma unroll
	for (int s = 0; s < K; s++) if (sc[s] == 0) load_data(s);
	int kk = 0, kbase = k;
	// phase B: plain fast steps; every 4 steps the lanes below scale 0 are rescaled, and a lane that reaches scale 0
	// fetches its ring data (its true terms before that are < 2^-340 of the result)
	while (k + 1 < nk) {
		bool pend = false;
#pragma unroll
		for (int s = 0; s < K; s++) pend |= (sc[s] < 0);
		if (!__any(pend)) break;
		for (int it = 0; it < 2 && k + 1 < nk; it++, k += 2) {
			const double4_t c0 = LDC(coef, k), c1 = LDC(coef, k+1);
			S0_ANA_PAIR(c0, c1)
		}
#pragma unroll
		for (int s = 0; s < K; s++)
			if (sc[s] < 0 && fabs(lam2[s]) > SC_BIG) {
				lam1[s] *= SC_SMALL; lam2[s] *= SC_SMALL;
				if (++sc[s] == 0) load_data(s);
			}
	}
	// phase C: every lane at scale 0 (or without data): next coefficients prefetched with scalar loads
	double4_t c0 = LDC(coef, k), c1 = LDC(coef, k+1);
#ifndef PXS_NO_PHASEC_UNROLL
	while (k + 3 < nk) {      // (two pairs per iteration on alternating row sets, see leg_syn_s0)
		double4_t n0 = LDC(coef, k+2), n1 = LDC(coef, k+3);
		S0_ANA_PAIR(c0, c1)
		k += 2;
		c0 = LDC(coef, k+2); c1 = LDC(coef, k+3);
		S0_ANA_PAIR(n0, n1)
		k += 2;
	}
#endif
	for (; k + 1 < nk; k += 2) {
		const double4_t n0 = LDC(coef, k+2), n1 = LDC(coef, k+3);
		S0_ANA_PAIR(c0, c1)
		c0 = n0; c1 = n1;
	}
	if (k < nk) {
		double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
		for (int s = 0; s < K; s++) { t0 = fma(lam2[s], d1r[s], t0); t1 = fma(lam2[s], d1i[s], t1); t2 = fma(lam2[s], d2r[s], t2); t3 = fma(lam2[s], d2i[s], t3); }
		LEG_RED_PUT(kk, t0, t1, t2, t3)
		kk++;
	}
	if (kk > 0) leg_flush(red, pout + 4*kbase, lane, kk, a.atomic);
}


// ---- batched spin-0 analysis as an FP64-MFMA GEMM (round 5) -----------------------------------------------------------------------
// For the maps of a batched call the per-m problem is  mom[k][map, c] = sum_ring p_k(ring) D[ring][map, c]  with the SAME p_k(ring) for
// every map (c = the four real right-hand sides of leg_ana_s0: re / im of the ring-pair sum, re / im of the difference x cos theta):
// the ring axis is the K dimension of v_mfma_f64_16x16x4_f64, M = 16 consecutive recurrence steps, N = 16 = 4 maps x 4 sides.
// The reference loops over the maps, one ducc0 call each (pixell/curvedsky.py:1038-1046); here a wave runs ONE Ishioka recurrence per
// ring pair (lane = ring pair, phases A / B as in leg_ana_s0; a lane below scale 0 contributes p = 0), parks 16 steps of it in a
// [16][64] LDS tile and issues 16 MFMAs per tile and group of 4 maps against B operands (the ring data, 16 x 2 VGPRs per group) that
// stay in registers for the whole l loop.  What the VALU form pays per map -- the recurrence (2 of 6 FMAs), the 64-lane
// reduce-scatter (10 of ~58 VALU per step) and the three-VGPR-operand FMA rate -- is paid once per 4 NG maps or not at all.
//  * Workgroup = 8 waves over 512 consecutive ring pairs, wave w the pairs [64 w, 64 w + 64): polar waves join at the tile where their
//    first lane reaches scale 0.  Tiles are aligned to multiples of 16 steps of the m; every wave adds its 16 x 16 accumulators
//    into an LDS tile (ds_add_f64) and after ONE barrier per tile the waves share out the flush: one global_atomic_add_f64 per
//    (chunk of 512 pairs, row, map) -- the count of leg_ana_s0<8>.
//  * Recurrence lane L is MFMA slot (kk, q) = (L >> 4, L & 15): the P row is written as 64 consecutive doubles and lane (i, kk)
//    reads its 16 A operands P[i][16 kk + q] from rows of 65 doubles -- conflict-free for ds_read_b64 and ds_read2_b64 alike.
//  * The ring data reach the B registers through the LDS: the 512 threads read the rows leg[map][m][ring] of 4 maps coalesced (one
//    ring pair per thread), park (sum, difference x cos) as 16 doubles per pair (17-double entries: lane (j, kk) of MFMA q then reads
//    entry 64 w + 16 kk + q, double j, conflict-free), and every lane picks its 16 operands.  (First form: per-lane gathers straight
//    from global memory -- 16 % of the kernel's wave time, tools/mm_time.sh.)
//  * Step coefficients come from a compact table (a, b) resp. (a, a + b) per step (LegTables::coef2), the 16 steps of the NEXT tile
//    requested with four s_load_dwordx16 before the MFMAs of the current one (first form: the 32-byte rows of the VALU kernels,
//    requested and awaited group by group -- four scalar-load round trips per tile, 38 % of the wave time).
// lab build (-DPXS_LAB_MMTIME): shader-clock time of the phases of leg_ana_s0_mm, summed over the waves (tools/mm_time.sh)
#if defined(PXS_LAB_MMTIME) && !defined(PXS_HOST_SIM)
__device__ unsigned long long mm_prof[16];
#define MM_T0 long long tprev_ = clock64(); unsigned long long tacc_[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define MM_TICK(i) { const long long tn_ = clock64(); tacc_[i] += (unsigned long long)(tn_ - tprev_); tprev_ = tn_; }
#define MM_TDUMP if (lane == 0) { for (int i_ = 0; i_ < 14; i_++) atomicAdd(&mm_prof[i_], tacc_[i_]); atomicAdd(&mm_prof[15], 1ull); }
#else
#define MM_T0
#define MM_TICK(i)
#define MM_TDUMP
#endif
template<int NG, int W> __global__ __launch_bounds__(64*W, 4) void leg_ana_s0_mm(const LegK a)
{
	PXS_SHARED(double, sh);
	constexpr int K = 1;
	MM_T0
	double* __restrict__ ptile = sh;                              // [W][16][MM_PSTRIDE]
	double* __restrict__ red = sh + W*16*MM_PSTRIDE;              // [2][4 NG][64]: the accumulators of a tile summed over the waves
	int* __restrict__ s_kmin = reinterpret_cast<int*>(sh + mm_lds_doubles(NG, W));
	const int tid = threadIdx.x, lane = tid & 63, w = PXS_UNIFORM_INT(tid >> 6);
	int wv, m, bb;
	if (!leg_block(a, wv, m, bb)) return;
	const long row0 = PXS_UNIFORM_LONG(a.row[m]);
	const int nk = (a.lmax - m)/2 + 1;
	const double4_t* __restrict__ coef = a.coef + row0;
	const int pbase = wv*64*W;
	const bool polar = [&] { const double c = a.cth[min(pbase + 64*(w + 1), a.npairs) - 1]; return c*c > PXS_POLAR_COS2; }();      // (per wave: its own 64 pairs, its own coefficient stream)
	double csq[K], lam1[K], lam2[K]; int sc[K];
	bool alive;
	{
		const int p = pbase + tid;      // pair of this recurrence lane: wave w owns the pairs [64 w, 64 w + 64) of the chunk
		const bool valid = p < a.npairs;
		const double x = valid ? a.cth[p] : 0.0, sth = valid ? a.sth[p] : 0.0;
		csq[0] = polar ? -sth*sth : x*x;
		alive = valid && ((double)m <= a.lmax*sth + a.ofs);
		lam1[0] = 0; lam2[0] = 0; sc[0] = 0;
		if (alive) { double mt; int e; pow_scaled(sth, m, mt, e); to_scaled(mt, e, lam2[0], sc[0]); }
	}
	if (tid == 0) *s_kmin = nk;
	__syncthreads();
	MM_TICK(0)
	// phase A, per wave: recurrence only until the first lane of the wave is at scale 0; kw = the first step this wave contributes to
	int k = 0;
	const bool wave_alive = __any(alive);
	if (wave_alive) { S0_PHASE_A }
	const int kw = wave_alive ? PXS_UNIFORM_INT(k) : nk + 16;
	MM_TICK(1)
	if (lane == 0) atomicMin(s_kmin, kw);
	__syncthreads();
	const int kmin = PXS_UNIFORM_INT(*s_kmin);
	MM_TICK(2)
	if (kmin >= nk) { MM_TDUMP return; }      // (workgroup-uniform) no ring of this chunk carries signal at this m
	// B operands through the LDS: thread = ring pair, 4 maps per round
	double breg[NG][16];
	{
		const int p = pbase + tid;
		const bool ok = p < a.npairs;
		const int rn = ok ? a.ring_n[p] : -1, rs = ok ? a.ring_s[p] : -1;
		const double x = ok ? a.cth[p] : 0.0;
		MM_TICK(8)
		double* __restrict__ ent = sh + tid*MM_ESTRIDE;
		const double* __restrict__ rd = sh + (64*w + 16*(lane >> 4))*MM_ESTRIDE + (lane & 15);
#pragma unroll
		for (int g = 0; g < NG; g++) {
			double2 vn[4], vs[4];
#pragma unroll
			for (int mm = 0; mm < 4; mm++) {
				const int map = (bb*NG + g)*4 + mm;
				const double2* __restrict__ in = a.leg + (long)map*a.leg_bs + (long)m*a.ld;
				const bool okm = map < a.nmaps;
				vn[mm] = (okm && rn >= 0) ? in[rn] : make_double2(0, 0); vs[mm] = (okm && rs >= 0) ? in[rs] : make_double2(0, 0);
			}
			MM_TICK(9)
			if (g > 0) __syncthreads();      // the reads of the previous round
#pragma unroll
			for (int mm = 0; mm < 4; mm++) {
				ent[4*mm + 0] = vn[mm].x + vs[mm].x; ent[4*mm + 1] = vn[mm].y + vs[mm].y;
				ent[4*mm + 2] = (vn[mm].x - vs[mm].x)*x; ent[4*mm + 3] = (vn[mm].y - vs[mm].y)*x;
			}
			__syncthreads();
			MM_TICK(10)
#pragma unroll
			for (int q = 0; q < 16; q++) breg[g][q] = rd[q*MM_ESTRIDE];
			MM_WAVE_SYNC();
			MM_TICK(11)
		}
		__syncthreads();
		for (int i = tid; i < 2*NG*4*64; i += 64*W) red[i] = 0.0;
		__syncthreads();
	}
	MM_TICK(3)
	const double* __restrict__ tab = reinterpret_cast<const double*>(polar ? a.coef2p : a.coef2) + 2*row0;      // (a, b') of step k at tab[2 k]
	bool pend = __any(sc[0] < 0);
	double* __restrict__ pmine = ptile + w*16*MM_PSTRIDE;
	const double* __restrict__ pread = pmine + (lane & 15)*MM_PSTRIDE + 16*(lane >> 4);
	double cf[32]; int cf_tile = -1;      // coefficients of the 16 steps of tile cf_tile, requested a tile ahead
	long ntile = 0;
	// flush of tile tf (after the barrier that ends it): register r of group g holds rows 4 r + lane / 16 of the tile, column lane % 16 =
	// 4 (map in the group) + side.  It is issued behind the MFMAs of the NEXT tile (the two reduction tiles alternate), off the path
	// from the barrier to that tile's recurrence.
	auto mm_flush = [&](int tf) {
		double* __restrict__ redf = red + (tf & 1)*NG*4*64;
		for (int c = w; c < 4*NG; c += W) {
			const int g = c >> 2, r = c & 3;
			double* rp = redf + c*64 + lane;
			const double v = *rp; *rp = 0.0;
			const int krow = 16*tf + 4*r + (lane >> 4), map = (bb*NG + g)*4 + ((lane & 15) >> 2);
			if (krow < nk && map < a.nmaps) {
				double* dst = a.mom + (long)map*a.mom_bs + 4*(row0 + krow) + (lane & 3);
#ifdef PXS_HOST_SIM
				atomicAdd(dst, v);
#elif defined(PXS_LAB_NOATOM)
				if (v == 12345.678) *dst = v;      // timing experiment (wrong results)
#else
				unsafeAtomicAdd(dst, v);
#endif
			}
		}
	};
	int tlast = -1;
	for (int t = kmin >> 4; 16*t < nk; t++) {
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
				if (kq >= kw && kq < nk) {
					p0 = lam2[0]; lam1[0] = fma(mm_coef(cf[8*q4 + 0], csq[0], cf[8*q4 + 1]), lam2[0], lam1[0]);
					p1 = lam1[0]; lam2[0] = fma(mm_coef(cf[8*q4 + 2], csq[0], cf[8*q4 + 3]), lam1[0], lam2[0]);
					p2 = lam2[0]; lam1[0] = fma(mm_coef(cf[8*q4 + 4], csq[0], cf[8*q4 + 5]), lam2[0], lam1[0]);
					p3 = lam1[0]; lam2[0] = fma(mm_coef(cf[8*q4 + 6], csq[0], cf[8*q4 + 7]), lam1[0], lam2[0]);
					if (pend) {      // phase B: lanes below scale 0 contribute nothing yet; rescale them every 4 steps
						if (sc[0] < 0) { p0 = p1 = p2 = p3 = 0.0; if (fabs(lam2[0]) > SC_BIG) { lam1[0] *= SC_SMALL; lam2[0] *= SC_SMALL; sc[0]++; } }
						pend = __any(sc[0] < 0);
					}
					// rows beyond the last step of this m stay out of the sums (their table rows belong to the next m)
					if (kq + 1 >= nk) p1 = 0.0;
					if (kq + 2 >= nk) p2 = 0.0;
					if (kq + 3 >= nk) p3 = 0.0;
				}
				pmine[(4*q4 + 0)*MM_PSTRIDE + lane] = p0; pmine[(4*q4 + 1)*MM_PSTRIDE + lane] = p1;
				pmine[(4*q4 + 2)*MM_PSTRIDE + lane] = p2; pmine[(4*q4 + 3)*MM_PSTRIDE + lane] = p3;
			}
			MM_WAVE_SYNC();
			MM_TICK(4)
			double av[4];
#pragma unroll
			for (int q = 0; q < 4; q++) av[q] = pread[q];
			MM_WAVE_SYNC();
			if (k0 + 16 < nk) {      // the rows of the next tile, on their way during the MFMAs (requested after the first A operands have landed)
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
			MM_TICK(5)
		}
		if (tlast >= 0) mm_flush(tlast);
		tlast = t;
		__syncthreads();
		MM_TICK(6)
	}
	if (tlast >= 0) mm_flush(tlast);
	MM_TDUMP
	PXS_COUNT(1, ntile*(NG*256L + 32L) + (wave_alive ? (long)kw*2 : 0L));
}


// ---- batched spin-0 synthesis as an FP64-MFMA GEMM (round 5) ----------------------------------------------------------------------
// The transpose of leg_ana_s0_mm: leg[ring][map, c] = sum_k p_k(ring) almt[k][map, c], c = the four real columns of alm_pre_s0 (even
// part re / im, odd part re / im).  M = 16 ring pairs, N = 16 = 4 maps x 4 columns, K = recurrence steps: the accumulators
// (64 ring pairs x 16 columns per group of 4 maps = 4 x 8 VGPRs) stay in registers for the whole l loop, one wave per workgroup
// and NO cross-wave step at all (every wave owns its rings).  A operands: lane (i, kk) of MFMA (rb, q) takes step q + 4 kk of ring
// pair 16 rb + i from the wave's [16][68] P tile (the same tile and recurrence as the analysis; rows of 68 doubles: the four steps of
// an MFMA lie 4 rows = 32 banks apart); B operands: the pre-scaled alm rows of the tile, one double per lane and MFMA step-quad,
// loaded a tile ahead (the 32 bytes per step and map the VALU kernel takes through the scalar cache).  At the end a lane holds one
// column of one ring pair: the quad (even re, even im, odd re, odd im) is combined across lanes into the north and south ring values.

template<int NG> __global__ __launch_bounds__(64, 4) void leg_syn_s0_mm(const LegK a)
{
	PXS_SHARED(double, pmine);      // [16][MMS_PSTRIDE]
	constexpr int K = 1;
	const int lane = threadIdx.x;
	int wv, m, bb;
	if (!leg_block(a, wv, m, bb)) return;
	const long row0 = PXS_UNIFORM_LONG(a.row[m]);
	const int nk = (a.lmax - m)/2 + 1;
	const double4_t* __restrict__ coef = a.coef + row0;
	const int pbase = wv*64;
	const bool polar = [&] { const double c = a.cth[min(pbase + 64, a.npairs) - 1]; return c*c > PXS_POLAR_COS2; }();
	double csq[K], lam1[K], lam2[K]; int sc[K];
	bool alive;
	{
		const int p = pbase + lane;
		const bool valid = p < a.npairs;
		const double x = valid ? a.cth[p] : 0.0, sth = valid ? a.sth[p] : 0.0;
		csq[0] = polar ? -sth*sth : x*x;
		alive = valid && ((double)m <= a.lmax*sth + a.ofs);
		lam1[0] = 0; lam2[0] = 0; sc[0] = 0;
		if (alive) { double mt; int e; pow_scaled(sth, m, mt, e); to_scaled(mt, e, lam2[0], sc[0]); }
	}
	mm_acc acc[NG][4];
#pragma unroll
	for (int g = 0; g < NG; g++)
#pragma unroll
		for (int rb = 0; rb < 4; rb++) { acc[g][rb][0] = 0; acc[g][rb][1] = 0; acc[g][rb][2] = 0; acc[g][rb][3] = 0; }
	long ntile = 0;
	// phase A: recurrence only until the first lane of the wave is at scale 0 (a wave without a live ring skips the loop below)
	int k = 0;
	const bool wave_alive = __any(alive);
	if (wave_alive) { S0_PHASE_A }
	const int kw = PXS_UNIFORM_INT(wave_alive ? k : nk + 16);      // (explicitly wave-uniform: left as a select, the loop below was compiled as divergent and the prefetched coefficient rows went to VGPRs)
	coef = (const double4_t*)PXS_UNIFORM_LONG((long)coef);
	{
		const double* __restrict__ tab = reinterpret_cast<const double*>(polar ? a.coef2p : a.coef2) + 2*row0;      // (a, b') of step k at tab[2 k]
		// B operand of MFMA step-quad q: lane (j, kk) holds column j & 3 of map 4 (bb NG + g) + (j >> 2) at step q + 4 kk of the tile
		const int jcol = lane & 15, kk4 = lane >> 4;
		const double* bsrc[NG]; bool bok[NG];
#pragma unroll
		for (int g = 0; g < NG; g++) {
			const int map = (bb*NG + g)*4 + (jcol >> 2);
			bok[g] = map < a.nmaps;
			bsrc[g] = a.almt + (long)(bok[g] ? map : 0)*a.almt_bs + 4*row0 + (jcol & 3) + 16*kk4;
		}
		auto load_b = [&](int k0, double (*b)[4]) {
#pragma unroll
			for (int g = 0; g < NG; g++)
#pragma unroll
				for (int q = 0; q < 4; q++) b[g][q] = (bok[g] && k0 + q + 4*kk4 < nk) ? bsrc[g][4L*(k0 + q)] : 0.0;
		};
		bool pend = __any(sc[0] < 0);
		const double* __restrict__ pread = pmine + 4*(lane >> 4)*MMS_PSTRIDE + (lane & 15);
		double cf[32];      // coefficients of the 16 steps of the tile, requested a tile ahead (every tile from the wave's first one on is run)
		double bcur[NG][4], bnxt[NG][4];
		load_b(16*(kw >> 4), bcur);
		if (kw < nk) {
#pragma unroll
			for (int i = 0; i < 32; i++) cf[i] = LDCD(tab, 32L*(kw >> 4) + i);
		}
		for (int t = kw >> 4; 16*t < nk; t++) {
			const int k0 = 16*t;
			ntile++;
#pragma unroll
			for (int q4 = 0; q4 < 4; q4++) {
				const int kq = k0 + 4*q4;
				double p0 = 0, p1 = 0, p2 = 0, p3 = 0;
				if (kq >= kw && kq < nk) {
					p0 = lam2[0]; lam1[0] = fma(mm_coef(cf[8*q4 + 0], csq[0], cf[8*q4 + 1]), lam2[0], lam1[0]);
					p1 = lam1[0]; lam2[0] = fma(mm_coef(cf[8*q4 + 2], csq[0], cf[8*q4 + 3]), lam1[0], lam2[0]);
					p2 = lam2[0]; lam1[0] = fma(mm_coef(cf[8*q4 + 4], csq[0], cf[8*q4 + 5]), lam2[0], lam1[0]);
					p3 = lam1[0]; lam2[0] = fma(mm_coef(cf[8*q4 + 6], csq[0], cf[8*q4 + 7]), lam1[0], lam2[0]);
					if (pend) {      // phase B: lanes below scale 0 contribute nothing yet; rescale them every 4 steps
						if (sc[0] < 0) { p0 = p1 = p2 = p3 = 0.0; if (fabs(lam2[0]) > SC_BIG) { lam1[0] *= SC_SMALL; lam2[0] *= SC_SMALL; sc[0]++; } }
						pend = __any(sc[0] < 0);
					}
					if (kq + 1 >= nk) p1 = 0.0;
					if (kq + 2 >= nk) p2 = 0.0;
					if (kq + 3 >= nk) p3 = 0.0;
				}
				pmine[(4*q4 + 0)*MMS_PSTRIDE + lane] = p0; pmine[(4*q4 + 1)*MMS_PSTRIDE + lane] = p1;
				pmine[(4*q4 + 2)*MMS_PSTRIDE + lane] = p2; pmine[(4*q4 + 3)*MMS_PSTRIDE + lane] = p3;
			}
			MM_WAVE_SYNC();
			double av[4];
#pragma unroll
			for (int rb = 0; rb < 4; rb++) av[rb] = pread[16*rb];
			MM_WAVE_SYNC();
			if (k0 + 16 < nk) {      // the rows of the next tile (coefficients and pre-scaled alm), on their way during the MFMAs
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
	// register r of acc[g][rb] at lane (i4 = lane / 16, j = lane % 16): ring pair 16 rb + 4 r + i4, column j = 4 (map in the group) + c
	const int c = lane & 3;
#pragma unroll
	for (int g = 0; g < NG; g++) {
		const int map = (bb*NG + g)*4 + ((lane & 15) >> 2);
		double* __restrict__ out = reinterpret_cast<double*>(a.leg + (long)(map < a.nmaps ? map : 0)*a.leg_bs + (long)m*a.ld) + (c & 1);
#pragma unroll
		for (int rb = 0; rb < 4; rb++)
#pragma unroll
			for (int r = 0; r < 4; r++) {
				const int p = pbase + 16*rb + 4*r + (lane >> 4);
				const bool valid = p < a.npairs && map < a.nmaps;
				const double v = acc[g][rb][r], o = MMS_XOR2(v);
				const double x = valid ? a.cth[p] : 0.0;
				// c = 0, 1: north ring, even + x odd; c = 2, 3: south ring, even - x odd (this lane holds the odd part)
				const double val = c < 2 ? fma(x, o, v) : fma(-x, v, o);
				const int ring = valid ? (c < 2 ? a.ring_n[p] : a.ring_s[p]) : -1;
				if (ring >= 0) out[2*ring] = val;
			}
	}
	PXS_COUNT(0, ntile*(NG*256L + 32L) + (wave_alive ? (long)kw*2 : 0L));
}

// ---- launchers ----
void launch_leg_syn_s0(int K, dim3 grid, hipStream_t st, const LegK& a) {
	// (ring pairs per lane the product's rules select: 4 and 2; lab builds -- PXS_K_* -- compile the others too)
#ifdef PXS_LAB
	if (K == 8) { hipLaunchKernelGGL(leg_syn_s0<8>, grid, dim3(64), 0, st, a); return; }
#endif
	if (K == 2) hipLaunchKernelGGL(leg_syn_s0<2>, grid, dim3(64), 0, st, a);
	else        hipLaunchKernelGGL(leg_syn_s0<4>, grid, dim3(64), 0, st, a);
}
void launch_leg_ana_s0(int K, dim3 grid, size_t lds, hipStream_t st, const LegK& a) {
#ifdef PXS_LAB
	if (K == 12) { hipLaunchKernelGGL(leg_ana_s0<12>, grid, dim3(64), lds, st, a); return; }
#endif
	if (K == 8)      hipLaunchKernelGGL(leg_ana_s0<8>, grid, dim3(64), lds, st, a);
	else if (K == 2) hipLaunchKernelGGL(leg_ana_s0<2>, grid, dim3(64), lds, st, a);
	else             hipLaunchKernelGGL(leg_ana_s0<4>, grid, dim3(64), lds, st, a);
}
void launch_leg_syn_s0_mm(int ng, dim3 grid, hipStream_t st, const LegK& a) {
	if (ng == 2) hipLaunchKernelGGL(leg_syn_s0_mm<2>, grid, dim3(64), mm_syn_lds(), st, a);
	else         hipLaunchKernelGGL(leg_syn_s0_mm<1>, grid, dim3(64), mm_syn_lds(), st, a);
}
template<int NG, int W> static void mm_launch1(dim3 grid, hipStream_t st, const LegK& a) {
	static const bool once = [] { (void)hipFuncSetAttribute((const void*)leg_ana_s0_mm<NG, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 256); return true; }(); (void)once;
	hipLaunchKernelGGL((leg_ana_s0_mm<NG, W>), grid, dim3(64*W), mm_ana_lds(NG, W), st, a);
#if defined(PXS_LAB_MMTIME) && !defined(PXS_HOST_SIM)
	{	unsigned long long h[16]; PXS_HIP(hipStreamSynchronize(st)); PXS_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(mm_prof), sizeof(h)));
		static const char* nm_[14] = {"init", "phaseA", "kmin_barrier", "B_final_barriers", "P_phase", "mfma+ds_add", "barrier", "flush", "B_index", "B_data_loads", "B_lds_write+barrier", "B_lds_read", "-", "-"};
		double tot = 0; for (int i = 0; i < 14; i++) tot += (double)h[i];
		fprintf(stderr, "[mm_prof] waves %llu, cycles per wave %.0f:", h[15], tot/std::max(1.0, (double)h[15]));
		for (int i = 0; i < 12; i++) fprintf(stderr, " %s %.1f%%", nm_[i], 100.0*h[i]/tot);
		fprintf(stderr, "\n"); memset(h, 0, sizeof(h)); PXS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(mm_prof), h, sizeof(h))); }
#endif
}
template<int W> static void mm_launch(int ng, dim3 grid, hipStream_t st, const LegK& a) { if (ng == 2) mm_launch1<2, W>(grid, st, a); else mm_launch1<1, W>(grid, st, a); }
void launch_leg_ana_s0_mm(int ng, int W, dim3 grid, hipStream_t st, const LegK& a) {
#ifdef PXS_LAB      /* (lab builds: other workgroup sizes, PXS_ANA_MM_W; measured at C4: 2 waves 95 ms, 4: 78, 8: 77, 16: 82 per 64 maps) */
	if (W == 16) { mm_launch<16>(ng, grid, st, a); return; }
	if (W == 4)  { mm_launch<4>(ng, grid, st, a); return; }
	if (W == 2)  { mm_launch<2>(ng, grid, st, a); return; }
#endif
	mm_launch<MM_WAVES>(ng, grid, st, a);
}

} // namespace pxs
