// Device-side pieces shared by the Legendre translation units (legendre.hip: host side, alm kernels, tables; leg_s0.hip: spin-0 kernels;
// leg_spin.hip: spin-s kernels): kernel arguments, block -> (m, ring chunk) order, extended-exponent helpers, the phase-A macros, the
// lane-sum reduction tile of the VALU analysis kernels and the MFMA helpers of the batched kernels.  Design notes: head of legendre.hip.
#pragma once
#include "legendre.hpp"
#include <cmath>
#include <algorithm>

namespace pxs {

static constexpr double SC_BIG   = 0x1p+400;
static constexpr double SC_SMALL = 0x1p-800;
static constexpr int    SC_STEP  = 800;

struct double4_t { double a, b, c, d; };
// Wave-uniform table rows are fetched through the constant address space: that makes them scalar loads (s_load_dwordx8)
// even in kernels that also store to global memory inside their loops.  Without it the analysis kernels, whose flush
// stores precede later row loads, got per-lane global_load broadcasts for every coefficient row (SQ_INSTS_SMEM 5.5e7
// against SQ_INSTS_VMEM_RD 2.2e9 for leg_ana_spin<6> at config 3; the synthesis kernels had 4e9 scalar loads).
#ifdef PXS_HOST_SIM
#define LDC(p, i) ((p)[i])
#else
typedef double pxs_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double4_t ldc_row(const double4_t* p, long i) {
	const __attribute__((address_space(4))) pxs_d4* c = (const __attribute__((address_space(4))) pxs_d4*)(unsigned long long)p;
	const pxs_d4 v = c[i]; double4_t r; r.a = v.x; r.b = v.y; r.c = v.z; r.d = v.w; return r;
}
#define LDC(p, i) ldc_row((p), (i))
#endif
// one double of a wave-uniform table through the constant address space (s_load_dwordx2)
#ifdef PXS_HOST_SIM
#define LDCD(p, i) ((p)[i])
#else
__device__ __forceinline__ double ldc_double(const double* p, long i) {
	const __attribute__((address_space(4))) double* c = (const __attribute__((address_space(4))) double*)(unsigned long long)p;
	return c[i];
}
#define LDCD(p, i) ldc_double((p), (i))
#endif

struct LegK {
	int lmax, mmax, spin, nm, npairs, nring, nwave;
	long ld;                            // row stride of leg[m][ring] (>= nring; rows padded to whole 128-byte lines)
	long nrows;
	const long* row; const double4_t* coef; const double* alpha;
	const int* ring_n; const int* ring_s; const double* cth; const double* sth; const double* sh2; const double* ch2;
	double* almt; double* part; double* mom;
	double2* leg;
	double ofs;
	int m0; long rowbase, rows_chunk;   // analysis processes m in chunks to bound the partial-moment scratch
	int nmc, xcd;                       // m count of this launch; XCD-aware block order on/off
	int* first;                         // analysis: see LegWork::first
	int atomic;                         // analysis: waves add their sums into mom (part = mom) instead of writing per-wave partial moments
	// recurrence seeds: the state of every chain at the end of phase A, per (m, wave): [nd][K][64] doubles, [ni][K][64] + 64 ints
	// (the first of the last 64: the step reached).  mode 0: off, 1: run phase A and record, 2: load instead of running it
	int seed_mode; double* seed_d; int* seed_i;
	// executed-work counters (profiling on: pxs_profile; null otherwise): count[0] synthesis, count[1] analysis, in FMA instructions
	// per lane: every wave adds (steps it ran) x K x (FMAs per ring pair and step: 6 per two degrees for spin 0, 12 per degree for
	// spin s; 2 / 4 in the recurrence-only phase A).  x 64 lanes x 2 = the FP64 flops the hardware executed in the recurrences and
	// accumulations (rings dropped as polar-dead and (wave, m) pairs skipped entirely are not in it, masked-off lanes are).
	double* count;
	// maps of a batched call in one launch: the waves of one m of ALL maps sit next to each other in an XCD's queue, so the maps
	// share the coefficient rows in L2 and the scalar cache.  Strides in elements of leg (double2), almt and mom (double).
	int nb; long leg_bs, almt_bs, mom_bs;
	int nmaps;                          // MFMA kernels (leg_ana_s0_mm): nb counts GROUPS of maps there, nmaps the maps themselves
	const double2* coef2; const double2* coef2p;      // compact step table (a, b) / (a, a + b) of the MFMA kernels (LegTables::coef2)
};
// (PXS_NCOUNT slots, one picked by the block index: 200 000 waves adding to ONE address cost ~10 ms per C3 step and 24 ms per C4 step)
#define PXS_NCOUNT 1024
#ifdef PXS_HOST_SIM
#define PXS_COUNT(dir, expr) if (a.count != nullptr && lane == 0) atomicAdd(a.count + 2*(blockIdx.x & (PXS_NCOUNT - 1)) + (dir), (double)(expr))
#else
#define PXS_COUNT(dir, expr) if (a.count != nullptr && lane == 0) unsafeAtomicAdd(a.count + 2*(blockIdx.x & (PXS_NCOUNT - 1)) + (dir), (double)(expr))
#endif

// Block -> (m, ring chunk).  Every wave of one m streams the same coefficient rows (32 B per l) through the
// scalar cache; workgroups are dealt round-robin to the 8 XCDs, each with a private L2.  With the plain
// (chunk, m) grid the nwave readers of a stream were spread over all XCDs and drifted apart, and FETCH_SIZE
// showed every one of them going to the fabric (49 GB per leg_syn_spin launch at config 3 = nwave x the
// table).  This order gives all chunks of one m the same `block % 8`, back to back in that XCD's queue, so
// one reader misses and the others hit in L2.
// (Workgroups of 2-4 independent waves of the same m -- to share the rows in the CU's scalar cache -- were measured twice:
// with __launch_bounds__(256) and the lane taken as threadIdx.x & 63 every kernel got slower even at one wave per workgroup
// (config 3: leg_syn 106 -> 113 ms, leg_ana 144 -> 151 ms), 4 waves per workgroup 128 / 165 ms.  One wave per workgroup stays.)
__device__ __forceinline__ bool leg_block(const LegK& a, int& wv, int& m, int& bb) {
	if (!a.xcd) { wv = blockIdx.x; m = blockIdx.y + a.m0; bb = blockIdx.z; return true; }
	const unsigned b = blockIdx.x, x = b & 7u, j = b >> 3;
	const unsigned per_m = (unsigned)a.nwave*(unsigned)a.nb;
	const unsigned ml = j / per_m, r = j - ml*per_m;
	bb = (int)(r / (unsigned)a.nwave);
	wv = (int)(r - (unsigned)bb*a.nwave);
	const unsigned mi = ml*8u + x;
	m = (int)mi + a.m0;
	return mi < (unsigned)a.nmc;
}
static inline dim3 leg_grid(const LegK& a) { return a.xcd ? dim3((unsigned)(8*((a.nmc+7)/8)*a.nwave*a.nb)) : dim3(a.nwave, a.nmc, a.nb); }

// ---------------------------------------------------------------------------------
// scaled powers
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void frexp_norm(double& m, int& e) { int d; m = frexp(m, &d); e += d; }

// x^n as mant * 2^e, mant in [0.5,1) (or 0)
__device__ __forceinline__ void pow_scaled(double x, int n, double& mant, int& e) {
	double rm = 0.5; int re = 1;
	int be = 0; double bm = frexp(x, &be);
	while (n) {
		if (n & 1) { rm *= bm; re += be; frexp_norm(rm, re); }
		bm *= bm; be *= 2; frexp_norm(bm, be);
		n >>= 1;
	}
	mant = rm; e = re;
}
// value = mant*2^e  ->  v*2^(800*scale), scale <= 0, |v| <= 2^400
__device__ __forceinline__ void to_scaled(double mant, int e, double& v, int& scale) {
	if (mant == 0.0) { v = 0.0; scale = 0; return; }
	int s = (e >= 0) ? (e + SC_STEP/2)/SC_STEP : -((-e + SC_STEP/2)/SC_STEP);
	if (s > 0) s = 0;
	v = ldexp(mant, e - SC_STEP*s); scale = s;
}

// coefficient adjusted accordingly (table columns c,d).  This keeps full relative precision near the
// poles, where x = cos(theta) rounds away the information about theta: there the recurrence sits at its double root (step coefficient
// 2 - (l theta)^2-ish) and an ABSOLUTE error eps in the coefficient grows like l^2 eps -- 4e-13 of the map rms on the rings next to the poles at
// lmax 240, 3e-12 at lmax 600 (tests/test_grid_fuzz.py found it), against 1e-14 elsewhere.  The form is exact algebra for every ring but cancels
// towards the equator (a (1 - sin^2) + b: the spin-0 equator ring of a wave forced into it went from 4e-14 to 7e-13 at lmax 240), so a wave takes it
// only if its MOST EQUATORIAL ring still has cos^2 > 0.1 (theta < 71.5 deg: at most one digit of the coefficient).  Until round 5 the bound was 1/2:
// a wave spans 256-512 ring pairs, so grids below 1024-2048 rings -- and the CC-grid detour of the synthesis up to lmax ~2000 -- never ran their
// polar rings in this form; with 0.1 that shrinks to 644-1288 rings (below which l^2 eps stays under ~4e-12).
#ifndef PXS_POLAR_COS2
#define PXS_POLAR_COS2 0.1
#endif
__device__ __forceinline__ bool leg_wave_polar(const LegK& a, int wv, int K) {
	const int last = min((wv+1)*K*64, a.npairs) - 1;   // most equatorial pair of the wave (wave-uniform; pairs are ordered pole first)
	const double c = a.cth[last];
	return c*c > PXS_POLAR_COS2;
}

// make a VGPR copy of a wave-uniform value once, so that v_fma_f64 can take it as the addend next to
// an SGPR multiplicand (gfx950 allows one scalar source per VALU op; without this the compiler
// re-materialises the constant for every use with two v_mov_b32)
#ifdef PXS_HOST_SIM
#define PXS_VCOPY(dst, src) double dst = (src)
#else
#define PXS_VCOPY(dst, src) double dst; asm("v_mov_b64 %0, %1" : "=v"(dst) : "s"(src))
#endif

// (An L2 prefetch of the coefficient streams via global_load_lds into an LDS sink was tried to hide SMEM
// miss latency and measured SLOWER on MI355X: leg_syn 10.8 -> 12.4 ms at config 2; removed.)

#ifdef PXS_HOST_SIM
#define PXS_UNIFORM_INT(x) (x)
#define PXS_UNIFORM_LONG(x) (x)
#else
#define PXS_UNIFORM_INT(x) __builtin_amdgcn_readfirstlane(x)
// (a wave-uniform table offset that the compiler keeps in VGPRs turns every coefficient row load of the loops below into a per-lane
// global_load: seen when the seed stores entered the kernels -- leg_syn 101 -> 123 ms at config 3)
#define PXS_UNIFORM_LONG(x) ((long)(((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long)(x) >> 32)) << 32) | (unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long)(x))))
#endif
// Recurrence seeds.  Phase A (recurrence only, no accumulation, until the first lane of the wave reaches scale 0) is the same
// for every transform on a plan: ~18 % of the steps of a live (wave, m) at a third of the cost of an accumulating step, i.e.
// ~5 % of the Legendre time, plus the sin^m start values.  The first launch on a ring set records the state it ends in, later
// launches load it: K x 20 bytes (spin 0) or K x 40 bytes (spin s) per lane.
#define S0_SEEDED_PHASE_A \
	if (a.seed_mode == 2) { \
		const double* sd = a.seed_d + ((long)m*a.nwave + wv)*(2*K*64); const int* si = a.seed_i + ((long)m*a.nwave + wv)*((K+1)*64); \
		k = PXS_UNIFORM_INT(si[K*64]); \
		_Pragma("unroll") for (int s = 0; s < K; s++) { lam1[s] = sd[s*64 + lane]; lam2[s] = sd[(K+s)*64 + lane]; sc[s] = si[s*64 + lane]; } \
	} else { \
		S0_PHASE_A \
		if (a.seed_mode == 1) { \
			double* sd = a.seed_d + ((long)m*a.nwave + wv)*(2*K*64); int* si = a.seed_i + ((long)m*a.nwave + wv)*((K+1)*64); \
			_Pragma("unroll") for (int s = 0; s < K; s++) { sd[s*64 + lane] = lam1[s]; sd[(K+s)*64 + lane] = lam2[s]; si[s*64 + lane] = sc[s]; } \
			if (lane == 0) si[K*64] = k; \
		} \
	}
#define SPIN_SEEDED_PHASE_A \
	if (a.seed_mode == 2) { \
		const double* sd = a.seed_d + ((long)m*a.nwave + wv)*(4*K*64); const int* si = a.seed_i + ((long)m*a.nwave + wv)*((2*K+1)*64); \
		j = PXS_UNIFORM_INT(si[2*K*64]); \
		_Pragma("unroll") for (int s = 0; s < K; s++) { \
			S.gp1[s] = sd[s*64 + lane]; S.gp2[s] = sd[(K+s)*64 + lane]; S.gm1[s] = sd[(2*K+s)*64 + lane]; S.gm2[s] = sd[(3*K+s)*64 + lane]; \
			S.scp[s] = si[s*64 + lane]; S.scm[s] = si[(K+s)*64 + lane]; } \
	} else { \
		SPIN_PHASE_A \
		if (a.seed_mode == 1) { \
			double* sd = a.seed_d + ((long)m*a.nwave + wv)*(4*K*64); int* si = a.seed_i + ((long)m*a.nwave + wv)*((2*K+1)*64); \
			_Pragma("unroll") for (int s = 0; s < K; s++) { \
				sd[s*64 + lane] = S.gp1[s]; sd[(K+s)*64 + lane] = S.gp2[s]; sd[(2*K+s)*64 + lane] = S.gm1[s]; sd[(3*K+s)*64 + lane] = S.gm2[s]; \
				si[s*64 + lane] = S.scp[s]; si[(K+s)*64 + lane] = S.scm[s]; } \
			if (lane == 0) si[2*K*64] = j; \
		} \
	}

// Phase A of the spin-0 kernels: no lane of the wave has reached scale 0 yet, so nothing is
// accumulated.  Four recurrence steps per iteration with the four coefficient rows fetched together;
// the rescale / activity test runs once per 4 steps (a chain grows by < 2^60 in 4 steps, far from
// overflow at 2^1024, and entering the accumulating phases a few steps late only drops terms < 2^-340).
#define S0_PHASE_A \
	while (k + 4 <= nk) { \
		bool act = false; \
		_Pragma("unroll") for (int s = 0; s < K; s++) act |= (sc[s] == 0 && lam2[s] != 0.0); \
		if (__any(act)) break; \
		const double4_t q0 = LDC(coef, k), q1 = LDC(coef, k+1), q2 = LDC(coef, k+2), q3 = LDC(coef, k+3); \
		const double b0 = polar ? q0.c : q0.b, b1 = polar ? q1.c : q1.b, b2 = polar ? q2.c : q2.b, b3 = polar ? q3.c : q3.b; \
		_Pragma("unroll") for (int s = 0; s < K; s++) { \
			lam1[s] = fma(fma(q0.a, csq[s], b0), lam2[s], lam1[s]); \
			lam2[s] = fma(fma(q1.a, csq[s], b1), lam1[s], lam2[s]); \
			lam1[s] = fma(fma(q2.a, csq[s], b2), lam2[s], lam1[s]); \
			lam2[s] = fma(fma(q3.a, csq[s], b3), lam1[s], lam2[s]); \
			if (sc[s] < 0 && fabs(lam2[s]) > SC_BIG) { lam1[s] *= SC_SMALL; lam2[s] *= SC_SMALL; sc[s]++; } \
		} \
		k += 4; \
	}

// Phase B history: it used to be a per-step loop with per-lane gating (cndmask) and a rescale test in every step:
// 207 instructions per step for leg_ana_spin<6> against 97 per step in the fast loop, ~18 % of the kernel time in a
// phase that covers ~8 % of the steps.  (Rejected before that: merging the gated steps into the fast pair loop behind
// a wave-uniform `if (pend)`: leg_syn 118 -> 172 ms at config 3; a branch-free pair-wise gated loop: VGPRs 124 -> 192.)
// Now phase B runs the ungated fast steps and only tests / rescales every 4 steps, see the kernels.

// Workgroups are ONE wave: lanes run in lockstep and a wave's LDS operations execute in order, so
// cross-lane visibility of the LDS tile only needs the LDS counter drained -- not an s_barrier, whose
// compiler-inserted s_waitcnt vmcnt(0) would also wait for the (slow, fire-and-forget) global store
// of the previous flush.
#ifdef PXS_HOST_SIM
#define PXS_WAVE_LDS_SYNC() __syncthreads()
#elif defined(PXS_LDS_NOWAIT)
#define PXS_WAVE_LDS_SYNC() asm volatile("" ::: "memory")
#else
#define PXS_WAVE_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif
// The sums over the rings of a wave (4 values per recurrence step) are collected for LEG_FSTEPS steps in an LDS tile and
// flushed together: output j = 4*step + row of the tile is then owned by lane j, which adds up its partial sums and
// contributes ONE value to a contiguous 512-byte store (or atomic add).  Measured on MI355X at config 3 (leg_ana per round
// trip, same box): no reduction at all 105.5 ms, lane swaps + LDS writes +22.7 ms, and the former flush (every 4 steps, 4
// lanes per output, two shuffles, 128-byte stores) +14 ms; this flush +3.3 ms.  What remains is the lane-swap stage: 6 swaps,
// 3 adds and a ds_write per step next to 48 FMAs, every one of them a 4-cycle VALU issue for a wave64 (10/58 = the measured
// share).  More ring pairs per lane would amortise it, but the kernels sit at the 3-waves-per-SIMD VGPR line already.
#define LEG_FSTEPS 16
#ifdef PXS_HOST_SIM
// simulator path: every lane writes its 4 sums, lane j adds row j over the 64 lanes
#define LEG_RED_STRIDE 66
#define LEG_RED_DOUBLES (4*LEG_FSTEPS*LEG_RED_STRIDE)
__device__ __forceinline__ double leg_flush_sum(const double* red, int lane) {
	const double* r = red + lane*LEG_RED_STRIDE;
	double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
	for (int i = 0; i < 64; i += 4) { s0 += r[i]; s1 += r[i+1]; s2 += r[i+2]; s3 += r[i+3]; }
	return (s0 + s1) + (s2 + s3);
}
__device__ __forceinline__ int leg_flush_col(int lane) { return lane; }
#define LEG_RED_PUT(kk, t0, t1, t2, t3) \
	red[((kk)*4+0)*LEG_RED_STRIDE + lane] = t0; red[((kk)*4+1)*LEG_RED_STRIDE + lane] = t1; \
	red[((kk)*4+2)*LEG_RED_STRIDE + lane] = t2; red[((kk)*4+3)*LEG_RED_STRIDE + lane] = t3;
#else
// MI355X path: reduce-scatter across lanes with the gfx950 lane-swap instructions.  Stage 1
// (v_permlane32_swap on the pairs (t0,t1), (t2,t3)) leaves sum(t0|t2) in lanes 0-31 and sum(t1|t3) in
// lanes 32-63; stage 2 (v_permlane16_swap) leaves ONE value per lane, already summed over the 4 lanes
// {l, l+16, l+32, l+48}: the four 16-lane rows of the wave hold t0, t2, t1, t3.  One ds_write_b64 per step (4x fewer
// LDS bytes than transposing all partial sums; the LDS write port was the limiter); row r of step kk goes to
// red[(4 kk + r)*18 .. +16], so that lane j = 4 kk + r reads its 16 partial sums as 8 aligned 16-byte words.
// (Tried and rejected: v_mfma_f64_4x4x4 with B = 1 as a lane adder: correct, but 8 dependent f64 MFMAs per step made the
// kernel matrix-pipe bound.  Round 3, with the lane layout from tools/mfma_probe.hip -- A at lane 16k+4b+i, B at 16k+4b+j, D at
// 16i+4b+j -- and B_r = [j == r]: four MFMAs accumulate the four sums of a step into ONE register, 4 issues + a ds_write instead
// of 9 VALU ops + a ds_write, 160 / 168 VGPRs: leg_ana_spin<4> 105.9 -> 112.9 ms, leg_ana_s0<8> 26.9 -> 31.0 ms at config 3: the
// f64 MFMA shares the FMA pipe's throughput on MI355X, it does not add to it. transposing all four sums through LDS: 145.1 against 142.1 ms.)
#define LEG_RED_STRIDE 18
#define LEG_RED_DOUBLES (4*LEG_FSTEPS*LEG_RED_STRIDE)
__device__ __forceinline__ void leg_swap32(double& a, double& b) {
	const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
	const auto r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
	const auto r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
	a = __hiloint2double(r1[0], r0[0]); b = __hiloint2double(r1[1], r0[1]);
}
__device__ __forceinline__ void leg_swap16(double& a, double& b) {
	const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
	const auto r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
	const auto r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
	a = __hiloint2double(r1[0], r0[0]); b = __hiloint2double(r1[1], r0[1]);
}
__device__ __forceinline__ double leg_flush_sum(const double* red, int lane) {
	const double2* r = reinterpret_cast<const double2*>(red + lane*LEG_RED_STRIDE);
	// two rounds of 4 loads, not unrolled: all 16 values at once cost 16-20 more VGPRs at the point where every chain is live
	// (leg_ana_s0<8> 174 VGPRs = 2 waves per SIMD instead of 3)
	double sum = 0;
#pragma unroll 1
	for (int h = 0; h < 8; h += 4) {
		const double2 a = r[h], b = r[h+1], c = r[h+2], d = r[h+3];
		sum += ((a.x + a.y) + (b.x + b.y)) + ((c.x + c.y) + (d.x + d.y));
	}
	return sum;
}
// lane j = 4 kk + r holds row r of step kk: rows are t0, t2, t1, t3
__device__ __forceinline__ int leg_flush_col(int lane) { const int r = lane & 3; return (lane & ~3) | ((r == 1) ? 2 : (r == 2) ? 1 : r); }
#ifdef PXS_EXP_NORED
#define LEG_RED_PUT(kk, t0, t1, t2, t3) { asm volatile("" :: "v"(t0), "v"(t1), "v"(t2), "v"(t3)); }     // timing experiment (wrong results)
#else
#define LEG_RED_PUT(kk, t0, t1, t2, t3) { \
	double a_ = t0, b_ = t1, c_ = t2, d_ = t3; \
	leg_swap32(a_, b_); leg_swap32(c_, d_); \
	double u_ = a_ + b_, v_ = c_ + d_; \
	leg_swap16(u_, v_); \
	red[((kk)*4 + (lane >> 4))*LEG_RED_STRIDE + (lane & 15)] = u_ + v_; }
#endif
#endif
// nkk steps of the tile -> dst[4 step + c] (c = 0..3: the sums t0..t3 of the step).  atomic: several waves add into the same
// rows (dst pre-zeroed); otherwise dst belongs to this wave alone
__device__ __forceinline__ void leg_flush(double* red, double* __restrict__ dst, int lane, int nkk, int atomic) {
#if defined(PXS_EXP_NORED) || defined(PXS_EXP_NOFLUSH)
	return;      // timing experiments (wrong results)
#endif
	PXS_WAVE_LDS_SYNC();
	if (lane < 4*nkk) {
		const double sum = leg_flush_sum(red, lane);
		double* q = dst + leg_flush_col(lane);
#ifdef PXS_HOST_SIM
		if (atomic) atomicAdd(q, sum); else *q = sum;
#else
		if (atomic) unsafeAtomicAdd(q, sum); else *q = sum;
#endif
	}
	PXS_WAVE_LDS_SYNC();
}


#define MM_PSTRIDE 65
#define MM_WAVES 8
#define MM_ESTRIDE 17
#ifdef PXS_HOST_SIM
struct mm_acc { double v[4]; double& operator[](int i) { return v[i]; } };
static inline mm_acc mm_mfma(double av, double bv, mm_acc c) {      // D[4r + lane/16][lane%16] += sum_kk A[i][kk] B[kk][j], A at lane i + 16 kk, B at lane j + 16 kk
	pxsim::BlockCtx* cx = pxsim::t_ctx; const int w = pxsim::wave_id(), l = pxsim::lane_id();
	uint64_t* s = cx->wslot->data() + (size_t)w*128;
	memcpy(&s[l], &av, 8); memcpy(&s[64 + l], &bv, 8); cx->wbar[w]->wait();
	for (int r = 0; r < 4; r++) {
		const int i = 4*r + (l >> 4), j = l & 15;
		double sum = c.v[r];
		for (int kk = 0; kk < 4; kk++) { double x, y; memcpy(&x, &s[i + 16*kk], 8); memcpy(&y, &s[64 + j + 16*kk], 8); sum = fma(x, y, sum); }
		c.v[r] = sum;
	}
	cx->wbar[w]->wait();
	return c;
}
static inline void mm_lds_add(double* p, double v) { atomicAdd(p, v); }
#define MM_WAVE_SYNC() pxsim::t_ctx->wbar[pxsim::wave_id()]->wait()
#else
typedef double mm_acc __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mm_acc mm_mfma(double av, double bv, mm_acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0); }
#ifdef PXS_LAB_NOLDSADD
__device__ __forceinline__ void mm_lds_add(double* p, double v) { *p = v; }      // timing experiment (wrong results)
#else
__device__ __forceinline__ void mm_lds_add(double* p, double v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#endif
#define MM_WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif
// step coefficient a x^2 + b' with both a and b' wave-uniform: gfx950 takes one scalar source per VALU op, so b' is copied to a VGPR
// right at its use (left to the compiler, the copies of all 16 steps of a tile were made early and lived in 64 VGPRs: spills)
__device__ __forceinline__ double mm_coef(double ca, double x2, double cb) { PXS_VCOPY(vb_, cb); return fma(ca, x2, vb_); }
// LDS: [W][16][MM_PSTRIDE] P tiles + [2][4 NG][64] reduction tiles (the staging area of the prologue, 64 W entries of MM_ESTRIDE doubles, lies over both)
__host__ __device__ constexpr int mm_lds_doubles(int NG, int W) { return W*16*MM_PSTRIDE + 2*NG*4*64 > 64*W*MM_ESTRIDE ? W*16*MM_PSTRIDE + 2*NG*4*64 : 64*W*MM_ESTRIDE; }
static inline size_t mm_ana_lds(int NG, int W) { return sizeof(double)*(size_t)mm_lds_doubles(NG, W) + 16; }

#define MMS_PSTRIDE 68
static inline size_t mm_syn_lds() { return sizeof(double)*16*MMS_PSTRIDE; }
#ifdef PXS_HOST_SIM
#define MMS_XOR2(v) __shfl_xor((v), 2)
#else
__device__ __forceinline__ double mms_xor2(double v) {      // value of lane ^ 2 (quad permute [2, 3, 0, 1])
	const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x4e, 0xf, 0xf, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x4e, 0xf, 0xf, true);
	return __hiloint2double(hi, lo);
}
#define MMS_XOR2(v) mms_xor2(v)
#endif

// launchers (defined next to the kernels; K = ring pairs per lane, ng = groups of 4 maps per wave / workgroup, W = waves per workgroup)
void launch_leg_syn_s0(int K, dim3 grid, hipStream_t st, const LegK& a);
void launch_leg_ana_s0(int K, dim3 grid, size_t lds, hipStream_t st, const LegK& a);
void launch_leg_syn_s0_mm(int ng, dim3 grid, hipStream_t st, const LegK& a);
void launch_leg_ana_s0_mm(int ng, int W, dim3 grid, hipStream_t st, const LegK& a);
void launch_leg_syn_spin(int K, dim3 grid, hipStream_t st, const LegK& a);
void launch_leg_ana_spin(int K, dim3 grid, size_t lds, hipStream_t st, const LegK& a);
void launch_leg_syn_spin_mm(int ng, dim3 grid, hipStream_t st, const LegK& a);
void launch_leg_ana_spin_mm(int ng, dim3 grid, hipStream_t st, const LegK& a);

} // namespace pxs
