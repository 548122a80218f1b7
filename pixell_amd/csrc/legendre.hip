// Legendre stage of the SHT for gfx950: alm <-> leg[comp][m][ring].
//
// Replaces ducc0's alm2leg / leg2alm (inside ducc0.sht.experimental.*; not in the reference
// tree) as reached from pixell/curvedsky.py:907-960, 1032-1084.
//
// Design (MI355X-first, plain FP64 FMA -- the contraction is 1-4 right-hand sides wide, too
// narrow for the 16x16x4 f64 MFMA):
//  * one wave64 per workgroup; a lane owns K ring PAIRS (theta, pi-theta) => K independent
//    recurrence chains of ILP per lane, north/south sharing one recurrence;
//  * spin 0: Ishioka-type two-step recurrence in x^2, p_{k+1} = (a_k x^2 + b_k) p_k + p_{k-1},
//    p_k ~ lambda_{m+2k+1,m}/x: 2 FMA of recurrence serve TWO degrees l, 4 FMA accumulate;
//    spin s: scaled three-term recurrence for (s)lambda and (-s)lambda, 4 FMA + 8 FMA per l;
//  * per-step coefficients (a,b) and the pre-scaled alm are wave-uniform: they are fetched with
//    scalar loads and fed to v_fma_f64 as SGPR operands, the recurrence state and the
//    accumulators stay in VGPRs for the whole l loop;
//  * extended exponent: lambda_mm ~ sin^m(theta) underflows for large m, so a chain starts as
//    (mantissa, scale) with value = mantissa * 2^(800*scale), is advanced without accumulating
//    until some lane of the wave reaches scale 0, then advanced with gated accumulation until all
//    lanes have, then runs the branch-free fast loop;
//  * analysis needs the sum over rings (lanes) for every l: the 4 sums of a step are reduce-scattered over the lanes with
//    v_permlane32_swap / v_permlane16_swap, one ds_write_b64 per step parks them in a tile of 16 steps; at a flush lane j owns
//    output j, adds its 16 partial sums and the wave issues one contiguous 512-byte global_atomic_add_f64 into the moments
//    (PXS_DETERMINISTIC=1: per-wave partial moments summed in wave order by reduce_partials instead).
#include "legendre.hpp"
#include <cmath>
#include <algorithm>
#include <thread>
#include <memory>

namespace pxs {

static constexpr double SC_BIG   = 0x1p+400;
static constexpr double SC_SMALL = 0x1p-800;
static constexpr int    SC_STEP  = 800;
// ring pairs per lane (K): defaults chosen by measurement on MI355X; PXS_K_SYN0/PXS_K_ANA0 (4|8) and
// PXS_K_SYNS/PXS_K_ANAS (2|3|4) override them for tuning runs
static int env_k(const char* name, int def, int lo, int hi) {
	const char* v = getenv(name); if (!v) return def;
	int k = atoi(v); return (k >= lo && k <= hi) ? k : def;
}
static int lab_k(const char* name, int def, int lo, int hi) {
	const char* v = lab_getenv(name); if (!v) return def;
	int k = atoi(v); return (k >= lo && k <= hi) ? k : def;
}
static int k_syn0() { static int k0 = lab_k("PXS_K_SYN0", 4, 2, 8); static int k = k0 >= 8 ? 8 : (k0 >= 4 ? 4 : 2); return k; }
static int k_ana0() { static int k0 = lab_k("PXS_K_ANA0", 8, 2, 12); static int k = k0 >= 12 ? 12 : (k0 >= 8 ? 8 : (k0 >= 4 ? 4 : 2)); return k; }
static int k_syns() { static int k = lab_k("PXS_K_SYNS", 3, 2, 4); return k; }
static int k_anas() { static int k = lab_k("PXS_K_ANAS", 4, 2, 6); return k; }   // 4: 149 VGPRs = 3 waves per SIMD (6: 227 = 2 waves; measured 146.9 vs 150.5 ms at config 3)
static int xcd_map() { static int k = lab_k("PXS_XCD_MAP", 1, 0, 1); return k; }

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

// ---------------------------------------------------------------------------------
// alm pre / post transforms
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double2 ld_alm(const void* alm, int dtype, long idx) {
	if (dtype == PX_C64) { float2 v = ((const float2*)alm)[idx]; return make_double2(v.x, v.y); }
	return ((const double2*)alm)[idx];
}
__device__ __forceinline__ void st_alm(void* alm, int dtype, long idx, double2 v) {
	if (dtype == PX_C64) ((float2*)alm)[idx] = make_float2((float)v.x, (float)v.y);
	else ((double2*)alm)[idx] = v;
}
__device__ __forceinline__ double eps_lm(int l, int m) {
	if (l <= m) return 0.0;
	double L = l, M = m;
	return sqrt((L*L - M*M)/(4.0*L*L - 1.0));
}

struct AlmK {
	int lmax, mmax, spin, deriv1, dtype;
	long nrows, cstride, lstride;
	const long* row; const double* alpha; const uint64_t* mstart;
	void* alm; double* almt; const double* mom;
	long alm_bs, almt_bs, mom_bs;      // batched calls: blockIdx.z = map; strides in alm elements / doubles
};
__device__ __forceinline__ void* alm_of_batch(const AlmK& a) { return (char*)a.alm + (size_t)(a.dtype == PX_C64 ? 8 : 16)*(size_t)blockIdx.z*(size_t)a.alm_bs; }

// Each thread takes ALM_U rows 256 apart and issues all their loads before the first store: with one row per thread a wave had 1 KB of
// loads in flight behind a chain of dependent scalar loads (alm_pre_spin 1.15 TB/s at config 3).
#define ALM_U 4
static inline dim3 alm_grid(int nrows, int nm, int nb) { return dim3((unsigned)((nrows + 256*ALM_U - 1)/(256*ALM_U)), (unsigned)nm, (unsigned)nb); }
// spin 0: almt[row(m)+k] = alpha_k * ( eps_{l+1} a_l + eps_{l+2} a_{l+2},  a_{l+1} ),  l = m+2k
__global__ __launch_bounds__(256) void alm_pre_s0(AlmK a) {
	const int m = blockIdx.y;
	const int nk = (a.lmax - m)/2 + 1;
	const int kb = blockIdx.x*(256*ALM_U) + threadIdx.x;
	if (kb >= nk) return;
	const long base = (long)a.mstart[m];
	const long row = a.row[m];
	const void* alm = alm_of_batch(a);
	double2 a0[ALM_U], a1[ALM_U], a2[ALM_U]; double al[ALM_U];
#pragma unroll
	for (int u = 0; u < ALM_U; u++) {
		const int k = kb + 256*u, l = m + 2*k;
		const bool ok = k < nk;
		a0[u] = ok ? ld_alm(alm, a.dtype, base + (long)l*a.lstride) : make_double2(0, 0);
		a1[u] = (ok && l+1 <= a.lmax) ? ld_alm(alm, a.dtype, base + (long)(l+1)*a.lstride) : make_double2(0, 0);
		a2[u] = (ok && l+2 <= a.lmax) ? ld_alm(alm, a.dtype, base + (long)(l+2)*a.lstride) : make_double2(0, 0);
		al[u] = ok ? a.alpha[row + k] : 0.0;
	}
#pragma unroll
	for (int u = 0; u < ALM_U; u++) {
		const int k = kb + 256*u, l = m + 2*k;
		if (k >= nk) break;
		const double e1 = eps_lm(l+1, m), e2 = eps_lm(l+2, m);
		double* o = a.almt + (long)blockIdx.z*a.almt_bs + 4*(row + k);
		o[0] = al[u]*(e1*a0[u].x + e2*a2[u].x); o[1] = al[u]*(e1*a0[u].y + e2*a2[u].y);
		o[2] = al[u]*a1[u].x;                   o[3] = al[u]*a1[u].y;
	}
}
// a_{m+2k} = eps_{l+1} alpha_k M1_k + eps_l alpha_{k-1} M1_{k-1};  a_{m+2k+1} = alpha_k M2_k
__global__ __launch_bounds__(256) void alm_post_s0(AlmK a) {
	const int m = blockIdx.y;
	const int nk = (a.lmax - m)/2 + 1;
	const int kb = blockIdx.x*(256*ALM_U) + threadIdx.x;
	if (kb >= nk) return;
	const long row = a.row[m];
	const long base = (long)a.mstart[m];
	void* alm = alm_of_batch(a);
	double M0[ALM_U], M1[ALM_U], M2[ALM_U], M3[ALM_U], P0[ALM_U], P1[ALM_U], al[ALM_U], alp[ALM_U];
#pragma unroll
	for (int u = 0; u < ALM_U; u++) {
		const int k = kb + 256*u;
		const bool ok = k < nk;
		const long r = row + k;
		const double* M = a.mom + (long)blockIdx.z*a.mom_bs + 4*r;
		M0[u] = ok ? M[0] : 0.0; M1[u] = ok ? M[1] : 0.0; M2[u] = ok ? M[2] : 0.0; M3[u] = ok ? M[3] : 0.0;
		al[u] = ok ? a.alpha[r] : 0.0;
		const bool prev = ok && k > 0;
		P0[u] = prev ? M[-4] : 0.0; P1[u] = prev ? M[-3] : 0.0; alp[u] = prev ? a.alpha[r-1] : 0.0;
	}
#pragma unroll
	for (int u = 0; u < ALM_U; u++) {
		const int k = kb + 256*u, l = m + 2*k;
		if (k >= nk) break;
		const double e1 = eps_lm(l+1, m), e0 = eps_lm(l, m);
		double2 v = make_double2(e1*al[u]*M0[u], e1*al[u]*M1[u]);
		if (k > 0) { v.x += e0*alp[u]*P0[u]; v.y += e0*alp[u]*P1[u]; }
		st_alm(alm, a.dtype, base + (long)l*a.lstride, v);
		if (l+1 <= a.lmax) st_alm(alm, a.dtype, base + (long)(l+1)*a.lstride, make_double2(al[u]*M2[u], al[u]*M3[u]));
	}
}
// spin s: rows l = l0..lmax; almt = beta_l * ( a+ = -(E+iB),  a- = -(-1)^s (E-iB) )
__global__ __launch_bounds__(256) void alm_pre_spin(AlmK a) {
	const int m = blockIdx.y;
	const int l0 = max(m, a.spin);
	const int lb = l0 + blockIdx.x*(256*ALM_U) + threadIdx.x;
	if (lb > a.lmax) return;
	const long base = (long)a.mstart[m], row = a.row[m];
	const void* alm = alm_of_batch(a);
	double2 E[ALM_U], B[ALM_U]; double be[ALM_U];
#pragma unroll
	for (int u = 0; u < ALM_U; u++) {
		const int l = lb + 256*u;
		const bool ok = l <= a.lmax;
		const long idx = base + (long)l*a.lstride;
		E[u] = ok ? ld_alm(alm, a.dtype, idx) : make_double2(0, 0);
		B[u] = (ok && !a.deriv1) ? ld_alm(alm, a.dtype, idx + a.cstride) : make_double2(0, 0);
		be[u] = ok ? a.alpha[row + (l - l0)] : 0.0;
	}
	const double sg = (a.spin & 1) ? -1.0 : 1.0;
#pragma unroll
	for (int u = 0; u < ALM_U; u++) {
		const int l = lb + 256*u;
		if (l > a.lmax) break;
		double2 e = E[u]; const double2 b = B[u];
		if (a.deriv1) { const double f = sqrt((double)l*(l+1.0)); e.x *= f; e.y *= f; }
		double* o = a.almt + (long)blockIdx.z*a.almt_bs + 4*(row + (l - l0));
		o[0] = -be[u]*(e.x - b.y); o[1] = -be[u]*(e.y + b.x);
		o[2] = -sg*be[u]*(e.x + b.y); o[3] = -sg*be[u]*(e.y - b.x);
	}
}
// E = -1/2 beta (mu+ + sg mu-),  B = i/2 beta (mu+ - sg mu-)
__global__ __launch_bounds__(256) void alm_post_spin(AlmK a) {
	const int m = blockIdx.y;
	const int l0 = max(m, a.spin);
	const int lb = blockIdx.x*(256*ALM_U) + threadIdx.x + min(m, l0);   // also zero-fill m <= l < l0
	if (lb > a.lmax) return;
	const long base = (long)a.mstart[m], row = a.row[m];
	void* alm = alm_of_batch(a);
	double M0[ALM_U], M1[ALM_U], M2[ALM_U], M3[ALM_U], be[ALM_U];
#pragma unroll
	for (int u = 0; u < ALM_U; u++) {
		const int l = lb + 256*u;
		const bool ok = l <= a.lmax && l >= l0;
		const long r = row + (l - l0);
		const double* M = a.mom + (long)blockIdx.z*a.mom_bs + 4*r;
		M0[u] = ok ? M[0] : 0.0; M1[u] = ok ? M[1] : 0.0; M2[u] = ok ? M[2] : 0.0; M3[u] = ok ? M[3] : 0.0;
		be[u] = ok ? a.alpha[r] : 0.0;
	}
	const double sg = (a.spin & 1) ? -1.0 : 1.0;
#pragma unroll
	for (int u = 0; u < ALM_U; u++) {
		const int l = lb + 256*u;
		if (l > a.lmax) break;
		const long idx = base + (long)l*a.lstride;
		// (rows m <= l < l0: be = 0 and the moments read as 0: E = B = 0)
		const double2 E = make_double2(-0.5*be[u]*(M0[u] + sg*M2[u]), -0.5*be[u]*(M1[u] + sg*M3[u]));
		const double2 B = make_double2(-0.5*be[u]*(M1[u] - sg*M3[u]),  0.5*be[u]*(M0[u] - sg*M2[u]));
		if (a.deriv1) { const double f = sqrt((double)l*(l+1.0)); st_alm(alm, a.dtype, idx, make_double2(f*E.x, f*E.y)); }
		else { st_alm(alm, a.dtype, idx, E); st_alm(alm, a.dtype, idx + a.cstride, B); }
	}
}

// mom[row] = sum over the waves that wrote that row.  A wave writes rows from the end of its phase A on and nothing at all
// when none of its rings is live for this m; `first` says which (36 % of the (wave, m) pairs are dead at config 3, and the
// 16 GiB partial buffer no longer needs a memset before every launch).
__global__ __launch_bounds__(256) void reduce_partials(const double* __restrict__ part, double* __restrict__ mom, const long* __restrict__ row,
		const int* __restrict__ first, int m0, int nmc, long rowbase, long rows_chunk, int nwave)
{
	const int mi = blockIdx.y;
	const long r0 = row[m0 + mi], nrow = row[m0 + mi + 1] - r0;
	const long i = (long)blockIdx.x*blockDim.x + threadIdx.x;      // 4*step + component
	if (i >= 4*nrow) return;
	const int step = (int)(i >> 2);
	const long off = (r0 - rowbase)*4 + i;
	double s = 0;
	for (int w = 0; w < nwave; w++) {
		const int f = first[w*nmc + mi];
		if (f > 0 && step >= f - 1) s += part[(long)w*rows_chunk*4 + off];
	}
	mom[r0*4 + i] = s;
}

// A wave is 'polar' when all its rings have cos^2 > PXS_POLAR_COS2: it then runs the recurrences in the variable
// -sin^2(theta) (spin 0) resp. -2 sin^2(theta/2) (spin s), with the constant term of the step
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
// step coefficient a x^2 + b' with both a and b' wave-uniform: gfx950 takes one scalar source per VALU op, so b' is copied to a VGPR
// right at its use (left to the compiler, the copies of all 16 steps of a tile were made early and lived in 64 VGPRs: spills)
__device__ __forceinline__ double mm_coef(double ca, double x2, double cb) { PXS_VCOPY(vb_, cb); return fma(ca, x2, vb_); }
// LDS: [W][16][MM_PSTRIDE] P tiles + [2][4 NG][64] reduction tiles (the staging area of the prologue, 64 W entries of MM_ESTRIDE doubles, lies over both)
__host__ __device__ constexpr int mm_lds_doubles(int NG, int W) { return W*16*MM_PSTRIDE + 2*NG*4*64 > 64*W*MM_ESTRIDE ? W*16*MM_PSTRIDE + 2*NG*4*64 : 64*W*MM_ESTRIDE; }
static inline size_t mm_ana_lds(int NG, int W) { return sizeof(double)*(size_t)mm_lds_doubles(NG, W) + 16; }

// compact step table: (a, b) or (a, a + b) of the rows of LegTables::coef; 32 rows of padding (a tile reads 16 steps whatever nk is)
__global__ __launch_bounds__(256) void coef2_kernel(const double4_t* __restrict__ coef, long nrows, double2* __restrict__ c2, double2* __restrict__ c2p) {
	const long i = (long)blockIdx.x*blockDim.x + threadIdx.x;
	if (i >= nrows + 32) return;
	if (i < nrows) { const double4_t c = coef[i]; c2[i] = make_double2(c.a, c.b); c2p[i] = make_double2(c.a, c.c); }
	else { c2[i] = make_double2(0, 0); c2p[i] = make_double2(0, 0); }
}

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

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
void LegProfile::begin(hipStream_t st, int stage) {
	if (!enabled) return;
	hipEvent_t e; PXS_HIP(hipEventCreate(&e)); PXS_HIP(hipEventRecord(e, st)); open_.push_back(e);
}
void LegProfile::end(hipStream_t st, int stage) {
	if (!enabled || open_.empty()) return;
	hipEvent_t e; PXS_HIP(hipEventCreate(&e)); PXS_HIP(hipEventRecord(e, st));
	recs.push_back(Rec{open_.back(), e, stage}); open_.pop_back();
}
void LegProfile::read(double* ms, int* counts, int nstage, bool reset) {
	for (int i = 0; i < nstage; i++) { ms[i] = 0; counts[i] = 0; }
	for (auto& r : recs) {
		PXS_HIP(hipEventSynchronize(r.b));
		float t = 0; PXS_HIP(hipEventElapsedTime(&t, r.a, r.b));
		if (r.stage >= 0 && r.stage < nstage) { ms[r.stage] += t; counts[r.stage]++; }
	}
	if (reset) { for (auto& r : recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); } recs.clear(); }
}
LegProfile::~LegProfile() { for (auto& r : recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); } for (auto e : open_) (void)hipEventDestroy(e); }

void RingSet::build(const std::vector<long double>& theta) {
	const long double PIl = 3.141592653589793238462643383279502884L;
	nring = (int)theta.size();
	std::vector<char> used(nring, 0);
	ring_n.clear(); ring_s.clear(); cth.clear(); sth.clear(); sh2.clear(); ch2.clear();
	// order pairs by colatitude of the northern member (pole first)
	std::vector<int> order(nring);
	for (int i = 0; i < nring; i++) order[i] = i;
	std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
		long double a = std::min(theta[x], PIl - theta[x]), b = std::min(theta[y], PIl - theta[y]); return a < b; });
	// fast path for symmetric ascending grids: partner of i is nring-1-i
	for (int oi = 0; oi < nring; oi++) {
		int i = order[oi];
		if (used[i]) continue;
		used[i] = 1;
		int partner = -1;
		int cand = nring-1-i;
		const long double tol = 1e-12L;
		if (cand != i && cand >= 0 && !used[cand] && fabsl(theta[cand] - (PIl - theta[i])) < tol) partner = cand;
		else {
			for (int j = 0; j < nring; j++) if (!used[j] && j != i && fabsl(theta[j] - (PIl - theta[i])) < tol) { partner = j; break; }
		}
		int in_ = i, is_ = partner;
		if (partner >= 0) { used[partner] = 1; if (theta[partner] < theta[i]) { in_ = partner; is_ = i; } }
		long double th = theta[in_];
		bool flipped = false;
		if (partner < 0 && th > PIl/2) { th = PIl - th; flipped = true; }   // lone southern ring: treat via its mirror
		if (flipped) { ring_n.push_back(-1); ring_s.push_back(in_); }
		else { ring_n.push_back(in_); ring_s.push_back(is_); }
		cth.push_back((double)cosl(th)); sth.push_back((double)sinl(th));
		sh2.push_back((double)sinl(th/2)); ch2.push_back((double)cosl(th/2));
	}
	npairs = (int)ring_n.size();
}
void RingSet::upload_all() {
	d_ring_n = upload(ring_n); d_ring_s = upload(ring_s); d_cth = upload(cth); d_sth = upload(sth);
	d_sh2 = upload(sh2); d_ch2 = upload(ch2);
}

void LegTables::build_host(int lmax_, int mmax_, int spin_) {
	lmax = lmax_; mmax = mmax_; spin = spin_;
	row.assign(mmax+2, 0);
	for (int m = 0; m <= mmax; m++) {
		long n = spin == 0 ? (lmax - m)/2 + 1 : std::max(0, lmax - std::max(m, spin) + 1);
		row[m+1] = row[m] + n;
	}
	nrows = row[mmax+1];
	// +4: the fast loops prefetch up to 3 rows ahead.  Uninitialised storage: every row is written below by the thread that owns its m
	// (a zero-filled std::vector costs a single-threaded first touch of 3 GB at lmax 10^4: more than the square roots)
	std::unique_ptr<double4_t[]> coef_(new double4_t[nrows + 4]); std::unique_ptr<double[]> alpha_(new double[nrows + 4]);
	double4_t* coef = coef_.get(); double* alpha = alpha_.get();
	for (long i = nrows; i < nrows + 4; i++) { coef[i] = double4_t{0, 0, 0, 0}; alpha[i] = 0.0; }
	typedef long double LDb;
	const LDb PIl = 3.141592653589793238462643383279502884L;
	// The rows of different m are independent (the sectoral start values are a short serial prefix): m is dealt cyclically to host
	// threads -- the table of a 21600-ring, lmax 10^4 plan is 7.5e7 rows of 4-6 long-double square roots, 8 s on one core.
	static const int nthr_env = [] { const char* e = getenv("PXS_TABLE_THREADS"); return e ? atoi(e) : 0; }();
	const int nthr = (int)std::max(1L, std::min<long>({(long)(nthr_env > 0 ? nthr_env : 32), (long)std::max(1u, std::thread::hardware_concurrency()), (long)mmax + 1, nrows/4096 + 1}));
	auto run_threads = [&](auto&& body) {      // body(m)
		if (nthr == 1) { for (int m = 0; m <= mmax; m++) body(m); return; }
		std::vector<std::thread> th;
		for (int t = 0; t < nthr; t++) th.emplace_back([&, t] { for (int m = t; m <= mmax; m += nthr) body(m); });
		for (auto& x : th) x.join();
	};
	if (spin == 0) {
		std::vector<LDb> cms(mmax+1);
		{ LDb cm = 1/sqrtl(4*PIl); for (int m = 0; m <= mmax; m++) { if (m > 0) cm = -cm*sqrtl((LDb)(2*m+1)/(LDb)(2*m)); cms[m] = cm; } }
		run_threads([&](int m) {
			const LDb cm = cms[m];
			auto eps = [&](int l) -> LDb { if (l <= m) return 0; LDb L = l, M = m; return sqrtl((L*L-M*M)/(4*L*L-1)); };
			const int nk = (lmax - m)/2 + 1;
			LDb a_prev = 0, a_cur = sqrtl((LDb)(2*m+3))*cm;   // alpha_{k-1}, alpha_k
			for (int k = 0; k < nk; k++) {
				const int lp = m + 2*k + 1;
				const LDb e2 = eps(lp+1)*eps(lp+1) + eps(lp)*eps(lp), f = eps(lp)*eps(lp-1), d = eps(lp+1)*eps(lp+2);
				const LDb a_next = (k == 0) ? a_cur/d : -f*a_prev/d;
				const LDb ak = a_cur/(a_next*d);
				coef[row[m]+k] = double4_t{(double)ak, (double)(-ak*e2), (double)(ak - ak*e2), 0.0};
				alpha[row[m]+k] = (double)a_cur;
				a_prev = a_cur; a_cur = a_next;
			}
		});
	} else {
		const int s = spin;
		// m < s: h(m) = (2s+1)(2s)!/((s+m)!(s-m)!) ; m >= s: c'_m^2 = g(m)/(4 pi 4^m)
		std::vector<LDb> nrm(mmax+1);
		{
			LDb h = 2*s+1; for (int i = 1; i <= s; i++) h = h*(LDb)(s+i)/(LDb)i;
			for (int m = 0; m < std::min(s, mmax+1); m++) { if (m > 0) h = h*(LDb)(s-m+1)/(LDb)(s+m); nrm[m] = sqrtl(h/(4*PIl)); }
			if (s <= mmax) {
				LDb c2 = (LDb)(2*s+1)/(4*PIl*powl(4.0L, s));
				nrm[s] = sqrtl(c2);
				for (int m = s+1; m <= mmax; m++) { c2 = c2*(LDb)(2*m+1)*(LDb)(2*m)/(4*(LDb)(m+s)*(LDb)(m-s)); nrm[m] = sqrtl(c2); }
			}
		}
		run_threads([&](int m) {
			const int l0 = std::max(m, s);
			const int nl = lmax - l0 + 1;
			if (nl <= 0) return;
			auto Sf = [&](int l) -> LDb { LDb L = l, M = m, Sp = s; return sqrtl((L*L-M*M)*(L*L-Sp*Sp)); };
			LDb b_prev = 0, b_cur = ((m & 1) ? -1 : 1)*nrm[m];
			for (int l = l0; l <= lmax; l++) {
				const LDb L = l;
				const LDb q = sqrtl((2*L+3)/(2*L+1))*(2*L+1);
				const LDb A = q*(L+1)/Sf(l+1);
				const LDb B = q*(LDb)m*(LDb)s/(L*Sf(l+1));
				const LDb C = (l > l0) ? sqrtl((2*L+3)/(2*L-1))*(L+1)*Sf(l)/(L*Sf(l+1)) : 0;
				const LDb b_next = (l == l0) ? A*b_cur : C*b_prev;
				{ const LDb ca = A*b_cur/b_next, cb = B*b_cur/b_next; coef[row[m]+(l-l0)] = double4_t{(double)ca, (double)cb, (double)(ca+cb), (double)(ca-cb)}; }
				alpha[row[m]+(l-l0)] = (double)b_cur;
				b_prev = b_cur; b_cur = b_next;
			}
		});
	}
	d_row = upload(row);
	d_coef.alloc(sizeof(double4_t)*(size_t)(nrows + 4)); PXS_HIP(hipMemcpy(d_coef.p, coef, d_coef.bytes, hipMemcpyHostToDevice));
	d_alpha.alloc(sizeof(double)*(size_t)(nrows + 4)); PXS_HIP(hipMemcpy(d_alpha.p, alpha, d_alpha.bytes, hipMemcpyHostToDevice));
}


// ---- recurrence tables on the GPU in double-double arithmetic (round 5) --------------------------------------------------------------
// The tables were built on host threads in long double (75 million rows of 4-6 square roots at lmax 10^4: 0.85 s of a 1.25 s cold
// call, followed by a 3 GB pageable upload).  The rows of one m are a serial recurrence, the m are independent: one GPU thread per m in
// double-double (106-bit) arithmetic builds the same rows in place in device memory in ~10 ms; the O(mmax) sectoral start values come from
// the host in long double as (hi, lo) pairs.  build_host (PXS_TABLES_HOST=1) remains as the reference the tests compare against.
// (error-free transforms: the compiler must not contract a product into a neighbouring add -- hipcc's default -ffp-contract=fast turned
// a - x*x into fma(-x, x, a) inside these and the device tables came out with 4-7e-15 errors instead of 2e-16, tools/tables_probe.hip)
#if defined(__clang__)
#define PXS_FP_STRICT _Pragma("clang fp contract(off)")
#else
#define PXS_FP_STRICT
#endif
struct dd { double h, l; };
__host__ __device__ __forceinline__ dd dd_norm(double a, double b) {
	PXS_FP_STRICT const double s = a + b; return dd{s, b - (s - a)}; }
__host__ __device__ __forceinline__ dd dd_of(double a) {
	PXS_FP_STRICT return dd{a, 0.0}; }
__host__ __device__ __forceinline__ dd dd_add(dd a, dd b) {
	PXS_FP_STRICT
	const double s = a.h + b.h, v = s - a.h, e = (a.h - (s - v)) + (b.h - v);
	return dd_norm(s, e + (a.l + b.l));
}
__host__ __device__ __forceinline__ dd dd_neg(dd a) {
	PXS_FP_STRICT return dd{-a.h, -a.l}; }
__host__ __device__ __forceinline__ dd dd_mul(dd a, dd b) {
	PXS_FP_STRICT
	const double p = a.h*b.h, e = fma(a.h, b.h, -p);
	return dd_norm(p, e + (a.h*b.l + a.l*b.h));
}
__host__ __device__ __forceinline__ dd dd_div(dd a, dd b) {
	PXS_FP_STRICT
	const double q1 = a.h/b.h;
	dd r = dd_add(a, dd_neg(dd_mul(dd_of(q1), b)));
	const double q2 = r.h/b.h;
	r = dd_add(r, dd_neg(dd_mul(dd_of(q2), b)));
	const double q3 = r.h/b.h;
	return dd_add(dd_norm(q1, q2), dd_of(q3));
}
__host__ __device__ __forceinline__ dd dd_sqrt(dd a) {
	PXS_FP_STRICT
	if (a.h <= 0.0) return dd{0.0, 0.0};
	const double x = sqrt(a.h);
	// one Newton step in double-double: x + (a - x^2) / (2 x)
	const double p = x*x, e = fma(x, x, -p);
	const dd r = dd_add(a, dd{-p, -e});
	return dd_norm(x, r.h/(2.0*x));
}
__host__ __device__ __forceinline__ double dd_val(dd a) {
	PXS_FP_STRICT return a.h + a.l; }

// spin 0: q(l) = eps_l^2 = (l^2 - m^2) / (4 l^2 - 1) (0 for l <= m); rows as in build_host
__global__ __launch_bounds__(64) void leg_tables_s0(int lmax, int mmax, const long* __restrict__ row, const dd* __restrict__ cms, double4_t* __restrict__ coef, double* __restrict__ alpha) {
	const int m = blockIdx.x*blockDim.x + threadIdx.x;
	if (m > mmax) return;
	auto q = [&](int l) -> dd { if (l <= m) return dd{0.0, 0.0}; const double L = l, M = m; return dd_div(dd_of(L*L - M*M), dd_of(4.0*L*L - 1.0)); };
	const int nk = (lmax - m)/2 + 1;
	dd a_prev = dd{0.0, 0.0}, a_cur = dd_mul(dd_sqrt(dd_of(2.0*m + 3.0)), cms[m]);
	// sliding window of q(lp - 1) ... q(lp + 2), lp = m + 2 k + 1
	dd qm1 = q(m), q0 = q(m + 1), qp1 = q(m + 2), qp2 = q(m + 3);
	const long r0 = row[m];
	for (int k = 0; k < nk; k++) {
		const dd e2 = dd_add(qp1, q0), f = dd_sqrt(dd_mul(q0, qm1)), d = dd_sqrt(dd_mul(qp1, qp2));
		const dd a_next = (k == 0) ? dd_div(a_cur, d) : dd_neg(dd_div(dd_mul(f, a_prev), d));
		const dd ak = dd_div(a_cur, dd_mul(a_next, d));
		const dd ake = dd_mul(ak, e2);
		coef[r0 + k] = double4_t{dd_val(ak), -dd_val(ake), dd_val(dd_add(ak, dd_neg(ake))), 0.0};
		alpha[r0 + k] = dd_val(a_cur);
		a_prev = a_cur; a_cur = a_next;
		const int lp = m + 2*k + 3;      // next step
		qm1 = qp1; q0 = qp2; qp1 = q(lp + 1); qp2 = q(lp + 2);
	}
}
// spin s: rows l = max(m, s) .. lmax as in build_host
__global__ __launch_bounds__(64) void leg_tables_spin(int lmax, int mmax, int s, const long* __restrict__ row, const dd* __restrict__ nrm, double4_t* __restrict__ coef, double* __restrict__ alpha) {
	const int m = blockIdx.x*blockDim.x + threadIdx.x;
	if (m > mmax) return;
	const int l0 = max(m, s), nl = lmax - l0 + 1;
	if (nl <= 0) return;
	auto Sf = [&](int l) -> dd { const double L = l, M = m, Sp = s; return dd_sqrt(dd_mul(dd_of(L*L - M*M), dd_of(L*L - Sp*Sp))); };
	dd b_prev = dd{0.0, 0.0}, b_cur = (m & 1) ? dd_neg(nrm[m]) : nrm[m];
	const long r0 = row[m];
	dd sf = Sf(l0);
	for (int l = l0; l <= lmax; l++) {
		const double L = l;
		const dd sf1 = Sf(l + 1);
		const dd qq = dd_mul(dd_sqrt(dd_div(dd_of(2*L + 3), dd_of(2*L + 1))), dd_of(2*L + 1));
		const dd A = dd_div(dd_mul(qq, dd_of(L + 1)), sf1);
		const dd B = dd_div(dd_mul(qq, dd_of((double)m*(double)s)), dd_mul(dd_of(L), sf1));
		dd b_next;
		if (l == l0) b_next = dd_mul(A, b_cur);
		else {
			const dd C = dd_div(dd_mul(dd_mul(dd_sqrt(dd_div(dd_of(2*L + 3), dd_of(2*L - 1))), dd_of(L + 1)), sf), dd_mul(dd_of(L), sf1));
			b_next = dd_mul(C, b_prev);
		}
		const dd ca = dd_div(dd_mul(A, b_cur), b_next), cb = dd_div(dd_mul(B, b_cur), b_next);
		coef[r0 + (l - l0)] = double4_t{dd_val(ca), dd_val(cb), dd_val(dd_add(ca, cb)), dd_val(dd_add(ca, dd_neg(cb)))};
		alpha[r0 + (l - l0)] = dd_val(b_cur);
		b_prev = b_cur; b_cur = b_next; sf = sf1;
	}
}

void LegTables::build(int lmax_, int mmax_, int spin_) {
	{ const char* e = getenv("PXS_TABLES_HOST"); if (e && atoi(e) != 0) { build_host(lmax_, mmax_, spin_); return; } }
	lmax = lmax_; mmax = mmax_; spin = spin_;
	row.assign(mmax+2, 0);
	for (int m = 0; m <= mmax; m++) {
		long n = spin == 0 ? (lmax - m)/2 + 1 : std::max(0, lmax - std::max(m, spin) + 1);
		row[m+1] = row[m] + n;
	}
	nrows = row[mmax+1];
	typedef long double LDb;
	const LDb PIl = 3.141592653589793238462643383279502884L;
	// sectoral start values (a serial product over m): host, long double, handed over as (hi, lo)
	std::vector<dd> start(mmax+1);
	auto put = [&](int m, LDb v) { const double h = (double)v; start[m] = dd{h, (double)(v - (LDb)h)}; };
	if (spin == 0) {
		LDb cm = 1/sqrtl(4*PIl);
		for (int m = 0; m <= mmax; m++) { if (m > 0) cm = -cm*sqrtl((LDb)(2*m+1)/(LDb)(2*m)); put(m, cm); }
	} else {
		const int s = spin;
		LDb h = 2*s+1; for (int i = 1; i <= s; i++) h = h*(LDb)(s+i)/(LDb)i;
		for (int m = 0; m < std::min(s, mmax+1); m++) { if (m > 0) h = h*(LDb)(s-m+1)/(LDb)(s+m); put(m, sqrtl(h/(4*PIl))); }
		if (s <= mmax) {
			LDb c2 = (LDb)(2*s+1)/(4*PIl*powl(4.0L, s));
			put(s, sqrtl(c2));
			for (int m = s+1; m <= mmax; m++) { c2 = c2*(LDb)(2*m+1)*(LDb)(2*m)/(4*(LDb)(m+s)*(LDb)(m-s)); put(m, sqrtl(c2)); }
		}
	}
	d_row = upload(row);
	DevBuf d_start = upload(start);
	d_coef.alloc(sizeof(double4_t)*(size_t)(nrows + 4)); d_alpha.alloc(sizeof(double)*(size_t)(nrows + 4));
	// (+4 rows: the fast loops prefetch up to 3 rows ahead)
	PXS_HIP(hipMemset((char*)d_coef.p + sizeof(double4_t)*(size_t)nrows, 0, sizeof(double4_t)*4)); PXS_HIP(hipMemset((char*)d_alpha.p + sizeof(double)*(size_t)nrows, 0, sizeof(double)*4));
	const dim3 grid((unsigned)((mmax + 1 + 63)/64));
	if (spin == 0) hipLaunchKernelGGL(leg_tables_s0, grid, dim3(64), 0, (hipStream_t)0, lmax, mmax, d_row.as<long>(), d_start.as<dd>(), d_coef.as<double4_t>(), d_alpha.as<double>());
	else           hipLaunchKernelGGL(leg_tables_spin, grid, dim3(64), 0, (hipStream_t)0, lmax, mmax, spin, d_row.as<long>(), d_start.as<dd>(), d_coef.as<double4_t>(), d_alpha.as<double>());
	PXS_HIP(hipGetLastError());
	PXS_HIP(hipDeviceSynchronize());      // (d_start goes out of scope; the tables are plan state, built once)
}

// doubles per map of the pre-scaled alm / the moments of a batched call
static long leg_almt_stride(const LegTables& tb) { return 4*(tb.nrows + 4); }
static long leg_mom_stride(const LegTables& tb) { return 4*std::max<long>(tb.nrows, 1); }
static LegK make_legk(const RingSet& rs, const LegTables& tb, LegWork& wk, double2* leg, long ld, int K, int nb = 1, long leg_bs = 0) {
	LegK a; memset(&a, 0, sizeof(a));
	a.lmax = tb.lmax; a.mmax = tb.mmax; a.spin = tb.spin; a.nm = tb.mmax+1; a.npairs = rs.npairs; a.nring = rs.nring;
	a.nwave = (rs.npairs + 64*K - 1)/(64*K);
	a.nrows = tb.nrows; a.row = tb.d_row.as<long>(); a.coef = tb.d_coef.as<double4_t>(); a.alpha = tb.d_alpha.as<double>();
	a.ring_n = rs.d_ring_n.as<int>(); a.ring_s = rs.d_ring_s.as<int>(); a.cth = rs.d_cth.as<double>(); a.sth = rs.d_sth.as<double>();
	a.sh2 = rs.d_sh2.as<double>(); a.ch2 = rs.d_ch2.as<double>();
	a.almt = wk.almt.as<double>(); a.part = wk.part.as<double>(); a.mom = wk.mom.as<double>();
	a.leg = leg; a.ld = ld > 0 ? ld : rs.nring;
	a.ofs = std::max(100.0, 0.01*tb.lmax);
	a.nmc = a.nm; a.xcd = xcd_map();
	a.count = wk.count_on ? wk.count.as<double>() : nullptr;
	a.nb = nb; a.leg_bs = leg_bs; a.almt_bs = leg_almt_stride(tb); a.mom_bs = leg_mom_stride(tb);
	PXS_REQUIRE((long)8*((a.nm + 7)/8)*a.nwave*nb < (1L << 31), "internal: Legendre grid too large for one launch");
	return a;
}

// maps one launch can take: all maps of a launch share one grid of 8 ceil(nm / 8) nwave blocks each (a large batch of small-ring,
// high-lmax maps goes out as several launches instead of tripping make_legk's grid check)
static int leg_max_batch(const RingSet& rs, const LegTables& tb, int K) {
	const long nwave = (rs.npairs + 64L*K - 1)/(64L*K), per = 8L*((tb.mmax + 1 + 7)/8)*std::max<long>(nwave, 1);
	return (int)std::max<long>(1, std::min<long>(1 << 20, ((1L << 31) - 1)/per));
}
// seeds of (ring set, spin, direction, K): allocate on first use if the plan's budget allows; returns the mode for this launch
static LegWork::Seeds* seeds_for(LegWork& wk, const RingSet& rs, const LegTables& tb, int dir, int K, LegK& a) {
	a.seed_mode = 0; a.seed_d = nullptr; a.seed_i = nullptr;
	{ const char* e = getenv("PXS_SEED_GB"); if (e) wk.seed_budget = (size_t)atol(e) << 30; }
	if (wk.seed_budget == 0) return nullptr;
	// loading a seed costs 20 / 40 bytes per lane and m, running phase A ~0.09 lmax recurrence steps: measured on MI355X the seeds
	// win at lmax 10^4 (C3 341.8 -> 327.6 ms per round trip) and for spin 2 at lmax 4000 (C2 28.7 -> 28.0 ms), and lose for spin 0
	// at lmax 4000 (C4: leg_syn 109.1 -> 113.9 ms per 64 maps)
	{	const char* e = getenv("PXS_SEED_MIN_LMAX");
		const int lmin = e ? atoi(e) : (tb.spin == 0 ? 6000 : 3000);
		if (tb.lmax < lmin) return nullptr; }
	LegWork::Seeds& sb = wk.seeds[std::make_tuple((const void*)&rs, tb.spin, dir, K)];
	if (sb.refused) return nullptr;
	const int nd = tb.spin == 0 ? 2 : 4, ni = tb.spin == 0 ? 1 : 2;
	const size_t slots = (size_t)a.nm*a.nwave;
	if (!sb.d.p) {
		const size_t bd = sizeof(double)*slots*nd*K*64, bi = sizeof(int)*slots*(ni*K + 1)*64;
		if (wk.seed_bytes + bd + bi > wk.seed_budget) { sb.refused = true; return nullptr; }
		sb.d.alloc(bd); sb.i.alloc(bi); wk.seed_bytes += bd + bi;
	}
	a.seed_d = sb.d.as<double>(); a.seed_i = sb.i.as<int>(); a.seed_mode = sb.ready ? 2 : 1;
	return &sb;
}
// after the recording launch: later launches on OTHER streams wait for it (a plan serves one call at a time, but the next call may
// come on another stream)
static void seeds_written(LegWork::Seeds* sb, hipStream_t st) {
	if (!sb || sb->ready) return;
	if (!sb->written) PXS_HIP(hipEventCreateWithFlags(&sb->written, hipEventDisableTiming));
	PXS_HIP(hipEventRecord(sb->written, st)); sb->wstream = st; sb->ready = true;
}
static void seeds_wait(LegWork::Seeds* sb, hipStream_t st) {
	if (sb && sb->ready && sb->written && st != sb->wstream) PXS_HIP(hipStreamWaitEvent(st, sb->written, 0));
}

// will the next launch of this kind record seeds (first use on a plan, budget permitting)?
static bool seeds_pending(LegWork& wk, const RingSet& rs, const LegTables& tb, int dir, int K) {
	LegK probe; memset(&probe, 0, sizeof(probe)); probe.nm = tb.mmax + 1; probe.nwave = (rs.npairs + 64*K - 1)/(64*K);
	LegWork::Seeds* sb = seeds_for(wk, rs, tb, dir, K, probe);
	return sb != nullptr && !sb->ready;
}

static AlmK make_almk(const LegTables& tb, LegWork& wk, const void* alm, int dtype, long cstride, const uint64_t* d_mstart, long lstride, int deriv1, long alm_bs) {
	AlmK k; memset(&k, 0, sizeof(k));
	k.alm_bs = alm_bs; k.almt_bs = leg_almt_stride(tb); k.mom_bs = leg_mom_stride(tb);
	k.lmax = tb.lmax; k.mmax = tb.mmax; k.spin = tb.spin; k.deriv1 = deriv1; k.dtype = dtype;
	k.nrows = tb.nrows; k.cstride = cstride; k.lstride = lstride; k.row = tb.d_row.as<long>(); k.alpha = tb.d_alpha.as<double>();
	k.mstart = d_mstart; k.alm = const_cast<void*>(alm); k.almt = wk.almt.as<double>(); k.mom = wk.mom.as<double>();
	return k;
}

static void ensure_coef2(hipStream_t st, const LegTables& tb) {      // compact step table (a, b) / (a, a + b), built by the first batched transform on the plan
	if (tb.d_coef2.p) return;
	tb.d_coef2.alloc(sizeof(double2)*(size_t)(tb.nrows + 32)); tb.d_coef2p.alloc(sizeof(double2)*(size_t)(tb.nrows + 32));
	hipLaunchKernelGGL(coef2_kernel, dim3((unsigned)((tb.nrows + 32 + 255)/256)), dim3(256), 0, st, tb.d_coef.as<double4_t>(), tb.nrows, tb.d_coef2.as<double2>(), tb.d_coef2p.as<double2>());
	// the table is plan state that later calls read on THEIR streams with no dependency on this launch: it is complete before the pointer is
	// handed out (once per plan and spin; the seeds order themselves with events, LegTables::build synchronises the device likewise)
	PXS_HIP(hipStreamSynchronize(st));
}
// Rings per lane of a small ring set.  (1) A wave of 64 K ring pairs takes the polar form of the recurrences (leg_wave_polar) only if its most equatorial ring
// stays within 71.5 degrees of the pole, so on a grid of a few hundred rings the default K leaves the rings next to the poles in the plain form (l^2 eps there):
// where a smaller compiled K makes the first wave eligible, take it.  (2) Up to 512 ring pairs (lmax ~1000) the launch is short of waves, not of work per wave
// -- (mmax + 1) x ceil(npairs / 64 K) waves for 1024 SIMDs -- and the smallest K is the fastest (tools/ksmall_ab.sh, profiles/r05_k_small_grids.txt: the reference's
// benchmark shape 900x1800, lmax 750: 0.365 -> 0.335 ms per round trip, its T/Q/U version 1.047 -> 0.938; at lmax 1500 the defaults are level, at 2500 ahead).
static int k_small_grid(const RingSet& rs, int kdef, std::initializer_list<int> smaller, int kmid = 0) {
#ifdef PXS_HOST_SIM
	const bool off = [] { const char* e = getenv("PXS_K_SMALL_OFF"); return e && atoi(e) != 0; }();      // (the host simulation runs small grids only: its tests switch the rule off to reach the default kernels)
#else
	static const bool off = [] { const char* e = lab_getenv("PXS_K_SMALL_OFF"); return e && atoi(e) != 0; }();
#endif
	if (off || rs.npairs <= 0) return kdef;
	if (rs.npairs <= 512) { int k = kdef; for (int c : smaller) k = std::min(k, c); return k; }
	// (3) up to 1400 ring pairs (lmax ~2500: 2700 rings) the synthesis kernels are still 5-12 % faster with K = 2 and the scalar analysis 2-4 % with K = 4; at lmax 4000 the defaults
	// are 20 % ahead (tools/kmid_ab.sh, tools/kbig_ab.sh, profiles/r05_k_mid_grids.txt)
	if (kmid > 0 && rs.npairs <= 1400) return kmid;
	auto eligible = [&](int k) { const int last = std::min(64*k, rs.npairs) - 1; return last >= 0 && rs.cth[last]*rs.cth[last] > PXS_POLAR_COS2; };
	if (eligible(kdef)) return kdef;
	for (int k : smaller) if (k < kdef && eligible(k)) return k;
	return kdef;
}
static int syn_mm_min() { static int v = [] { const char* e = getenv("PXS_SYN_MM_MIN"); const int x = e ? atoi(e) : 4; return x <= 0 ? (1 << 30) : std::max(2, x); }(); return v; }
void leg_synthesis(hipStream_t st, const RingSet& rs, const LegTables& tb, LegWork& wk,
                   const void* alm, int alm_dtype, long alm_cstride, const uint64_t* d_mstart, long lstride,
                   double2* leg, int deriv1, LegProfile* prof, long ld, int nb, long alm_bstride, long leg_bstride)
{
	PXS_REQUIRE(alm_dtype == PX_C64 || alm_dtype == PX_C128, "alm must be complex64 or complex128");
	PXS_REQUIRE(nb >= 1, "leg_synthesis: nb must be >= 1");
	wk.almt.ensure(sizeof(double)*(size_t)leg_almt_stride(tb)*nb);
	AlmK ak = make_almk(tb, wk, alm, alm_dtype, alm_cstride, d_mstart, lstride, deriv1, alm_bstride);
	const int nm = tb.mmax+1;
	const int K = tb.spin == 0 ? k_small_grid(rs, k_syn0(), {2}, 2) : k_small_grid(rs, k_syns(), {2}, 2);
	if (tb.spin == 0) hipLaunchKernelGGL(alm_pre_s0, alm_grid(tb.lmax/2 + 1, nm, nb), dim3(256), 0, st, ak);
	else              hipLaunchKernelGGL(alm_pre_spin, alm_grid(tb.lmax + 1, nm, nb), dim3(256), 0, st, ak);
	// maps [b0, b0 + n) in one launch
	auto launch = [&](int b0, int n) {
		LegK a = make_legk(rs, tb, wk, leg + (size_t)b0*leg_bstride, ld, K, n, leg_bstride);
		a.almt += (size_t)b0*a.almt_bs;
		LegWork::Seeds* sb = seeds_for(wk, rs, tb, 0, K, a); seeds_wait(sb, st);
		if (prof) prof->begin(st, 0);
		if (tb.spin == 0) {
			// (ring pairs per lane the product's rules select: 4 and 2; lab builds -- PXS_K_* -- compile the others too)
#ifdef PXS_LAB
			if (K == 8)      hipLaunchKernelGGL(leg_syn_s0<8>, leg_grid(a), dim3(64), 0, st, a); else
#endif
			if (K == 2) hipLaunchKernelGGL(leg_syn_s0<2>, leg_grid(a), dim3(64), 0, st, a);
			else             hipLaunchKernelGGL(leg_syn_s0<4>, leg_grid(a), dim3(64), 0, st, a);
		} else {
#ifdef PXS_LAB
			if (K == 4)      hipLaunchKernelGGL(leg_syn_spin<4>, leg_grid(a), dim3(64), 0, st, a); else
#endif
			if (K == 3) hipLaunchKernelGGL(leg_syn_spin<3>, leg_grid(a), dim3(64), 0, st, a);
			else             hipLaunchKernelGGL(leg_syn_spin<2>, leg_grid(a), dim3(64), 0, st, a);
		}
		if (prof) prof->end(st, 0);
		seeds_written(sb, st);
	};
	int b0 = 0;
	// spin 0, 4 or more maps (PXS_SYN_MM_MIN): the FP64-MFMA form (leg_syn_s0_mm), 8 maps per wave; a remainder of <= 4 maps in 4-map waves,
	// a single left-over map through the VALU kernel.  Each map's result equals its single-map call to rounding, not bit for bit.
	if (tb.spin == 0 && nb >= syn_mm_min()) {
		int nmm = nb; if (nb % 8 == 1) nmm = nb - 1;
		ensure_coef2(st, tb);
		auto launch_mm = [&](int m0, int nmaps, int ng) {
			const int per = 4*ng, ngroups = (nmaps + per - 1)/per;
			LegK a = make_legk(rs, tb, wk, leg + (size_t)m0*leg_bstride, ld, 1, ngroups, leg_bstride);
			a.almt += (size_t)m0*a.almt_bs; a.nmaps = nmaps; a.coef2 = tb.d_coef2.as<double2>(); a.coef2p = tb.d_coef2p.as<double2>();
			if (prof) prof->begin(st, 0);
			if (ng == 2) hipLaunchKernelGGL(leg_syn_s0_mm<2>, leg_grid(a), dim3(64), mm_syn_lds(), st, a);
			else         hipLaunchKernelGGL(leg_syn_s0_mm<1>, leg_grid(a), dim3(64), mm_syn_lds(), st, a);
			if (prof) prof->end(st, 0);
		};
		static const int ngmax = lab_k("PXS_SYN_MM_NG", 2, 1, 2);      // groups of 4 maps per wave (tuning)
		const int mper = 4*ngmax, gmax = std::max(1, leg_max_batch(rs, tb, 1)), r = nmm % mper, n8 = nmm - r;
		for (int m0 = 0; m0 < n8; m0 += mper*gmax) launch_mm(m0, std::min(mper*gmax, n8 - m0), ngmax);
		if (r > 4) launch_mm(n8, r, 2); else if (r > 0) launch_mm(n8, r, 1);
		b0 = nmm;
	}
	// spin s, 4 or more maps (Q/U pairs of a stack of maps): leg_syn_spin_mm, 8 maps per wave, a remainder of <= 4 in 4-map waves, a lone left-over map
	// through the VALU kernel
	if (tb.spin > 0 && nb >= syn_mm_min()) {
		const int nmm = nb - (nb % 8 == 1 ? 1 : 0);
		ensure_coef2(st, tb);
		auto launch_mm = [&](int m0, int nmaps, int ng) {
			const int per = 4*ng, ngroups = (nmaps + per - 1)/per;
			LegK a = make_legk(rs, tb, wk, leg + (size_t)m0*leg_bstride, ld, 1, ngroups, leg_bstride);
			a.nwave = (rs.npairs + 31)/32;      // (a wave covers 32 ring pairs: one chain per half-wave)
			a.almt += (size_t)m0*a.almt_bs; a.nmaps = nmaps; a.coef2 = tb.d_coef2.as<double2>(); a.coef2p = tb.d_coef2p.as<double2>();
			PXS_REQUIRE((long)8*((a.nm + 7)/8)*a.nwave*ngroups < (1L << 31), "internal: Legendre grid too large for one launch");
			if (prof) prof->begin(st, 0);
			if (ng == 2) hipLaunchKernelGGL(leg_syn_spin_mm<2>, leg_grid(a), dim3(64), mm_syn_lds(), st, a);
			else         hipLaunchKernelGGL(leg_syn_spin_mm<1>, leg_grid(a), dim3(64), mm_syn_lds(), st, a);
			if (prof) prof->end(st, 0);
		};
		const int gmax = std::max(1, leg_max_batch(rs, tb, 1)/2), r = nmm % 8, n8 = nmm - r;
		for (int m0 = 0; m0 < n8; m0 += 8*gmax) launch_mm(m0, std::min(8*gmax, n8 - m0), 2);
		if (r > 4) launch_mm(n8, r, 2); else if (r > 0) launch_mm(n8, r, 1);
		b0 = nmm;
	}
	// (the launch that records the recurrence seeds takes one map, so that only one wave writes each seed)
	if (nb - b0 > 1 && seeds_pending(wk, rs, tb, 0, K)) { launch(b0, 1); b0 += 1; }
	for (const int nmax = leg_max_batch(rs, tb, K); b0 < nb; b0 += nmax) launch(b0, std::min(nmax, nb - b0));
	PXS_HIP(hipGetLastError());
}

// spin-0 analysis of nb >= PXS_ANA_MM_MIN (4) maps: the FP64-MFMA form (leg_ana_s0_mm), 8 maps per workgroup; a remainder of <= 4 maps
// takes 4-map workgroups, a single left-over map the VALU kernel.  Summation order differs from the single-map kernel: a batched
// call equals its single-map calls to rounding (1e-13), not bit for bit.
static int ana_mm_min() { static int v = [] { const char* e = getenv("PXS_ANA_MM_MIN"); const int x = e ? atoi(e) : 4; return x <= 0 ? (1 << 30) : std::max(2, x); }(); return v; }
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
static void leg_analysis_mm(hipStream_t st, const RingSet& rs, const LegTables& tb, LegWork& wk,
                  const double2* leg, void* alm, int alm_dtype, long alm_cstride, const uint64_t* d_mstart, long lstride,
                  LegProfile* prof, long ld, int nb, long alm_bstride, long leg_bstride)
{
	static const int W = [] { const int v = lab_k("PXS_ANA_MM_W", MM_WAVES, 2, 16); return v >= 16 ? 16 : (v >= 8 ? 8 : (v >= 4 ? 4 : 2)); }();      // waves per workgroup: 8 (lab builds: 2 | 4 | 8 | 16)
	const int nm = tb.mmax+1;
	const long n4 = leg_mom_stride(tb);
	const size_t aesz = alm_dtype == PX_C64 ? 8 : 16;
	int nmm = nb;
	if (nb % 8 == 1) nmm = nb - 1;      // (a lone map in a 4-map workgroup costs more than the VALU kernel)
	wk.mom.ensure(sizeof(double)*(size_t)n4*nmm);
	PXS_HIP(hipMemsetAsync(wk.mom.p, 0, sizeof(double)*(size_t)n4*nmm, st));
	ensure_coef2(st, tb);
	auto launch = [&](int b0, int nmaps, int ng) {      // maps [b0, b0 + nmaps) in groups of 4 ng
		const int per = 4*ng, ngroups = (nmaps + per - 1)/per;
		LegK a = make_legk(rs, tb, wk, const_cast<double2*>(leg) + (size_t)b0*leg_bstride, ld, W, ngroups, leg_bstride);
		a.mom = wk.mom.as<double>() + (size_t)b0*a.mom_bs; a.part = a.mom; a.atomic = 1; a.nmaps = nmaps;
		a.coef2 = tb.d_coef2.as<double2>(); a.coef2p = tb.d_coef2p.as<double2>();
		if (prof) prof->begin(st, 1);
		const dim3 grid = leg_grid(a);
#ifdef PXS_LAB      /* (lab builds: other workgroup sizes, PXS_ANA_MM_W; measured at C4: 2 waves 95 ms, 4: 78, 8: 77, 16: 82 per 64 maps) */
		if (W == 16)     mm_launch<16>(ng, grid, st, a);
		else if (W == 4) mm_launch<4>(ng, grid, st, a);
		else if (W == 2) mm_launch<2>(ng, grid, st, a);
		else
#endif
		mm_launch<8>(ng, grid, st, a);
		if (prof) prof->end(st, 1);
	};
	const int gmax = std::max(1, leg_max_batch(rs, tb, W));      // groups one launch can take (grid limit)
	static const int ngmax = lab_k("PXS_ANA_MM_NG", 2, 1, 2);      // groups of 4 maps per workgroup (tuning)
	const int mper = 4*ngmax, r = nmm % mper, n8 = nmm - r;
	for (int b0 = 0; b0 < n8; b0 += mper*gmax) launch(b0, std::min(mper*gmax, n8 - b0), ngmax);
	if (r > 4) launch(n8, r, 2); else if (r > 0) launch(n8, r, 1);
	AlmK ak = make_almk(tb, wk, alm, alm_dtype, alm_cstride, d_mstart, lstride, 0, alm_bstride);
	hipLaunchKernelGGL(alm_post_s0, alm_grid(tb.lmax/2 + 1, nm, nmm), dim3(256), 0, st, ak);
	PXS_HIP(hipGetLastError());
	if (nmm < nb)
		leg_analysis(st, rs, tb, wk, leg + (size_t)nmm*leg_bstride, (char*)alm + aesz*(size_t)nmm*alm_bstride, alm_dtype, alm_cstride, d_mstart, lstride, 0, prof, ld, 1, 0, 0);
}

// spin-s analysis of nb >= PXS_ANA_MM_MIN (4) maps: leg_ana_spin_mm, 8 maps (Q/U pairs) per workgroup, a remainder of <= 4 in 4-map workgroups, a lone
// left-over map through the VALU kernel
static void leg_analysis_spin_mm(hipStream_t st, const RingSet& rs, const LegTables& tb, LegWork& wk,
                  const double2* leg, void* alm, int alm_dtype, long alm_cstride, const uint64_t* d_mstart, long lstride,
                  LegProfile* prof, long ld, int nb, long alm_bstride, long leg_bstride)
{
	constexpr int W = MM_WAVES;
	const int nm = tb.mmax+1;
	const long n4 = leg_mom_stride(tb);
	const size_t aesz = alm_dtype == PX_C64 ? 8 : 16;
	const int nmm = nb - (nb % 8 == 1 ? 1 : 0);      // (a lone left-over map costs more in a 4-map workgroup than in the VALU kernel)
	wk.mom.ensure(sizeof(double)*(size_t)n4*nmm);
	PXS_HIP(hipMemsetAsync(wk.mom.p, 0, sizeof(double)*(size_t)n4*nmm, st));
	ensure_coef2(st, tb);
	static const bool once = [] { (void)hipFuncSetAttribute((const void*)leg_ana_spin_mm<2, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 256);
		(void)hipFuncSetAttribute((const void*)leg_ana_spin_mm<1, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 256); return true; }(); (void)once;
	auto launch = [&](int b0, int nmaps, int ng) {
		const int per = 4*ng, ngroups = (nmaps + per - 1)/per;
		// (a workgroup covers 32 W ring pairs: nwave of the block mapping counts those)
		LegK a = make_legk(rs, tb, wk, const_cast<double2*>(leg) + (size_t)b0*leg_bstride, ld, 1, ngroups, leg_bstride);
		a.nwave = (rs.npairs + 32*W - 1)/(32*W);
		a.mom = wk.mom.as<double>() + (size_t)b0*a.mom_bs; a.part = a.mom; a.atomic = 1; a.nmaps = nmaps;
		a.coef2 = tb.d_coef2.as<double2>(); a.coef2p = tb.d_coef2p.as<double2>();
		if (prof) prof->begin(st, 1);
		if (ng == 2) hipLaunchKernelGGL((leg_ana_spin_mm<2, W>), leg_grid(a), dim3(64*W), mm_ana_lds(2, W), st, a);
		else         hipLaunchKernelGGL((leg_ana_spin_mm<1, W>), leg_grid(a), dim3(64*W), mm_ana_lds(1, W), st, a);
		if (prof) prof->end(st, 1);
	};
	const int gmax = std::max(1, leg_max_batch(rs, tb, 2)), r = nmm % 8, n8 = nmm - r;
	for (int b0 = 0; b0 < n8; b0 += 8*gmax) launch(b0, std::min(8*gmax, n8 - b0), 2);
	if (r > 4) launch(n8, r, 2); else if (r > 0) launch(n8, r, 1);
	AlmK ak = make_almk(tb, wk, alm, alm_dtype, alm_cstride, d_mstart, lstride, 0, alm_bstride);
	hipLaunchKernelGGL(alm_post_spin, alm_grid(tb.lmax + 1, nm, nmm), dim3(256), 0, st, ak);
	PXS_HIP(hipGetLastError());
	if (nmm < nb)
		leg_analysis(st, rs, tb, wk, leg + (size_t)nmm*leg_bstride, (char*)alm + aesz*(size_t)nmm*alm_bstride, alm_dtype, alm_cstride, d_mstart, lstride, 0, prof, ld, 1, 0, 0);
}

void leg_analysis(hipStream_t st, const RingSet& rs, const LegTables& tb, LegWork& wk,
                  const double2* leg, void* alm, int alm_dtype, long alm_cstride, const uint64_t* d_mstart, long lstride,
                  int deriv1, LegProfile* prof, long ld, int nb, long alm_bstride, long leg_bstride)
{
	PXS_REQUIRE(alm_dtype == PX_C64 || alm_dtype == PX_C128, "alm must be complex64 or complex128");
	PXS_REQUIRE(nb >= 1, "leg_analysis: nb must be >= 1");
	if (tb.spin > 0 && !deriv1 && nb >= ana_mm_min() && !wk.deterministic) { leg_analysis_spin_mm(st, rs, tb, wk, leg, alm, alm_dtype, alm_cstride, d_mstart, lstride, prof, ld, nb, alm_bstride, leg_bstride); return; }
	if (tb.spin == 0 && nb >= ana_mm_min() && !wk.deterministic) { leg_analysis_mm(st, rs, tb, wk, leg, alm, alm_dtype, alm_cstride, d_mstart, lstride, prof, ld, nb, alm_bstride, leg_bstride); return; }
	const int K = tb.spin == 0 ? k_small_grid(rs, k_ana0(), {4, 2}, 4) : k_small_grid(rs, k_anas(), {3, 2});
	const int nm = tb.mmax+1;
	const int nwave = (rs.npairs + 64*K - 1)/(64*K);
	const long n4 = leg_mom_stride(tb);
	{	// the ordered (bitwise repeatable) scheme and the launch that records the seeds take one map at a time
		const bool one_by_one = wk.deterministic || seeds_pending(wk, rs, tb, 1, K);
		if (nb > 1 && one_by_one) {
			const size_t aesz = alm_dtype == PX_C64 ? 8 : 16;
			const int first = wk.deterministic ? nb : 1;
			for (int b = 0; b < first; b++)
				leg_analysis(st, rs, tb, wk, leg + (size_t)b*leg_bstride, (char*)alm + aesz*(size_t)b*alm_bstride, alm_dtype, alm_cstride, d_mstart, lstride, deriv1, prof, ld, 1, 0, 0);
			if (first < nb)
				leg_analysis(st, rs, tb, wk, leg + (size_t)first*leg_bstride, (char*)alm + aesz*(size_t)first*alm_bstride, alm_dtype, alm_cstride, d_mstart, lstride, deriv1, prof, ld, nb - first, alm_bstride, leg_bstride);
			return;
		}
	}
	if (const int nmax = leg_max_batch(rs, tb, K); nb > nmax) {      // (grid limit of one launch)
		const size_t aesz = alm_dtype == PX_C64 ? 8 : 16;
		for (int b = 0; b < nb; b += nmax)
			leg_analysis(st, rs, tb, wk, leg + (size_t)b*leg_bstride, (char*)alm + aesz*(size_t)b*alm_bstride, alm_dtype, alm_cstride, d_mstart, lstride, deriv1, prof, ld, std::min(nmax, nb - b), alm_bstride, leg_bstride);
		return;
	}
	wk.mom.ensure(sizeof(double)*(size_t)n4*nb);
	// Default: the waves of one m add their sums straight into mom with global_atomic_add_f64 -- the blocks of one m run back
	// to back on one XCD (leg_block), so the row they share sits in that XCD's L2 while they do.  The order of those additions
	// is not fixed: results repeat to rounding, not bit for bit.  pxs_plan_option("deterministic", 1) (or PXS_DETERMINISTIC=1 when the plan is made) selects the former scheme instead
	// (per-wave partial moments in scratch, summed in wave order by reduce_partials), for callers that need bitwise repeats.
	const bool atomic = !wk.deterministic;
	std::vector<int> cuts; cuts.push_back(0);
	if (atomic) {
		cuts.push_back(nm);
		PXS_HIP(hipMemsetAsync(wk.mom.p, 0, sizeof(double)*(size_t)n4*nb, st));
	} else {
		// chunk m so that the per-wave partial moments stay below part_budget bytes
		const size_t budget = wk.part_budget;
		for (int m = 0; m < nm;) {
			int m1 = m+1;
			while (m1 < nm && (size_t)(tb.row[m1+1]-tb.row[m])*32*nwave <= budget) m1++;
			cuts.push_back(m1); m = m1;
		}
		size_t maxrows = 1;
		for (size_t c = 0; c+1 < cuts.size(); c++) maxrows = std::max<size_t>(maxrows, (size_t)(tb.row[cuts[c+1]]-tb.row[cuts[c]]));
		wk.part.ensure(sizeof(double)*4*maxrows*nwave);
		// sized once, before the first launch: growing a buffer inside the chunk loop would free memory that kernels of the
		// previous chunk may still use (and hipFree synchronises the device)
		size_t maxm = 1;
		for (size_t c = 0; c+1 < cuts.size(); c++) maxm = std::max<size_t>(maxm, (size_t)(cuts[c+1]-cuts[c]));
		wk.first.ensure(sizeof(int)*(size_t)nwave*maxm);
	}
	const size_t sh = sizeof(double)*LEG_RED_DOUBLES;
	LegWork::Seeds* seeds = nullptr;
	for (size_t c = 0; c+1 < cuts.size(); c++) {
		const int m0 = cuts[c], m1 = cuts[c+1];
		const long rows = tb.row[m1]-tb.row[m0];
		if (rows <= 0) continue;
		LegK a = make_legk(rs, tb, wk, const_cast<double2*>(leg), ld, K, nb, leg_bstride);
		a.m0 = m0; a.rowbase = tb.row[m0]; a.rows_chunk = rows; a.nmc = m1-m0; a.atomic = atomic ? 1 : 0;
		seeds = seeds_for(wk, rs, tb, 1, K, a); seeds_wait(seeds, st);
		if (atomic) { a.part = (double*)wk.mom.p; a.rowbase = 0; a.rows_chunk = 0; a.first = nullptr; }
		else {
			PXS_HIP(hipMemsetAsync(wk.first.p, 0, sizeof(int)*(size_t)nwave*(m1-m0), st));
			a.first = wk.first.as<int>();
		}
		if (prof) prof->begin(st, 1);
		const dim3 grid = leg_grid(a);
		if (tb.spin == 0) {
#ifdef PXS_LAB
			if (K == 12)     hipLaunchKernelGGL(leg_ana_s0<12>, grid, dim3(64), sh, st, a); else
#endif
			if (K == 8) hipLaunchKernelGGL(leg_ana_s0<8>, grid, dim3(64), sh, st, a);
			else if (K == 2) hipLaunchKernelGGL(leg_ana_s0<2>, grid, dim3(64), sh, st, a);
			else             hipLaunchKernelGGL(leg_ana_s0<4>, grid, dim3(64), sh, st, a);
		} else {
#ifdef PXS_LAB
			if (K >= 6)      hipLaunchKernelGGL(leg_ana_spin<6>, grid, dim3(64), sh, st, a);
			else if (K == 5) hipLaunchKernelGGL(leg_ana_spin<5>, grid, dim3(64), sh, st, a); else
#endif
			if (K >= 4) hipLaunchKernelGGL(leg_ana_spin<4>, grid, dim3(64), sh, st, a);
			else if (K == 3) hipLaunchKernelGGL(leg_ana_spin<3>, grid, dim3(64), sh, st, a);
			else             hipLaunchKernelGGL(leg_ana_spin<2>, grid, dim3(64), sh, st, a);
		}
		if (prof) prof->end(st, 1);
		if (atomic) continue;
		const long maxrow = tb.row[m0+1] - tb.row[m0];       // rows per m shrink with m
		hipLaunchKernelGGL(reduce_partials, dim3((unsigned)((4*maxrow+255)/256), m1-m0), dim3(256), 0, st, (const double*)wk.part.p,
			(double*)wk.mom.p, tb.d_row.as<long>(), (const int*)wk.first.p, m0, m1-m0, tb.row[m0], rows, a.nwave);
	}
	seeds_written(seeds, st);
	AlmK ak = make_almk(tb, wk, alm, alm_dtype, alm_cstride, d_mstart, lstride, deriv1, alm_bstride);
	if (tb.spin == 0) hipLaunchKernelGGL(alm_post_s0, alm_grid(tb.lmax/2 + 1, nm, nb), dim3(256), 0, st, ak);
	else              hipLaunchKernelGGL(alm_post_spin, alm_grid(tb.lmax + 1, nm, nb), dim3(256), 0, st, ak);
	PXS_HIP(hipGetLastError());
}

} // namespace pxs
