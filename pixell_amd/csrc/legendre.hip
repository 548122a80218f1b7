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
// Translation units: this file holds the host side (tables, seeds, dispatch), the alm pre / post kernels and the table kernels; leg_s0.hip and leg_spin.hip the
// Legendre kernels themselves, legendre_dev.hpp what they share.
#include "legendre_dev.hpp"
#include <thread>
#include <memory>

namespace pxs {

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


// compact step table: (a, b) or (a, a + b) of the rows of LegTables::coef; 32 rows of padding (a tile reads 16 steps whatever nk is)
__global__ __launch_bounds__(256) void coef2_kernel(const double4_t* __restrict__ coef, long nrows, double2* __restrict__ c2, double2* __restrict__ c2p) {
	const long i = (long)blockIdx.x*blockDim.x + threadIdx.x;
	if (i >= nrows + 32) return;
	if (i < nrows) { const double4_t c = coef[i]; c2[i] = make_double2(c.a, c.b); c2p[i] = make_double2(c.a, c.c); }
	else { c2[i] = make_double2(0, 0); c2p[i] = make_double2(0, 0); }
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
		if (tb.spin == 0) launch_leg_syn_s0(K, leg_grid(a), st, a);
		else              launch_leg_syn_spin(K, leg_grid(a), st, a);
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
			launch_leg_syn_s0_mm(ng, leg_grid(a), st, a);
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
			launch_leg_syn_spin_mm(ng, leg_grid(a), st, a);
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
		launch_leg_ana_s0_mm(ng, W, leg_grid(a), st, a);
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
	auto launch = [&](int b0, int nmaps, int ng) {
		const int per = 4*ng, ngroups = (nmaps + per - 1)/per;
		// (a workgroup covers 32 W ring pairs: nwave of the block mapping counts those)
		LegK a = make_legk(rs, tb, wk, const_cast<double2*>(leg) + (size_t)b0*leg_bstride, ld, 1, ngroups, leg_bstride);
		a.nwave = (rs.npairs + 32*W - 1)/(32*W);
		a.mom = wk.mom.as<double>() + (size_t)b0*a.mom_bs; a.part = a.mom; a.atomic = 1; a.nmaps = nmaps;
		a.coef2 = tb.d_coef2.as<double2>(); a.coef2p = tb.d_coef2p.as<double2>();
		if (prof) prof->begin(st, 1);
		launch_leg_ana_spin_mm(ng, leg_grid(a), st, a);
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
		if (tb.spin == 0) launch_leg_ana_s0(K, grid, sh, st, a);
		else              launch_leg_ana_spin(K, grid, sh, st, a);
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
