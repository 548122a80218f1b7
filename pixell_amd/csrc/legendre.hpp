// Legendre stage of the SHT on gfx950 (alm <-> leg[m][ring]); see legendre.hip (host side, design notes), leg_s0.hip / leg_spin.hip (kernels), legendre_dev.hpp.
#pragma once
#include <map>
#include <tuple>
#include "common.hpp"

namespace pxs {

// Ring set the Legendre kernels iterate: north/south mirror pairs share one recurrence.
struct RingSet {
	int nring = 0;               // rings in leg[m][ring]
	int npairs = 0;
	std::vector<int> ring_n, ring_s;        // leg ring index of the northern / southern member (-1: none)
	std::vector<double> cth, sth, sh2, ch2; // of the northern member: cos, sin, sin(theta/2), cos(theta/2)
	DevBuf d_ring_n, d_ring_s, d_cth, d_sth, d_sh2, d_ch2;
	void build(const std::vector<long double>& theta);   // pairs rings theta <-> pi-theta
	void upload_all();
};

// Per-(lmax,mmax,spin) recurrence tables (host long double -> device double)
struct LegTables {
	int lmax = 0, mmax = 0, spin = 0;
	std::vector<long> row;      // row[m]: first row of m; rows = k-steps (spin 0: two l per row) or l-steps (spin s)
	long nrows = 0;
	DevBuf d_row, d_coef, d_alpha;   // coef[row] = (a,b); alpha[row] = scaling folded into the alm
	mutable DevBuf d_coef2, d_coef2p; // spin 0, batched analysis (leg_ana_s0_mm): compact rows (a, b) / (a, a + b), built on first use
	void build(int lmax, int mmax, int spin);        // on the GPU, double-double (legendre.hip)
	void build_host(int lmax, int mmax, int spin);   // host threads, long double: the reference (PXS_TABLES_HOST=1, tests)
};

// optional per-stage device timers (hipEvents on the launch stream); stage ids: see pxsht.h PXS_STAGE_*
struct LegProfile {
	struct Rec { hipEvent_t a, b; int stage; };
	std::vector<Rec> recs; std::vector<hipEvent_t> open_;
	bool enabled = false;
	void begin(hipStream_t st, int stage);
	void end(hipStream_t st, int stage);
	void read(double* ms, int* counts, int nstage, bool reset);
	~LegProfile();
};

struct LegWork {     // scratch owned by the SHT plan
	size_t part_budget = size_t(2) << 30;   // bytes of per-wave partial moments per launch of the ordered analysis (PXS_PART_GB overrides): m goes in chunks that fit
	bool deterministic = false;             // ordered (bitwise repeatable) analysis sums: pxs_plan_option("deterministic"); default from PXS_DETERMINISTIC when the plan is made
	DevBuf almt;     // [nrows][4] doubles
	DevBuf part;     // [nwave][nrows][4] doubles (analysis partial moments)
	DevBuf mom;      // [nrows][4] reduced moments
	DevBuf first;    // [nwave][m chunk] int: 1 + first row a wave wrote for that m (0: the wave had no live ring and wrote nothing)
	// recurrence seeds (legendre_dev.hpp, S0_SEEDED_PHASE_A): one set per (ring set, spin, direction, K), recorded by the first launch
	struct Seeds { DevBuf d, i; bool ready = false, refused = false; hipEvent_t written = nullptr; hipStream_t wstream = nullptr;
	               ~Seeds() { if (written) (void)hipEventDestroy(written); } };
	std::map<std::tuple<const void*, int, int, int>, Seeds> seeds;
	size_t seed_budget = size_t(16) << 30, seed_bytes = 0;     // PXS_SEED_GB; 0 turns the seeds off
	DevBuf count; bool count_on = false;                       // executed-work counters of the Legendre kernels (LegK::count), while profiling
};


// alm[(c) * alm_cstride + mstart[m] + l*lstride] -> leg[(c*nm + m)*ld + ring]  (c = 0 or 0,1; ld = 0: nring)
// nb maps in one launch: map b reads alm + b*alm_bstride (alm elements) and writes leg + b*leg_bstride (double2 elements)
void leg_synthesis(hipStream_t st, const RingSet& rs, const LegTables& tb, LegWork& wk,
                   const void* alm, int alm_dtype, long alm_cstride, const uint64_t* d_mstart, long lstride,
                   double2* leg, int deriv1, LegProfile* prof = nullptr, long ld = 0, int nb = 1, long alm_bstride = 0, long leg_bstride = 0);
// transpose of leg_synthesis (no weights)
void leg_analysis(hipStream_t st, const RingSet& rs, const LegTables& tb, LegWork& wk,
                  const double2* leg, void* alm, int alm_dtype, long alm_cstride, const uint64_t* d_mstart, long lstride,
                  int deriv1, LegProfile* prof = nullptr, long ld = 0, int nb = 1, long alm_bstride = 0, long leg_bstride = 0);

} // namespace pxs
