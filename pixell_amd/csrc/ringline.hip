// Single-kernel synthesis ring FFT for gfx950: ONE workgroup turns the spectra h[ring][m] of a ring pair into its two map rows --
// Hermitian extension of both rows as one complex line of nphi points (z = h_a + i h_b), backward transform with the line resident in
// registers and the LDS (regfft_dev.hpp), real part -> ring 2q, imaginary part -> ring 2q + 1 -- where the stage chain
// (fftchain.hip, MS1 / MS2: a four-step transform through HBM) writes and reads an intermediate of 16 bytes per pixel pair.  HBM
// traffic per pair: the two rows of h in, the two map rows out.  Same arithmetic statement as StRingS1::load / StRingS2::store; compiled
// for an explicit list of ring lengths (RING_CONFIGS), everything else runs the stage chain.
// (The analysis direction was built the same way and measured: its output leg[m][ring] is ring-contiguous per m, so a workgroup that owns
// two rings writes 32-byte pieces of 128-byte lines.  With the four workgroups that complete a line placed on one XCD in the same
// iteration the pieces do merge in that XCD's L2 -- WRITE_SIZE 1.26x the bytes of leg, not 4x -- but a store instruction that touches 64
// lines costs what the saved intermediate gains: 3.46 ms per 8 maps against 3.48 for the two-stage chain
// (profiles/r06_ring_line_analysis_ab.txt; the kernel is in the history, commit "Analysis ring FFT as one kernel").  map2leg keeps MA1 / MA2.)
// Replaces the ring FFTs inside ducc0's synthesis_2d / adjoint_analysis_2d (pixell/curvedsky.py:907-924).
#include "fftchain.hpp"
#include "chain_dev.hpp"
#include "regfft_dev.hpp"
#include <algorithm>
#include <map>
#include <utility>
#include <memory>
#include <vector>

namespace pxs {

struct RingArgs {
	const double2* tw; int ntw;
	const double2* h; long ldh, hcomp; int mmax, npair, nring, ntask; FastDiv dnp;
	MapAddr m;
};

static constexpr int rl_twlen(int count) { return RF_TWL + (count + RF_TWL - 1)/RF_TWL; }
template<class S, int NT, int P = 0> constexpr int rl_slots() {
	if constexpr (P >= S::NP) return 0; else { constexpr int a = RfPassT<S, P, NT>::slots, b = rl_slots<S, NT, P + 1>(); return a > b ? a : b; }
}
template<int NT_, int MMAX_CAP, class S_> struct RingCfg {
	static constexpr int NT = NT_, X = S_::N, MCAP = MMAX_CAP;      // ring length; the rows of h hold at most MCAP + 1 coefficients
	using S = S_;
	static constexpr int PMAX = rl_slots<S, NT>();
	static constexpr int ntw = rl_twlen(X);
	static constexpr int words = (X > 4*(MCAP + 1) ? X : 4*(MCAP + 1)) + 16;      // doubles: one component of the line, or both rows of h (complex)
	static constexpr size_t lds = sizeof(double2)*ntw + sizeof(double)*words + 16;
	static_assert(2*MCAP < X, "the Hermitian image of the coefficients overlaps them");
};

#ifndef PXS_HOST_SIM
#define PXS_RL_BOUNDS __launch_bounds__(CFG::NT)
#else
#define PXS_RL_BOUNDS
#endif
template<class CFG> __global__ PXS_RL_BOUNDS void ring_line_kernel(const RingArgs a)
{
	constexpr int NT = CFG::NT, PMAX = CFG::PMAX, X = CFG::X;
	using F = RegFft<NT, PMAX>; using S = typename CFG::S;
	using P0 = RfPassT<S, 0, NT>; using PL = RfPassT<S, S::NP - 1, NT>;
	PXS_SHARED(double2, lds);
	double2* tws = lds;
	double* line = reinterpret_cast<double*>(lds + CFG::ntw);
	double2* rows = lds + CFG::ntw;
	const int tid = threadIdx.x;
	for (int k = tid; k < CFG::ntw; k += NT) tws[k] = a.tw[k];
	constexpr int NL = (CFG::MCAP + 1 + NT - 1)/NT;      // coefficients of a row per thread
	const int nm = a.mmax + 1;
	for (int task = blockIdx.x; task < a.ntask; task += gridDim.x) {
		const int comp = (int)fdiv((uint32_t)task, a.dnp), q = task - comp*a.npair;
		const bool two = 2*q + 1 < a.nring;
		const double2* r0 = a.h + ((long)comp*a.hcomp + 2*q)*a.ldh;
		// both rows of h into the LDS (coalesced, every coefficient read from memory once; all loads in flight before the first wait)
		double2 raw[2][NL];
		sfor<0, 2*NL>([&](auto Q) RF_INL {
			constexpr int qq = RF_IDX(Q), row = qq/NL, u = qq % NL;
			const int i = tid + NT*u;
			const bool ok = i < nm && (row == 0 || two);
			const double2 x = r0[(ok ? (long)row*a.ldh : 0) + (ok ? i : 0)];
			raw[row][u] = ok ? x : make_double2(0, 0);
		});
		RF_BARRIER();
		sfor<0, 2*NL>([&](auto Q) RF_INL { constexpr int qq = RF_IDX(Q), row = qq/NL, u = qq % NL; const int i = tid + NT*u; if (i < nm) rows[row*nm + i] = raw[row][u]; });
		RF_BARRIER();
		// the line: bin k <- coefficient m = k (k <= mmax) or the conjugate of m = X - k; z = h_a + i h_b; conjugated for the backward
		// transform (backward = conj forward conj)
		double2 v[PMAX];
		F::template fill<P0>(v, tid, [&](int k) {
			int m = k; bool cj = false, zero = false;
			if (k > a.mmax) { m = X - k; cj = true; if (m > a.mmax) { zero = true; m = 0; } }
			double2 ha = rows[m], hb = rows[nm + m];
			if (m == 0) { ha.y = 0; hb.y = 0; }
			if (cj) { ha.y = -ha.y; hb.y = -hb.y; }
			const double2 z = make_double2(ha.x - hb.y, ha.y + hb.x);
			return zero ? make_double2(0, 0) : make_double2(z.x, -z.y);
		});
		F::template run<S>(v, tid, line, tws);
		// pixel x = b + nb j of output j of butterfly b of the last pass: real part -> ring 2q, minus the imaginary part (the conj of
		// the backward transform) -> ring 2q + 1
		{	const int bi = (int)fdiv((uint32_t)comp, a.m.dncb);
			const long o0 = (long)bi*a.m.bstride + (long)(comp - bi*a.m.ncb)*a.m.cstride + a.m.off0 + (2L*q)*a.m.rstride;
			sfor<0, PL::slots>([&](auto C) RF_INL {
				constexpr int c = RF_IDX(C), i = c/PL::R, j = c % PL::R;
				const int b = tid + NT*i;
				if ((i + 1)*NT <= PL::nb || b < PL::nb) {
					const long o = o0 + (long)(b + PL::nb*j)*a.m.pstride;
					wr_real(a.m.ptr, a.m.dtype, o, v[c].x);
					if (two) wr_real(a.m.ptr, a.m.dtype, o + a.m.rstride, -v[c].y);
				}
			});
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct RingEntry { long X; int mcap, nt, ntw; size_t lds; void (*launch)(const RingArgs&, size_t, long, hipStream_t); };
template<class CFG> static void launch_ring(const RingArgs& a, size_t lds, long nwg, hipStream_t st) {
#ifndef PXS_HOST_SIM
	static const bool once = [] { (void)hipFuncSetAttribute((const void*)ring_line_kernel<CFG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024); return true; }();
	(void)once;
#endif
	hipLaunchKernelGGL((ring_line_kernel<CFG>), dim3((unsigned)nwg), dim3(CFG::NT), lds, st, a);
}
template<class CFG> static RingEntry ring_entry() {
	static_assert(CFG::lds <= 160*1024 - 256, "the ring pair does not fit the LDS");
	return RingEntry{CFG::X, CFG::MCAP, CFG::NT, CFG::ntw, CFG::lds, &launch_ring<CFG>};
}
// Ring lengths compiled in (a plan takes the kernel when nphi matches and mmax is within the configuration's cap):
//   10 800 pixels per ring (BASELINE C2 / C4, mmax <= 4000): 16 15 15 3 on 1024 threads, <= 16 points per thread
#ifdef PXS_HOST_SIM
using RingSim = RingCfg<64, 300, RfSeq<12, 10, 6>>;      // (the simulator: 720 pixels per ring on 64 OS threads)
static const RingEntry RING_CONFIGS[] = { ring_entry<RingSim>() };
#else
using RingC4 = RingCfg<1024, 4000, RfSeq<16, 15, 15, 3>>;
static const RingEntry RING_CONFIGS[] = { ring_entry<RingC4>() };
#endif

struct RingLineState { std::map<std::pair<int, long>, DevBuf> tw; std::map<int, int> ncu; };      // twiddle tables and CU counts per (device, ring length)
static std::mutex g_ring_mu;
static RingLineState& ring_state() { static RingLineState s; return s; }

// PXS_RING_LINE=0 keeps the stage chain (read per call: the tests compare the two paths)
static bool ring_line_enabled() { const char* e = getenv("PXS_RING_LINE"); return e ? atoi(e) != 0 : true; }

static const RingEntry* ring_find(long nphi, int mmax) {
	if (!ring_line_enabled()) return nullptr;
	for (const RingEntry& c : RING_CONFIGS) if (c.X == nphi && mmax <= c.mcap) return &c;
	return nullptr;
}
bool FftChain::line_h2map_takes(long nphi, int mmax) { return ring_find(nphi, mmax) != nullptr; }

static void ring_tables(const RingEntry* e, const double2*& tw, int& ncu);

bool FftChain::line_h2map(hipStream_t st, const double2* h, long ldh, const MapDesc& m, int nc, int mmax, long hcomp)
{
	const RingEntry* e = ring_find(m.nphi, mmax);
	if (!e) return false;
	const double2* tw; int ncu; ring_tables(e, tw, ncu);
	const long npair = (m.nring + 1)/2;
	RingArgs a; memset(&a, 0, sizeof(a));
	a.tw = tw; a.ntw = e->ntw; a.h = h; a.ldh = ldh; a.hcomp = hcomp > 0 ? hcomp : m.nring; a.mmax = mmax; a.npair = (int)npair; a.nring = m.nring;
	const long ntask = (long)nc*npair;
	PXS_REQUIRE(ntask < (1L << 31), "internal: ring line grid too large");
	a.ntask = (int)ntask; a.dnp = make_fastdiv((uint32_t)npair);
	a.m = map_addr(m);
	const long per_cu = std::max<long>(1, std::min<long>(2048/e->nt, (long)(160*1024)/(long)e->lds));
	e->launch(a, e->lds, std::min<long>(ntask, (long)ncu*per_cu), st);
	PXS_HIP(hipGetLastError());
	return true;
}

static void ring_tables(const RingEntry* e, const double2*& tw, int& ncu)
{
	{	std::lock_guard<std::mutex> g(g_ring_mu);
		RingLineState& s = ring_state();
		int dev = 0;
#ifndef PXS_HOST_SIM
		PXS_HIP(hipGetDevice(&dev));
#endif
		DevBuf& b = s.tw[std::make_pair(dev, e->X)];
		if (!b.p) {
			std::vector<double2> t((size_t)e->ntw, make_double2(1, 0));
			const long double tpi = 6.283185307179586476925286766559L;
			for (int l = 0; l < RF_TWL; l++) { const long double ang = tpi*(long double)(l % e->X)/(long double)e->X; t[l] = make_double2((double)cosl(ang), (double)(-sinl(ang))); }
			for (long hh = 0; hh*RF_TWL < e->X; hh++) { const long double ang = tpi*(long double)((hh*RF_TWL) % e->X)/(long double)e->X; t[RF_TWL + hh] = make_double2((double)cosl(ang), (double)(-sinl(ang))); }
			b = upload(t);
		}
		tw = b.as<double2>();
		int& n = s.ncu[dev];
		if (n == 0) {
#ifdef PXS_HOST_SIM
			n = 2;
#else
			hipDeviceProp_t pr; PXS_HIP(hipGetDeviceProperties(&pr, dev)); n = std::max(1, pr.multiProcessorCount);
#endif
		}
		ncu = n;
	}
}

} // namespace pxs
