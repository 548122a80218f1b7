// Single-kernel theta resampling for gfx950: ONE workgroup carries a pair of columns (m, m+1) of `leg` through the whole chain
// of the analysis -- mirror-pair extension, FFT_N, shift / resize, IFFT_M, pointwise table, FFT_M, truncation to |k| <= lmax,
// IFFT_Ncc, separation of the pair, weights -- with the line resident in registers and the LDS (regfft_dev.hpp) instead of five
// kernels that pass it through HBM four times (fftchain.hip, RA1-RA5).  HBM traffic per pair: the two rows of `leg` in, the two
// rows of `leg_cc` out.  Same arithmetic statement as FftChain::to_cc / from_cc_adjoint (the stage functors there), which stay
// for every other size: the engine is compiled for an explicit list of circle sizes (LINE_CONFIGS below; the radix sequences are
// template parameters, see regfft_dev.hpp for why), and lines beyond ~16 000 points (C3: 43 200) do not fit a CU at all.
// Replaces ducc0's resample_to_prepared_CC inside analysis_2d / adjoint_synthesis_2d (pixell/curvedsky.py:1046, 924).
#include "fftchain.hpp"
#include "chain_dev.hpp"
#include "regfft_dev.hpp"
#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <tuple>
#include <vector>

namespace pxs {

struct LineArgs {
	const double2* tw; int ntw; // two-level twiddles of N, M, Ncc and of the shift phase, copied to the LDS once per workgroup
	int has_ph;                // the shift e^{-i k theta_0} = W_2N^{c k} is not the identity (c = mir_c != 0)
	// the pair of columns (PairSrc of the stage chain, loaded through the LDS one row at a time)
	const double2* leg; long ldleg, cstride; int nr, mir_c; const double2* wring;
	int lmax;
	const double2* sigma;
	const double* sig_half;    // sigma real and mirror-symmetric (the weights table of the default analysis): sigma[i].x, i <= M/2; else null
	int nr_out, a_odd, ncol, npair, ntask; FastDiv dnp;
	double2* out; long ld, ocstride; const double2* w; double scale;
#ifdef PXS_LAB_TL_TIME      /* lab builds: shader-clock time per phase, summed over the lines of workgroup 0 (tools/tl_lab.sh) */
	unsigned long long* prof;
#endif
};
#ifdef PXS_LAB_TL_TIME
#define TL_T(k) do { if (blockIdx.x == 0 && tid0 == CFG::NT - 64) { const unsigned long long t_ = clock64(); a.prof[k] += t_ - t0_; t0_ = t_; } } while (0)
#define TL_T0() unsigned long long t0_ = clock64()
#else
#define TL_T(k) do {} while (0)
#define TL_T0() do {} while (0)
#endif

// slot jp of a spectrum of X2 points <- bin of a spectrum of X1 points (StResize::mid of fftchain.hip, the non-transposed rule)
struct ResizeRule {
	int X1, X2, kmax, nyq;
	__device__ __forceinline__ int src(int jp, int& ak, bool& neg) const {
		const int kap = (2*jp <= X2) ? jp : jp - X2;
		ak = kap < 0 ? -kap : kap; neg = kap < 0;
		if ((kmax >= 0 && ak > kmax) || 2*ak > X1) return -1;
		return kap >= 0 ? kap : kap + X1;
	}
};

static constexpr int tl_twlen(int count) { return RF_TWL + (count + RF_TWL - 1)/RF_TWL; }
static constexpr int tl_max(int a, int b) { return a > b ? a : b; }
template<class S, int NT, int P = 0> constexpr int tl_slots() {
	if constexpr (P >= S::NP) return 0; else return tl_max(RfPassT<S, P, NT>::slots, tl_slots<S, NT, P + 1>());
}

// one compiled configuration: threads per workgroup and the radix sequences of the circles (SMi: the backward transform on M; when it
// ends with the radix SMf starts with, the pointwise step between them needs no exchange; SMf = RfSeq<>: no middle circle -- from_cc_adjoint)
template<int NT_, class SN_, class SMi_, class SMf_, class SC_> struct LineCfg {
	static constexpr int NT = NT_;
	using SN = SN_; using SMi = SMi_; using SMf = SMf_; using SC = SC_;
	static constexpr bool MID = SMf::NP > 0;
	static constexpr int N = SN::N, M = MID ? SMf::N : 0, Ncc = SC::N;
	static constexpr int PMAX = tl_max(tl_max(tl_slots<SN, NT>(), tl_slots<SC, NT>()), tl_max(tl_slots<SMi, NT>(), tl_slots<SMf, NT>()));
	// the workgroup's twiddle table in the LDS: N, M, Ncc, then the shift phase W_2N^{c k}, k <= N/2
	static constexpr int twN = 0, twM = twN + tl_twlen(N), twC = twM + (MID ? tl_twlen(M) : 0), twP = twC + tl_twlen(Ncc), ntw = twP + tl_twlen(N/2 + 1);
	static constexpr int words = tl_max(tl_max(N, M), 2*Ncc) + 32;      // doubles of the line area (the swizzle of regfft_dev.hpp permutes within blocks of 16)
	static constexpr size_t lds = sizeof(double2)*ntw + sizeof(double)*words + 16;
	static_assert(!MID || SMi::N == SMf::N, "the two transforms on M differ in length");
	static_assert(N/2 + 1 <= words/2, "a row of leg does not fit the line area");
	static_assert(!MID || (M % 2 == 0 && M/2 <= words/2), "half of the pointwise table does not fit the line area");
};

template<class CFG> struct LineOps {
	static constexpr int NT = CFG::NT, PMAX = CFG::PMAX;
	using F = RegFft<NT, PMAX>;
	using Regs = double2 (&)[PMAX];
	// the spectrum the registers hold (outputs of the last pass PP) -> through the LDS and the resize rule -> inputs of the first pass PN
	// of the BACKWARD transform that follows (conjugated: backward = conj forward conj).  ph: two-level table of the shift phase in the
	// LDS (lo[128], hi[]), or null.
	template<class PP, class PN> static __device__ __forceinline__ void resize(Regs v, int tid, double* line, const ResizeRule rr, const double2* ph) {
		static_assert(PN::slots <= PMAX, "pass needs more register slots than the kernel has");
		RF_BARRIER();
		F::template write_comp<PP, 0>(v, tid, line);
		RF_BARRIER();
		RF_OPAQUE(tid);
		sfor<0, PN::slots>([&](auto C) RF_INL {
			constexpr int c = RF_IDX(C), i = c/PN::R, kk = c % PN::R;
			int ak; bool neg; const bool ok = (i + 1)*NT <= PN::nb || tid < PN::nb - NT*i;
			const int k = rr.src(ok ? tid + (NT*i + PN::nb*kk) : 0, ak, neg);
			const double x = line[F::pad(k >= 0 ? k : 0)];
			v[c].x = k >= 0 ? x : 0.0;
			RF_FENCE_SLOT(c);
		});
		RF_BARRIER();
		F::template write_comp<PP, 1>(v, tid, line);
		RF_BARRIER();
		RF_OPAQUE(tid);
		sfor<0, PN::slots>([&](auto C) RF_INL {
			constexpr int c = RF_IDX(C), i = c/PN::R, kk = c % PN::R;
			int ak; bool neg; const bool ok = (i + 1)*NT <= PN::nb || tid < PN::nb - NT*i;
			const int k = rr.src(ok ? tid + (NT*i + PN::nb*kk) : 0, ak, neg);
			const double y = line[F::pad(k >= 0 ? k : 0)];
			double2 x = make_double2(v[c].x, k >= 0 ? y : 0.0);
			if (rr.nyq && 2*ak == rr.X1) x = cscale(x, 0.5);
			if (ph) { double2 t = cmul(ph[ak & (RF_TWL - 1)], ph[RF_TWL + (ak >> 7)]); if (neg) t.y = -t.y; x = cmul(x, t); }
			v[c] = make_double2(x.x, -x.y);
			RF_FENCE_SLOT(c);
		});
	}
	// mirror-pair extension of columns (2 pr, 2 pr + 1) (PairSrc::get of the stage chain): each row goes through the LDS once -- coalesced
	// loads, every element read from memory once -- and lands in the registers in the read pattern of pass PN.  The loads of BOTH rows are
	// issued before the first wait (a loop over the row, one load per trip, paid the memory latency six times per row: 16 000 clocks).
	template<class PN> static __device__ __forceinline__ void load_pair(Regs v, int tid, double2* line2, const LineArgs& a, int comp, int pr) {
		static_assert(PN::slots <= PMAX, "pass needs more register slots than the kernel has");
		constexpr int N = CFG::N, NL = (N/2 + 1 + NT - 1)/NT;      // row elements per thread
		const int ca = 2*pr;
		const double2* ra = a.leg + (long)comp*a.cstride + (long)ca*a.ldleg;
		const bool two = ca + 1 < a.ncol;      // (an odd number of columns: the last pair has one)
		// row 0 is loaded first; the loads of row 1 are issued while row 0 sits in the LDS, and wait in registers through its gather
		// (both rows up front held 12 complex values per thread next to the line that is being filled: spills)
		double2 raw[NL], raw1[NL];
		auto load_row = [&](double2 (&dst)[NL], int row) RF_INL {
			sfor<0, NL>([&](auto U) RF_INL {
				constexpr int u = RF_IDX(U);
				const int i = tid + NT*u;
				const bool ok = i < a.nr && (row == 0 || two);
				double2 x = ra[(ok ? (long)row*a.ldleg : 0) + (ok ? i : 0)];
				if (a.wring) x = cscale(x, a.wring[ok ? i : 0].x);
				dst[u] = ok ? x : make_double2(0, 0);
			});
		};
		load_row(raw, 0);
		sfor<0, 2>([&](auto RW) RF_INL {
			constexpr int row = RF_IDX(RW);
			const bool odd = row == 0 ? a.a_odd != 0 : a.a_odd == 0;      // this column is odd under the reflection
			RF_BARRIER();
			sfor<0, NL>([&](auto U) RF_INL { constexpr int u = RF_IDX(U); const int i = tid + NT*u; if (i < a.nr) line2[i] = row == 0 ? raw[u] : raw1[u]; });
			if constexpr (row == 0) load_row(raw1, 1);
			RF_BARRIER();
			RF_OPAQUE(tid);
			sfor<0, PN::slots>([&](auto C) RF_INL {
				constexpr int c = RF_IDX(C), i = c/PN::R, kk = c % PN::R;
				const bool ok = (i + 1)*NT <= PN::nb || tid < PN::nb - NT*i;
				const int j = ok ? tid + (NT*i + PN::nb*kk) : 0;
				int src = j; bool mir = false;
				if (j >= a.nr) { src = N - j - a.mir_c; if (src < 0) src += N; mir = true; }
				const int tj = 2*j + a.mir_c;
				double2 x = line2[src];
				if (odd) {
					if (tj == 0 || tj == N || tj == 2*N) x = make_double2(0, 0);      // the sample is its own mirror image
					else if (mir) x = make_double2(-x.x, -x.y);
				}
				v[c] = row == 0 ? x : cadd(v[c], x);
				RF_FENCE_SLOT(c);
			});
		});
	}
};

#ifndef PXS_HOST_SIM
#define PXS_TL_BOUNDS __launch_bounds__(CFG::NT)
#else
#define PXS_TL_BOUNDS
#endif
template<class CFG> __global__ PXS_TL_BOUNDS void theta_line_kernel(const LineArgs a)
{
	constexpr int NT = CFG::NT, PMAX = CFG::PMAX;
	using F = RegFft<NT, PMAX>;
	using L = LineOps<CFG>;
	using SN = typename CFG::SN; using SMi = typename CFG::SMi; using SMf = typename CFG::SMf; using SC = typename CFG::SC;
	PXS_SHARED(double2, lds);
	double2* tws = lds;
	double* line = reinterpret_cast<double*>(lds + CFG::ntw);
	double2* line2 = lds + CFG::ntw;
	const int tid0 = threadIdx.x;
	for (int k = tid0; k < CFG::ntw; k += NT) tws[k] = a.tw[k];
	const double2* ph = a.has_ph ? tws + CFG::twP : nullptr;
#if !defined(PXS_HOST_SIM) && !defined(PXS_LAB_TL_NOSTAGGER)
	// Every line takes the same time, so workgroups that start together stay in step: all CUs then load their rows in the same
	// microseconds (256 x 173 KB = 44 MB per burst, ~11 us of the memory system for what is 385 GB/s on average) and idle the memory
	// for the rest of the line.  The start of workgroup w is delayed by w/256 of a line time (~130 us: s_sleep 127 = 8128 clocks).
	for (int k = 0; k < (int)(blockIdx.x & 255)/8; k++) __builtin_amdgcn_s_sleep(127);
#endif
	for (int task = blockIdx.x; task < a.ntask; task += gridDim.x) {
		const int comp = (int)fdiv((uint32_t)task, a.dnp), pr = task - comp*a.npair;
		// (the thread index, opaque per line: the per-thread addresses of everything a line touches -- table entries, rows, outputs -- are loop
		// invariants, and hoisted out of this loop they occupy registers that are spilled in the prologue and reloaded on every line)
		int tid = tid0; RF_OPAQUE(tid);
		double2 v[PMAX];
		TL_T0();
		L::template load_pair<RfPassT<SN, 0, NT>>(v, tid, line2, a, comp, pr);
		TL_T(0);
		F::template run<SN>(v, tid, line, tws + CFG::twN);
		TL_T(1);
		using NL = RfPassT<SN, SN::NP - 1, NT>;
		if constexpr (CFG::MID) {
			constexpr int N = CFG::N, M = CFG::M;
			ResizeRule r1; r1.X1 = N; r1.X2 = M; r1.kmax = M > N ? -1 : M/2 - 1; r1.nyq = M > N ? 1 : 0;
			L::template resize<NL, RfPassT<SMi, 0, NT>>(v, tid, line, r1, ph);
			TL_T(2);
			F::template run<SMi>(v, tid, line, tws + CFG::twM);
			TL_T(3);
			{	// the pointwise table on the samples where they are; if the backward transform ends with the radix the forward one starts
				// with, the registers are in place for it, else one exchange
				using ML = RfPassT<SMi, SMi::NP - 1, NT>; using MF = RfPassT<SMf, 0, NT>;
				// The table goes through the LDS (the line area is free between two exchanges): with one global load per register slot the
				// compiler spilled the line around the loads and the step took 44 000 of the 296 000 clocks of a line.  A real, mirror-symmetric
				// table (the quadrature weights of the default analysis) goes in at once as M/2 + 1 doubles; any other table half a circle
				// at a time.
				constexpr int M = CFG::M, H = M/2;
				RF_OPAQUE(tid);
				if (a.sig_half) {
					constexpr int NH = (H + 1 + NT - 1)/NT;
					double raw[NH];
					sfor<0, NH>([&](auto U) RF_INL { constexpr int u = RF_IDX(U); const int i = tid + NT*u; raw[u] = a.sig_half[i <= H ? i : 0]; });
					RF_BARRIER();
					sfor<0, NH>([&](auto U) RF_INL { constexpr int u = RF_IDX(U); const int i = tid + NT*u; if (i <= H) line[i] = raw[u]; });
					RF_BARRIER();
					F::template pointwise<ML>(v, tid, [&](double2 x, int idx) { const double g = line[idx <= H ? idx : M - idx]; return make_double2(x.x*g, -x.y*g); });
				} else {
#pragma unroll 1
					for (int half = 0; half < 2; half++) {
						const double2* sg = a.sigma + half*H;
						RF_BARRIER();
						for (int i = tid; i < H; i += NT) line2[i] = sg[i];
						RF_BARRIER();
						const int lo = half*H;
						F::template pointwise<ML>(v, tid, [&](double2 x, int idx) {
							const bool in = (unsigned)(idx - lo) < (unsigned)H;
							const double2 t = cmul(cconj(x), line2[in ? idx - lo : 0]);
							return in ? t : x; });
					}
				}
				if constexpr (ML::R != MF::R) F::template exchange<ML, MF>(v, tid, line); }
			TL_T(4);
			F::template run<SMf>(v, tid, line, tws + CFG::twM);
			TL_T(5);
			ResizeRule r2; r2.X1 = M; r2.X2 = CFG::Ncc; r2.kmax = a.lmax; r2.nyq = 0;
			L::template resize<RfPassT<SMf, SMf::NP - 1, NT>, RfPassT<SC, 0, NT>>(v, tid, line, r2, nullptr);
			TL_T(6);
		} else {
			ResizeRule r1; r1.X1 = CFG::N; r1.X2 = CFG::Ncc; r1.kmax = a.lmax; r1.nyq = 0;
			L::template resize<NL, RfPassT<SC, 0, NT>>(v, tid, line, r1, ph);
		}
		F::template run<SC>(v, tid, line, tws + CFG::twC);
		TL_T(7);
		// the circle of Ncc points, both components (the line area is sized for it), then the separation of the pair by reflection
		// symmetry (StSplit<0> of the stage chain)
		RF_BARRIER();
		F::template write_c128<RfPassT<SC, SC::NP - 1, NT>>(v, tid, line2);
		RF_BARRIER();
		RF_OPAQUE(tid);
		{	constexpr int Ncc = CFG::Ncc;
			const int ca = 2*pr;
			double2* oc = a.out + (long)comp*a.ocstride;
			for (int t = tid; t < a.nr_out; t += NT) {
				int tm = Ncc - t; if (tm >= Ncc) tm -= Ncc;
				const double2 z = cconj(line2[F::pad(t)]);
				double2 ev, od;
				if (tm == t) { ev = z; od = make_double2(0, 0); }
				else {
					const double2 y = cconj(line2[F::pad(tm)]);
					ev = make_double2(0.5*(z.x + y.x), 0.5*(z.y + y.y)); od = make_double2(0.5*(z.x - y.x), 0.5*(z.y - y.y));
				}
				const double2 va = a.a_odd ? od : ev, vb = a.a_odd ? ev : od;
				const double f = a.scale*(a.w ? a.w[t].x : 1.0);
				oc[(long)ca*a.ld + t] = cscale(va, f);
				if (ca + 1 < a.ncol) oc[(long)(ca + 1)*a.ld + t] = cscale(vb, f);
			}
		}
		TL_T(8);
#ifdef PXS_LAB_TL_TIME
		if (blockIdx.x == 0 && tid0 == CFG::NT - 64) a.prof[9] += 1;
#endif
	}
}

// ---------------------------------------------------------------------------------------------------------------
// host side: the compiled configurations, tables, launch
// ---------------------------------------------------------------------------------------------------------------
struct LineEntry {       // what the host needs to know about a compiled configuration
	long N, M, Ncc; int nt, ntw; size_t lds; int twN, twM, twC, twP;
	void (*launch)(const LineArgs&, size_t lds, long nwg, hipStream_t st);
};
template<class CFG> static void launch_cfg(const LineArgs& a, size_t lds, long nwg, hipStream_t st) {
#ifndef PXS_HOST_SIM
	static const bool once = [] { (void)hipFuncSetAttribute((const void*)theta_line_kernel<CFG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024); return true; }();
	(void)once;
#endif
	hipLaunchKernelGGL((theta_line_kernel<CFG>), dim3((unsigned)nwg), dim3(CFG::NT), lds, st, a);
}
template<class CFG> static LineEntry entry_of() {
	static_assert(CFG::lds <= 160*1024 - 256, "the line does not fit the LDS");
	return LineEntry{CFG::N, CFG::M, CFG::Ncc, CFG::NT, CFG::ntw, CFG::lds, CFG::twN, CFG::twM, CFG::twC, CFG::twP, &launch_cfg<CFG>};
}

// The configurations compiled in.  A plan takes the engine when its circles (N, M, N_cc) match one of them exactly; everything else
// runs the stage chains.  Per grid: to_cc (with the middle circle M = 2 N_cc of the default analysis) and from_cc_adjoint (without).
//   5400 rings x lmax 4000 (BASELINE C2 / C4): N = 10 800, N_cc = 2 good_size_complex(4001) = 8064, M = 16 128: 1024 threads x <= 21 points
#ifdef PXS_HOST_SIM
// (the simulator runs every lane as its own context: 64-thread workgroups on a grid of 360 rings, lmax 250 -- every feature of the large
// configuration at a fifteenth of its size: 2 and 3 butterflies per thread, radix 7, M > N)
using CfgSimA = LineCfg<64, RfSeq<12, 10, 6>, RfSeq<7, 9, 16>, RfSeq<16, 9, 7>, RfSeq<12, 7, 6>>;
using CfgSimB = LineCfg<64, RfSeq<12, 10, 6>, RfSeq<>, RfSeq<>, RfSeq<12, 7, 6>>;
static const LineEntry LINE_CONFIGS[] = { entry_of<CfgSimA>(), entry_of<CfgSimB>() };
#else
// (four passes per transform, radices up to 16: the composite radices run the register-lean butterfly of regfft_dev.hpp -- with the
// generic one these sequences spilled 70-190 times per transform and C4's to_cc took 13.7 ms per 8 maps; five-pass sequences of
// radices up to 9, which did not spill, 7.2 ms)
using CfgC4A = LineCfg<1024, RfSeq<16, 15, 15, 3>, RfSeq<7, 9, 16, 16>, RfSeq<16, 16, 9, 7>, RfSeq<16, 8, 9, 7>>;
using CfgC4B = LineCfg<1024, RfSeq<16, 15, 15, 3>, RfSeq<>, RfSeq<>, RfSeq<16, 8, 9, 7>>;
static const LineEntry LINE_CONFIGS[] = { entry_of<CfgC4A>(), entry_of<CfgC4B>() };
#endif

struct ThetaLine {
	std::map<std::tuple<long, long, long, int>, DevBuf> tw;      // twiddle tables per (configuration, shift)
	std::map<std::pair<const void*, long>, DevBuf> sig;          // per pointwise table seen: its real half if it is real and mirror-symmetric (else empty)
	int ncu = 0;
};

// PXS_THETA_LINE=0 keeps the stage chains (read per call: the tests compare the two paths)
static bool line_enabled() { const char* e = getenv("PXS_THETA_LINE"); return e ? atoi(e) != 0 : true; }

bool FftChain::line_takes(const ThetaPlan& tp, bool has_mid) {
	if (!tp.ok || !line_enabled()) return false;
	for (const LineEntry& c : LINE_CONFIGS) if (c.N == tp.N && c.Ncc == tp.Ncc && c.M == (has_mid ? tp.M : 0)) return true;
	return false;
}

bool FftChain::line_analysis(hipStream_t st, const ThetaPlan& tp, bool has_mid, const double2* leg, long ldleg, int nr, int mir_c, double2* leg_cc, long ldcc, int ncc,
                             int nc, int nm, int spin, int lmax, const double2* ph_shift, const double2* sigma, const double2* w, const double2* wring)
{
	if (!line_enabled()) return false;
	const LineEntry* e = nullptr;
	for (const LineEntry& c : LINE_CONFIGS) if (c.N == tp.N && c.Ncc == tp.Ncc && c.M == (has_mid ? tp.M : 0)) e = &c;
	if (!e || nr > tp.N/2 + 1 || ncc != tp.Ncc/2 + 1) return false;
	const double2* tw; const double* sig_half = nullptr;
	{	std::lock_guard<std::mutex> g(mu_);
		if (!tl_) tl_ = std::make_shared<ThetaLine>();
		DevBuf& b = tl_->tw[std::make_tuple(e->N, e->M, e->Ncc, mir_c)];
		if (!b.p) {	// per length lo[l] = W_n^l, l < 128, hi[h] = W_n^{128 h}; the shift phase e^{-i k theta_0} = W_2N^{c k}, k <= N/2
			std::vector<double2> t((size_t)e->ntw, make_double2(1, 0));
			const long double tpi = 6.283185307179586476925286766559L;
			auto put = [&](int off, long n, long mult, long count) {
				for (int l = 0; l < RF_TWL; l++) { const long double ang = tpi*(long double)((mult*l) % n)/(long double)n; t[off + l] = make_double2((double)cosl(ang), (double)(-sinl(ang))); }
				for (long h = 0; h*RF_TWL < count; h++) { const long double ang = tpi*(long double)((mult*h*RF_TWL) % n)/(long double)n; t[off + RF_TWL + h] = make_double2((double)cosl(ang), (double)(-sinl(ang))); }
			};
			put(e->twN, e->N, 1, e->N); if (e->M > 0) put(e->twM, e->M, 1, e->M); put(e->twC, e->Ncc, 1, e->Ncc);
			put(e->twP, 2*e->N, mir_c, e->N/2 + 1);
			b = upload(t);
		}
		tw = b.as<double2>();
		if (has_mid) {	// the pointwise table: is it real and mirror-symmetric (the quadrature weights of the default analysis are)?  Looked at once
			// per table -- a table belongs to its plan and is never rewritten -- with one copy to the host.
			auto key = std::make_pair((const void*)sigma, (long)e->M);
			auto it = tl_->sig.find(key);
			if (it == tl_->sig.end()) {
				std::vector<double2> hs((size_t)e->M);
				PXS_HIP(hipStreamSynchronize(st));
				PXS_HIP(hipMemcpy(hs.data(), sigma, sizeof(double2)*hs.size(), hipMemcpyDeviceToHost));
				bool sym = true;
				for (long i = 0; i < e->M && sym; i++) sym = hs[i].y == 0.0 && hs[i].x == hs[(e->M - i) % e->M].x;
				DevBuf hb;
				if (sym) { std::vector<double> half((size_t)e->M/2 + 1); for (size_t i = 0; i < half.size(); i++) half[i] = hs[i].x; hb = upload(half); }
				it = tl_->sig.emplace(key, std::move(hb)).first;
			}
			sig_half = it->second.as<double>();
		}
		if (tl_->ncu == 0) {
#ifdef PXS_HOST_SIM
			tl_->ncu = 2;
#else
			int dev = 0; PXS_HIP(hipGetDevice(&dev));
			hipDeviceProp_t pr; PXS_HIP(hipGetDeviceProperties(&pr, dev)); tl_->ncu = std::max(1, pr.multiProcessorCount);
#endif
		}
	}
	(void)ph_shift;      // (the table of e^{-i k theta_0}: the engine keeps its own two-level form in the LDS)
	const long npair = (nm + 1)/2;
	LineArgs a; memset(&a, 0, sizeof(a));
	a.tw = tw; a.ntw = e->ntw; a.has_ph = mir_c != 0 ? 1 : 0;
	a.leg = leg; a.cstride = (long)nm*ldleg; a.ldleg = ldleg; a.nr = nr; a.mir_c = mir_c; a.wring = wring;
	a.lmax = lmax; a.sigma = sigma; a.sig_half = sig_half;
#ifdef PXS_LAB_TL_NOSIGHALF
	a.sig_half = nullptr;
#endif
	a.nr_out = ncc; a.a_odd = spin & 1; a.ncol = nm; a.npair = (int)npair; a.dnp = make_fastdiv((uint32_t)npair);
	const long ntask = (long)nc*npair;
	PXS_REQUIRE(ntask < (1L << 31), "internal: theta line grid too large");
	a.ntask = (int)ntask;
	a.out = leg_cc; a.ld = ldcc; a.ocstride = (long)nm*ldcc; a.w = w; a.scale = 1.0;
	const long per_cu = std::max<long>(1, std::min<long>(2048/e->nt, (long)(160*1024)/(long)e->lds));
	long nwg = std::min<long>(ntask, (long)tl_->ncu*per_cu);
	if (const char* ev = lab_getenv("PXS_TL_NWG")) nwg = std::max<long>(1, std::min<long>(nwg, atol(ev)));      // (lab builds: fewer workgroups, to see a line's phases without the other CUs' traffic)
#ifdef PXS_LAB_TL_TIME
	static DevBuf prof(16*sizeof(unsigned long long));
	PXS_HIP(hipMemsetAsync(prof.p, 0, prof.bytes, st)); a.prof = prof.as<unsigned long long>();
#endif
	e->launch(a, e->lds, nwg, st);
	PXS_HIP(hipGetLastError());
#ifdef PXS_LAB_TL_TIME
	{	unsigned long long h[16]; PXS_HIP(hipStreamSynchronize(st)); PXS_HIP(hipMemcpy(h, prof.p, sizeof(h), hipMemcpyDeviceToHost));
		static const char* nm[9] = {"load", "fftN", "resize1", "fftMi", "sigma", "fftMf", "resize2", "fftC", "split+store"};
		double tot = 0; for (int k = 0; k < 9; k++) tot += (double)h[k];
		fprintf(stderr, "[pxsht lab] theta line, workgroup 0, last wave, %llu lines, shader clocks per line:", h[9]);
		for (int k = 0; k < 9; k++) fprintf(stderr, " %s %.0f (%.0f%%)", nm[k], (double)h[k]/std::max<double>(1, (double)h[9]), 100.0*h[k]/std::max(tot, 1.0));
		fprintf(stderr, "\n"); }
#endif
	return true;
}

} // namespace pxs
