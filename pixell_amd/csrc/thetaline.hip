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
	int nr_out, a_odd, ncol, npair, ntask; FastDiv dnp;
	double2* out; long ld, ocstride; const double2* w; double scale;
};

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
	static constexpr int pad(int i) { return i + (i >> 4); }
	static constexpr int words = tl_max(tl_max(pad(N), pad(M)), 2*pad(Ncc)) + 8;      // doubles of the line area
	static constexpr size_t lds = sizeof(double2)*ntw + sizeof(double)*words + 16;
	static_assert(!MID || SMi::N == SMf::N, "the two transforms on M differ in length");
	static_assert(N/2 + 1 <= words/2, "a row of leg does not fit the line area");
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
	// loads, every element read from memory once -- and lands in the registers in the read pattern of pass PN
	template<class PN> static __device__ __forceinline__ void load_pair(Regs v, int tid, double2* line2, const LineArgs& a, int comp, int pr) {
		static_assert(PN::slots <= PMAX, "pass needs more register slots than the kernel has");
		constexpr int N = CFG::N;
		const int ca = 2*pr;
		const double2* ra = a.leg + (long)comp*a.cstride + (long)ca*a.ldleg;
		const int nrow = (ca + 1 < a.ncol) ? 2 : 1;
#pragma unroll 1
		for (int row = 0; row < 2; row++) {
			const double2* r = ra + (long)row*a.ldleg;
			const bool odd = row == 0 ? a.a_odd != 0 : a.a_odd == 0;      // this column is odd under the reflection
			RF_BARRIER();
			if (row < nrow) for (int i = tid; i < a.nr; i += NT) { double2 x = r[i]; if (a.wring) x = cscale(x, a.wring[i].x); line2[i] = x; }
			else for (int i = tid; i < a.nr; i += NT) line2[i] = make_double2(0, 0);      // (an odd number of columns: the last pair has one)
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
		}
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
	const int tid = threadIdx.x;
	for (int k = tid; k < CFG::ntw; k += NT) tws[k] = a.tw[k];
	const double2* ph = a.has_ph ? tws + CFG::twP : nullptr;
	for (int task = blockIdx.x; task < a.ntask; task += gridDim.x) {
		const int comp = (int)fdiv((uint32_t)task, a.dnp), pr = task - comp*a.npair;
		double2 v[PMAX];
		L::template load_pair<RfPassT<SN, 0, NT>>(v, tid, line2, a, comp, pr);
		F::template run<SN>(v, tid, line, tws + CFG::twN);
		using NL = RfPassT<SN, SN::NP - 1, NT>;
		if constexpr (CFG::MID) {
			constexpr int N = CFG::N, M = CFG::M;
			ResizeRule r1; r1.X1 = N; r1.X2 = M; r1.kmax = M > N ? -1 : M/2 - 1; r1.nyq = M > N ? 1 : 0;
			L::template resize<NL, RfPassT<SMi, 0, NT>>(v, tid, line, r1, ph);
			F::template run<SMi>(v, tid, line, tws + CFG::twM);
			{	// the pointwise table on the samples where they are; if the backward transform ends with the radix the forward one starts
				// with, the registers are in place for it, else one exchange
				using ML = RfPassT<SMi, SMi::NP - 1, NT>; using MF = RfPassT<SMf, 0, NT>;
				const double2* sg = a.sigma;
				F::template pointwise<ML>(v, tid, [&](double2 x, int idx) { return cmul(cconj(x), sg[idx]); });
				if constexpr (ML::R != MF::R) F::template exchange<ML, MF>(v, tid, line); }
			F::template run<SMf>(v, tid, line, tws + CFG::twM);
			ResizeRule r2; r2.X1 = M; r2.X2 = CFG::Ncc; r2.kmax = a.lmax; r2.nyq = 0;
			L::template resize<RfPassT<SMf, SMf::NP - 1, NT>, RfPassT<SC, 0, NT>>(v, tid, line, r2, nullptr);
		} else {
			ResizeRule r1; r1.X1 = CFG::N; r1.X2 = CFG::Ncc; r1.kmax = a.lmax; r1.nyq = 0;
			L::template resize<NL, RfPassT<SC, 0, NT>>(v, tid, line, r1, ph);
		}
		F::template run<SC>(v, tid, line, tws + CFG::twC);
		// the circle of Ncc points, both components (the line area is sized for it), then the separation of the pair by reflection
		// symmetry (StSplit<0> of the stage chain)
		RF_BARRIER();
		F::template write_c128<RfPassT<SC, SC::NP - 1, NT>>(v, tid, line2);
		RF_BARRIER();
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
// (the simulator runs one OS thread per lane: 64-thread workgroups on a grid of 360 rings, lmax 250 -- every feature of the large
// configuration at a fifteenth of its size: 2 and 3 butterflies per thread, radix 7, M > N)
using CfgSimA = LineCfg<64, RfSeq<12, 10, 6>, RfSeq<7, 9, 16>, RfSeq<16, 9, 7>, RfSeq<12, 7, 6>>;
using CfgSimB = LineCfg<64, RfSeq<12, 10, 6>, RfSeq<>, RfSeq<>, RfSeq<12, 7, 6>>;
static const LineEntry LINE_CONFIGS[] = { entry_of<CfgSimA>(), entry_of<CfgSimB>() };
#else
// (radices up to 9 with two or three butterflies per thread: with radix 16 / 15 the compiler needs ~2x the registers of a butterfly --
// inputs, outputs and the twiddle powers at once -- and spills 60-170 times per transform at the 128 registers of a 1024-thread
// workgroup; these sequences spill 0-8 times.  A fifth pass per transform costs one more LDS exchange.)
using CfgC4A = LineCfg<1024, RfSeq<8, 6, 5, 5, 9>, RfSeq<8, 8, 6, 6, 7>, RfSeq<8, 8, 6, 6, 7>, RfSeq<8, 8, 6, 3, 7>>;
using CfgC4B = LineCfg<1024, RfSeq<8, 6, 5, 5, 9>, RfSeq<>, RfSeq<>, RfSeq<8, 8, 6, 3, 7>>;
static const LineEntry LINE_CONFIGS[] = { entry_of<CfgC4A>(), entry_of<CfgC4B>() };
#endif

struct ThetaLine {
	std::map<std::tuple<long, long, long, int>, DevBuf> tw;      // twiddle tables per (configuration, shift)
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
	const double2* tw;
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
	a.lmax = lmax; a.sigma = sigma;
	a.nr_out = ncc; a.a_odd = spin & 1; a.ncol = nm; a.npair = (int)npair; a.dnp = make_fastdiv((uint32_t)npair);
	const long ntask = (long)nc*npair;
	PXS_REQUIRE(ntask < (1L << 31), "internal: theta line grid too large");
	a.ntask = (int)ntask;
	a.out = leg_cc; a.ld = ldcc; a.ocstride = (long)nm*ldcc; a.w = w; a.scale = 1.0;
	const long per_cu = std::max<long>(1, std::min<long>(2048/e->nt, (long)(160*1024)/(long)e->lds));
	const long nwg = std::min<long>(ntask, (long)tl_->ncu*per_cu);
	e->launch(a, e->lds, nwg, st);
	PXS_HIP(hipGetLastError());
	return true;
}

} // namespace pxs
