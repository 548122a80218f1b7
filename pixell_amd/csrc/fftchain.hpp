// Fused FFT chains of the SHT for gfx950 (see fftchain.hip): ring FFTs (map <-> leg) and the exact theta resampling
// between the map's rings and the minimal Clenshaw-Curtis grid, as sequences of LDS kernels in which every intermediate
// array is written once and read once.
#pragma once
#include "fft.hpp"
#include <tuple>
#include <map>
#include <mutex>
#include <memory>

namespace pxs {

struct Lds2;

// factorisation N = a*b of a four-step transform: pass 1 = a-point transforms over the residues mod b,
// pass 2 = b-point transforms producing the residues mod a
struct Split { long a = 0, b = 0; };

struct ThetaPlan {            // sizes of the theta chains of a grid plan
	bool ok = false;
	long N = 0, M = 0, Ncc = 0;      // map circle, fine circle (|sin| product), CC circle
	long g = 0, bN = 0, g2 = 0, ac = 0;     // analysis: N = g*bN, M = g*g2 (both ways), Ncc = ac*g
	long gs = 0, bs = 0, aNs = 0;            // synthesis: Ncc = gs*bs, N = aNs*gs
};

class FftChain {
public:
	explicit FftChain(FftContext* fc) : fc_(fc) {}
	// can the engine run its LDS passes on lines of this length (radices 2,3,4,5, length <= 512)?
	static bool sub_ok(long n);
	static bool sub_ok_theta(long n); // ... of the theta stages: radix 7 as well
	static long pad8(long n) { return (n + 7) & ~7L; }
	// ring FFT split of nphi for analysis (map -> leg) and synthesis (h -> map); false: no usable factorisation
	bool plan_rings(long nphi);
	// theta chain sizes for a grid with N circle samples; picks Ncc (even, > 2 lmax + 1) and M (> N + 2 lmax)
	static ThetaPlan plan_theta(long N, int lmax);
	static long ducc_ncc(int lmax);     // 2 good_size_complex(lmax + 1): the circle of the CC grid ducc0 runs its Legendre stage on
	bool rings_ok() const { return ra_.a > 0; }
	std::string describe() const { return "analysis " + std::to_string(ra_.a) + "x" + std::to_string(ra_.b) + ", synthesis " + std::to_string(rs_.a) + "x" + std::to_string(rs_.b); }

	struct MapDesc { const void* ptr; int dtype; long cstride, ring_off0, ring_stride, pix_stride; int nring; long nphi;
	                 long bstride = 0; int ncb = 0; };   // batches: component index k of a call = b*ncb + c lives at b*bstride + c*cstride (ncb = 0: no batch axis)
	// map -> leg[c][m][ring] * tab[m] * scale   (two real rings per complex transform; needs 2 mmax < nphi)
	void map2leg(hipStream_t st, const MapDesc& m, int nc, int mmax, double2* leg, long ldleg, const double2* tab, double scale);
	// h[c][ring][m] (row stride ldh) -> map
	// (hcomp: rows of h per component when the map's rings are a row range of a larger h; 0 = the map's ring count)
	void h2map(hipStream_t st, const double2* h, long ldh, const MapDesc& m, int nc, int mmax, long hcomp = 0);
	// leg on the map's rings -> quadrature-weighted leg on the CC grid (columns paired by parity)
	void to_cc(hipStream_t st, const ThetaPlan& tp, const double2* leg, long ldleg, int nr, int mir_c, double2* leg_cc, long ldcc, int ncc,
	           int nc, int nm, int spin, int lmax, const double2* ph_shift, const double2* sigma, const double2* wcc);
	// band-limited leg on the CC grid -> h[c][ring][m] * conj(tab[m]) * scale on the map's rings
	void from_cc(hipStream_t st, const ThetaPlan& tp, const double2* leg_cc, long ldcc, int ncc, double2* h, long ldh, int nr, int mir_c,
	             int nc, int nm, int spin, int lmax, const double2* ph_up, const double2* tab, double scale, const double2* wring = nullptr);      // wring: optional weight (.x) per output ring
	// exact transpose of from_cc for grids without self-mirrored rings: leg on the map's rings -> leg on the CC grid (w: 1/N_cc, half at the poles)
	void from_cc_adjoint(hipStream_t st, const ThetaPlan& tp, const double2* leg, long ldleg, int nr, int mir_c, double2* leg_cc, long ldcc, int ncc,
	                     int nc, int nm, int spin, int lmax, const double2* ph_shift, const double2* w, const double2* wring = nullptr);      // wring: optional weight (.x) per input ring
	// exact adjoint of to_cc: leg on the CC grid -> h[c][ring][m] * conj(tab[m]) * scale on the map's rings (whalf: the to_cc weights, halved off the poles)
	void to_cc_adjoint(hipStream_t st, const ThetaPlan& tp, const double2* leg_cc, long ldcc, int ncc, double2* h, long ldh, int nr, int mir_c,
	                   int nc, int nm, int spin, int lmax, const double2* ph_shift, const double2* sigma, const double2* whalf, const double2* tab, double scale);
	// 2-D FFT of real [npre][ny][nx] (float32 / float64) into complex128 [npre][ny][nx]; false if nx or ny has no usable factorisation
	bool fft2_real(hipStream_t st, const void* in, int in_dtype, double2* out, long npre, long ny, long nx, bool forward, double scale);
	// 2-D FFT of complex128 [npre][ny][nx] (in == out allowed); false if nx or ny has no usable factorisation
	bool fft2_c2c(hipStream_t st, const double2* in, double2* out, long npre, long ny, long nx, bool forward, double scale);
	// single-kernel form of to_cc (has_mid) / from_cc_adjoint for lines that fit a CU (thetaline.hip): one workgroup per pair of columns,
	// no HBM intermediates.  false: not eligible (sizes, radices) or switched off (PXS_THETA_LINE=0) -- the caller runs the stage chain.
	bool line_analysis(hipStream_t st, const ThetaPlan& tp, bool has_mid, const double2* leg, long ldleg, int nr, int mir_c, double2* leg_cc, long ldcc, int ncc,
	                   int nc, int nm, int spin, int lmax, const double2* ph_shift, const double2* sigma, const double2* w, const double2* wring);
	static bool line_takes(const ThetaPlan& tp, bool has_mid);      // ... would it (then the call needs no theta scratch)
	// single-kernel form of h2map for ring lengths compiled into ringline.hip: one workgroup per ring pair, no HBM intermediate
	bool line_h2map(hipStream_t st, const double2* h, long ldh, const MapDesc& m, int nc, int mmax, long hcomp);
	static bool line_h2map_takes(long nphi, int mmax);
	size_t scratch_bytes() const { return s1_.bytes + s2_.bytes; }
	// scratch a call needs, so that the plan can size it before the first launch of the call (kind 0: to_cc, 1: from_cc_adjoint, 2: from_cc, 3: to_cc_adjoint)
	static void theta_scratch(const ThetaPlan& tp, int nm, int nc, int kind, size_t& b1, size_t& b2);
	void ring_scratch(long nring, int nc, bool analysis, size_t& b1, int mmax = -1) const;      // (mmax given: 0 for a synthesis that ringline.hip takes)
	void reserve(size_t b1, size_t b2) { s1_.ensure(b1); s2_.ensure(b2); }
private:
	const double2* small_tw(long X, int n, int T);
	template<class S> void set_tiles(S& s, int T, long nlines, long X);
	template<class S> int tile_lines_for(long n_a, long n_b, long nlines, int mult, long tab_pts = -1);
	template<class S> void launch_any(S& s, long nblk, hipStream_t st);
	std::mutex mu_;
	std::map<std::tuple<long, int, int>, DevBuf> stw_;
	FftContext* fc_;
	Split ra_, rs_;           // ring FFT splits: analysis, synthesis
	long nphi_ = 0;
	DevBuf s1_, s2_;          // ping-pong scratch
	std::shared_ptr<struct ThetaLine> tl_;     // plans of the single-kernel theta engine
};

// where the pixels of a map live, in the form the ring-FFT kernels take it (fftchain.hip, ringline.hip)
struct MapAddr { void* ptr; int dtype; long cstride, bstride, off0, rstride, pstride; int nring, ncb; FastDiv dncb; };   // "component" index = b*ncb + c
inline MapAddr map_addr(const FftChain::MapDesc& m) {
	MapAddr a; a.ptr = const_cast<void*>(m.ptr); a.dtype = m.dtype; a.cstride = m.cstride; a.bstride = m.bstride; a.off0 = m.ring_off0; a.rstride = m.ring_stride; a.pstride = m.pix_stride; a.nring = m.nring;
	a.ncb = m.ncb > 0 ? m.ncb : (1 << 30); a.dncb = make_fastdiv((uint32_t)a.ncb);
	return a;
}

} // namespace pxs
