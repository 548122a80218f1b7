// Device-side building blocks of the LDS FFT kernels (shared by fft.hip and fftchain.hip).
#pragma once
#include "common.hpp"
#include "fft_radix_tw.hpp"

namespace pxs {

static constexpr int FFT_NLOC_MAX = 2048;   // longest line done in one LDS pass
static constexpr int FFT_LDS_PTS  = 4096;   // complex points of LDS per workgroup (64 KiB)
static constexpr int FFT_MAXFAC   = 16;

struct PassDesc { int R; int L; int tws; FastDiv dL; FastDiv dnb; };

// LDS line buffer addressing hook (see below)
#define LPAD(i) (i)   /* padding ((i)+((i)>>4)) measured neutral-to-slower on MI355X: global latency, not LDS banks, bounds this kernel */

__device__ __forceinline__ uint32_t fdiv(uint32_t x, FastDiv f) { return f.d <= 1 ? x : __umulhi(x, f.mul); }
__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x*b.x - a.y*b.y, a.x*b.y + a.y*b.x); }
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x+b.x, a.y+b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x-b.x, a.y-b.y); }
__device__ __forceinline__ double2 cconj(double2 a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ double2 mulmi(double2 a) { return make_double2(a.y, -a.x); }  // a * (-i)

template<int R> __device__ __forceinline__ void butterfly(double2* v);
template<> __device__ __forceinline__ void butterfly<2>(double2* v) {
	double2 a = v[0], b = v[1]; v[0] = cadd(a, b); v[1] = csub(a, b);
}
template<> __device__ __forceinline__ void butterfly<3>(double2* v) {
	const double s = 0.86602540378443864676;
	double2 t1 = cadd(v[1], v[2]);
	double2 t2 = make_double2(v[0].x - 0.5*t1.x, v[0].y - 0.5*t1.y);
	double2 d = csub(v[1], v[2]);
	double2 t3 = make_double2(s*d.y, -s*d.x);   // -i*s*d
	v[0] = cadd(v[0], t1); v[1] = cadd(t2, t3); v[2] = csub(t2, t3);
}
template<> __device__ __forceinline__ void butterfly<4>(double2* v) {
	double2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
	double2 t2 = cadd(v[1], v[3]), t3 = mulmi(csub(v[1], v[3]));
	v[0] = cadd(t0, t2); v[1] = cadd(t1, t3); v[2] = csub(t0, t2); v[3] = csub(t1, t3);
}
template<> __device__ __forceinline__ void butterfly<5>(double2* v) {
	const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;
	const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;
	double2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]);
	double2 t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
	double2 a = v[0];
	double2 m1 = make_double2(a.x + c1*t1.x + c2*t2.x, a.y + c1*t1.y + c2*t2.y);
	double2 m2 = make_double2(a.x + c2*t1.x + c1*t2.x, a.y + c2*t1.y + c1*t2.y);
	double2 u1 = make_double2(s1*t3.x + s2*t4.x, s1*t3.y + s2*t4.y);
	double2 u2 = make_double2(s2*t3.x - s1*t4.x, s2*t3.y - s1*t4.y);
	double2 iu1 = mulmi(u1), iu2 = mulmi(u2);   // -i*u
	v[0] = make_double2(a.x + t1.x + t2.x, a.y + t1.y + t2.y);
	v[1] = cadd(m1, iu1); v[4] = csub(m1, iu1);
	v[2] = cadd(m2, iu2); v[3] = csub(m2, iu2);
}

// (radix 7: only in the theta stages of the chain kernels -- ducc0's ring counts good_size_complex(lmax + 1) often carry a factor 7,
// e.g. 10080 for lmax = 10000 and 4032 for lmax = 4000)
template<> __device__ __forceinline__ void butterfly<7>(double2* v) {
	const double c1 = 0.62348980185873353053, c2 = -0.22252093395631440429, c3 = -0.90096886790241912624;
	const double s1 = 0.78183148246802980871, s2 = 0.97492791218182360702, s3 = 0.43388373911755812048;
	const double2 a = v[0];
	const double2 t1 = cadd(v[1], v[6]), t2 = cadd(v[2], v[5]), t3 = cadd(v[3], v[4]);
	const double2 d1 = csub(v[1], v[6]), d2 = csub(v[2], v[5]), d3 = csub(v[3], v[4]);
	const double2 m1 = make_double2(a.x + c1*t1.x + c2*t2.x + c3*t3.x, a.y + c1*t1.y + c2*t2.y + c3*t3.y);
	const double2 m2 = make_double2(a.x + c2*t1.x + c3*t2.x + c1*t3.x, a.y + c2*t1.y + c3*t2.y + c1*t3.y);
	const double2 m3 = make_double2(a.x + c3*t1.x + c1*t2.x + c2*t3.x, a.y + c3*t1.y + c1*t2.y + c2*t3.y);
	const double2 u1 = make_double2(s1*d1.x + s2*d2.x + s3*d3.x, s1*d1.y + s2*d2.y + s3*d3.y);
	const double2 u2 = make_double2(s2*d1.x - s3*d2.x - s1*d3.x, s2*d1.y - s3*d2.y - s1*d3.y);
	const double2 u3 = make_double2(s3*d1.x - s1*d2.x + s2*d3.x, s3*d1.y - s1*d2.y + s2*d3.y);
	const double2 iu1 = mulmi(u1), iu2 = mulmi(u2), iu3 = mulmi(u3);   // -i*u
	v[0] = make_double2(a.x + t1.x + t2.x + t3.x, a.y + t1.y + t2.y + t3.y);
	v[1] = cadd(m1, iu1); v[6] = csub(m1, iu1);
	v[2] = cadd(m2, iu2); v[5] = csub(m2, iu2);
	v[3] = cadd(m3, iu3); v[4] = csub(m3, iu3);
}

// workgroup barrier that only waits for LDS traffic: __syncthreads() also drains vmcnt, which would stall on the
// global loads the pipelined kernel keeps in flight for its next tile
#ifdef PXS_HOST_SIM
#define PXS_LDS_BARRIER() __syncthreads()
#else
#define PXS_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif


// Sub-transform view used by the chain kernels (fftchain.hip): one in-place mixed-radix DIT FFT of T lines of n points
// (line stride ns) held in LDS; input must sit at its digit-reversed slot perm[j].
struct LdsFft {
	int n, nfac, ns, generic;
	const PassDesc* pass; const int* perm; const double2* tw;   // device tables of the engine (FftSub)
	FastDiv dn;
};

template<int R, int NT> __device__ __forceinline__ void radix_pass_t(double2* buf, const double2* tw, int n, int ns, int T, const PassDesc& ps) {
	const int nb = n / R;
	const int total = T*nb;
	for (int b = threadIdx.x; b < total; b += NT) {
		const uint32_t t = fdiv(b, ps.dnb);
		const uint32_t bb = b - t*nb;
		const uint32_t blk = fdiv(bb, ps.dL);
		const uint32_t q = bb - blk*ps.L;
		const uint32_t p0 = t*ns + blk*ps.L*R + q;
		double2 v[R];
#pragma unroll
		for (int i = 0; i < R; i++) v[i] = buf[p0 + i*ps.L];
		if (ps.L > 1) {
			const int step = q*ps.tws;
#pragma unroll
			for (int i = 1; i < R; i++) v[i] = cmul(v[i], tw[i*step]);
		}
		butterfly<R>(v);
#pragma unroll
		for (int i = 0; i < R; i++) buf[p0 + i*ps.L] = v[i];
	}
}

// Composite radices R = A*B (6, 8, 9, 10, 12, 15, 16) done in registers: B butterflies of radix A over stride B, the constant
// twiddles W_R^{n2 k1}, A butterflies of radix B.  Output X[k1 + A k2] ends up in v[B k1 + k2].  One LDS round trip (and one
// barrier) then does the work of two passes of the plain radices.
// Only radices up to PXS_COMP_MAXR are compiled in.  Measured on MI355X (C3 bench, FFT stages of a round trip, same box):
// plain radices 112.5 ms (48-54 VGPRs), up to 8: 108.4 ms (62-67 VGPRs), up to 10: 107.9 ms (75-79 VGPRs), up to 16: 139 ms --
// at 107-112 VGPRs a 512-thread workgroup fits only twice on a CU and even tiles that never take the radix-16 path slow
// down by a third.  9 keeps every chain kernel at <= 69 VGPRs.  lds_fft<NT, MAXR> lets a kernel leave out the larger ones: the
// ring stages of fftchain.hip compile radices up to 8 only and fit 64 VGPRs (8 waves per SIMD): ring FFTs 49.3 -> 46.4 ms at C3,
// 68.2 -> 60.8 ms at C4.
#ifndef PXS_COMP_MAXR
#define PXS_COMP_MAXR 9
#endif
template<int R> struct RadixTw;
template<> struct RadixTw<6>  { static __device__ __forceinline__ double2 w(int j) { return make_double2(PXS_RTW6[2*j],  PXS_RTW6[2*j+1]); } };
template<> struct RadixTw<8>  { static __device__ __forceinline__ double2 w(int j) { return make_double2(PXS_RTW8[2*j],  PXS_RTW8[2*j+1]); } };
template<> struct RadixTw<9>  { static __device__ __forceinline__ double2 w(int j) { return make_double2(PXS_RTW9[2*j],  PXS_RTW9[2*j+1]); } };
template<> struct RadixTw<10> { static __device__ __forceinline__ double2 w(int j) { return make_double2(PXS_RTW10[2*j], PXS_RTW10[2*j+1]); } };
template<> struct RadixTw<12> { static __device__ __forceinline__ double2 w(int j) { return make_double2(PXS_RTW12[2*j], PXS_RTW12[2*j+1]); } };
template<> struct RadixTw<15> { static __device__ __forceinline__ double2 w(int j) { return make_double2(PXS_RTW15[2*j], PXS_RTW15[2*j+1]); } };
template<> struct RadixTw<16> { static __device__ __forceinline__ double2 w(int j) { return make_double2(PXS_RTW16[2*j], PXS_RTW16[2*j+1]); } };

template<int A, int B> __device__ __forceinline__ void butterfly_comp(double2* v) {
#pragma unroll
	for (int n2 = 0; n2 < B; n2++) {
		double2 t[A];
#pragma unroll
		for (int n1 = 0; n1 < A; n1++) t[n1] = v[B*n1 + n2];
		butterfly<A>(t);
#pragma unroll
		for (int k1 = 0; k1 < A; k1++) v[B*k1 + n2] = (n2*k1 == 0) ? t[k1] : cmul(t[k1], RadixTw<A*B>::w(n2*k1));
	}
#pragma unroll
	for (int k1 = 0; k1 < A; k1++) butterfly<B>(v + B*k1);
}

template<int A, int B, int NT> __device__ __forceinline__ void radix_pass_comp(double2* buf, const double2* tw, int n, int ns, int T, const PassDesc& ps) {
	constexpr int R = A*B;
	const int nb = n / R;
	const int total = T*nb;
	for (int b = threadIdx.x; b < total; b += NT) {
		const uint32_t t = fdiv(b, ps.dnb);
		const uint32_t bb = b - t*nb;
		const uint32_t blk = fdiv(bb, ps.dL);
		const uint32_t q = bb - blk*ps.L;
		const uint32_t p0 = t*ns + blk*ps.L*R + q;
		double2 v[R];
#pragma unroll
		for (int i = 0; i < R; i++) v[i] = buf[p0 + i*ps.L];
		if (ps.L > 1) {
			const int step = q*ps.tws;
#pragma unroll
			for (int i = 1; i < R; i++) v[i] = cmul(v[i], tw[i*step]);
		}
		butterfly_comp<A, B>(v);
#pragma unroll
		for (int k = 0; k < R; k++) buf[p0 + k*ps.L] = v[B*(k % A) + k/A];
	}
}

// all passes of f on T lines (radices 2,3,4,5 and the composite ones); ends with a barrier
template<int NT, int MAXR = PXS_COMP_MAXR, bool R7 = true> __device__ __forceinline__ void lds_fft(double2* buf, const double2* tw, const LdsFft& f, int T) {
	static_assert(MAXR <= PXS_COMP_MAXR, "composite radix not compiled in");
	for (int p = 0; p < f.nfac; p++) {
		const PassDesc ps = f.pass[p];
		switch (ps.R) {
			case 2: radix_pass_t<2, NT>(buf, tw, f.n, f.ns, T, ps); break;
			case 3: radix_pass_t<3, NT>(buf, tw, f.n, f.ns, T, ps); break;
			case 4: radix_pass_t<4, NT>(buf, tw, f.n, f.ns, T, ps); break;
			case 5: radix_pass_t<5, NT>(buf, tw, f.n, f.ns, T, ps); break;
			case 6: if constexpr (MAXR >= 6) radix_pass_comp<3, 2, NT>(buf, tw, f.n, f.ns, T, ps); break;
#ifndef PXS_NO_RADIX7   /* (A/B builds without the radix-7 code: tools/fft2_ab.sh) */
			case 7: if constexpr (MAXR >= 9 && R7) radix_pass_t<7, NT>(buf, tw, f.n, f.ns, T, ps); break;      // (theta stages only; FftContext::sub plans a 7 only for maxr >= 9)
#endif
			case 8: if constexpr (MAXR >= 8) radix_pass_comp<4, 2, NT>(buf, tw, f.n, f.ns, T, ps); break;
			case 9: if constexpr (MAXR >= 9) radix_pass_comp<3, 3, NT>(buf, tw, f.n, f.ns, T, ps); break;
#if PXS_COMP_MAXR >= 10
			case 10: if constexpr (MAXR >= 10) radix_pass_comp<5, 2, NT>(buf, tw, f.n, f.ns, T, ps); break;
#endif
#if PXS_COMP_MAXR >= 12
			case 12: if constexpr (MAXR >= 12) radix_pass_comp<4, 3, NT>(buf, tw, f.n, f.ns, T, ps); break;
#endif
#if PXS_COMP_MAXR >= 15
			case 15: if constexpr (MAXR >= 15) radix_pass_comp<5, 3, NT>(buf, tw, f.n, f.ns, T, ps); break;
#endif
#if PXS_COMP_MAXR >= 16
			case 16: if constexpr (MAXR >= 16) radix_pass_comp<4, 4, NT>(buf, tw, f.n, f.ns, T, ps); break;
#endif
			default: break;
		}
		PXS_LDS_BARRIER();
	}
}

} // namespace pxs
