// Device-side pieces shared by the fused chain stages (fftchain.hip) and the single-kernel theta engine (thetaline.hip).
#pragma once
#include "fft_dev.hpp"

namespace pxs {

__device__ __forceinline__ double2 cscale(double2 a, double f) { return make_double2(a.x*f, a.y*f); }
__device__ __forceinline__ double2 rd_real(const void* p, int dtype, long off) {
	return dtype == PX_F32 ? make_double2((double)((const float*)p)[off], 0.0) : make_double2(((const double*)p)[off], 0.0);
}
__device__ __forceinline__ void wr_real(void* p, int dtype, long off, double v) {
	if (dtype == PX_F32) ((float*)p)[off] = (float)v; else ((double*)p)[off] = v;
}

// value of circle sample j of the packed pair of columns (2p, 2p+1): even + odd extension (cf. LD_MIRROR_PAIR in fft.hip)
struct PairSrc {
	const double2* leg; long ld; int nr; int N; int mir_c; int a_odd; int ncol; long cstride;      // cstride: elements between the components of a launch
	const double2* w;        // optional per-ring weight (.x), applied to a ring sample and to its mirror image
	int plain, conj;         // plain: no packing, no extension: column p itself (the 2-D FFTs); conj: conjugated (backward transform as conj FFT conj)
	__device__ __forceinline__ double2 get(int comp, int p, int j) const {
		if (plain) { const double2 v = leg[(long)comp*cstride + (long)p*ld + j]; return conj ? cconj(v) : v; }
		int src = j; bool mir = false;
		if (j >= nr) { src = N - j - mir_c; if (src < 0) src += N; mir = true; }
		const int tj = 2*j + mir_c;
		const bool selfm = tj == 0 || tj == N || tj == 2*N;       // the sample is its own mirror image
		const int ca = 2*p;
		const double2* lc = leg + (long)comp*cstride;
		double2 va = lc[(long)ca*ld + src];
		double2 vb = (ca + 1 < ncol) ? lc[(long)(ca + 1)*ld + src] : make_double2(0, 0);
		double2& vo = a_odd ? va : vb;
		if (selfm) vo = make_double2(0, 0);
		else if (mir) { vo.x = -vo.x; vo.y = -vo.y; }
		const double2 sum = cadd(va, vb);
		return w ? cscale(sum, w[src].x) : sum;
	}
};

} // namespace pxs
