// Streaming alm post-processing on the GPU: alm2cl, lmul (almxfl), lmatmul.
// Replaces cython/cmisc_core.c:16-110 (alm2cl_*), :159-182 (lmul_*), :185-274 (lmatmul_*) of the
// reference, as reached through alm_info.alm2cl / alm_info.lmul (pixell/curvedsky.py:451-474) and
// curvedsky.almxfl / alm2cl (:630-712).  HBM-bound: every alm element is touched once; lanes run
// along l so that loads of one m-column are contiguous (lstride 1).
#include "../../include/pxsht.h"
#include "common.hpp"
#include <map>
#include <mutex>

namespace pxs {
const char* get_last_error();

__device__ __forceinline__ double2 ld_c(const void* p, int dtype, long i) {
	if (dtype == PX_C64) { float2 v = ((const float2*)p)[i]; return make_double2(v.x, v.y); }
	return ((const double2*)p)[i];
}
__device__ __forceinline__ void st_c(void* p, int dtype, long i, double2 v) {
	if (dtype == PX_C64) ((float2*)p)[i] = make_float2((float)v.x, (float)v.y);
	else ((double2*)p)[i] = v;
}

// cl[l] = 2/(2l+1) * ( a1_l0.re*a2_l0.re/2 + sum_{m>=1} Re(a1_lm conj(a2_lm)) )    (cmisc_core.c:16-46)
__global__ __launch_bounds__(256) void alm2cl_kernel(int lmax, int mmax, const uint64_t* __restrict__ mstart, long lstride,
		const void* __restrict__ a1, const void* __restrict__ a2, int dtype, void* __restrict__ cl, int cl_dtype, int acc_f32)
{
	const int l = blockIdx.x*blockDim.x + threadIdx.x;
	if (l > lmax) return;
	const int mtop = min(l, mmax);
	if (dtype == PX_C64) {
		// single precision alm: the reference forms each term in float (cmisc_core.c:27,33 / :58,64) and only the
		// running sum is float (alm2cl_sp) or double (alm2cl_sp_to_dp); mirrored here, without fma contraction
		const float2* __restrict__ p1 = (const float2*)a1; const float2* __restrict__ p2 = (const float2*)a2;
		const long i0 = (long)mstart[0] + l*lstride;
		const float t0 = __fmul_rn(p1[i0].x, p2[i0].x)/2;
		float sf = t0; double sd = t0;
		for (int m = 1; m <= mtop; m++) {
			const long i = (long)mstart[m] + l*lstride;
			const float2 x = p1[i], y = p2[i];
			const float t = __fadd_rn(__fmul_rn(x.x, y.x), __fmul_rn(x.y, y.y));
			if (acc_f32) sf = __fadd_rn(sf, t); else sd += t;
		}
		if (acc_f32) { sf = (float)(sf*(2.0/(2*l+1))); if (cl_dtype == PX_F32) ((float*)cl)[l] = sf; else ((double*)cl)[l] = sf; }
		else { sd *= 2.0/(2*l+1); if (cl_dtype == PX_F32) ((float*)cl)[l] = (float)sd; else ((double*)cl)[l] = sd; }
		return;
	}
	double s;
	{ const double2 x = ld_c(a1, dtype, (long)mstart[0] + l*lstride), y = ld_c(a2, dtype, (long)mstart[0] + l*lstride); s = x.x*y.x/2; }
	for (int m = 1; m <= mtop; m++) {
		const long i = (long)mstart[m] + l*lstride;
		const double2 x = ld_c(a1, dtype, i), y = ld_c(a2, dtype, i);
		s += x.x*y.x + x.y*y.y;
	}
	s *= 2.0/(2*l+1);
	if (cl_dtype == PX_F32) ((float*)cl)[l] = (float)s; else ((double*)cl)[l] = s;
}

// double-accumulating variants, split over m: block (l tile, m chunk) -> part[chunk][l]; a second kernel adds the chunks in a fixed
// order (deterministic) and applies 2/(2l+1).  The one-thread-per-l kernel above walks 10^4 dependent loads per thread with only
// lmax+1 threads on the chip; this one exposes (lmax+1)(mmax+1)/MCH-fold parallelism and streams the alm once.
static constexpr int ALM2CL_MCH = 32;
__global__ __launch_bounds__(256) void alm2cl_part_kernel(int lmax, int mmax, const uint64_t* __restrict__ mstart, long lstride,
		const void* __restrict__ a1, const void* __restrict__ a2, int dtype, double* __restrict__ part)
{
	const int l = blockIdx.x*blockDim.x + threadIdx.x;
	const int m0 = blockIdx.y*ALM2CL_MCH;
	if (l > lmax) return;
	const int m1 = min(min(l, mmax), m0 + ALM2CL_MCH - 1);
	double s = 0;
	for (int m = m0; m <= m1; m++) {
		const long i = (long)mstart[m] + l*lstride;
		if (dtype == PX_C64) {     // terms formed in float as the reference does (cmisc_core.c:58,64), summed in double
			const float2 x = ((const float2*)a1)[i], y = ((const float2*)a2)[i];
			s += (m == 0) ? (double)(__fmul_rn(x.x, y.x)/2) : (double)__fadd_rn(__fmul_rn(x.x, y.x), __fmul_rn(x.y, y.y));
		} else {
			const double2 x = ((const double2*)a1)[i], y = ((const double2*)a2)[i];
			s += (m == 0) ? x.x*y.x/2 : x.x*y.x + x.y*y.y;
		}
	}
	part[(long)blockIdx.y*(lmax+1) + l] = s;
}
__global__ __launch_bounds__(256) void alm2cl_sum_kernel(int lmax, int mmax, const double* __restrict__ part, void* __restrict__ cl, int cl_dtype)
{
	const int l = blockIdx.x*blockDim.x + threadIdx.x;
	if (l > lmax) return;
	const int nch = min(l, mmax)/ALM2CL_MCH + 1;          // chunks beyond m = l hold nothing for this l
	double s = 0;
	for (int c = 0; c < nch; c++) s += part[(long)c*(lmax+1) + l];
	s *= 2.0/(2*l+1);
	if (cl_dtype == PX_F32) ((float*)cl)[l] = (float)s; else ((double*)cl)[l] = s;
}

// out[a][lm] = sum_b lmat[a][b][l] in[b][lm]   (N = M = 1: plain lmul); lmat rows shorter than lmax+1 read as 0 (cmisc_core.c:159-274)
__global__ __launch_bounds__(256) void lmatmul_kernel(int N, int M, int lmax, int mmax, const uint64_t* __restrict__ mstart, long lstride,
		const void* __restrict__ in, long in_cstride, void* __restrict__ out, long out_cstride, int dtype,
		const double* __restrict__ lmat, int nl)
{
	const int l = blockIdx.x*blockDim.x + threadIdx.x;
	const int m = blockIdx.y;
	if (l > lmax || l < m) return;
	const long i = (long)mstart[m] + l*lstride;
	// every input component of this (l,m) is read before any output is written, so out may alias in
	// (the reference's work arrays, cmisc_core.c:193-214)
	if (dtype == PX_C64) {   // lmatmul_sp (cmisc_core.c:231-274): float filter, float products, float sums
		const float2* __restrict__ pin = (const float2*)in; float2* __restrict__ pout = (float2*)out;
		float2 v[8];
		for (int b = 0; b < M; b++) v[b] = pin[b*in_cstride + i];
		for (int a = 0; a < N; a++) {
			float re = 0, im = 0;
			for (int b = 0; b < M; b++) {
				const float f = l < nl ? (float)lmat[((long)a*M + b)*nl + l] : 0.0f;
				re = __fadd_rn(re, __fmul_rn(f, v[b].x)); im = __fadd_rn(im, __fmul_rn(f, v[b].y));
			}
			pout[a*out_cstride + i] = make_float2(re, im);
		}
		return;
	}
	const double2* __restrict__ pin = (const double2*)in; double2* __restrict__ pout = (double2*)out;
	double2 v[8];
	for (int b = 0; b < M; b++) v[b] = pin[b*in_cstride + i];
	for (int a = 0; a < N; a++) {
		double re = 0, im = 0;
		for (int b = 0; b < M; b++) {
			const double f = l < nl ? lmat[((long)a*M + b)*nl + l] : 0.0;
			re += f*v[b].x; im += f*v[b].y;
		}
		pout[a*out_cstride + i] = make_double2(re, im);
	}
}
} // namespace pxs

using namespace pxs;
#define PXS_TRY try {
#define PXS_CATCH } catch (const pxs::Error& e) { pxs::set_last_error(e.what()); return e.code; } \
	catch (const std::exception& e) { pxs::set_last_error(e.what()); return pxs::PXS_ERR_ARG; } return 0;

extern "C" {

int pxa_alm2cl(int lmax, int mmax, const uint64_t* d_mstart, int64_t lstride, const void* alm1, const void* alm2, int alm_dtype,
               void* cl, int cl_dtype, int device, void* stream)
{
	PXS_TRY
	PXS_REQUIRE(lmax >= 0 && mmax >= 0 && mmax <= lmax && d_mstart && alm1 && alm2 && cl, "pxa_alm2cl: bad arguments");
	PXS_REQUIRE(alm_dtype == PX_C64 || alm_dtype == PX_C128, "pxa_alm2cl: alm must be complex64 or complex128");
	PXS_REQUIRE(cl_dtype == PX_F32 || cl_dtype == PX_F64, "pxa_alm2cl: cl must be float32 or float64");
	PXS_REQUIRE(!(alm_dtype == PX_C128 && cl_dtype == PX_F32), "pxa_alm2cl: float32 spectrum of double precision alm is not supported");
	PXS_HIP(hipSetDevice(device));
	const bool acc_f32 = alm_dtype == PX_C64 && cl_dtype == PX_F32;
	if (acc_f32 || lmax < 256) {
		// float running sum in the reference's order (alm2cl_sp), or a spectrum too short to be worth two launches
		hipLaunchKernelGGL(alm2cl_kernel, dim3((lmax+256)/256), dim3(256), 0, (hipStream_t)stream, lmax, mmax, d_mstart, (long)lstride,
			alm1, alm2, alm_dtype, cl, cl_dtype, acc_f32 ? 1 : 0);
	} else {
		static std::mutex mu; static std::map<int, DevBuf> scratch;
		const int nch = mmax/ALM2CL_MCH + 1;
		double* part;
		{ std::lock_guard<std::mutex> g(mu); DevBuf& b = scratch[device]; b.ensure(sizeof(double)*(size_t)nch*(lmax+1)); part = b.as<double>(); }
		hipLaunchKernelGGL(alm2cl_part_kernel, dim3((lmax+256)/256, nch), dim3(256), 0, (hipStream_t)stream, lmax, mmax, d_mstart, (long)lstride,
			alm1, alm2, alm_dtype, part);
		hipLaunchKernelGGL(alm2cl_sum_kernel, dim3((lmax+256)/256), dim3(256), 0, (hipStream_t)stream, lmax, mmax, (const double*)part, cl, cl_dtype);
	}
	PXS_HIP(hipGetLastError());
	PXS_CATCH
}

int pxa_lmatmul(int N, int M, int lmax, int mmax, const uint64_t* d_mstart, int64_t lstride,
                const void* alm_in, int64_t in_cstride, void* alm_out, int64_t out_cstride, int alm_dtype,
                const double* d_lmat, int nl, int device, void* stream)
{
	PXS_TRY
	PXS_REQUIRE(N >= 1 && M >= 1 && M <= 8 && lmax >= 0 && mmax >= 0 && mmax <= lmax && d_mstart && alm_in && alm_out && d_lmat && nl >= 0, "pxa_lmatmul: bad arguments");
	PXS_REQUIRE(alm_dtype == PX_C64 || alm_dtype == PX_C128, "pxa_lmatmul: alm must be complex64 or complex128");
	PXS_HIP(hipSetDevice(device));
	hipLaunchKernelGGL(lmatmul_kernel, dim3((lmax+256)/256, mmax+1), dim3(256), 0, (hipStream_t)stream, N, M, lmax, mmax, d_mstart, (long)lstride,
		alm_in, (long)in_cstride, alm_out, (long)out_cstride, alm_dtype, d_lmat, nl);
	PXS_HIP(hipGetLastError());
	PXS_CATCH
}

} // extern "C"
