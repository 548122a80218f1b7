// Flat-sky harmonic helpers around the 2-D map FFT: the Q/U <-> E/B rotation of enmap.map2harm / harm2map
// (pixell/enmap.py:1358-1389, queb_rotmat :1391-1400), enmap.calc_ps2d (:1959-2011) and the radial binning of enmap.lbin
// (:2526-2556).  All are one streaming pass over the harmonic map; the reference materialises a [2,2,ny,nx] rotation
// matrix and an [ny,nx] |l| map on the host for them, here the angle and |l| come from the two l axes on the fly.
#include "../../include/pxsht.h"
#include "common.hpp"

namespace pxs {

__device__ __forceinline__ double2 ld_cx(const void* p, int dtype, long i) {
	if (dtype == PX_C64) { float2 v = ((const float2*)p)[i]; return make_double2(v.x, v.y); }
	return ((const double2*)p)[i];
}
__device__ __forceinline__ void st_cx(void* p, int dtype, long i, double2 v) {
	if (dtype == PX_C64) ((float2*)p)[i] = make_float2((float)v.x, (float)v.y);
	else ((double2*)p)[i] = v;
}

// (a, b) <- (c a - s b, s a + c b), c + i s = e^{i spin atan2(sign lx, ly)}
__global__ __launch_bounds__(256) void rotate_queb_kernel(int ny, int nx, const double* __restrict__ ly, const double* __restrict__ lx,
		int spin, double sign, void* __restrict__ a, void* __restrict__ b, int dtype)
{
	const int x = blockIdx.x*blockDim.x + threadIdx.x, y = blockIdx.y;
	if (x >= nx) return;
	const double ang = spin*atan2(sign*lx[x], ly[y]);
	double s, c; sincos(ang, &s, &c);
	const long i = (long)y*nx + x;
	const double2 va = ld_cx(a, dtype, i), vb = ld_cx(b, dtype, i);
	st_cx(a, dtype, i, make_double2(c*va.x - s*vb.x, c*va.y - s*vb.y));
	st_cx(b, dtype, i, make_double2(s*va.x + c*vb.x, s*va.y + c*vb.y));
}

// out = Re(a conj(b))
__global__ __launch_bounds__(256) void ps2d_kernel(long n, const void* __restrict__ a, const void* __restrict__ b, int dtype, void* __restrict__ out, int odtype)
{
	const long i = (long)blockIdx.x*blockDim.x + threadIdx.x;
	if (i >= n) return;
	const double2 va = ld_cx(a, dtype, i), vb = ld_cx(b, dtype, i);
	double r;
	if (dtype == PX_C64) r = (double)(__fadd_rn(__fmul_rn((float)va.x, (float)vb.x), __fmul_rn((float)va.y, (float)vb.y)));   // numpy's complex64 product
	else r = va.x*vb.x + va.y*vb.y;
	if (odtype == PX_F32) ((float*)out)[i] = (float)r; else ((double*)out)[i] = r;
}

// sums of map, |l| and counts per bin floor(|l| / bsize); bins >= nbin are dropped (enmap._bin_helper, enmap.py:2533-2556).
// A workgroup bins a 64 x 64 pixel tile: its pixels fall into a narrow range of |l| rings (~100 bins for 4096 pixels), so the
// sums are collected in an LDS histogram that starts at the tile's smallest bin and only the touched bins go to memory with one
// atomic each -- one global atomic per pixel and array (700 M at 10800 x 21600) made this the longest stage of the C5 pipeline.
#define LBIN_TILE 64
#define LBIN_LDS 512
#ifdef PXS_HOST_SIM
#define PXS_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#else
#define PXS_ATOMIC_ADD(p, v) unsafeAtomicAdd((p), (v))
#endif
// bin = floor(sqrt(l2) / bsize) exactly as the double-precision expression gives it, without paying for the correctly rounded square root and
// the division on every pixel (they made the kernel compute-bound: 1.05 ms for 1.9 GB at 10800 x 21600): v_rsq_f64 + one Newton step is good to
// ~1e-14, the product with 1 / bsize to ~1e-15; only a quotient within 1e-9 of a bin edge takes the exact expression.
__device__ __forceinline__ long lbin_bin(double l2, double bsize, double inv) {
	if (!(l2 > 0.0)) return 0;
#ifdef PXS_HOST_SIM
	const double r = 1.0/sqrt(l2);
#else
	const double r = __builtin_amdgcn_rsq(l2);
#endif
	double sq = l2*r;
	sq = fma(fma(-sq, sq, l2), 0.5*r, sq);
	const double q = sq*inv;
	const long b = (long)q;
	const double fr = q - (double)b, tol = 1e-9*(1.0 + q);
	if (fr < tol || fr > 1.0 - tol || !(q < 4e18)) return (long)floor(sqrt(l2)/bsize);
	return b;
}
__global__ __launch_bounds__(256) void lbin_kernel(int ny, int nx, const double* __restrict__ ly, const double* __restrict__ lx, double bsize, int nbin,
		const void* __restrict__ map, int dtype, int with_l, double* __restrict__ osum, double* __restrict__ olsum, double* __restrict__ ohit)
{
	PXS_SHARED(double, hist);          // [3][LBIN_LDS] sums, |l| sums, counts; then one int: the tile's smallest bin
	int* bmin = reinterpret_cast<int*>(hist + 3*LBIN_LDS);
	const int tid = threadIdx.x, tx = tid & (LBIN_TILE - 1), ty = tid >> 6;
	const int x = blockIdx.x*LBIN_TILE + tx, y0 = blockIdx.y*LBIN_TILE;
	for (int k = tid; k < 3*LBIN_LDS; k += 256) hist[k] = 0.0;
	if (tid == 0) *bmin = 0x7fffffff;
	__syncthreads();
	const double lxx = x < nx ? lx[x] : 0.0, lx2 = lxx*lxx, inv = 1.0/bsize;
	// the bins of this thread's 16 pixels, computed once (-1: outside the map or the bin range)
	long bins[LBIN_TILE/4];
	int lo = 0x7fffffff;
#pragma unroll
	for (int jj = 0; jj < LBIN_TILE/4; jj++) {
		const int y = y0 + ty + 4*jj;
		long bin = -1;
		if (x < nx && y < ny) { const double lyy = ly[y]; bin = lbin_bin(lyy*lyy + lx2, bsize, inv); if (bin < lo) lo = (int)bin; }
		bins[jj] = bin;
	}
	if (lo != 0x7fffffff) atomicMin(bmin, lo);
	__syncthreads();
	const int base = *bmin;
#pragma unroll
	for (int jj = 0; jj < LBIN_TILE/4; jj++) {
		const int y = y0 + ty + 4*jj;
		const long bin = bins[jj];
		if (bin < 0 || bin >= nbin) continue;
		const long i = (long)y*nx + x;
		const double v = dtype == PX_F32 ? (double)((const float*)map)[i] : ((const double*)map)[i];
		const long k = bin - base;
		if (k < LBIN_LDS) {
			PXS_ATOMIC_ADD(hist + k, v);
			if (with_l) { const double lyy = ly[y]; PXS_ATOMIC_ADD(hist + LBIN_LDS + k, sqrt(lyy*lyy + lx2)); PXS_ATOMIC_ADD(hist + 2*LBIN_LDS + k, 1.0); }
		} else {	// (a tile wider than the LDS histogram: coarse pixels with very fine bins)
			PXS_ATOMIC_ADD(osum + bin, v);
			if (with_l) { const double lyy = ly[y]; PXS_ATOMIC_ADD(olsum + bin, sqrt(lyy*lyy + lx2)); PXS_ATOMIC_ADD(ohit + bin, 1.0); }
		}
	}
	__syncthreads();
	for (int k = tid; k < LBIN_LDS; k += 256) {
		const long bin = (long)base + k;
		if (bin >= nbin) break;
		// (without the counts a zero sum cannot tell an untouched bin from a cancelling one: adding 0 is harmless)
		if (with_l) { const double h = hist[2*LBIN_LDS + k]; if (h != 0.0) { PXS_ATOMIC_ADD(osum + bin, hist[k]); PXS_ATOMIC_ADD(olsum + bin, hist[LBIN_LDS + k]); PXS_ATOMIC_ADD(ohit + bin, h); } }
		else if (hist[k] != 0.0) PXS_ATOMIC_ADD(osum + bin, hist[k]);
	}
}

// sum[bin[i]] += map[i] for 0 <= bin[i] < nbin: binning by a per-pixel bin table (enmap.rbin, and enmap.lbin with a transform of |l|: there the bin of a
// pixel is a geometry table the host makes once, enmap.py:2512-2556 _bin_helper).  A block takes 4096 consecutive pixels, collects them in an LDS histogram of
// the LBIN_LDS bins from its smallest one on and adds that to the global sums; pixels further out add to the global sums themselves.
#define BIN_PER_THREAD 16
__global__ __launch_bounds__(256) void bin_index_kernel(long n, const int* __restrict__ bin, int nbin, const void* __restrict__ map, int dtype, double* __restrict__ osum)
{
	PXS_SHARED(double, hist);          // [LBIN_LDS] sums; then one int: the block's smallest bin
	int* bmin = reinterpret_cast<int*>(hist + LBIN_LDS);
	const int tid = threadIdx.x;
	const long i0 = (long)blockIdx.x*256*BIN_PER_THREAD;
	for (int k = tid; k < LBIN_LDS; k += 256) hist[k] = 0.0;
	if (tid == 0) *bmin = 0x7fffffff;
	__syncthreads();
	int b[BIN_PER_THREAD];
	int lo = 0x7fffffff;
#pragma unroll
	for (int jj = 0; jj < BIN_PER_THREAD; jj++) {
		const long i = i0 + (long)jj*256 + tid;
		int v = i < n ? bin[i] : -1;
		if (v >= nbin) v = -1;
		b[jj] = v;
		if (v >= 0 && v < lo) lo = v;
	}
	if (lo != 0x7fffffff) atomicMin(bmin, lo);
	__syncthreads();
	const int base = *bmin;
#pragma unroll
	for (int jj = 0; jj < BIN_PER_THREAD; jj++) {
		if (b[jj] < 0) continue;
		const long i = i0 + (long)jj*256 + tid;
		const double v = dtype == PX_F32 ? (double)((const float*)map)[i] : ((const double*)map)[i];
		const int k = b[jj] - base;
		if (k < LBIN_LDS) PXS_ATOMIC_ADD(hist + k, v); else PXS_ATOMIC_ADD(osum + b[jj], v);
	}
	__syncthreads();
	for (int k = tid; k < LBIN_LDS; k += 256) {
		const long bb = (long)base + k;
		if (bb >= nbin) break;
		if (hist[k] != 0.0) PXS_ATOMIC_ADD(osum + bb, hist[k]);
	}
}

// data[i] *= vec[(i / inner) % n]: multiply along one axis of a contiguous complex array (fft.shift's phase ramps, fft.py:347-368)
__global__ __launch_bounds__(256) void mul_axis_kernel(long total, long n, long inner, void* __restrict__ data, int dtype, const double2* __restrict__ vec)
{
	const long i = (long)blockIdx.x*blockDim.x + threadIdx.x;
	if (i >= total) return;
	const double2 w = vec[(i/inner) % n];
	const double2 v = ld_cx(data, dtype, i);
	st_cx(data, dtype, i, make_double2(v.x*w.x - v.y*w.y, v.x*w.y + v.y*w.x));
}

} // namespace pxs

using namespace pxs;
#define PXS_TRY try {
#define PXS_CATCH } catch (const pxs::Error& e) { pxs::set_last_error(e.what()); return e.code; } \
	catch (const std::exception& e) { pxs::set_last_error(e.what()); return pxs::PXS_ERR_ARG; } return 0;

extern "C" {

int pxm_rotate_queb(int ny, int nx, const double* d_ly, const double* d_lx, int spin, int inverse_sign,
                    void* a, void* b, int dtype, int device, void* stream)
{
	PXS_TRY
	PXS_REQUIRE(ny > 0 && nx > 0 && d_ly && d_lx && a && b, "pxm_rotate_queb: bad arguments");
	PXS_REQUIRE(dtype == PX_C64 || dtype == PX_C128, "pxm_rotate_queb: maps must be complex64 or complex128");
	PXS_HIP(hipSetDevice(device));
	hipLaunchKernelGGL(rotate_queb_kernel, dim3((nx+255)/256, ny), dim3(256), 0, (hipStream_t)stream, ny, nx, d_ly, d_lx, spin,
		inverse_sign ? -1.0 : 1.0, a, b, dtype);
	PXS_HIP(hipGetLastError());
	PXS_CATCH
}

int pxm_ps2d(int64_t n, const void* a, const void* b, int dtype, void* out, int out_dtype, int device, void* stream)
{
	PXS_TRY
	PXS_REQUIRE(n >= 0 && a && b && out, "pxm_ps2d: bad arguments");
	PXS_REQUIRE(dtype == PX_C64 || dtype == PX_C128, "pxm_ps2d: inputs must be complex64 or complex128");
	PXS_REQUIRE(out_dtype == PX_F32 || out_dtype == PX_F64, "pxm_ps2d: output must be float32 or float64");
	PXS_HIP(hipSetDevice(device));
	if (n > 0) hipLaunchKernelGGL(ps2d_kernel, dim3((unsigned)((n+255)/256)), dim3(256), 0, (hipStream_t)stream, (long)n, a, b, dtype, out, out_dtype);
	PXS_HIP(hipGetLastError());
	PXS_CATCH
}

int pxm_lbin(int ny, int nx, const double* d_ly, const double* d_lx, double bsize, int nbin,
             const void* map, int dtype, double* d_sum, double* d_lsum, double* d_hit, int device, void* stream)
{
	PXS_TRY
	PXS_REQUIRE(ny > 0 && nx > 0 && d_ly && d_lx && map && d_sum && bsize > 0 && nbin >= 0, "pxm_lbin: bad arguments");
	PXS_REQUIRE(dtype == PX_F32 || dtype == PX_F64, "pxm_lbin: map must be float32 or float64");
	PXS_REQUIRE((d_lsum == nullptr) == (d_hit == nullptr), "pxm_lbin: give both or none of lsum, hit");
	PXS_HIP(hipSetDevice(device));
	if (nbin > 0) hipLaunchKernelGGL(lbin_kernel, dim3((nx + LBIN_TILE - 1)/LBIN_TILE, (ny + LBIN_TILE - 1)/LBIN_TILE), dim3(256), sizeof(double)*3*LBIN_LDS + 16, (hipStream_t)stream, ny, nx, d_ly, d_lx, bsize, nbin,
		map, dtype, d_lsum ? 1 : 0, d_sum, d_lsum, d_hit);
	PXS_HIP(hipGetLastError());
	PXS_CATCH
}

int pxm_bin_index(int64_t n, const int32_t* d_bin, int nbin, const void* map, int dtype, double* d_sum, int device, void* stream)
{
	PXS_TRY
	PXS_REQUIRE(n >= 0 && nbin >= 0 && (n == 0 || (d_bin && map)) && (nbin == 0 || d_sum), "pxm_bin_index: bad arguments");
	PXS_REQUIRE(dtype == PX_F32 || dtype == PX_F64, "pxm_bin_index: map must be float32 or float64");
	PXS_HIP(hipSetDevice(device));
	if (n > 0 && nbin > 0) hipLaunchKernelGGL(bin_index_kernel, dim3((unsigned)((n + 256*BIN_PER_THREAD - 1)/(256*BIN_PER_THREAD))), dim3(256), sizeof(double)*LBIN_LDS + 16, (hipStream_t)stream,
		(long)n, (const int*)d_bin, nbin, map, dtype, d_sum);
	PXS_HIP(hipGetLastError());
	PXS_CATCH
}

int pxm_mul_axis(int64_t total, int64_t n, int64_t inner, void* data, int dtype, const void* d_vec, int device, void* stream)
{
	PXS_TRY
	PXS_REQUIRE(total >= 0 && n > 0 && inner > 0 && data && d_vec, "pxm_mul_axis: bad arguments");
	PXS_REQUIRE(dtype == PX_C64 || dtype == PX_C128, "pxm_mul_axis: data must be complex64 or complex128");
	PXS_HIP(hipSetDevice(device));
	if (total > 0) hipLaunchKernelGGL(mul_axis_kernel, dim3((unsigned)((total+255)/256)), dim3(256), 0, (hipStream_t)stream, (long)total, (long)n, (long)inner, data, dtype, (const double2*)d_vec);
	PXS_HIP(hipGetLastError());
	PXS_CATCH
}

} // extern "C"
