// C-ABI entry points for the N-d FFT (include/pxsht.h: pxf_*), built on FftContext.
#include "../../include/pxsht.h"
#include "fft.hpp"
#include <map>
#include <mutex>
#include <memory>
#include <algorithm>

namespace pxs {

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }
const char* get_last_error() { return g_last_error.c_str(); }

FftContext& fft_context(int device) {
	static std::mutex mu; static std::map<int, std::unique_ptr<FftContext>> ctx;
	std::lock_guard<std::mutex> g(mu);
	auto& p = ctx[device];
	if (!p) p.reset(new FftContext(device));
	return *p;
}

struct AxisDim { long n, is, os; };

// transform along one axis of an N-d array; other dims become line dims (<= 3 after merging,
// extra leading dims looped on the host)
static void fft_axis(FftContext& fc, hipStream_t st, long n, bool forward, std::vector<AxisDim> dims, long is_e, long os_e,
                     FftLoad ld, FftStore stf) {
	// drop singleton dims, merge mergeable neighbours (outer,inner): outer.s == inner.s*inner.n for both in and out
	std::vector<AxisDim> d;
	for (auto& x : dims) if (x.n > 1) d.push_back(x);
	for (size_t k = 0; k + 1 < d.size();) {
		if (d[k].is == d[k+1].is*d[k+1].n && d[k].os == d[k+1].os*d[k+1].n) { d[k+1].n *= d[k].n; d.erase(d.begin()+k); }
		else k++;
	}
	// pick tile dim: smallest |input stride|
	FftDims fd; fd.is_e = is_e; fd.os_e = os_e;
	if (!d.empty()) {
		size_t best = 0;
		for (size_t k = 1; k < d.size(); k++) if (std::abs(d[k].is) < std::abs(d[best].is)) best = k;
		fd.n_i = d[best].n; fd.is_i = d[best].is; fd.os_i = d[best].os; d.erase(d.begin()+best);
	}
	if (!d.empty()) { fd.n_o1 = d.back().n; fd.is_o1 = d.back().is; fd.os_o1 = d.back().os; d.pop_back(); }
	if (!d.empty()) { fd.n_o2 = d.back().n; fd.is_o2 = d.back().is; fd.os_o2 = d.back().os; d.pop_back(); }
	// remaining dims: host loop
	std::vector<long> idx(d.size(), 0);
	auto esz = [](int dt) { return dt == PX_F32 ? 4 : dt == PX_F64 ? 8 : dt == PX_C64 ? 8 : 16; };
	while (true) {
		long ioff = 0, ooff = 0;
		for (size_t k = 0; k < d.size(); k++) { ioff += idx[k]*d[k].is; ooff += idx[k]*d[k].os; }
		FftLoad l2 = ld; FftStore s2 = stf;
		l2.ptr = (const char*)ld.ptr + ioff*esz(ld.dtype); s2.ptr = (char*)stf.ptr + ooff*esz(stf.dtype);
		fc.exec(st, n, forward, fd, l2, s2);
		size_t k = 0;
		for (; k < d.size(); k++) { if (++idx[k] < d[k].n) break; idx[k] = 0; }
		if (k == d.size()) break;
	}
}

} // namespace pxs

using namespace pxs;

#define PXS_TRY try {
#define PXS_CATCH } catch (const pxs::Error& e) { pxs::set_last_error(e.what()); return e.code; } \
	catch (const std::exception& e) { pxs::set_last_error(e.what()); return pxs::PXS_ERR_ARG; } return 0;

extern "C" {

const char* pxs_last_error(void) { return pxs::get_last_error(); }
const char* pxs_version(void) {
#ifdef PXS_HOST_SIM
	return "pxsht 0.1 HOSTSIM (test-only CPU emulation; not a product path)";
#else
	return "pxsht 0.1 gfx950";
#endif
}
int pxf_fft_supported(int64_t n) { return FftContext::supported(n) ? 1 : 0; }
int64_t pxf_fft_good_size(int64_t n) { return FftContext::good_size(n); }

int pxf_fft_nd(int ndim, const int64_t* shape, const int64_t* istride, const int64_t* ostride,
               int naxes, const int* axes_in, int kind, int forward, double scale,
               int in_dtype, int out_dtype, const void* in, void* out, int device, void* stream)
{
	PXS_TRY
	PXS_REQUIRE(ndim >= 1 && ndim <= 16 && naxes >= 1 && naxes <= ndim, "pxf_fft_nd: bad ndim/naxes");
	PXS_REQUIRE(kind >= 0 && kind <= 3, "pxf_fft_nd: kind must be 0 (c2c), 1 (r2c), 2 (c2r) or 3 (DCT-I)");
	if (kind == 3) PXS_REQUIRE(in_dtype <= PX_F64 && out_dtype <= PX_F64, "DCT-I needs real in, real out");
	if (kind == 0) PXS_REQUIRE(out_dtype >= PX_C64, "c2c needs complex output (real input is read with zero imaginary part)");
	if (kind == 1) PXS_REQUIRE(in_dtype <= PX_F64 && out_dtype >= PX_C64, "r2c needs real in, complex out");
	if (kind == 2) PXS_REQUIRE(in_dtype >= PX_C64 && out_dtype <= PX_F64, "c2r needs complex in, real out");
	std::vector<int> axes(axes_in, axes_in+naxes);
	for (auto& a : axes) { if (a < 0) a += ndim; PXS_REQUIRE(a >= 0 && a < ndim, "pxf_fft_nd: axis out of range"); }
	for (int k = 0; k < ndim; k++) if (shape[k] == 0) return 0;
	for (int a : axes) {
		std::string why;
		if (kind == 3) {   // DCT-I of n points = real part of the FFT of the even extension to 2(n-1) points
			if (shape[a] < 2) throw Error(PXS_ERR_ARG, "DCT-I needs at least 2 points along each axis");
			if (!FftContext::supported(2*(shape[a]-1), &why)) throw Error(PXS_ERR_UNSUPPORTED, "DCT-I of " + std::to_string(shape[a]) + " points: " + why);
		} else if (!FftContext::supported(shape[a], &why)) throw Error(PXS_ERR_UNSUPPORTED, why);
	}
	PXS_HIP(hipSetDevice(device));
	FftContext& fc = fft_context(device);
	hipStream_t st = (hipStream_t)stream;
	const int last = axes.back();
	const long nlast = shape[last], nh = nlast/2 + 1;
	// complex-domain shape (half spectrum along `last` for r2c / c2r)
	std::vector<long> cshape(shape, shape+ndim);
	if (kind != 0) cshape[last] = nh;
	auto other_dims = [&](int ax, const std::vector<long>& shp, const int64_t* is, const int64_t* os) {
		std::vector<AxisDim> d;
		for (int k = 0; k < ndim; k++) if (k != ax) d.push_back({shp[k], (long)is[k], (long)os[k]});
		return d;
	};
	if (kind == 3) {
		// FFTW_REDFT00 (pixell/fft.py:211-231, the transform behind enmap.fft(dct=True)): y_k = x_0 + (-1)^k x_{n-1} + 2 sum x_j cos(pi jk/(n-1)),
		// i.e. Re FFT_{2(n-1)} of the even mirror extension; the extension is a load functor, only the first n bins are stored
		std::vector<long> rshape(shape, shape+ndim);
		for (int t = 0; t < naxes; t++) {
			int ax = axes[naxes-1-t];
			bool first = (t == 0), lastpass = (t == naxes-1);
			FftLoad ld; FftStore stf;
			ld.ptr = first ? in : out; ld.dtype = first ? in_dtype : out_dtype;
			ld.mode = LD_MIRROR; ld.ne = shape[ax]; ld.mir_c = 0; ld.par0 = 0; ld.par_step = 0;
			stf.ptr = out; stf.dtype = out_dtype; stf.ne = shape[ax]; stf.scale = lastpass ? scale : 1.0;
			const int64_t* is = first ? istride : ostride;
			fft_axis(fc, st, 2*(shape[ax]-1), true, other_dims(ax, rshape, is, ostride), is[ax], ostride[ax], ld, stf);
		}
	} else if (kind == 0) {
		for (int t = 0; t < naxes; t++) {
			int ax = axes[naxes-1-t];
			bool first = (t == 0), lastpass = (t == naxes-1);
			FftLoad ld; FftStore stf;
			ld.ptr = first ? in : out; ld.dtype = first ? in_dtype : out_dtype;
			stf.ptr = out; stf.dtype = out_dtype; stf.scale = lastpass ? scale : 1.0;
			const int64_t* is = first ? istride : ostride;
			fft_axis(fc, st, shape[ax], forward != 0, other_dims(ax, cshape, is, ostride), is[ax], ostride[ax], ld, stf);
		}
	} else if (kind == 1) {
		// real axis first (numpy.fft.rfftn order), then c2c over the remaining axes in place on out
		{
			FftLoad ld; FftStore stf;
			ld.ptr = in; ld.dtype = in_dtype; stf.ptr = out; stf.dtype = out_dtype; stf.ne = nh;
			stf.scale = naxes == 1 ? scale : 1.0;
			std::vector<long> rshape(shape, shape+ndim);
			fft_axis(fc, st, nlast, forward != 0, other_dims(last, rshape, istride, ostride), istride[last], ostride[last], ld, stf);
		}
		for (int t = 1; t < naxes; t++) {
			int ax = axes[naxes-1-t];
			FftLoad ld; FftStore stf;
			ld.ptr = out; ld.dtype = out_dtype; stf.ptr = out; stf.dtype = out_dtype; stf.scale = (t == naxes-1) ? scale : 1.0;
			fft_axis(fc, st, shape[ax], forward != 0, other_dims(ax, cshape, ostride, ostride), ostride[ax], ostride[ax], ld, stf);
		}
	} else {
		// c2r: complex transforms over all but the last axis into scratch, then Hermitian c2r on the last axis
		const void* src = in; int src_dtype = in_dtype; std::vector<int64_t> sstride(istride, istride+ndim);
		DevBuf scratch;
		if (naxes > 1) {
			size_t tot = 1; for (int k = 0; k < ndim; k++) tot *= cshape[k];
			scratch.alloc(tot*sizeof(double2));
			std::vector<int64_t> cs(ndim); long acc = 1;
			for (int k = ndim-1; k >= 0; k--) { cs[k] = acc; acc *= cshape[k]; }
			for (int t = 0; t < naxes-1; t++) {
				int ax = axes[naxes-2-t];
				FftLoad ld; FftStore stf;
				ld.ptr = (t == 0) ? in : scratch.p; ld.dtype = (t == 0) ? in_dtype : PX_C128;
				stf.ptr = scratch.p; stf.dtype = PX_C128;
				const int64_t* is = (t == 0) ? istride : cs.data();
				fft_axis(fc, st, shape[ax], forward != 0, other_dims(ax, cshape, is, cs.data()), is[ax], cs[ax], ld, stf);
			}
			src = scratch.p; src_dtype = PX_C128; sstride = cs;
		}
		FftLoad ld; FftStore stf;
		ld.ptr = src; ld.dtype = src_dtype; ld.mode = LD_HERM; ld.ne = nh;
		stf.ptr = out; stf.dtype = out_dtype; stf.scale = scale;
		std::vector<long> rshape(shape, shape+ndim);
		// note: LD_HERM reads indices < nh directly and conj(N-e) otherwise: for even N the Nyquist bin e=N/2 < nh is read directly
		fft_axis(fc, st, nlast, forward != 0, other_dims(last, rshape, sstride.data(), ostride), sstride[last], ostride[last], ld, stf);
		if (naxes > 1) PXS_HIP(hipStreamSynchronize(st));   // scratch is freed on return
	}
	PXS_CATCH
}

} // extern "C"
