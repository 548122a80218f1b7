// C-ABI entry points for the N-d FFT (include/pxsht.h: pxf_*), built on FftContext.
#include "../../include/pxsht.h"
#include "fft.hpp"
#include "fftchain.hpp"
#include <map>
#include <mutex>
#include <memory>
#include <algorithm>
#include <tuple>
#include <cmath>

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
// ---- Bluestein (chirp-z) for lengths the mixed-radix engine cannot factor (a prime factor > 2048) -------------------
// X_k = w_k sum_j (x_j w_j) conj(w)_{k-j}, w_j = e^{-+ i pi j^2/n}: two FFTs of a 5-smooth length M >= 2n-1 with the chirp
// multiplications and the zero padding as load / store functors (numpy, the reference's fallback engine, takes any n).
struct BluePlan { long n = 0, M = 0; DevBuf w, wout, bhat; };
static const BluePlan& blue_plan(FftContext& fc, int device, hipStream_t st, long n, bool forward) {
	static std::mutex mu; static std::map<std::tuple<int, long, int>, std::unique_ptr<BluePlan>> plans;
	std::lock_guard<std::mutex> g(mu);
	auto& p = plans[std::make_tuple(device, n, forward ? 1 : 0)];
	if (p) return *p;
	p.reset(new BluePlan());
	typedef long double LD;
	const LD pi = 3.141592653589793238462643383279502884L;
	p->n = n; p->M = FftContext::good_size(2*n - 1);
	const long M = p->M;
	const LD sgn = forward ? -1.0L : 1.0L;
	std::vector<double2> w(n), wout(n), b(M, make_double2(0, 0));
	for (long j = 0; j < n; j++) {
		const long q = (long)(((unsigned long long)j*(unsigned long long)j) % (unsigned long long)(2*n));   // j^2 mod 2n keeps the phase accurate
		const LD a = pi*(LD)q/(LD)n;
		const LD c = cosl(a), s_ = sgn*sinl(a);
		w[j] = make_double2((double)c, (double)s_);
		wout[j] = make_double2((double)(c/(LD)M), (double)(s_/(LD)M));        // w_k / M: the backward transform below is unnormalised
		b[j] = make_double2((double)c, (double)(-s_));
		if (j > 0) b[M - j] = b[j];
	}
	p->w = upload(w); p->wout = upload(wout);
	DevBuf db = upload(b);
	p->bhat.alloc(sizeof(double2)*M);
	FftDims d; d.n_i = 1; d.is_e = 1; d.os_e = 1;
	FftLoad ld; ld.ptr = db.p; FftStore sf; sf.ptr = p->bhat.p;
	fc.exec(st, M, true, d, ld, sf);
	PXS_HIP(hipStreamSynchronize(st));      // db is freed on return
	return *p;
}

static void fft_axis(FftContext& fc, hipStream_t st, long n, bool forward, std::vector<AxisDim> dims, long is_e, long os_e,
                     FftLoad ld, FftStore stf);

static std::mutex g_blue_mu; static std::map<std::pair<int, hipStream_t>, std::unique_ptr<DevBuf>> g_blue_scratch;
// chain engine (and its scratch) of the 2-D real -> complex fast path, per stream like the other scratch
// (each entry owns the transposed intermediate of its last transform -- 15 GB for a complex 21600 x 43200 map -- so the table is
// bounded: beyond PXF_F2_MAX_STREAMS (4) the entry used longest ago is dropped; hipFree waits for the kernels that may still use it)
// Entries are shared_ptr: the caller holds one for the duration of its transform, so an eviction (or fft_release_stream) on another
// thread only drops the table's reference -- the chain and its scratch go when the last user is done.  The cap counts the streams of
// ONE device (a process driving 8 GPUs keeps 4 per GPU), and an entry somebody holds is never the one evicted.
struct F2Entry { std::shared_ptr<FftChain> ch; unsigned long stamp = 0; };
static std::mutex g_f2_mu; static std::map<std::pair<int, hipStream_t>, F2Entry> g_f2; static unsigned long g_f2_clock = 0;
static std::shared_ptr<FftChain> f2_chain(int device, hipStream_t st, FftContext& fc) {
	std::lock_guard<std::mutex> g(g_f2_mu);
	static const size_t cap = [] { const char* e = getenv("PXF_F2_MAX_STREAMS"); return (size_t)std::max(1, e ? atoi(e) : 4); }();
	auto key = std::make_pair(device, st);
	auto it = g_f2.find(key);
	if (it == g_f2.end()) {
		for (;;) {
			size_t ndev = 0; auto old = g_f2.end();
			for (auto j = g_f2.begin(); j != g_f2.end(); ++j) if (j->first.first == device) {
				ndev++;
				if (j->second.ch.use_count() == 1 && (old == g_f2.end() || j->second.stamp < old->second.stamp)) old = j;      // (idle: only the table holds it)
			}
			if (ndev < cap || old == g_f2.end()) break;
			g_f2.erase(old);
		}
		it = g_f2.emplace(key, F2Entry()).first;
		it->second.ch = std::make_shared<FftChain>(&fc);
	}
	it->second.stamp = ++g_f2_clock;
	return it->second.ch;
}
// scratch of the multi-axis c2r transforms, per stream like the other scratch (grows only; no synchronisation in the call path)
static std::mutex g_c2r_mu; static std::map<std::pair<int, hipStream_t>, std::unique_ptr<DevBuf>> g_c2r_scratch;
// a stream is about to be destroyed: free the scratch keyed by it (a recycled handle must not inherit a stale buffer)
void fft_release_stream(int device, hipStream_t st) {
	{ std::lock_guard<std::mutex> g(g_blue_mu); g_blue_scratch.erase(std::make_pair(device, st)); }
	{ std::lock_guard<std::mutex> g(g_f2_mu); g_f2.erase(std::make_pair(device, st)); }
	{ std::lock_guard<std::mutex> g(g_c2r_mu); g_c2r_scratch.erase(std::make_pair(device, st)); }
	fft_context(device).release_stream(st);
}

static void bluestein_axis(FftContext& fc, int device, hipStream_t st, long n, bool forward, std::vector<AxisDim> dims, long is_e, long os_e,
                           FftLoad ld, FftStore stf) {
	const bool c2r = ld.mode == LD_HERM && !ld.herm_fold;      // Hermitian half spectrum in, real line out
	if ((ld.mode != LD_PLAIN && !c2r) || ld.mul || stf.mul) throw Error(PXS_ERR_UNSUPPORTED, "FFT length " + std::to_string(n) + " has a prime factor > 2048: only plain c2c / r2c / c2r transforms are available for it (Bluestein)");
	const BluePlan& bp = blue_plan(fc, device, st, n, forward);
	const long M = bp.M;
	// dense scratch [lines][M]
	std::vector<AxisDim> d1 = dims, d2 = dims;
	long lines = 1;
	for (size_t k = dims.size(); k-- > 0;) { d1[k].os = lines*M; d2[k].is = lines*M; lines *= dims[k].n; }
	// scratch per stream, grown on demand and kept (growing frees the old block: hipFree waits for the device); owners of private
	// streams give it back with fft_release_stream before they destroy the stream
	DevBuf* sp;
	{ std::lock_guard<std::mutex> g(g_blue_mu); auto& u = g_blue_scratch[std::make_pair(device, st)]; if (!u) u.reset(new DevBuf()); sp = u.get(); }
	DevBuf& scratch = *sp; scratch.ensure(sizeof(double2)*(size_t)lines*M);
	{	// y = FFT_M(x w, zero padded)
		FftLoad l1 = ld; l1.mul = bp.w.as<double2>();
		if (c2r) l1.herm_n = n;                                   // Hermitian extension of n points (times the chirp), zero padded to M
		else l1.ne = (ld.ne >= 0 && ld.ne < n) ? ld.ne : n;
		FftStore s1; s1.ptr = scratch.p; s1.dtype = PX_C128;
		fft_axis(fc, st, M, true, d1, is_e, 1, l1, s1);
	}
	{	// X = w/M * IFFT_M(y bhat), first n (or ne) outputs
		FftLoad l2; l2.ptr = scratch.p; l2.dtype = PX_C128; l2.mul = bp.bhat.as<double2>();
		FftStore s2 = stf; s2.mul = bp.wout.as<double2>(); s2.ne = (stf.ne >= 0 && stf.ne < n) ? stf.ne : n;
		fft_axis(fc, st, M, false, d2, 1, os_e, l2, s2);
	}
}

static int g_fft_device = 0;
static void fft_axis(FftContext& fc, hipStream_t st, long n, bool forward, std::vector<AxisDim> dims, long is_e, long os_e,
                     FftLoad ld, FftStore stf) {
	if (!FftContext::supported(n)) { bluestein_axis(fc, g_fft_device, st, n, forward, dims, is_e, os_e, ld, stf); return; }
	{	// the generic radix pass costs n*p for a prime factor p: beyond ~128 two chirp FFTs of a smooth length are cheaper
		// (healpix ring lengths 4k: alm2map_healpix at nside 2048, lmax 4096, 3 components 270 -> 94 ms)
		static const long pmin = [] { const char* e = lab_getenv("PXS_BLUESTEIN_MINPRIME"); return e ? atol(e) : 128L; }();
		const bool plain = (ld.mode == LD_PLAIN || (ld.mode == LD_HERM && !ld.herm_fold)) && !ld.mul && !stf.mul && ld.shift == 0 && stf.shift == 0 && !stf.conj_out
			&& stf.two_sided_k < 0 && !stf.real_pair;
		long m = n, big = 1;
		for (long q = 2; q*q <= m; q++) while (m % q == 0) { big = std::max(big, q); m /= q; }
		if (m > 1) big = std::max(big, m);
		if (plain && big > pmin) { bluestein_axis(fc, g_fft_device, st, n, forward, dims, is_e, os_e, ld, stf); return; }
	}
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

// nlines dense complex lines of n points each, in -> out (may alias), any n (Bluestein where the mixed-radix engine has no
// factorisation): the ring FFTs of ring sets with per-ring lengths (sht.hip, general rings)
void fft_dense_lines(int device, hipStream_t st, long n, bool forward, long nlines, const double2* in, double2* out) {
	if (nlines <= 0 || n <= 0) return;
	if (n == 1) { if (in != out) PXS_HIP(hipMemcpyAsync(out, in, sizeof(double2)*(size_t)nlines, hipMemcpyDeviceToDevice, st)); return; }
	FftContext& fc = fft_context(device);
	g_fft_device = device;
	std::vector<AxisDim> dims; dims.push_back(AxisDim{nlines, n, n});
	FftLoad ld; ld.ptr = in; FftStore sf; sf.ptr = out;
	fft_axis(fc, st, n, forward, dims, 1, 1, ld, sf);
}

// ---- r2r (DCT / DST) as functor-wrapped complex FFTs ---------------------------------------------------------------
struct R2RPlan { long N = 0; bool forward = true, mirror = false; int mir_c = 0; long ld_shift = 0, st_shift = 0; double scale = 1.0; DevBuf ld_mul, st_mul; };
long r2r_length(int kind, long n) {
	switch (kind) { case 3: return 2*(n-1); case 7: return 2*(n+1); default: return 2*n; }
}
const R2RPlan& r2r_plan(int device, int kind, long n) {
	static std::mutex mu; static std::map<std::tuple<int, int, long>, std::unique_ptr<R2RPlan>> plans;
	std::lock_guard<std::mutex> g(mu);
	auto& p = plans[std::make_tuple(device, kind, n)];
	if (p) return *p;
	p.reset(new R2RPlan());
	typedef long double LD;
	const LD pi = 3.141592653589793238462643383279502884L;
	p->N = r2r_length(kind, n);
	auto tab = [&](long len, auto f) { std::vector<double2> t(len); for (long e = 0; e < len; e++) { LD re, im; f(e, re, im); t[e] = make_double2((double)re, (double)im); } return upload(t); };
	const LD h = pi/(2*(LD)n);      // half-sample phase step pi/(2n)
	switch (kind) {
	case 3:  /* REDFT00  DCT-I   */ p->mirror = true; p->mir_c = 0; break;
	case 4:  /* REDFT10  DCT-II  y_k = Re(e^{-i pi k/2n} Z_k), Z = FFT_2n of the half-sample mirror extension */
		p->mirror = true; p->mir_c = 1;
		p->st_mul = tab(n, [&](long k, LD& re, LD& im) { re = cosl(h*k); im = -sinl(h*k); }); break;
	case 5:  /* REDFT01  DCT-III y_k = 2 Re sum_j c_j x_j e^{i pi j/2n} e^{+2 pi i jk/2n}, c_0 = 1/2 */
		p->forward = false; p->scale = 2;
		p->ld_mul = tab(n, [&](long j, LD& re, LD& im) { const LD c = j == 0 ? 0.5L : 1.0L; re = c*cosl(h*j); im = c*sinl(h*j); }); break;
	case 6:  /* REDFT11  DCT-IV  y_k = 2 Re[e^{-i pi (k+1/2)/2n} sum_j x_j e^{-i pi j/2n} e^{-2 pi i jk/2n}] */
		p->scale = 2;
		p->ld_mul = tab(n, [&](long j, LD& re, LD& im) { re = cosl(h*j); im = -sinl(h*j); });
		p->st_mul = tab(n, [&](long k, LD& re, LD& im) { re = cosl(h*(k+0.5L)); im = -sinl(h*(k+0.5L)); }); break;
	case 7: { /* RODFT00 DST-I   y_k = 2 Re[i e^{-2 pi i q/N} F_q], q = k+1, F = FFT_N of the zero-padded line */
		p->scale = 2; p->st_shift = 1;
		const LD w = 2*pi/(LD)p->N;
		p->st_mul = tab(n+1, [&](long q, LD& re, LD& im) { re = sinl(w*q); im = cosl(w*q); }); break; }   // i e^{-ia} = sin a + i cos a
	case 8:  /* RODFT10  DST-II  y_k = 2 Re[i e^{-i pi q/2n} F_q], q = k+1 */
		p->scale = 2; p->st_shift = 1;
		p->st_mul = tab(n+1, [&](long q, LD& re, LD& im) { re = sinl(h*q); im = cosl(h*q); }); break;
	case 9:  /* RODFT01  DST-III y_k = 2 Re[i sum_j c_j x_j e^{-i pi (j+1)/2n} e^{-2 pi i (j+1)k/2n}], c_{n-1} = 1/2: element j sits at j+1 */
		p->scale = 2; p->ld_shift = 1;
		p->ld_mul = tab(n+1, [&](long e, LD& re, LD& im) { const LD c = e == n ? 0.5L : 1.0L; re = c*cosl(h*e); im = -c*sinl(h*e); });
		p->st_mul = tab(n, [&](long, LD& re, LD& im) { re = 0; im = 1; }); break;
	case 10: /* RODFT11  DST-IV  y_k = 2 Re[i e^{-i pi (k+1/2)/2n} sum_j x_j e^{-i pi j/2n} e^{-2 pi i jk/2n}] */
		p->scale = 2;
		p->ld_mul = tab(n, [&](long j, LD& re, LD& im) { re = cosl(h*j); im = -sinl(h*j); });
		p->st_mul = tab(n, [&](long k, LD& re, LD& im) { re = sinl(h*(k+0.5L)); im = cosl(h*(k+0.5L)); }); break;
	default: throw Error(PXS_ERR_ARG, "unknown r2r kind");
	}
	return *p;
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
	PXS_REQUIRE(kind >= 0 && kind <= 10, "pxf_fft_nd: kind must be 0 (c2c), 1 (r2c), 2 (c2r) or 3..10 (DCT-I..IV, DST-I..IV)");
	if (kind >= 3) PXS_REQUIRE(in_dtype <= PX_F64 && out_dtype <= PX_F64, "DCT/DST need real in, real out");
	if (kind == 0) PXS_REQUIRE(out_dtype >= PX_C64, "c2c needs complex output (real input is read with zero imaginary part)");
	if (kind == 1) PXS_REQUIRE(in_dtype <= PX_F64 && out_dtype >= PX_C64, "r2c needs real in, complex out");
	if (kind == 2) PXS_REQUIRE(in_dtype >= PX_C64 && out_dtype <= PX_F64, "c2r needs complex in, real out");
	std::vector<int> axes(axes_in, axes_in+naxes);
	for (auto& a : axes) { if (a < 0) a += ndim; PXS_REQUIRE(a >= 0 && a < ndim, "pxf_fft_nd: axis out of range"); }
	for (int k = 0; k < ndim; k++) if (shape[k] == 0) return 0;
	for (int a : axes) {
		std::string why;
		if (kind >= 3) {
			const long N = r2r_length(kind, shape[a]);
			if (N < 2) throw Error(PXS_ERR_ARG, "DCT-I needs at least 2 points along each axis");
			if (!FftContext::supported(N, &why)) throw Error(PXS_ERR_UNSUPPORTED, "DCT/DST of " + std::to_string(shape[a]) + " points: " + why);
		}
	}
	PXS_HIP(hipSetDevice(device));
	FftContext& fc = fft_context(device);
	g_fft_device = device;
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
	if (kind >= 3) {
		// FFTW r2r kinds (pixell/fft.py:211-290): each is the real part of a complex FFT of an extension of the line --
		// mirror extension or zero padding with a half-sample phase ramp as a load functor, the other phase ramp, the bin
		// offset and "2 Re" / "-2 Im" in the store functor.  Only the first n bins are stored.
		std::vector<long> rshape(shape, shape+ndim);
		for (int t = 0; t < naxes; t++) {
			int ax = axes[naxes-1-t];
			bool first = (t == 0), lastpass = (t == naxes-1);
			const R2RPlan& rp = r2r_plan(device, kind, shape[ax]);
			FftLoad ld; FftStore stf;
			ld.ptr = first ? in : out; ld.dtype = first ? in_dtype : out_dtype;
			ld.mode = rp.mirror ? LD_MIRROR : LD_PLAIN; ld.ne = shape[ax]; ld.mir_c = rp.mir_c; ld.par0 = 0; ld.par_step = 0;
			ld.shift = rp.ld_shift; ld.mul = rp.ld_mul.p ? rp.ld_mul.as<double2>() : nullptr;
			stf.ptr = out; stf.dtype = out_dtype; stf.ne = shape[ax] + rp.st_shift; stf.shift = rp.st_shift;
			stf.mul = rp.st_mul.p ? rp.st_mul.as<double2>() : nullptr;
			stf.scale = (lastpass ? scale : 1.0)*rp.scale;
			const int64_t* is = first ? istride : ostride;
			fft_axis(fc, st, rp.N, rp.forward, other_dims(ax, rshape, is, ostride), is[ax], ostride[ax], ld, stf);
		}
	} else if (kind == 0) {
		{	// 2-D transforms over the last two axes of dense arrays (enmap.fft / ifft) through the chain stages.  Real maps: two rows per
			// complex line, the Hermitian half carried through the column passes; complex input: rows into a transposed intermediate, columns back
			const long minpix = [] { const char* e = getenv("PXS_FFT2_FAST_MINPIX"); return e ? atol(e) : (1L << 16); }();    // (-1: never; read per call: the tests switch it)
			const bool real_in = in_dtype <= PX_F64;
			bool dense = ndim >= 2 && naxes == 2 && ((axes[0] == ndim-2 && axes[1] == ndim-1) || (axes[0] == ndim-1 && axes[1] == ndim-2)) && (in != out || !real_in)
				&& (real_in || in_dtype == PX_C128) && out_dtype == PX_C128 && minpix >= 0 && (long)shape[ndim-1]*shape[ndim-2] >= minpix;
			long acc = 1, npre = 1;
			for (int k = ndim-1; k >= 0 && dense; k--) { dense = istride[k] == acc && ostride[k] == acc; acc *= shape[k]; if (k < ndim-2) npre *= shape[k]; }
			if (dense) {
				const std::shared_ptr<FftChain> ch = f2_chain(device, st, fc);
				if (real_in ? ch->fft2_real(st, in, in_dtype, (double2*)out, npre, shape[ndim-2], shape[ndim-1], forward != 0, scale)
				            : ch->fft2_c2c(st, (const double2*)in, (double2*)out, npre, shape[ndim-2], shape[ndim-1], forward != 0, scale)) return 0;
			}
		}
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
		if (naxes > 1) {
			size_t tot = 1; for (int k = 0; k < ndim; k++) tot *= cshape[k];
			DevBuf* sp;
			{ std::lock_guard<std::mutex> g(g_c2r_mu); auto& u = g_c2r_scratch[std::make_pair(device, st)]; if (!u) u.reset(new DevBuf()); sp = u.get(); }
			sp->ensure(tot*sizeof(double2));
			DevBuf& scratch = *sp;
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
	}
	PXS_CATCH
}

} // extern "C"
