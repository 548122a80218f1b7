// SHT plans and pipelines (C-ABI pxs_* of include/pxsht.h) for gfx950.
//
//   synthesis        alm --Legendre--> leg[m][ring] --transpose*phase--> h[ring][m] --c2r ring FFT--> map
//   adjoint synth.   map --r2c ring FFT (pruned to m<=mmax)--> h[ring][m] --transpose*phase--> leg --Legendre^T--> alm
//   analysis_2d      map --ring FFT--> leg on the map's rings --theta resampling--> leg on a minimal
//                    Clenshaw-Curtis grid (lmax+2.. rings) incl. exact |sin| quadrature --Legendre^T--> alm
//   adj. analysis    the exact transpose of the above
//
// The theta resampling integrates the trigonometric interpolant exactly: mirror-extend each m
// column to the full circle with parity (-1)^(m+s), FFT, resize the spectrum onto a fine grid
// of M > N + 2 lmax points, multiply by the truncated Fourier series of |sin theta| there, FFT
// back, keep |k| <= lmax and evaluate on the CC grid.  All index remaps are fused into the FFT
// load/store functors (fft.hpp).
// flip_y / flip_x (curvedsky.map2buffer, curvedsky.py:1384-1411) are negative strides here.
#include "../../include/pxsht.h"
#include "fft.hpp"
#include "legendre.hpp"
#include "fftchain.hpp"
#include <map>
#include <memory>
#include <cmath>
#include <mutex>
#include <algorithm>

namespace pxs {

FftContext& fft_context(int device);
void fft_dense_lines(int device, hipStream_t st, long n, bool forward, long nlines, const double2* in, double2* out);   // api_fft.hip
void fft_release_stream(int device, hipStream_t st);                                                                  // api_fft.hip
const char* get_last_error();

typedef long double LDb;
static const LDb PIl = 3.141592653589793238462643383279502884L;

struct GridInfo { long N; int c; bool ok; };
static GridInfo grid_info(const std::string& g, int n) {
	if (g == "CC")     return {2L*n-2, 0, true};
	if (g == "F1")     return {2L*n,   1, true};
	if (g == "MW")     return {2L*n-1, 1, true};
	if (g == "MWflip") return {2L*n-1, 0, true};
	// Driscoll-Healy (north pole + n-1 interior rings of spacing pi/n) and Fejer-2 (n interior rings of spacing pi/(n+1)): the
	// circle has sample points without a ring (a pole), so there is no interpolant to integrate: analysis on these grids is plain
	// quadrature with Fejer's second rule, exact up to get_ducc_maxlmax (curvedsky.py:1349-1353)
	if (g == "DH")     return {2L*n,   0, true};
	if (g == "F2")     return {2L*n+2, 2, true};
	return {0, 0, false};
}
static bool grid_weights_only(const std::string& g) { return g == "DH" || g == "F2"; }
// Fejer's second rule on K interior nodes theta_j = j pi/(K+1): w_j = 4 sin(theta_j)/(K+1) sum_{k=1}^{(K+1)/2} sin((2k-1) theta_j)/(2k-1)
// (ring weights, sum 2); the inner sum by the Chebyshev recurrence sin((2k+1)t) = 2 cos(2t) sin((2k-1)t) - sin((2k-3)t)
static std::vector<LDb> fejer2_weights(int K) {
	std::vector<LDb> w(K);
	for (int j = 1; j <= K; j++) {
		const LDb t = (LDb)j*PIl/(K+1), c2 = 2*cosl(2*t);
		LDb sm = -sinl(t), s0 = sinl(t), acc = 0;          // sin(-t), sin(t)
		for (int k = 1; k <= (K+1)/2; k++) { acc += s0/(2*k-1); const LDb sn = c2*s0 - sm; sm = s0; s0 = sn; }
		w[j-1] = 4*sinl(t)/(K+1)*acc;
	}
	return w;
}
static int grid_maxlmax(const std::string& g, int n) {
	if (g == "CC") return n-2;
	if (g == "DH") return (n-2)/2;
	if (g == "F2") return (n-1)/2;
	return n-1;
}
static std::vector<LDb> grid_theta(const std::string& g, int n) {
	GridInfo gi = grid_info(g, n);
	std::vector<LDb> th(n);
	for (int j = 0; j < n; j++) th[j] = (LDb)gi.c*PIl/gi.N + 2*PIl*j/gi.N;
	return th;
}
// Fourier coefficients of |sin|: s_q = -(2/pi)/(q^2-1) (q even), 0 (q odd)
static double abs_sin_coef(long q) { if (q & 1) return 0.0; LDb Q = q; return (double)(-(2/PIl)/(Q*Q-1)); }

// ---- glue kernels ----------------------------------------------------------------------
// out[b][c][r] = in[b][r][c] * tab[c] (optionally conj(tab)); 32x32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_mul(const double2* __restrict__ in, double2* __restrict__ out,
		int nr, int nc, long in_bstride, long out_bstride, const double2* __restrict__ tab, int conj_tab, double scale)
{
	PXS_SHARED(double2, tile);   // [32][33]
	const int b = blockIdx.z;
	const int r0 = blockIdx.y*32, c0 = blockIdx.x*32;
	const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
	in += (long)b*in_bstride; out += (long)b*out_bstride;
	for (int j = ty; j < 32; j += 8) {
		const int r = r0 + j, c = c0 + tx;
		if (r < nr && c < nc) tile[j*33 + tx] = in[(long)r*nc + c];
	}
	__syncthreads();
	for (int j = ty; j < 32; j += 8) {
		const int c = c0 + j, r = r0 + tx;
		if (r < nr && c < nc) {
			double2 v = tile[tx*33 + j];
			if (tab) { double2 t = tab[c]; if (conj_tab) t.y = -t.y; v = make_double2(v.x*t.x - v.y*t.y, v.x*t.y + v.y*t.x); }
			v.x *= scale; v.y *= scale;
			out[(long)c*nr + r] = v;
		}
	}
}
// out[b][r][c] = in[b][c][r] * tab[c]: same kernel with roles swapped is enough (tab indexed by the OUTPUT column)
__global__ __launch_bounds__(256) void transpose_mul_outcol(const double2* __restrict__ in, double2* __restrict__ out,
		int nr, int nc, long in_bstride, long out_bstride, const double2* __restrict__ tab, int conj_tab, double scale, long ldin, long ldout)
{
	// in[b][c][r] (nc rows of length nr) -> out[b][r][c]
	PXS_SHARED(double2, tile);
	const int b = blockIdx.z;
	const int r0 = blockIdx.y*32, c0 = blockIdx.x*32;
	const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
	in += (long)b*in_bstride; out += (long)b*out_bstride;
	for (int j = ty; j < 32; j += 8) {
		const int c = c0 + j, r = r0 + tx;
		if (r < nr && c < nc) tile[j*33 + tx] = in[(long)c*ldin + r];
	}
	__syncthreads();
	for (int j = ty; j < 32; j += 8) {
		const int r = r0 + j, c = c0 + tx;
		if (r < nr && c < nc) {
			double2 v = tile[tx*33 + j];
			if (tab) { double2 t = tab[c]; if (conj_tab) t.y = -t.y; v = make_double2(v.x*t.x - v.y*t.y, v.x*t.y + v.y*t.x); }
			v.x *= scale; v.y *= scale;
			out[(long)r*ldout + c] = v;
		}
	}
}

// ring pairs were transformed as one complex line z = a + i b; rows hold Z[0..k] then Z[n-1..n-k] (k = mmax).
// X_a[m] = (Z[m] + conj(Z[n-m]))/2, X_b[m] = -i (Z[m] - conj(Z[n-m]))/2  ->  leg[b][m][2q], leg[b][m][2q+1], times tab[m]*scale.
__global__ __launch_bounds__(256) void unpack_pair_transpose(const double2* __restrict__ in, double2* __restrict__ out,
		int nr, int nm, long in_bstride, long out_bstride, const double2* __restrict__ tab, double scale)
{
	PXS_SHARED(double2, tile);   // two [32][33] tiles
	double2* tp = tile; double2* tm = tile + 32*33;
	const int b = blockIdx.z;
	const int q0 = blockIdx.y*32, m0 = blockIdx.x*32;
	const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
	const int npair = (nr + 1)/2, k = nm - 1, rowlen = 2*k + 1;
	in += (long)b*in_bstride; out += (long)b*out_bstride;
	for (int j = ty; j < 32; j += 8) {
		const int q = q0 + j, m = m0 + tx;
		if (q < npair && m < nm) {
			tp[j*33 + tx] = in[(long)q*rowlen + m];
			tm[j*33 + tx] = in[(long)q*rowlen + (m == 0 ? 0 : k + m)];
		}
	}
	__syncthreads();
	for (int j = ty; j < 32; j += 8) {
		const int m = m0 + j, q = q0 + tx;
		if (q < npair && m < nm) {
			const double2 zp = tp[tx*33 + j], zm = tm[tx*33 + j];
			// conj(zm) = (zm.x, -zm.y)
			double2 xa = make_double2(0.5*(zp.x + zm.x), 0.5*(zp.y - zm.y));
			double2 d  = make_double2(zp.x - zm.x, zp.y + zm.y);
			double2 xb = make_double2(0.5*d.y, -0.5*d.x);                 // -i d / 2
			const double2 t = tab[m];
			xa = make_double2((xa.x*t.x - xa.y*t.y)*scale, (xa.x*t.y + xa.y*t.x)*scale);
			xb = make_double2((xb.x*t.x - xb.y*t.y)*scale, (xb.x*t.y + xb.y*t.x)*scale);
			out[(long)m*nr + 2*q] = xa;
			if (2*q + 1 < nr) out[(long)m*nr + 2*q + 1] = xb;
		}
	}
}

// aliased rings (mmax >= nphi): leg[b][m][r] = h[b][r][m mod nphi] * tab[m]   (rare; simple gather)
__global__ __launch_bounds__(256) void gather_alias(const double2* __restrict__ in, double2* __restrict__ out,
		int nr, int nm, int ncin, int nphi, long in_bstride, long out_bstride, const double2* __restrict__ tab, double scale)
{
	const long idx = (long)blockIdx.x*blockDim.x + threadIdx.x;
	const int b = blockIdx.y;
	if (idx >= (long)nr*nm) return;
	const int m = (int)(idx / nr), r = (int)(idx - (long)m*nr);
	double2 v = in[(long)b*in_bstride + (long)r*ncin + (m % nphi)];
	const double2 t = tab[m];
	v = make_double2((v.x*t.x - v.y*t.y)*scale, (v.x*t.y + v.y*t.x)*scale);
	out[(long)b*out_bstride + idx] = v;
}

// ---- general ring sets (per-ring nphi, phi0, ringstart: healpix, profile rings; ducc's synthesis / adjoint_synthesis contract,
// curvedsky.py:328-349, 396-403, 537, 553, 936-960) ---------------------------------------------------------------------
// Rings are grouped by length; z holds one complex line per ring, group after group, so that every group is a dense
// [rings][n] block for one batched FFT.  A block of 256 threads works on 256 consecutive elements of one ring (blk_ring, blk_k0).
struct GenK {
	int nring, nm, nc; long ldleg, leg_cstride, zc, map_cstride, pix_stride;
	const int* blk_ring; const int* blk_k0; const int* nphi; const long* zoff; const long* rstart; const double* phi0;
	double scale;
};
// ring spectrum with aliasing: H[k] = sum_{m = k mod n, m <= mmax} f_m leg[m][r] e^{i m phi0_r}, f_0 = 1, f_m = 2: the real ring is
// Re of the backward DFT of H
__global__ __launch_bounds__(256) void gen_fold(const GenK g, const double2* __restrict__ leg, double2* __restrict__ z)
{
	const int r = g.blk_ring[blockIdx.x], k = g.blk_k0[blockIdx.x] + (int)threadIdx.x, c = blockIdx.y;
	const int n = g.nphi[r];
	if (k >= n) return;
	const double ph = g.phi0[r];
	const double2* col = leg + (long)c*g.leg_cstride + r;
	double ar = 0, ai = 0;
	for (int m = k; m < g.nm; m += n) {
		const double2 a = col[(long)m*g.ldleg];
		double sn, cs; sincos((double)m*ph, &sn, &cs);
		const double f = m ? 2.0 : 1.0;
		ar += f*(a.x*cs - a.y*sn); ai += f*(a.x*sn + a.y*cs);
	}
	z[(long)c*g.zc + g.zoff[r] + k] = make_double2(ar, ai);
}
template<class T> __global__ __launch_bounds__(256) void gen_scatter(const GenK g, const double2* __restrict__ z, T* __restrict__ map)
{
	const int r = g.blk_ring[blockIdx.x], j = g.blk_k0[blockIdx.x] + (int)threadIdx.x, c = blockIdx.y;
	if (j >= g.nphi[r]) return;
	map[(long)c*g.map_cstride + g.rstart[r] + (long)j*g.pix_stride] = (T)z[(long)c*g.zc + g.zoff[r] + j].x;
}
template<class T> __global__ __launch_bounds__(256) void gen_gather(const GenK g, const T* __restrict__ map, double2* __restrict__ z)
{
	const int r = g.blk_ring[blockIdx.x], j = g.blk_k0[blockIdx.x] + (int)threadIdx.x, c = blockIdx.y;
	if (j >= g.nphi[r]) return;
	z[(long)c*g.zc + g.zoff[r] + j] = make_double2((double)map[(long)c*g.map_cstride + g.rstart[r] + (long)j*g.pix_stride], 0.0);
}
// leg[m][r] = scale e^{-i m phi0_r} F_r[m mod n_r]
__global__ __launch_bounds__(256) void gen_unfold(const GenK g, const double2* __restrict__ z, double2* __restrict__ leg)
{
	const int r = blockIdx.x*256 + (int)threadIdx.x, c = blockIdx.z;
	if (r >= g.nring) return;
	const int n = g.nphi[r]; const double ph = g.phi0[r];
	const double2* line = z + (long)c*g.zc + g.zoff[r];
	for (int m = blockIdx.y; m < g.nm; m += gridDim.y) {
		const double2 a = line[m % n];
		double sn, cs; sincos((double)m*ph, &sn, &cs);
		leg[(long)c*g.leg_cstride + (long)m*g.ldleg + r] = make_double2(g.scale*(a.x*cs + a.y*sn), g.scale*(a.y*cs - a.x*sn));
	}
}

// leg[line][ring] *= w[ring] (DH / F2 analysis: plain quadrature weights)
__global__ __launch_bounds__(256) void scale_rings(double2* __restrict__ leg, long nlines, int nr, long ld, const double2* __restrict__ w)
{
	const long idx = (long)blockIdx.x*blockDim.x + threadIdx.x;
	if (idx >= nlines*nr) return;
	const long line = idx / nr; const int j = (int)(idx - line*nr);
	double2& v = leg[line*ld + j]; const double f = w[j].x;
	v.x *= f; v.y *= f;
}

// transpose of the parity mirror extension: out[line][j] = in[line][j] + sgn * in[line][mirror(j)], j < nr
// (self-mirrored samples -- pole rings -- are kept for even parity and dropped for odd parity)
__global__ __launch_bounds__(256) void fold_mirror(const double2* __restrict__ in, double2* __restrict__ out,
		int nr, long N, int c, long nlines, int par0)
{
	const long idx = (long)blockIdx.x*blockDim.x + threadIdx.x;
	if (idx >= nlines*nr) return;
	const long line = idx / nr; const int j = (int)(idx - line*nr);
	const bool odd = ((line + par0) & 1) != 0;
	long mj = N - j - c; if (mj >= N) mj -= N; if (mj < 0) mj += N;
	double2 v = in[line*N + j];
	if (mj == j) { if (odd) v = make_double2(0, 0); }
	else { const double2 w = in[line*N + mj]; if (odd) { v.x -= w.x; v.y -= w.y; } else { v.x += w.x; v.y += w.y; } }
	out[line*nr + j] = v;
}

// undo the pair packing of the theta-FFT chain: Z holds even(theta)+odd(theta) on the full circle of N points;
// out[line 2p + (even? ..)][j] = (Z[j] +- Z[mirror(j)])/2 * w[j] for the real rings j < nr.
__global__ __launch_bounds__(256) void split_pair(const double2* __restrict__ Z, double2* __restrict__ out,
		int nr, long N, int c, long npairs, long nlines, int par0, const double2* __restrict__ w, double scale)
{
	const long idx = (long)blockIdx.x*blockDim.x + threadIdx.x;
	if (idx >= npairs*nr) return;
	const long pr = idx / nr; const int j = (int)(idx - pr*nr);
	long mj = N - j - c; if (mj >= N) mj -= N; if (mj < 0) mj += N;
	const double2 z = Z[pr*N + j];
	double2 ev, od;
	if (mj == j) { ev = z; od = make_double2(0, 0); }
	else { const double2 y = Z[pr*N + mj]; ev = make_double2(0.5*(z.x + y.x), 0.5*(z.y + y.y)); od = make_double2(0.5*(z.x - y.x), 0.5*(z.y - y.y)); }
	const double f = scale*(w ? w[j].x : 1.0);
	ev.x *= f; ev.y *= f; od.x *= f; od.y *= f;
	const bool a_odd = (par0 & 1) != 0;
	const long la = 2*pr, lb = la + 1;
	out[la*nr + j] = a_odd ? od : ev;
	if (lb < nlines) out[lb*nr + j] = a_odd ? ev : od;
}

// split_pair fused with the transpose of leg2map (synthesis through the CC grid): Z[pair][N] -> h[ring][m] * conj(tab[m]) * scale,
// through a 32 (m) x 32 (ring) LDS tile; saves writing and re-reading leg[m][ring] (2 x 10 GB at config 3)
__global__ __launch_bounds__(256) void split_pair_transposed(const double2* __restrict__ Z, double2* __restrict__ out,
		int nr, int nm, long N, int c, int par0, const double2* __restrict__ tab, double scale)
{
	PXS_SHARED(double2, tile);                       // [32 m][33]
	const int r0 = blockIdx.y*32, m0 = blockIdx.x*32;
	const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // tx: ring within tile, ty: 0..7
	const bool a_odd = (par0 & 1) != 0;              // parity of column 0; m0 is even, so of every even column
	for (int pj = ty; pj < 16; pj += 8) {            // 16 pairs = 32 columns m0 + 2 pj, m0 + 2 pj + 1
		const long pr = (m0 >> 1) + pj; const int j = r0 + tx;
		double2 ev = make_double2(0, 0), od = make_double2(0, 0);
		if (j < nr && 2*pr < nm) {
			long mj = N - j - c; if (mj >= N) mj -= N; if (mj < 0) mj += N;
			const double2 z = Z[pr*N + j];
			if (mj == j) ev = z;
			else { const double2 y = Z[pr*N + mj]; ev = make_double2(0.5*(z.x + y.x), 0.5*(z.y + y.y)); od = make_double2(0.5*(z.x - y.x), 0.5*(z.y - y.y)); }
		}
		tile[(2*pj)*33 + tx] = a_odd ? od : ev;
		tile[(2*pj+1)*33 + tx] = a_odd ? ev : od;
	}
	__syncthreads();
	for (int j = ty; j < 32; j += 8) {
		const int r = r0 + j, m = m0 + tx;
		if (r < nr && m < nm) {
			const double2 v = tile[tx*33 + j];
			const double2 t = tab[m];
			out[(long)r*nm + m] = make_double2((v.x*t.x + v.y*t.y)*scale, (v.y*t.x - v.x*t.y)*scale);     // * conj(tab[m]) * scale
		}
	}
}

} // namespace pxs

using namespace pxs;

struct pxs_plan {
	// a plan serves one call at a time (it owns the scratch of the call, and the analysis option is read several times in a call):
	// the entry points that run or reconfigure it take this lock, so two host threads on one cached plan queue up instead of interleaving
	std::mutex call_mu;
	int device = 0;
	bool is_grid = false;
	std::string geometry;
	int nring = 0, nphi = 0;
	// rows of a larger F1 grid (declination bands through pxs_plan_rings): the grid has nfull rings, the map's first ring is its
	// ring row0.  Such plans share the CC-grid machinery of the grid plans for synthesis and its adjoint.  Grid plans: nfull = nring.
	int nfull = 0, row0 = 0; bool band = false;
	double phi0 = 0;
	long ring_off0 = 0, ring_stride = 0, pix_stride = 1;   // user-map offset of (ring r, pixel x) = ring_off0 + r*ring_stride + x*pix_stride
	int lmax = 0, mmax = 0; long lstride = 1;
	DevBuf d_mstart;
	RingSet rs_map, rs_cc;
	std::map<int, std::unique_ptr<LegTables>> tables;
	LegWork wk;
	DevBuf leg, leg2, hbuf, phase;
	// analysis resampling (grid plans)
	long N = 0; int mir_c = 0; long M = 0, Ncc = 0; int ncc = 0;
	DevBuf ph_shift, ph_up, sigma, wcc, wadj, whalf, b1, b2;
	// general ring sets (gen_* kernels): rings of equal length are one dense block of z
	struct GenGroup { long n, count, zoff; };
	bool general = false; std::vector<GenGroup> groups; long npixz = 0;
	DevBuf g_blk_ring, g_blk_k0, g_nphi, g_zoff, g_rstart, g_phi0, gz; long g_nblk = 0;
	std::vector<hipStream_t> gstreams; std::vector<hipEvent_t> gjoin; hipEvent_t gfork = nullptr;
	~pxs_plan() {
		for (auto s_ : gstreams) { pxs::fft_release_stream(device, s_); (void)hipStreamDestroy(s_); }
		for (auto e : gjoin) (void)hipEventDestroy(e);
		if (gfork) (void)hipEventDestroy(gfork);
	}
	DevBuf wring;                // DH / F2 grids: per-ring quadrature weight / nphi (analysis = weighted adjoint synthesis)
	int ana_weights = 2;         // pxs_plan_option("analysis"): 2 = ducc0's route (default: the fine-CC form; ring weights on CC grids with ntheta >= 2 lmax + 2),
	                             // 0 = interpolant (|sin| series on the M circle), 1 = ring weights + adjoint synthesis where ntheta >= 2 lmax + 2;
	                             // PXS_ANALYSIS=ducc0|interpolant|weights presets it
	// the fine-CC form of the analysis (ducc0's resample_to_prepared_CC as published: the theta-interpolant, low-passed where the grid
	// has more than 2 N_cc circle samples, is evaluated on the CC grid of N_cc + 1 rings, multiplied by that grid's quadrature weights
	// and carried to the N_cc/2 + 1 rings of the Legendre stage by the transposed upsampling): the chain of the interpolant form with
	// M = 2 N_cc and the weights as the pointwise table
	ThetaPlan tpf; DevBuf sigma_f, wcc_f, whalf_f; long Mf = 0;      // tpf.ok: through the fused chains; Mf > 0: the tables exist (the unfused engine takes any plan)
	DevBuf wgrid, wgrid_ext, wunit_ext;     // ... its weights: get_gridweights / nphi per ring (built on first use); _ext: self-mirrored rings doubled (ring_weights_ext)
	bool syn_via_cc = false, syn_via_cc0 = false;      // synthesis through the CC grid: spin s / spin 0 (the recurrence of spin 0 is 4x cheaper per ring, the resampling is not)
	bool ring_pairs = true;      // transform two real rings per complex FFT (PXS_RING_PAIRS=0 disables)
	FftContext* fc = nullptr;
	// fused FFT chains (fftchain.hip).  chain_rings: ring FFTs through MA1/MA2, MS1/MS2; tp.ok: theta resampling through RA1-5 / RS1-3.
	// The chain paths use row strides padded to whole 128-byte lines for leg / leg_cc / h; the older unfused paths (kept for
	// aliased rings, lengths without a usable factorisation, the adjoint of the analysis) use dense rows.  A stride is chosen per call.
	std::unique_ptr<FftChain> chain; ThetaPlan tp; bool chain_rings = false;
	bool chain_theta() const { return chain_rings && tp.ok; }
	long ld_map() const { return chain_rings ? FftChain::pad8(nring) : nring; }
	long ld_cc()  const { return chain_theta() ? FftChain::pad8(ncc) : ncc; }
	long ld_h()   const { return chain_rings ? FftChain::pad8(mmax + 1) : mmax + 1; }
	FftChain::MapDesc map_desc(const void* map, int dtype, long cstride, long bstride = 0, int ncb = 0) const {
		FftChain::MapDesc m; m.ptr = map; m.dtype = dtype; m.cstride = cstride; m.ring_off0 = ring_off0; m.ring_stride = ring_stride; m.pix_stride = pix_stride; m.nring = nring; m.nphi = nphi;
		m.bstride = bstride; m.ncb = ncb;
		return m; }
	LegProfile prof;
	size_t resample_chunk_bytes = size_t(1) << 40;    // per intermediate buffer of the theta-FFT chain (chunking to stay in the
	                                                  // Infinity Cache was measured slower: 22.4 -> 26.2 ms at config 2; PXS_RESAMPLE_MB re-enables it)

	LegTables& table(int spin) {
		auto& p = tables[spin];
		if (!p) { p.reset(new LegTables()); p->build(lmax, mmax, spin); }
		return *p;
	}
};

namespace {

void plan_common(pxs_plan* p, int lmax, int mmax, const uint64_t* mstart, int64_t lstride, int device) {
	PXS_REQUIRE(lmax >= 0 && mmax >= 0 && mmax <= lmax, "need 0 <= mmax <= lmax");
	PXS_REQUIRE(mstart != nullptr, "mstart is required");
	PXS_HIP(hipSetDevice(device));
	p->device = device; p->lmax = lmax; p->mmax = mmax; p->lstride = lstride;
	std::vector<uint64_t> ms(mstart, mstart+mmax+1);
	p->d_mstart = upload(ms);
	p->fc = &fft_context(device);
	{ const char* e = getenv("PXS_FFT_TEMP_MB"); if (e) p->fc->temp_budget = (size_t)atol(e) << 20; }
	{ const char* e = lab_getenv("PXS_RING_PAIRS"); if (e) p->ring_pairs = atoi(e) != 0; }
	{	// scratch for the per-wave partial moments of the analysis: 16 GiB where the device has room (MI355X: 288 GB),
		// never more than 1/8 of what is free now.  Measured at config 3: 4 GiB 448.8, 8 GiB 442.7, 16 GiB 438.0, 24 GiB 437.7 ms.
		size_t fr = 0, tot = 0;
		if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr > 0) p->wk.part_budget = std::min<size_t>(size_t(2) << 30, std::max<size_t>(fr/8, size_t(256) << 20));      // (ordered analysis only: <= 2 GB of partial moments, m in chunks)
		const char* e = getenv("PXS_PART_GB"); if (e) p->wk.part_budget = (size_t)atol(e) << 30;
		{ const char* d = getenv("PXS_DETERMINISTIC"); p->wk.deterministic = d && atoi(d) != 0; }      // default of pxs_plan_option("deterministic")
	}
	{ const char* e = getenv("PXS_RESAMPLE_MB"); if (e) p->resample_chunk_bytes = (size_t)atol(e) << 20; }
	{ const char* e = getenv("PXS_ANALYSIS"); if (e) { const std::string v(e); p->ana_weights = v == "weights" ? 1 : (v == "ducc0" ? 2 : (v == "interpolant" ? 0 : p->ana_weights)); } }
	if (p->general) return;      // per-ring lengths and phases: see setup_general
	std::string why;
	if (!FftContext::supported(p->nphi, &why)) throw Error(PXS_ERR_UNSUPPORTED, why);
	// e^{-i m phi0}
	std::vector<double2> ph(mmax+1);
	for (int m = 0; m <= mmax; m++) { LDb a = (LDb)m*(LDb)p->phi0; ph[m] = make_double2((double)cosl(a), (double)(-sinl(a))); }
	p->phase = upload(ph);
	{	static const bool use = [] { const char* e = lab_getenv("PXS_CHAIN"); return e ? atoi(e) != 0 : true; }();
		p->chain.reset(new FftChain(p->fc));
		p->chain_rings = use && p->ring_pairs && 2L*mmax < p->nphi && p->chain->plan_rings(p->nphi);
		if (getenv("PXS_CHAIN_VERBOSE")) fprintf(stderr, "[pxsht] ring chain nphi=%d mmax=%d: %s\n", p->nphi, mmax, p->chain_rings ? p->chain->describe().c_str() : "off");
	}
}

void setup_resampling(pxs_plan* p) {
	if (p->nfull == 0) p->nfull = p->nring;
	GridInfo gi = grid_info(p->geometry, p->nfull);
	p->N = gi.N; p->mir_c = gi.c;
	const int lmax = p->lmax;
	p->Ncc = FftContext::good_size(std::max<long>(2L*lmax + 2, 4));
	if (p->Ncc & 1) p->Ncc = FftContext::good_size(p->Ncc + 1);
	while (p->Ncc & 1) p->Ncc = FftContext::good_size(p->Ncc + 1);
	{	// ducc0's own N_cc = 2 good_size_complex(lmax + 1) where the FFT engine takes it and twice it (the fused chains decide for themselves below)
		static const bool ducc_size = [] { const char* e = lab_getenv("PXS_THETA_DUCC_NCC"); return e ? atoi(e) != 0 : true; }();
		const long nd = FftChain::ducc_ncc(lmax);
		if (ducc_size && nd >= 4 && FftContext::supported(nd) && FftContext::supported(2*nd)) p->Ncc = nd;
	}
	p->M = FftContext::good_size(p->N + 2L*lmax + 2);
	{ const char* e = lab_getenv("PXS_M_FINE"); if (e && atol(e) >= p->M && FftContext::supported(atol(e))) p->M = atol(e); }   // experiments: a larger fine grid with friendlier factors
	if (p->chain_rings) {	// the fused theta chains need N, M and Ncc to share a modulus: let their planner pick M and Ncc
		p->tp = FftChain::plan_theta(p->N, lmax);
		if (p->tp.ok) { p->Ncc = p->tp.Ncc; p->M = p->tp.M; }
		if (getenv("PXS_CHAIN_VERBOSE")) fprintf(stderr, "[pxsht] theta chain N=%ld lmax=%d: ok=%d g=%ld bN=%ld g2=%ld (M=%ld) ac=%ld (Ncc=%ld) | syn gs=%ld bs=%ld aNs=%ld\n",
			p->N, lmax, (int)p->tp.ok, p->tp.g, p->tp.bN, p->tp.g2, p->tp.M, p->tp.ac, p->tp.Ncc, p->tp.gs, p->tp.bs, p->tp.aNs);
	}
	p->ncc = (int)(p->Ncc/2 + 1);
	std::string why;
	if (!FftContext::supported(p->N, &why)) throw Error(PXS_ERR_UNSUPPORTED, why);
	// CC ring set
	std::vector<LDb> th(p->ncc);
	for (int j = 0; j < p->ncc; j++) th[j] = 2*PIl*j/p->Ncc;
	th[p->ncc-1] = PIl;
	p->rs_cc.build(th); p->rs_cc.upload_all();
	// shift phases e^{-i k theta0}, k = 0..N/2
	const LDb th0 = (LDb)gi.c*PIl/gi.N;
	std::vector<double2> ps(p->N/2 + 1);
	for (long k = 0; k <= p->N/2; k++) { LDb a = (LDb)k*th0; ps[k] = make_double2((double)cosl(a), (double)(-sinl(a))); }
	p->ph_shift = upload(ps);
	{ std::vector<double2> pu(ps.size()); for (size_t k = 0; k < ps.size(); k++) pu[k] = make_double2(ps[k].x, -ps[k].y); p->ph_up = upload(pu); }
	// synthesis: Legendre on the minimal CC grid + exact Fourier upsampling in theta pays once the map has clearly more rings
	// (measured: at nring / ncc = 1.33 -- C2, C4 -- the detour pays for spin 2 and costs 11 ms per 64 scalar maps at C4)
	// (a band: what counts is the ring-pair slots of its own rings, unpaired rings take a whole slot)
	{ const char* e = lab_getenv("PXS_SYN_VIA_CC"); const long rings = p->band ? 2L*p->rs_map.npairs : p->nring;
	  p->syn_via_cc = e ? atoi(e) != 0 : (rings > p->ncc + p->ncc/4);
	  p->syn_via_cc0 = e ? atoi(e) != 0 : (rings > p->ncc + p->ncc/2); }
	// sigma_i = sum_{|q|<=Ks} s_q e^{i q theta_i} on the M grid, via one device FFT
	const long Ks = lmax + p->N/2;
	PXS_REQUIRE(2*Ks < p->M, "internal: fine grid too small");
	std::vector<double2> spec(p->M, make_double2(0, 0));
	for (long q = 0; q <= Ks; q++) { double s = abs_sin_coef(q); spec[q].x = s; if (q > 0) spec[p->M - q].x = s; }
	DevBuf dspec = upload(spec);
	p->sigma.alloc(sizeof(double2)*p->M);
	FftDims d; d.n_i = 1; d.is_e = 1; d.os_e = 1;
	FftLoad ld; ld.ptr = dspec.p; FftStore st; st.ptr = p->sigma.p;
	p->fc->exec(nullptr, p->M, false, d, ld, st);
	PXS_HIP(hipStreamSynchronize(nullptr));
	// CC weights incl. all FFT normalisations: (pi/nphi)(2pi/Ncc) eps_j / (N M)
	std::vector<double2> w(p->ncc);
	for (int j = 0; j < p->ncc; j++) {
		LDb e = (j == 0 || j == p->ncc-1) ? 1 : 2;
		w[j] = make_double2((double)((PIl/p->nphi)*(2*PIl/p->Ncc)*e/((LDb)p->N*(LDb)p->M)), 0.0);
	}
	p->wcc = upload(w);
	{	// the same weights for the fused adjoint of the analysis (FftChain::to_cc_adjoint): halved off the two pole rings
		std::vector<double2> wh(w); for (int j = 1; j + 1 < p->ncc; j++) wh[j].x *= 0.5;
		p->whalf = upload(wh); }
	{	// weights of the transposed theta upsampling (FftChain::from_cc_adjoint): 1/N_cc, half at the two pole rings
		std::vector<double2> wa(p->ncc);
		for (int j = 0; j < p->ncc; j++) wa[j] = make_double2(((j == 0 || j == p->ncc-1) ? 0.5 : 1.0)/(double)p->Ncc, 0.0);
		p->wadj = upload(wa); }
	if (FftContext::supported(2*p->Ncc)) {	// the fine-CC form: M = 2 N_cc (through the fused chains: = g * (2 ac))
		if (p->tp.ok && FftChain::sub_ok_theta(2*p->tp.ac)) { p->tpf = p->tp; p->tpf.g2 = 2*p->tp.ac; p->tpf.M = 2*p->Ncc; } else p->tpf.ok = false;
		const long Mf = p->Mf = 2*p->Ncc; const int nf = (int)(Mf/2 + 1);
		std::vector<double> gw(nf);
		if (pxs_gridweights("CC", nf, gw.data()) != 0) throw Error(PXS_ERR_ARG, get_last_error());
		// table on the circle: (Mf / 2 pi) x the weight of the full-circle rule int g |sin| = sum_i t_i g(theta_i): ring weight
		// (sum over the rings = 2) at both images of a ring, twice that at the two poles, which the circle holds once
		std::vector<double2> sg(Mf);
		for (long i = 0; i < Mf; i++) {
			const long r = std::min(i, Mf - i);
			const LDb wr = (LDb)gw[r]/(2*PIl)*((r == 0 || r == Mf/2) ? 2 : 1);
			sg[i] = make_double2((double)(wr*(LDb)Mf/(2*PIl)), 0.0);
		}
		p->sigma_f = upload(sg);
		const double f = (double)p->M/(double)Mf;       // the FFT normalisation 1/(N M) of the weights, for this M
		std::vector<double2> wf(w), whf(w);
		for (int j = 0; j < p->ncc; j++) { wf[j].x *= f; whf[j].x *= (j == 0 || j == p->ncc-1) ? f : 0.5*f; }
		p->wcc_f = upload(wf); p->whalf_f = upload(whf);
	}
}

// tables of a general ring set: groups of equal length, z offsets, the block table of the gen_* kernels
void setup_general(pxs_plan* p, int nring, const uint64_t* nphi, const double* phi0, const uint64_t* ringstart) {
	std::vector<int> order(nring);
	for (int r = 0; r < nring; r++) { order[r] = r; PXS_REQUIRE(nphi[r] >= 1 && nphi[r] < (1ull << 31), "pxs_plan_rings: bad nphi"); }
	std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return nphi[a] < nphi[b]; });
	std::vector<int> np(nring), blk_ring, blk_k0; std::vector<long> zoff(nring), rstart(nring); std::vector<double> ph(phi0, phi0 + nring);
	long off = 0;
	p->groups.clear();
	for (int i = 0; i < nring; i++) {
		const int r = order[i];
		np[r] = (int)nphi[r]; rstart[r] = (long)ringstart[r]; zoff[r] = off;
		if (p->groups.empty() || p->groups.back().n != (long)nphi[r]) p->groups.push_back(pxs_plan::GenGroup{(long)nphi[r], 0, off});
		p->groups.back().count++;
		for (long k0 = 0; k0 < (long)nphi[r]; k0 += 256) { blk_ring.push_back(r); blk_k0.push_back((int)k0); }
		off += (long)nphi[r];
	}
	p->npixz = off; p->g_nblk = (long)blk_ring.size();
	p->g_blk_ring = upload(blk_ring); p->g_blk_k0 = upload(blk_k0); p->g_nphi = upload(np); p->g_zoff = upload(zoff);
	p->g_rstart = upload(rstart); p->g_phi0 = upload(ph);
	if (p->groups.size() >= 8) {
		static const int ns = [] { const char* e = lab_getenv("PXS_GEN_STREAMS"); return e ? std::max(0, atoi(e)) : 16; }();
		p->gstreams.resize(ns); p->gjoin.resize(ns);
		for (int i = 0; i < ns; i++) { PXS_HIP(hipStreamCreateWithFlags(&p->gstreams[i], hipStreamNonBlocking)); PXS_HIP(hipEventCreateWithFlags(&p->gjoin[i], hipEventDisableTiming)); }
		PXS_HIP(hipEventCreateWithFlags(&p->gfork, hipEventDisableTiming));
	}
}
static GenK gen_args(pxs_plan* p, int nc, long ldleg, long map_cstride, double scale) {
	GenK g; g.nring = p->nring; g.nm = p->mmax + 1; g.nc = nc; g.ldleg = ldleg; g.leg_cstride = (long)(p->mmax + 1)*ldleg; g.zc = p->npixz;
	g.map_cstride = map_cstride; g.pix_stride = p->pix_stride;
	g.blk_ring = p->g_blk_ring.as<int>(); g.blk_k0 = p->g_blk_k0.as<int>(); g.nphi = p->g_nphi.as<int>(); g.zoff = p->g_zoff.as<long>();
	g.rstart = p->g_rstart.as<long>(); g.phi0 = p->g_phi0.as<double>(); g.scale = scale;
	return g;
}
// one batched FFT per (component, length).  A healpix map has 2 nside lengths with two to four rings each: the launches are
// dealt round-robin to side streams that fork from / join the caller's stream, so that the many small transforms overlap
// (nside 2048, lmax 4096, 3 components: 282 ms on one stream, 94 ms on 16)
static void gen_ffts(pxs_plan* p, hipStream_t st, int nc, bool forward) {
	const size_t njobs = (size_t)nc*p->groups.size();
	const size_t ns = njobs >= 16 ? p->gstreams.size() : 0;
	if (ns) {
		PXS_HIP(hipEventRecord(p->gfork, st));
		for (size_t i = 0; i < ns; i++) PXS_HIP(hipStreamWaitEvent(p->gstreams[i], p->gfork, 0));
	}
	size_t job = 0;
	for (int c = 0; c < nc; c++)
		for (const auto& gr : p->groups) {
			double2* zl = p->gz.as<double2>() + (size_t)c*p->npixz + gr.zoff;
			fft_dense_lines(p->device, ns ? p->gstreams[job % ns] : st, gr.n, forward, gr.count, zl, zl);
			job++;
		}
	for (size_t i = 0; i < ns; i++) { PXS_HIP(hipEventRecord(p->gjoin[i], p->gstreams[i])); PXS_HIP(hipStreamWaitEvent(st, p->gjoin[i], 0)); }
}

int ncomp_of(int spin, int mode, bool alm_side) {
	if (mode == PXS_MODE_DERIV1) return alm_side ? 1 : 2;
	return spin == 0 ? 1 : 2;
}

// ring FFT: user map -> hbuf[c][ring][m] -> leg[c][m][ring] * e^{-i m phi0} * scale
// (ld: row stride of leg; the unfused path below only writes dense rows, ld == nring)
void map2leg(pxs_plan* p, hipStream_t st, const void* map, int map_dtype, long map_cstride, int nc, double2* leg, double scale, long ldl, long map_bstride = 0, int ncb = 0) {
	p->prof.begin(st, PXS_STAGE_RING_FFT);
	const int nm = p->mmax+1, nr = p->nring;
	if (p->general) {
		PXS_REQUIRE(ncb == 0, "internal: general ring sets take one map per call");
		p->gz.ensure(sizeof(double2)*(size_t)nc*p->npixz);
		const GenK g = gen_args(p, nc, ldl, map_cstride, scale);
		const dim3 grid((unsigned)p->g_nblk, nc);
		if (map_dtype == PX_F32) hipLaunchKernelGGL(gen_gather<float>, grid, dim3(256), 0, st, g, (const float*)map, p->gz.as<double2>());
		else                     hipLaunchKernelGGL(gen_gather<double>, grid, dim3(256), 0, st, g, (const double*)map, p->gz.as<double2>());
		gen_ffts(p, st, nc, true);
		hipLaunchKernelGGL(gen_unfold, dim3((nr + 255)/256, std::min(nm, 4096), nc), dim3(256), 0, st, g, (const double2*)p->gz.p, leg);
		p->prof.end(st, PXS_STAGE_RING_FFT);
		PXS_HIP(hipGetLastError());
		return;
	}
	if (p->chain_rings) {
		p->chain->map2leg(st, p->map_desc(map, map_dtype, map_cstride, map_bstride, ncb), nc, p->mmax, leg, ldl, p->phase.as<double2>(), scale);
		p->prof.end(st, PXS_STAGE_RING_FFT);
		return;
	}
	PXS_REQUIRE(ldl == nr && ncb == 0, "internal: unfused ring FFT needs dense rows and single maps");
	auto esz = [](int dt) { return dt == PX_F32 ? 4 : 8; };
	if (2L*p->mmax < p->nphi && p->ring_pairs) {
		// two real rings per complex transform; the pruned two-sided spectrum (|k| <= mmax) is unpacked in the transpose
		const int npair = (nr + 1)/2; const long rowlen = 2L*p->mmax + 1;
		p->hbuf.ensure(sizeof(double2)*(size_t)nc*std::max<size_t>((size_t)npair*rowlen, (size_t)nr*nm));
		FftDims d; d.n_i = npair; d.is_i = p->ring_stride; d.os_i = rowlen; d.n_o1 = nc; d.is_o1 = map_cstride; d.os_o1 = (long)npair*rowlen;
		d.is_e = p->pix_stride; d.os_e = 1;
		FftLoad ld; ld.ptr = (const char*)map + esz(map_dtype)*p->ring_off0; ld.dtype = map_dtype; ld.mode = LD_REAL_PAIR; ld.pair_lines = nr;
		FftStore sf; sf.ptr = p->hbuf.p; sf.two_sided_k = p->mmax; sf.compact_two_sided = 1;
		p->fc->exec(st, p->nphi, true, d, ld, sf);
		dim3 grid((nm+31)/32, (npair+31)/32, nc);
		hipLaunchKernelGGL(unpack_pair_transpose, grid, dim3(256), sizeof(double2)*2*32*33, st, (const double2*)p->hbuf.p, leg, nr, nm,
			(long)npair*rowlen, (long)nr*nm, (const double2*)p->phase.p, scale);
		p->prof.end(st, PXS_STAGE_RING_FFT);
		PXS_HIP(hipGetLastError());
		return;
	}
	const int ncin = std::min(nm, p->nphi);           // distinct FFT bins needed (m >= nphi alias onto m mod nphi)
	p->hbuf.ensure(sizeof(double2)*(size_t)nc*nr*nm);
	FftDims d; d.n_i = nr; d.is_i = p->ring_stride; d.os_i = ncin; d.n_o1 = nc; d.is_o1 = map_cstride; d.os_o1 = (long)nr*ncin;
	d.is_e = p->pix_stride; d.os_e = 1;
	FftLoad ld; ld.ptr = (const char*)map + esz(map_dtype)*p->ring_off0; ld.dtype = map_dtype;
	FftStore sf; sf.ptr = p->hbuf.p; sf.ne = ncin;
	p->fc->exec(st, p->nphi, true, d, ld, sf);
	if (ncin == nm) {
		dim3 grid((nm+31)/32, (nr+31)/32, nc);
		hipLaunchKernelGGL(transpose_mul, grid, dim3(256), sizeof(double2)*32*33, st, (const double2*)p->hbuf.p, leg, nr, nm,
			(long)nr*nm, (long)nr*nm, (const double2*)p->phase.p, 0, scale);
	} else {
		dim3 grid((unsigned)(((long)nr*nm + 255)/256), nc);
		hipLaunchKernelGGL(gather_alias, grid, dim3(256), 0, st, (const double2*)p->hbuf.p, leg, nr, nm, ncin, p->nphi,
			(long)nr*ncin, (long)nr*nm, (const double2*)p->phase.p, scale);
	}
	p->prof.end(st, PXS_STAGE_RING_FFT);
	PXS_HIP(hipGetLastError());
}

// leg[c][m][ring] * e^{+i m phi0} -> hbuf[c][ring][m] -> c2r ring FFT -> user map
// (ldleg: row stride of leg; hbuf rows are ld_h() long)
void leg2map(pxs_plan* p, hipStream_t st, const double2* leg, long ldleg, void* map, int map_dtype, long map_cstride, int nc, bool have_h = false, long map_bstride = 0, int ncb = 0) {
	p->prof.begin(st, PXS_STAGE_RING_FFT);
	const int nm = p->mmax+1, nr = p->nring;
	if (p->general) {
		PXS_REQUIRE(ncb == 0 && !have_h, "internal: general ring sets take one map per call");
		p->gz.ensure(sizeof(double2)*(size_t)nc*p->npixz);
		const GenK g = gen_args(p, nc, ldleg, map_cstride, 1.0);
		const dim3 grid((unsigned)p->g_nblk, nc);
		hipLaunchKernelGGL(gen_fold, grid, dim3(256), 0, st, g, leg, p->gz.as<double2>());
		gen_ffts(p, st, nc, false);
		if (map_dtype == PX_F32) hipLaunchKernelGGL(gen_scatter<float>, grid, dim3(256), 0, st, g, (const double2*)p->gz.p, (float*)map);
		else                     hipLaunchKernelGGL(gen_scatter<double>, grid, dim3(256), 0, st, g, (const double2*)p->gz.p, (double*)map);
		p->prof.end(st, PXS_STAGE_RING_FFT);
		PXS_HIP(hipGetLastError());
		(void)nm;
		return;
	}
	const long ldh = p->ld_h();
	p->hbuf.ensure(sizeof(double2)*(size_t)nc*(have_h && p->band ? p->nfull : nr)*ldh);
	if (!have_h) {      // (the CC synthesis path has written hbuf already, see resample_from_cc / FftChain::from_cc)
		dim3 grid((nm+31)/32, (nr+31)/32, nc);
		hipLaunchKernelGGL(transpose_mul_outcol, grid, dim3(256), sizeof(double2)*32*33, st, leg, (double2*)p->hbuf.p, nr, nm,
			(long)nm*ldleg, (long)nr*ldh, (const double2*)p->phase.p, 1, 1.0, ldleg, ldh);
	}
	if (p->chain_rings) {
		// (have_h on a band plan: h holds all nfull rings of the grid, the map's rings start at row0)
		const bool full_h = have_h && p->band;
		p->chain->h2map(st, p->hbuf.as<double2>() + (full_h ? (size_t)p->row0*ldh : 0), ldh, p->map_desc(map, map_dtype, map_cstride, map_bstride, ncb), nc, p->mmax,
			full_h ? p->nfull : 0);
		p->prof.end(st, PXS_STAGE_RING_FFT);
		return;
	}
	PXS_REQUIRE(ncb == 0, "internal: unfused ring FFT handles single maps");
	auto esz = [](int dt) { return dt == PX_F32 ? 4 : 8; };
	if (2L*p->mmax < p->nphi && p->ring_pairs) {
		// two rings per complex transform: Z = X_a + i X_b, real part -> ring 2q, imaginary part -> ring 2q+1
		FftDims d; d.n_i = (nr + 1)/2; d.is_i = nm; d.os_i = p->ring_stride; d.n_o1 = nc; d.is_o1 = (long)nr*nm; d.os_o1 = map_cstride;
		d.is_e = 1; d.os_e = p->pix_stride;
		FftLoad ld; ld.ptr = p->hbuf.p; ld.mode = LD_HERM_PAIR; ld.ne = nm; ld.pair_lines = nr;
		FftStore sf; sf.ptr = (char*)map + esz(map_dtype)*p->ring_off0; sf.dtype = map_dtype; sf.real_pair = 1; sf.pair_lines = nr;
		p->fc->exec(st, p->nphi, false, d, ld, sf);
	} else {
		FftDims d; d.n_i = nr; d.is_i = nm; d.os_i = p->ring_stride; d.n_o1 = nc; d.is_o1 = (long)nr*nm; d.os_o1 = map_cstride;
		d.is_e = 1; d.os_e = p->pix_stride;
		FftLoad ld; ld.ptr = p->hbuf.p; ld.mode = LD_HERM; ld.ne = nm; ld.herm_fold = 1;
		FftStore sf; sf.ptr = (char*)map + esz(map_dtype)*p->ring_off0; sf.dtype = map_dtype;
		p->fc->exec(st, p->nphi, false, d, ld, sf);
	}
	p->prof.end(st, PXS_STAGE_RING_FFT);
	PXS_HIP(hipGetLastError());
}

// leg on the map's rings [c][m][nring] -> weighted leg on the CC grid [c][m][ncc].
// Columns m and m+1 have opposite theta-parity, so their mirror extensions are the even and the odd part
// of ONE sequence: each pair shares the whole 4-FFT chain (every stage is linear and commutes with the
// reflection theta -> -theta) and is separated again at the end -- half the FFT work.
// (M, sigma, w: the fine circle, the pointwise table on it and the CC weights of the form of the analysis -- the |sin| series on
// M > N + 2 lmax points, or ducc0's route: the CC weights on M = 2 N_cc points, with the spectrum cut to |k| < M/2 when M <= N)
void resample_to_cc(pxs_plan* p, hipStream_t st, const double2* leg_in, double2* leg_cc, int nc, int spin, long M, const double2* sigma, const double2* wcc) {
	const int nm = p->mmax+1, nr = p->nring;
	const long npair_all = (nm + 1)/2;
	const long chunk = std::max<long>(32, std::min<long>(npair_all, (long)(p->resample_chunk_bytes/(sizeof(double2)*M))));
	p->b1.ensure(sizeof(double2)*(size_t)chunk*std::max(p->N, p->Ncc));
	p->b2.ensure(sizeof(double2)*(size_t)chunk*M);
	p->prof.begin(st, PXS_STAGE_RESAMPLE);
	for (int c = 0; c < nc; c++)
	for (long p0 = 0; p0 < npair_all; p0 += chunk) {
		const long np = std::min<long>(chunk, npair_all - p0);
		const long m0 = 2*p0, nlines = std::min<long>(2*np, nm - m0);
		{	// (a) packed mirror extension of lines (2i, 2i+1), forward FFT_N
			FftDims d; d.n_i = np; d.is_i = nr; d.os_i = p->N; d.is_e = 1; d.os_e = 1;
			FftLoad ld; ld.ptr = leg_in + ((size_t)c*nm + m0)*nr; ld.mode = LD_MIRROR_PAIR; ld.ne = nr; ld.mir_c = p->mir_c;
			ld.par0 = (spin + (int)m0) & 1; ld.pair_lines = nlines;
			FftStore sf; sf.ptr = p->b1.p;
			p->fc->exec(st, p->N, true, d, ld, sf);
		}
		{	// (b) shift to theta0 = 0, pad to M, backward FFT_M, multiply by the |sin| series
			FftDims d; d.n_i = np; d.is_i = p->N; d.os_i = M; d.is_e = 1; d.os_e = 1;
			FftLoad ld; ld.ptr = p->b1.p; ld.mode = LD_SPEC; ld.ne = p->N; ld.nyq_half = M > p->N ? 1 : 0; ld.kmax = M > p->N ? -1 : M/2 - 1; ld.mul = p->ph_shift.as<double2>();
			FftStore sf; sf.ptr = p->b2.p; sf.mul = sigma;
			p->fc->exec(st, M, false, d, ld, sf);
		}
		{	// (c) forward FFT_M in place; only |k| <= lmax are needed
			FftDims d; d.n_i = np; d.is_i = M; d.os_i = M; d.is_e = 1; d.os_e = 1;
			FftLoad ld; ld.ptr = p->b2.p;
			FftStore sf; sf.ptr = p->b2.p; sf.two_sided_k = p->lmax;
			p->fc->exec(st, M, true, d, ld, sf);
		}
		{	// (d) truncate to |k| <= lmax, backward FFT_Ncc onto the full CC circle
			FftDims d; d.n_i = np; d.is_i = M; d.os_i = p->Ncc; d.is_e = 1; d.os_e = 1;
			FftLoad ld; ld.ptr = p->b2.p; ld.mode = LD_SPEC; ld.ne = M; ld.kmax = p->lmax;
			FftStore sf; sf.ptr = p->b1.p;
			p->fc->exec(st, p->Ncc, false, d, ld, sf);
		}
		{	// (e) separate the pair by reflection symmetry, keep rings 0..ncc-1, apply the quadrature weights
			const long tot = np*p->ncc;
			hipLaunchKernelGGL(split_pair, dim3((unsigned)((tot+255)/256)), dim3(256), 0, st, (const double2*)p->b1.p,
				leg_cc + ((size_t)c*nm + m0)*p->ncc, p->ncc, p->Ncc, 0, np, nlines, (spin + (int)m0) & 1, wcc, 1.0);
		}
	}
	p->prof.end(st, PXS_STAGE_RESAMPLE);
	PXS_HIP(hipGetLastError());
}

// exact transpose of resample_to_cc: leg on the CC grid [c][m][ncc] -> leg on the map's rings [c][m][nring]
void resample_to_cc_adjoint(pxs_plan* p, hipStream_t st, const double2* leg_cc, double2* leg_out, int nc, int spin, long M, const double2* sigma, const double2* wcc) {
	const int nm = p->mmax+1, nr = p->nring;
	const long chunk = std::max<long>(32, std::min<long>(nm, (long)(p->resample_chunk_bytes/(sizeof(double2)*M))));
	p->b1.ensure(sizeof(double2)*(size_t)chunk*std::max(p->N, p->Ncc));
	p->b2.ensure(sizeof(double2)*(size_t)chunk*M);
	p->prof.begin(st, PXS_STAGE_RESAMPLE);
	for (int c = 0; c < nc; c++)
	for (long m0 = 0; m0 < nm; m0 += chunk) {
		const long nl = std::min<long>(chunk, nm - m0);
		{	// (d)^H: weights, zero-extend the rings to the Ncc circle, forward FFT_Ncc
			FftDims d; d.n_i = nl; d.is_i = p->ncc; d.os_i = p->Ncc; d.is_e = 1; d.os_e = 1;
			FftLoad ld; ld.ptr = leg_cc + ((size_t)c*nm + m0)*p->ncc; ld.ne = p->ncc; ld.mul = wcc;
			FftStore sf; sf.ptr = p->b1.p;
			p->fc->exec(st, p->Ncc, true, d, ld, sf);
		}
		{	// (c)^H: embed |k| <= lmax into the M spectrum, backward FFT_M, multiply by the |sin| series
			FftDims d; d.n_i = nl; d.is_i = p->Ncc; d.os_i = M; d.is_e = 1; d.os_e = 1;
			FftLoad ld; ld.ptr = p->b1.p; ld.mode = LD_SPEC; ld.ne = p->Ncc; ld.kmax = p->lmax;
			FftStore sf; sf.ptr = p->b2.p; sf.mul = sigma;
			p->fc->exec(st, M, false, d, ld, sf);
		}
		{	// (b)^H first half: forward FFT_M in place (only |k| <= N/2 needed)
			FftDims d; d.n_i = nl; d.is_i = M; d.os_i = M; d.is_e = 1; d.os_e = 1;
			FftLoad ld; ld.ptr = p->b2.p;
			FftStore sf; sf.ptr = p->b2.p; sf.two_sided_k = M > p->N ? p->N/2 : -1;
			p->fc->exec(st, M, true, d, ld, sf);
		}
		{	// (b)^H second half + (a)^H FFT: truncate M -> N with the Nyquist combination and conj phase, backward FFT_N
			// (M <= N: the transpose of the low pass, zero padding of |k| < M/2)
			FftDims d; d.n_i = nl; d.is_i = M; d.os_i = p->N; d.is_e = 1; d.os_e = 1;
			FftLoad ld; ld.ptr = p->b2.p; ld.mode = LD_SPEC_ADJ; ld.ne = M; ld.nyq_half = M > p->N ? 1 : 0; ld.kmax = M > p->N ? -1 : M/2 - 1; ld.mul = p->ph_shift.as<double2>();
			FftStore sf; sf.ptr = p->b1.p;
			p->fc->exec(st, p->N, false, d, ld, sf);
		}
		{	// (a)^H: fold the mirror images back onto the rings
			const long tot = nl*nr;
			hipLaunchKernelGGL(fold_mirror, dim3((unsigned)((tot+255)/256)), dim3(256), 0, st, (const double2*)p->b1.p,
				leg_out + ((size_t)c*nm + m0)*nr, nr, p->N, p->mir_c, nl, (spin + (int)m0) & 1);
		}
	}
	p->prof.end(st, PXS_STAGE_RESAMPLE);
	PXS_HIP(hipGetLastError());
}

// band-limited leg on the CC grid [c][m][ncc] -> leg on the map's rings [c][m][nring] (exact for degree <= lmax);
// columns (m, m+1) packed as in resample_to_cc
void resample_from_cc(pxs_plan* p, hipStream_t st, const double2* leg_cc, double2* leg_out, int nc, int spin, double2* h_out = nullptr) {
	const int nm = p->mmax+1, nr = p->nring;
	const long npair_all = (nm + 1)/2;
	const long chunk = std::max<long>(32, std::min<long>(npair_all, (long)(p->resample_chunk_bytes/(sizeof(double2)*p->N))));
	p->b1.ensure(sizeof(double2)*(size_t)chunk*std::max(p->N, p->Ncc));
	p->b2.ensure(sizeof(double2)*(size_t)chunk*p->N);
	p->prof.begin(st, PXS_STAGE_RESAMPLE);
	for (int c = 0; c < nc; c++)
	for (long p0 = 0; p0 < npair_all; p0 += chunk) {
		const long np = std::min<long>(chunk, npair_all - p0);
		const long m0 = 2*p0, nlines = std::min<long>(2*np, nm - m0);
		{	// packed mirror extension of the CC rings to the full circle, forward FFT_Ncc
			FftDims d; d.n_i = np; d.is_i = p->ncc; d.os_i = p->Ncc; d.is_e = 1; d.os_e = 1;
			FftLoad ld; ld.ptr = leg_cc + ((size_t)c*nm + m0)*p->ncc; ld.mode = LD_MIRROR_PAIR; ld.ne = p->ncc; ld.mir_c = 0;
			ld.par0 = (spin + (int)m0) & 1; ld.pair_lines = nlines;
			FftStore sf; sf.ptr = p->b1.p;
			p->fc->exec(st, p->Ncc, true, d, ld, sf);
		}
		{	// keep |k| <= lmax, shift to the target grid's theta0, backward FFT_N onto the full circle
			FftDims d; d.n_i = np; d.is_i = p->Ncc; d.os_i = p->N; d.is_e = 1; d.os_e = 1;
			FftLoad ld; ld.ptr = p->b1.p; ld.mode = LD_SPEC; ld.ne = p->Ncc; ld.kmax = p->lmax; ld.mul = p->ph_up.as<double2>();
			FftStore sf; sf.ptr = p->b2.p;
			p->fc->exec(st, p->N, false, d, ld, sf);
		}
		if (h_out && np == npair_all) {	// separate the pair straight into the ring-major layout of the ring FFT (times e^{+i m phi0})
			dim3 grid((nm+31)/32, (nr+31)/32);
			hipLaunchKernelGGL(split_pair_transposed, grid, dim3(256), sizeof(double2)*32*33, st, (const double2*)p->b2.p,
				h_out + (size_t)c*nr*nm, nr, nm, p->N, p->mir_c, spin & 1, (const double2*)p->phase.p, 1.0/(double)p->Ncc);
		} else {	// separate the pair, keep the real rings
			const long tot = np*nr;
			hipLaunchKernelGGL(split_pair, dim3((unsigned)((tot+255)/256)), dim3(256), 0, st, (const double2*)p->b2.p,
				leg_out + ((size_t)c*nm + m0)*nr, nr, p->N, p->mir_c, np, nlines, (spin + (int)m0) & 1, (const double2*)nullptr, 1.0/(double)p->Ncc);
		}
	}
	p->prof.end(st, PXS_STAGE_RESAMPLE);
	PXS_HIP(hipGetLastError());
}

} // namespace

#define PXS_TRY try {
#define PXS_CATCH } catch (const pxs::Error& e) { pxs::set_last_error(e.what()); return e.code; } \
	catch (const std::exception& e) { pxs::set_last_error(e.what()); return pxs::PXS_ERR_ARG; } return 0;

extern "C" {

int pxs_grid_maxlmax(const char* geometry, int ntheta) { return grid_maxlmax(geometry ? geometry : "", ntheta); }

int pxs_gridweights(const char* geometry, int ntheta, double* out) {
	PXS_TRY
	PXS_REQUIRE(geometry && out && ntheta > 0, "pxs_gridweights: bad arguments");
	GridInfo gi = grid_info(geometry, ntheta);
	if (!gi.ok) throw Error(PXS_ERR_UNSUPPORTED, std::string("gridweights: unsupported geometry '") + geometry + "' (CC, F1, MW, MWflip, DH, F2)");
	if (grid_weights_only(geometry)) {
		const bool dh = std::string(geometry) == "DH";
		const std::vector<LDb> w = fejer2_weights(dh ? ntheta-1 : ntheta);
		if (dh) out[0] = 0;
		for (size_t j = 0; j < w.size(); j++) out[j + (dh ? 1 : 0)] = (double)(w[j]*2*PIl);
		return 0;
	}
	const long N = gi.N; const int n = ntheta;
	const LDb th0 = (LDb)gi.c*PIl/N;
	std::vector<LDb> v(N);
	const long K = (N-1)/2;
	if (N <= 4096) {
		for (long j = 0; j < N; j++) {
			LDb th = th0 + 2*PIl*j/N, s = (LDb)abs_sin_coef(0);
			for (long k = 2; k <= K; k += 2) s += 2*(-(2/PIl)/((LDb)k*k-1))*cosl(k*th);
			if (N % 2 == 0 && ((N/2) % 2) == 0) s += (-(2/PIl)/((LDb)(N/2)*(N/2)-1))*cosl((N/2)*(th-th0))*cosl((N/2)*th0);
			v[j] = s*PIl/N;
		}
	} else {
		// the same cosine series as one backward DFT of length N on the device (the direct sum is N^2/4 long-double cosines:
		// 60 s for the 21600-ring grid, paid by every map2alm of a declination band through quad_weights)
		std::vector<double2> c(N, make_double2(0, 0));
		c[0].x = (double)abs_sin_coef(0);
		for (long k = 2; k <= K; k += 2) {
			const LDb sk = -(2/PIl)/((LDb)k*k-1), a = (LDb)k*th0;
			c[k] = make_double2((double)(sk*cosl(a)), (double)(sk*sinl(a)));
			c[N-k] = make_double2(c[k].x, -c[k].y);
		}
		if (N % 2 == 0 && ((N/2) % 2) == 0) c[N/2] = make_double2((double)((-(2/PIl)/((LDb)(N/2)*(N/2)-1))*cosl((N/2)*th0)), 0.0);
		int dev = 0; PXS_HIP(hipGetDevice(&dev));
		DevBuf dc = upload(c);
		fft_dense_lines(dev, nullptr, N, false, 1, dc.as<double2>(), dc.as<double2>());
		PXS_HIP(hipStreamSynchronize(nullptr));
		PXS_HIP(hipMemcpy(c.data(), dc.p, sizeof(double2)*N, hipMemcpyDeviceToHost));
		for (long j = 0; j < N; j++) v[j] = (LDb)c[j].x*PIl/N;
	}
	for (int j = 0; j < n; j++) out[j] = 0;
	for (long jp = 0; jp < N; jp++) { long r = jp < n ? jp : ((-jp - gi.c) % N + N) % N; out[r] += (double)(v[jp]*2*PIl); }
	PXS_CATCH
}

int pxs_plan_grid2d(pxs_plan** plan, const char* geometry, int ntheta, int nphi, double phi0,
                    int flip_y, int flip_x, int lmax, int mmax, const uint64_t* mstart, int64_t lstride, int device)
{
	PXS_TRY
	PXS_REQUIRE(plan && geometry && ntheta > 0 && nphi > 0, "pxs_plan_grid2d: bad arguments");
	GridInfo gi = grid_info(geometry, ntheta);
	if (!gi.ok) throw Error(PXS_ERR_UNSUPPORTED, std::string("unsupported 2d geometry '") + geometry + "' (supported: CC, F1, MW, MWflip, DH, F2)");
	std::unique_ptr<pxs_plan> p(new pxs_plan());
	p->is_grid = true; p->geometry = geometry; p->nring = ntheta; p->nphi = nphi; p->phi0 = phi0;
	p->ring_stride = flip_y ? -(long)nphi : (long)nphi;
	p->pix_stride = flip_x ? -1 : 1;
	p->ring_off0 = (flip_y ? (long)(ntheta-1)*nphi : 0) + (flip_x ? (long)nphi-1 : 0);
	plan_common(p.get(), lmax, mmax, mstart, lstride, device);
	p->rs_map.build(grid_theta(geometry, ntheta)); p->rs_map.upload_all();
	if (grid_weights_only(geometry)) {	// quadrature weights per pixel for the analysis (pxs_analysis)
		std::vector<double> w(ntheta);
		if (pxs_gridweights(geometry, ntheta, w.data()) != 0) throw Error(PXS_ERR_ARG, get_last_error());
		std::vector<double2> wd(ntheta);
		for (int j = 0; j < ntheta; j++) wd[j] = make_double2(w[j]/nphi, 0.0);
		p->wring = upload(wd);
	} else if (lmax <= grid_maxlmax(geometry, ntheta)) setup_resampling(p.get());
	*plan = p.release();
	PXS_CATCH
}

int pxs_plan_rings(pxs_plan** plan, int nring, const double* theta, const uint64_t* nphi,
                   const double* phi0, const uint64_t* ringstart, int64_t pixstride,
                   int lmax, int mmax, const uint64_t* mstart, int64_t lstride, int device)
{
	PXS_TRY
	PXS_REQUIRE(plan && nring > 0 && theta && nphi && phi0 && ringstart, "pxs_plan_rings: bad arguments");
	std::unique_ptr<pxs_plan> p(new pxs_plan());
	p->is_grid = false; p->nring = nring; p->nphi = (int)nphi[0]; p->phi0 = phi0[0];
	p->ring_off0 = (long)ringstart[0];
	p->ring_stride = nring > 1 ? (long)ringstart[1] - (long)ringstart[0] : (long)nphi[0];
	// equal rings at equal spacing (every CAR map) take the fused / paired ring FFTs; anything else (healpix, profile rings,
	// masked ring subsets) the general path: per-ring length, phase and offset
	for (int r = 0; r < nring; r++)
		if ((long)nphi[r] != p->nphi || std::fabs(phi0[r] - phi0[0]) > 1e-13 || (long)ringstart[r] != p->ring_off0 + r*p->ring_stride) p->general = true;
	if (!FftContext::supported(p->nphi)) p->general = true;          // equal rings of a length with a prime factor > 2048: Bluestein lives on the general path
	{ const char* e = getenv("PXS_GENERAL_RINGS"); if (e && atoi(e) != 0) p->general = true; }      // (tests: the general path on uniform rings)
	p->pix_stride = pixstride;
	plan_common(p.get(), lmax, mmax, mstart, lstride, device);
	if (p->general) setup_general(p.get(), nring, nphi, phi0, ringstart);
	std::vector<LDb> th(nring);
	for (int r = 0; r < nring; r++) th[r] = theta[r];
	p->rs_map.build(th); p->rs_map.upload_all();
	// Rows of a Fejer-1 grid (a declination band of a CAR map, curvedsky.py:843-873): theta_r = (row0 + r + 1/2) pi / n.  Synthesis
	// and its adjoint can then run their Legendre stage on the ~lmax+2 CC rings of that grid instead of the band's own ring-pair
	// slots (an unpaired ring costs a whole slot); PXS_BAND_VIA_CC=0 turns it off.
	{	static const bool on = [] { const char* e = lab_getenv("PXS_BAND_VIA_CC"); return e ? atoi(e) != 0 : true; }();
		if (on && !p->general && p->chain_rings && nring >= 2) {
			const LDb d = th[1] - th[0];
			const long n = d > 0 ? (long)llroundl(PIl/d) : 0;
			const long k0 = n > 0 ? (long)llroundl(th[0]*n/PIl - 0.5L) : -1;
			bool ok = n > nring && k0 >= 0 && k0 + nring <= n && lmax <= n - 1 && n < (1L << 30) && FftContext::supported(2*n);
			for (int r = 0; ok && r < nring; r++) ok = fabsl(th[r] - ((LDb)(k0 + r) + 0.5L)*PIl/n) < 1e-11L;
			if (ok) {
				p->band = true; p->geometry = "F1"; p->nfull = (int)n; p->row0 = (int)k0;
				setup_resampling(p.get());
				if (!p->chain_theta()) { p->band = false; p->nfull = 0; p->row0 = 0; p->ncc = 0; }     // (no fused theta chain for this size: stay on the band's own rings)
			}
		}
	}
	*plan = p.release();
	PXS_CATCH
}

void pxs_plan_destroy(pxs_plan* plan) { delete plan; }

int pxs_plan_option(pxs_plan* p, const char* name, int64_t value) {
	PXS_TRY
	PXS_REQUIRE(p && name, "pxs_plan_option: null argument");
	std::lock_guard<std::mutex> plan_lock(p->call_mu);
	if (std::string(name) == "analysis") {
		PXS_REQUIRE(value >= 0 && value <= 2, "pxs_plan_option: analysis takes 0 (interpolant), 1 (weights) or 2 (ducc0)");
		p->ana_weights = (int)value;
	} else if (std::string(name) == "deterministic") {      // analysis sums in a fixed order: bitwise repeatable results (default 0: atomic adds, repeatable to ~1e-14)
		PXS_REQUIRE(value == 0 || value == 1, "pxs_plan_option: deterministic takes 0 or 1");
		p->wk.deterministic = value != 0;
	} else if (std::string(name) == "build_tables") {      // build the recurrence tables of spin `value` now rather than in the first transform (cold-start accounting of bench.py)
		PXS_REQUIRE(value >= 0 && value <= p->lmax + 1, "pxs_plan_option: build_tables takes a spin");
		PXS_HIP(hipSetDevice(p->device));
		(void)p->table((int)value);
	} else throw Error(PXS_ERR_ARG, std::string("pxs_plan_option: unknown option '") + name + "'");
	PXS_CATCH
}

static int ana_form_now(const pxs_plan* p);
static int theta_line_now(const pxs_plan* p);
int pxs_plan_query(const pxs_plan* p, const char* name, int64_t* value) {
	PXS_TRY
	PXS_REQUIRE(p && name && value, "pxs_plan_query: null argument");
	const std::string n(name);
	if (n == "analysis_form") *value = ana_form_now(p);
	else if (n == "ncc_circle") *value = p->ncc > 0 ? p->Ncc : 0;
	else if (n == "ducc_ncc_circle") *value = FftChain::ducc_ncc(p->lmax);
	else if (n == "theta_line") *value = theta_line_now(p);
	else throw Error(PXS_ERR_ARG, std::string("pxs_plan_query: unknown name '") + name + "'");
	PXS_CATCH
}

int pxs_plan_info(const pxs_plan* p, int* nsyn, int* nana, int64_t* scratch) {
	if (!p) return PXS_ERR_ARG;
	if (nsyn) *nsyn = (p->syn_via_cc && p->ncc > 0) ? p->ncc : p->nring;
	if (nana) *nana = p->ncc > 0 ? p->ncc : p->nring;
	if (scratch) *scratch = (int64_t)(p->leg.bytes + p->leg2.bytes + p->hbuf.bytes + p->b1.bytes + p->b2.bytes + p->wk.almt.bytes + p->wk.part.bytes + p->wk.mom.bytes);
	return 0;
}

/* planner of the fused theta chains (diagnostics / tests): out = {ok, g, bN, g2, M, ac, Ncc, gs, bs, aNs} */
int pxs_debug_theta_plan(int64_t N, int lmax, int64_t* out) {
	const ThetaPlan t = FftChain::plan_theta(N, lmax);
	const int64_t v[10] = {t.ok, t.g, t.bN, t.g2, t.M, t.ac, t.Ncc, t.gs, t.bs, t.aNs};
	for (int i = 0; i < 10; i++) out[i] = v[i];
	return 0;
}

/* diagnostics (tools/chain_lab.py): average ms of one fused-chain stage group of the plan on scratch data.
 * kind 0: map2leg (MA1, MA2), 1: h2map (MS1, MS2), 2: to_cc (RA1-5), 3: from_cc (RS1-3), 4: from_cc_adjoint */
int pxs_debug_chain(pxs_plan* p, int kind, int nc, int spin, int reps, double* ms) {
	PXS_TRY
	PXS_REQUIRE(p && ms && nc >= 1 && reps >= 1 && kind >= 0 && kind <= 4, "pxs_debug_chain: bad arguments");
	PXS_REQUIRE(p->chain_rings && (kind < 2 || p->chain_theta()), "pxs_debug_chain: the plan has no fused chain for this stage");
	PXS_HIP(hipSetDevice(p->device));
	const size_t nm = (size_t)p->mmax + 1, nr = (size_t)p->nring, c16 = sizeof(double2);
	const long ldm = FftChain::pad8(p->nring), ldc = p->ld_cc(), ldh = p->ld_h();
	DevBuf dmap;
	if (kind < 2) { dmap.alloc(sizeof(double)*(size_t)nc*nr*p->nphi); PXS_HIP(hipMemset(dmap.p, 0, dmap.bytes)); }
	p->leg.ensure(c16*nc*nm*ldm); p->leg2.ensure(c16*nc*nm*std::max<long>(ldc, 8)); p->hbuf.ensure(c16*nc*nr*ldh);
	PXS_HIP(hipMemset(p->leg.p, 0, p->leg.bytes)); PXS_HIP(hipMemset(p->leg2.p, 0, p->leg2.bytes)); PXS_HIP(hipMemset(p->hbuf.p, 0, p->hbuf.bytes));
	hipStream_t st = nullptr;
	auto run = [&]() {
		const FftChain::MapDesc md = p->map_desc(dmap.p, PX_F64, (long)nr*p->nphi);
		switch (kind) {
		case 0: p->chain->map2leg(st, md, nc, p->mmax, p->leg.as<double2>(), ldm, p->phase.as<double2>(), 1.0); break;
		case 1: p->chain->h2map(st, p->hbuf.as<double2>(), ldh, md, nc, p->mmax); break;
		case 2: { const bool f = p->ana_weights != 0 && p->Mf > 0 && p->tpf.ok;      // (the form the plan's analysis option selects)
			p->chain->to_cc(st, f ? p->tpf : p->tp, p->leg.as<double2>(), ldm, p->nring, p->mir_c, p->leg2.as<double2>(), ldc, p->ncc, nc, (int)nm, spin, p->lmax,
				p->ph_shift.as<double2>(), (f ? p->sigma_f : p->sigma).as<double2>(), (f ? p->wcc_f : p->wcc).as<double2>()); } break;
		case 3: p->chain->from_cc(st, p->tp, p->leg2.as<double2>(), ldc, p->ncc, p->hbuf.as<double2>(), ldh, p->nring, p->mir_c, nc, (int)nm, spin, p->lmax,
				p->ph_up.as<double2>(), p->phase.as<double2>(), 1.0/(double)p->Ncc); break;
		default: p->chain->from_cc_adjoint(st, p->tp, p->leg.as<double2>(), ldm, p->nring, p->mir_c, p->leg2.as<double2>(), ldc, p->ncc, nc, (int)nm, spin, p->lmax,
				p->ph_shift.as<double2>(), p->wadj.as<double2>()); break;
		}
	};
	run(); PXS_HIP(hipStreamSynchronize(st));
	hipEvent_t e0, e1; PXS_HIP(hipEventCreate(&e0)); PXS_HIP(hipEventCreate(&e1));
	PXS_HIP(hipEventRecord(e0, st));
	for (int i = 0; i < reps; i++) run();
	PXS_HIP(hipEventRecord(e1, st)); PXS_HIP(hipEventSynchronize(e1));
	float t = 0; PXS_HIP(hipEventElapsedTime(&t, e0, e1));
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	*ms = (double)t/reps;
	PXS_CATCH
}

int pxs_profile(pxs_plan* p, int enable) {
	PXS_TRY
	PXS_REQUIRE(p, "pxs_profile: null plan");
	p->prof.enabled = enable != 0;
	if (enable && !p->wk.count.p) { PXS_HIP(hipSetDevice(p->device)); p->wk.count.alloc(2*1024*sizeof(double)); PXS_HIP(hipMemset(p->wk.count.p, 0, p->wk.count.bytes)); }
	p->wk.count_on = enable != 0;
	PXS_CATCH
}

int pxs_profile_flops(pxs_plan* p, double* flops, int reset) {
	PXS_TRY
	PXS_REQUIRE(p && flops, "pxs_profile_flops: null argument");
	flops[0] = flops[1] = 0;
	if (!p->wk.count.p) return 0;
	PXS_HIP(hipSetDevice(p->device));
	PXS_HIP(hipDeviceSynchronize());
	std::vector<double> c(p->wk.count.bytes/sizeof(double)); PXS_HIP(hipMemcpy(c.data(), p->wk.count.p, p->wk.count.bytes, hipMemcpyDeviceToHost));
	for (size_t i = 0; i + 1 < c.size(); i += 2) { flops[0] += c[i]*128.0; flops[1] += c[i+1]*128.0; }     // FMA instructions per lane x 64 lanes x 2 flops
	if (reset) PXS_HIP(hipMemset(p->wk.count.p, 0, p->wk.count.bytes));
	PXS_CATCH
}

int pxs_profile_read(pxs_plan* p, double* ms, int* counts, int reset) {
	PXS_TRY
	PXS_REQUIRE(p && ms && counts, "pxs_profile_read: null argument");
	p->prof.read(ms, counts, PXS_NSTAGE, reset != 0);
	PXS_CATCH
}

// How a pxs_analysis call runs.  Decided HERE, once, for reserve_call, the batch chunking and analysis_core.
enum AnaPath {
	ANA_RING_WEIGHTS,   // quadrature weights on the map's rings + (adjoint) synthesis there: DH / F2 grids; the weights option on grids without the CC detour
	ANA_CC_WEIGHTS,     // the weights option on F1 grids with fused chains: ring weights, transposed theta upsampling, Legendre stage on the CC grid
	ANA_CHAIN,          // exact quadrature of the theta-interpolant through the fused chains (to_cc / to_cc_adjoint)
	ANA_UNFUSED };      // ... through the generic FFT engine on dense rows, one map at a time
static bool adj_ana_fused() { const char* e = getenv("PXS_ADJ_ANA_FUSED"); return e ? atoi(e) != 0 : true; }      // (read per call: the tests switch it)
static AnaPath ana_path(const pxs_plan* p, int adjoint) {
	if (p->wring.p) return ANA_RING_WEIGHTS;
	// ring weights on the map's own rings: the weights option, and ducc0's route on CC grids with >= 2 lmax + 2 rings (its
	// resample_to_prepared_CC multiplies such a grid by its weights directly: need_first_resample is false for it)
	if ((p->ana_weights == 1 || (p->ana_weights == 2 && p->geometry == "CC")) && (long)p->nring >= 2L*p->lmax + 2)
		return (p->chain_theta() && p->ncc > 0) ? ANA_CC_WEIGHTS : ANA_RING_WEIGHTS;
	const bool fine = p->ana_weights != 0 && p->Mf > 0;        // the fine-CC form (default; the weights option on a grid below 2 lmax + 2 rings takes it too)
	if (p->chain_theta() && (fine ? p->tpf.ok : (!adjoint || adj_ana_fused()))) return ANA_CHAIN;
	return ANA_UNFUSED;      // (also: the fine-CC form on a plan whose chains cannot hold the circle of 2 N_cc points)
}
// tables of the interpolating forms of the analysis (fused chains or the unfused engine): the fine-CC form, else the interpolant form
struct AnaSet { const ThetaPlan* tp; const double2* sigma; const double2* wcc; const double2* whalf; long M; bool fine; };
static AnaSet ana_set(const pxs_plan* p) {
	if (p->ana_weights != 0 && p->Mf > 0) return AnaSet{&p->tpf, p->sigma_f.as<double2>(), p->wcc_f.as<double2>(), p->whalf_f.as<double2>(), p->Mf, true};
	return AnaSet{&p->tp, p->sigma.as<double2>(), p->wcc.as<double2>(), p->whalf.as<double2>(), p->M, false};
}
// 1: the theta resampling of pxs_analysis (the plan's current form) runs as ONE kernel per call (thetaline.hip), 0: as the stage chain
static int theta_line_now(const pxs_plan* p) {
	if (!p->chain_theta()) return 0;
	const AnaPath path = ana_path(p, 0);
	if (path == ANA_CHAIN) return FftChain::line_takes(*ana_set(p).tp, true) ? 1 : 0;
	if (path == ANA_CC_WEIGHTS) return FftChain::line_takes(p->tp, false) ? 1 : 0;
	return 0;
}
static int ana_form_now(const pxs_plan* p) {
	const AnaPath path = ana_path(p, 0);
	if (path == ANA_RING_WEIGHTS || path == ANA_CC_WEIGHTS) return 1;
	return ana_set(p).fine ? 2 : 0;
}
// per-ring weight / nphi of the ring-weights paths (DH / F2: fixed at plan time; the weights option: built on first use)
static const double2* ring_weights(pxs_plan* p) {
	if (p->wring.p) return p->wring.as<double2>();
	if (!p->wgrid.p) {
		std::vector<double> w(p->nring);
		if (pxs_gridweights(p->geometry.c_str(), p->nring, w.data()) != 0) throw Error(PXS_ERR_ARG, get_last_error());
		std::vector<double2> wd(p->nring);
		for (int j = 0; j < p->nring; j++) wd[j] = make_double2(w[j]/p->nphi, 0.0);
		p->wgrid = upload(wd);
	}
	return p->wgrid.as<double2>();
}
// ... for the transposed theta upsampling (FftChain::from_cc_adjoint): its first stage extends the weighted ring samples to the circle
// by reflection, which holds a self-mirrored ring (the pole rings of CC, one of MW / MWflip) once and every other ring twice -- the
// zero extension it stands for is half the sum of the symmetric and the antisymmetric one, so the self-mirrored rings count double
static const double2* ring_weights_ext(pxs_plan* p) {
	const double2* w = ring_weights(p);
	if (p->geometry == "F1") return w;
	if (!p->wgrid_ext.p) {
		std::vector<double2> wd(p->nring);
		PXS_HIP(hipMemcpy(wd.data(), w, sizeof(double2)*p->nring, hipMemcpyDeviceToHost));
		for (int r = 0; r < p->nring; r++) { const long t = 2L*r + p->mir_c; if (t == 0 || t == p->N || t == 2*p->N) wd[r].x *= 2; }
		p->wgrid_ext = upload(wd);
	}
	return p->wgrid_ext.as<double2>();
}

// unit weights for the transposed theta upsampling on grids with self-mirrored rings (see ring_weights_ext); null on F1 grids
static const double2* unit_weights_ext(pxs_plan* p) {
	if (p->geometry == "F1" || p->band) return nullptr;
	if (!p->wunit_ext.p) {
		std::vector<double2> wd(p->nring, make_double2(1, 0));
		for (int r = 0; r < p->nring; r++) { const long t = 2L*r + p->mir_c; if (t == 0 || t == p->N || t == 2*p->N) wd[r].x = 2; }
		p->wunit_ext = upload(wd);
	}
	return p->wunit_ext.as<double2>();
}

// nb maps of one call (nb > 1 only on the fused-chain paths): ring FFTs of all maps in one launch each, theta chains and
// Legendre kernels map by map on the plan's scratch
static void synthesis_core(pxs_plan* p, int spin, int mode, int adjoint, int nb, void* alm, int alm_dtype, long alm_cstride, long alm_bstride,
                           void* map, int map_dtype, long map_cstride, long map_bstride, hipStream_t st)
{
	const int ncm = ncomp_of(spin, mode, false), nca = ncomp_of(spin, mode, true), nct = nb*ncm;
	const int nm = p->mmax+1, nr = p->nring;
	LegTables& tb = p->table(spin);
	const bool th = p->chain_theta();
	const long ldm = p->chain_rings ? FftChain::pad8(nr) : nr;
	const int ncb = nb > 1 ? ncm : 0;
	(void)nca;
	p->leg.ensure(sizeof(double2)*(size_t)nct*nm*ldm);
	if (!adjoint) {
		if ((p->is_grid || (p->band && th)) && (spin == 0 ? p->syn_via_cc0 : p->syn_via_cc) && p->ncc > 0) {
			const long ldc = p->ld_cc();
			p->leg2.ensure(sizeof(double2)*(size_t)nct*nm*ldc);
			leg_synthesis(st, p->rs_cc, tb, p->wk, alm, alm_dtype, alm_cstride, p->d_mstart.as<uint64_t>(), p->lstride,
				p->leg2.as<double2>(), mode == PXS_MODE_DERIV1, &p->prof, ldc, nb, alm_bstride, (long)ncm*nm*ldc);
			if (th) {	// fused chain: CC grid -> ring spectra of the map's rings, written ring-major for the ring FFT
				const long ldh = p->ld_h();
				const int nrh = p->band ? p->nfull : nr;          // (a band: h for every ring of the grid, the ring FFTs take its rows)
				p->hbuf.ensure(sizeof(double2)*(size_t)nct*nrh*ldh);
				p->prof.begin(st, PXS_STAGE_RESAMPLE);
				p->chain->from_cc(st, p->tp, p->leg2.as<double2>(), ldc, p->ncc, p->hbuf.as<double2>(), ldh, nrh, p->mir_c, nct, nm, spin, p->lmax,
					p->ph_up.as<double2>(), p->phase.as<double2>(), 1.0/(double)p->Ncc);
				p->prof.end(st, PXS_STAGE_RESAMPLE);
				leg2map(p, st, nullptr, nr, map, map_dtype, map_cstride, nct, true, map_bstride, ncb);
			} else {
				PXS_REQUIRE(nb == 1, "internal: batched call on an unfused path");
				static const bool fuse = [] { const char* e = lab_getenv("PXS_FUSE_SPLIT"); return e ? atoi(e) != 0 : true; }();
				const bool via_h = fuse && !p->chain_rings;     // (the unfused transposing split writes dense h rows)
				if (via_h) p->hbuf.ensure(sizeof(double2)*(size_t)ncm*nr*nm);
				resample_from_cc(p, st, p->leg2.as<double2>(), p->leg.as<double2>(), ncm, spin, via_h ? p->hbuf.as<double2>() : nullptr);
				leg2map(p, st, p->leg.as<double2>(), nr, map, map_dtype, map_cstride, ncm, via_h);
			}
		} else {
			leg_synthesis(st, p->rs_map, tb, p->wk, alm, alm_dtype, alm_cstride, p->d_mstart.as<uint64_t>(), p->lstride,
				p->leg.as<double2>(), mode == PXS_MODE_DERIV1, &p->prof, ldm, nb, alm_bstride, (long)ncm*nm*ldm);
			leg2map(p, st, p->leg.as<double2>(), ldm, map, map_dtype, map_cstride, nct, false, map_bstride, ncb);
		}
	} else {
		// the transpose of the synthesis through the CC grid, where that is the cheaper synthesis: grids (self-mirrored rings of
		// CC / MW / MWflip count double in the reflection that stands for the zero extension, unit_weights_ext) and bands of F1 grids
		// (rows outside the band are zero)
		static const bool adj_cc = [] { const char* e = lab_getenv("PXS_ADJ_VIA_CC"); return e ? atoi(e) != 0 : true; }();
		const bool via = adj_cc && (p->is_grid || p->band) && th && (p->is_grid || p->geometry == "F1") && (spin == 0 ? p->syn_via_cc0 : p->syn_via_cc) && p->ncc > 0;
		if (via && p->band) {
			const long ldf = FftChain::pad8(p->nfull);
			p->leg.ensure(sizeof(double2)*(size_t)nct*nm*ldf);
			PXS_HIP(hipMemsetAsync(p->leg.p, 0, sizeof(double2)*(size_t)nct*nm*ldf, st));
			map2leg(p, st, map, map_dtype, map_cstride, nct, p->leg.as<double2>() + p->row0, 1.0, ldf, map_bstride, ncb);
		} else map2leg(p, st, map, map_dtype, map_cstride, nct, p->leg.as<double2>(), 1.0, ldm, map_bstride, ncb);
		if (via) {
			const long ldc = p->ld_cc(), ldin = p->band ? FftChain::pad8(p->nfull) : ldm;
			p->leg2.ensure(sizeof(double2)*(size_t)nct*nm*ldc);
			p->prof.begin(st, PXS_STAGE_RESAMPLE);
			p->chain->from_cc_adjoint(st, p->tp, p->leg.as<double2>(), ldin, p->band ? p->nfull : nr, p->mir_c, p->leg2.as<double2>(), ldc, p->ncc, nct, nm, spin, p->lmax,
				p->ph_shift.as<double2>(), p->wadj.as<double2>(), unit_weights_ext(p));
			p->prof.end(st, PXS_STAGE_RESAMPLE);
			leg_analysis(st, p->rs_cc, tb, p->wk, p->leg2.as<double2>(), alm, alm_dtype, alm_cstride, p->d_mstart.as<uint64_t>(), p->lstride,
				mode == PXS_MODE_DERIV1, &p->prof, ldc, nb, alm_bstride, (long)ncm*nm*ldc);
			return;
		}
		leg_analysis(st, p->rs_map, tb, p->wk, p->leg.as<double2>(), alm, alm_dtype, alm_cstride, p->d_mstart.as<uint64_t>(), p->lstride,
			mode == PXS_MODE_DERIV1, &p->prof, ldm, nb, alm_bstride, (long)ncm*nm*ldm);
	}
}

// Scratch of a call is sized HERE, before its first launch (the fused-chain paths every BASELINE configuration takes; a plan's
// buffers only grow, so this allocates on the first call of a kind and when a later call brings a larger batch).  The `ensure`
// calls further down are then no-ops; they remain as the allocation points of the rarely taken paths (general ring sets,
// unfused FFTs, the deterministic analysis), where a growing buffer is freed between launches through hipFree's device sync.
static void reserve_call(pxs_plan* p, int spin, int mode, bool synthesis, bool adjoint, int nb) {
	if (p->general || !p->chain_rings) return;
	const int ncm = synthesis ? ncomp_of(spin, mode, false) : (spin == 0 ? 1 : 2), nct = nb*ncm;
	const size_t nm = (size_t)p->mmax + 1, nr = (size_t)p->nring, c16 = sizeof(double2);
	const bool th = p->chain_theta();
	const size_t ldm = (size_t)FftChain::pad8(p->nring), ldc = (size_t)p->ld_cc(), ldh = (size_t)p->ld_h();
	const bool via_cc = (p->is_grid || (p->band && th)) && (spin == 0 ? p->syn_via_cc0 : p->syn_via_cc) && p->ncc > 0;
	LegTables& tb = p->table(spin);
	size_t c1 = 0, c2 = 0, r1 = 0;
	auto theta = [&](int kind) {
		if (!th) return;
		const ThetaPlan& tpk = (kind == 0 || kind == 3) ? *ana_set(p).tp : p->tp;
		if (kind <= 1 && FftChain::line_takes(tpk, kind == 0)) return;      // the single-kernel engine (thetaline.hip) has no intermediates
		FftChain::theta_scratch(tpk, (int)nm, nct, kind, c1, c2); };
	if (synthesis && !adjoint) {                        // alm -> map
		p->wk.almt.ensure(sizeof(double)*4*(tb.nrows + 4)*nb);
		p->leg.ensure(c16*nct*nm*ldm);
		if (via_cc) { p->leg2.ensure(c16*nct*nm*ldc); if (th) { p->hbuf.ensure(c16*nct*(p->band ? (size_t)p->nfull : nr)*ldh); theta(2); } }
		else p->hbuf.ensure(c16*nct*nr*ldh);
		p->chain->ring_scratch(p->nring, nct, false, r1, p->mmax);
	} else if (synthesis) {                             // map -> alm, transpose of the synthesis
		p->wk.mom.ensure(sizeof(double)*4*std::max<long>(tb.nrows, 1)*nb);
		const bool via = (p->is_grid || p->band) && th && (p->is_grid || p->geometry == "F1") && via_cc;
		p->leg.ensure(c16*nct*nm*(via && p->band ? (size_t)FftChain::pad8(p->nfull) : ldm));
		if (via) { p->leg2.ensure(c16*nct*nm*ldc); theta(1); }
		p->chain->ring_scratch(p->nring, nct, true, r1);
	} else {                                            // pxs_analysis: the same path decision as analysis_core
		const AnaPath path = ana_path(p, adjoint);
		if (path == ANA_UNFUSED) return;
		if (adjoint) p->wk.almt.ensure(sizeof(double)*4*(tb.nrows + 4)*nb); else p->wk.mom.ensure(sizeof(double)*4*std::max<long>(tb.nrows, 1)*nb);
		if (path == ANA_RING_WEIGHTS) { p->leg.ensure(c16*nct*nm*ldm); if (adjoint) p->hbuf.ensure(c16*nct*nr*ldh); }
		else if (path == ANA_CC_WEIGHTS) { p->leg2.ensure(c16*nct*nm*ldc); if (adjoint) { p->hbuf.ensure(c16*nct*nr*ldh); theta(2); } else { p->leg.ensure(c16*nct*nm*ldm); theta(1); } }
		else if (!adjoint) { p->leg.ensure(c16*nct*nm*ldm); p->leg2.ensure(c16*nct*nm*ldc); theta(0); }      // analysis_2d
		else { p->leg2.ensure(c16*nct*nm*ldc); p->hbuf.ensure(c16*nct*nr*ldh); theta(3); }                   // adjoint_analysis_2d (fused transposed chain)
		p->chain->ring_scratch(p->nring, nct, !adjoint, r1, p->mmax);
	}
	p->chain->reserve(std::max(c1, r1), c2);
}

// maps per pass of a batched call: bounded by the scratch the plan may hold (leg + leg_cc + h per map, and the chain scratch)
static int batch_chunk(const pxs_plan* p, int nbatch, int ncm) {
	if (nbatch <= 1 || !p->chain_rings) return 1;           // the unfused paths take one map at a time
	const size_t per_map = sizeof(double2)*(size_t)ncm*((size_t)(p->mmax+1)*(p->nring + (p->ncc > 0 ? p->ncc : 0))*2 + (size_t)p->nring*p->nphi);
	// scratch budget of one pass: PXS_BATCH_GB, default 64 GB (MI355X: 288 GB; 8 maps of config 5 = 56 GB, so that the batched Legendre kernels get
	// their 8-map groups there; it was 32 in rounds 2-4), never more than 40 % of what is free on the device now beyond what the plan holds already
	static const size_t budget_env = [] { const char* e = getenv("PXS_BATCH_GB"); return (size_t)(e ? atol(e) : 0) << 30; }();
	size_t budget = budget_env ? budget_env : (size_t)64 << 30;
	if (!budget_env) {
		size_t fr = 0, tot = 0;
		const size_t held = p->leg.bytes + p->leg2.bytes + p->hbuf.bytes + (p->chain ? p->chain->scratch_bytes() : 0);
		if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr > 0) budget = std::min(budget, std::max<size_t>((size_t)4 << 30, held + (size_t)(0.4*(double)fr)));
	}
	const int cap = (int)std::max<size_t>(1, std::min<size_t>((size_t)nbatch, budget/std::max<size_t>(per_map, 1)));
	if (ncm == 1 && cap >= 8 && nbatch > cap) {      // scalar maps: the batched Legendre analysis works on groups of 8 maps (leg_ana_s0_mm)
		const int cap8 = cap & ~7, npass = (nbatch + cap8 - 1)/cap8;
		return std::min(cap8, (((nbatch + npass - 1)/npass + 7)/8)*8);
	}
	const int npass = (nbatch + cap - 1)/cap;
	return (nbatch + npass - 1)/npass;      // equal passes (64 maps at 15 per pass: 13 x 4 + 12, not 15 x 4 + 4 with a last pass at 27 % of the batch dimension)
}

int pxs_synthesis(pxs_plan* p, int spin, int mode, int adjoint, int nbatch,
                  void* alm, int alm_dtype, int64_t alm_cstride, int64_t alm_bstride,
                  void* map, int map_dtype, int64_t map_cstride, int64_t map_bstride, void* stream)
{
	PXS_TRY
	PXS_REQUIRE(p && alm && map, "pxs_synthesis: null argument");
	std::lock_guard<std::mutex> plan_lock(p->call_mu);
	PXS_REQUIRE(spin >= 0 && spin <= p->lmax + 1, "pxs_synthesis: bad spin");
	PXS_REQUIRE(mode == PXS_MODE_STANDARD || (mode == PXS_MODE_DERIV1 && spin == 1), "DERIV1 needs spin 1");
	PXS_REQUIRE(map_dtype == PX_F32 || map_dtype == PX_F64, "map must be float32 or float64");
	PXS_REQUIRE(nbatch >= 1, "pxs_synthesis: nbatch must be >= 1");
	PXS_HIP(hipSetDevice(p->device));
	hipStream_t st = (hipStream_t)stream;
	const size_t aesz = alm_dtype == PX_C64 ? 8 : 16, mesz = map_dtype == PX_F32 ? 4 : 8;
	// (a grid whose ring FFTs are chained but whose theta resampling is not -- 2 ntheta with a prime factor >= 7 -- takes the
	// CC detour through the unfused resampling, one map at a time)
	const bool cc_unfused = !adjoint && (p->is_grid || p->band) && (spin == 0 ? p->syn_via_cc0 : p->syn_via_cc) && p->ncc > 0 && !p->chain_theta();
	const int chunk = cc_unfused ? 1 : batch_chunk(p, nbatch, ncomp_of(spin, mode, false));
	reserve_call(p, spin, mode, true, adjoint != 0, std::min(chunk, nbatch));
	for (int b0 = 0; b0 < nbatch; b0 += chunk) {
		const int nb = std::min(chunk, nbatch - b0);
		synthesis_core(p, spin, mode, adjoint, nb, (char*)alm + aesz*(size_t)b0*alm_bstride, alm_dtype, alm_cstride, alm_bstride,
			(char*)map + mesz*(size_t)b0*map_bstride, map_dtype, map_cstride, map_bstride, st);
	}
	PXS_CATCH
}

static void analysis_core(pxs_plan* p, int spin, int adjoint, int nb, void* map, int map_dtype, long map_cstride, long map_bstride,
                          void* alm, int alm_dtype, long alm_cstride, long alm_bstride, hipStream_t st)
{
	const int nc = spin == 0 ? 1 : 2, nct = nb*nc;
	const int nm = p->mmax+1, nr = p->nring;
	LegTables& tb = p->table(spin);
	const AnaPath path = ana_path(p, adjoint);
	if (getenv("PXS_CHAIN_VERBOSE")) fprintf(stderr, "[pxsht] analysis path %d (0 ring weights, 1 weights via the CC grid, 2 chain, 3 unfused), option %d, adjoint %d\n", (int)path, p->ana_weights, adjoint);
	if (path == ANA_RING_WEIGHTS) {	// DH / F2 (and the weights option off the CC detour): analysis = adjoint synthesis of the weighted map (its adjoint: synthesis, then the weights)
		const double2* wr = ring_weights(p);
		const long ldw = p->chain_rings ? FftChain::pad8(nr) : nr;
		const int ncbw = nb > 1 ? nc : 0;
		p->leg.ensure(sizeof(double2)*(size_t)nct*nm*ldw);
		const long tot = (long)nct*nm*nr;
		if (!adjoint) {
			map2leg(p, st, map, map_dtype, map_cstride, nct, p->leg.as<double2>(), 1.0, ldw, map_bstride, ncbw);
			hipLaunchKernelGGL(scale_rings, dim3((unsigned)((tot+255)/256)), dim3(256), 0, st, p->leg.as<double2>(), (long)nct*nm, nr, ldw, wr);
			leg_analysis(st, p->rs_map, tb, p->wk, p->leg.as<double2>(), alm, alm_dtype, alm_cstride,
				p->d_mstart.as<uint64_t>(), p->lstride, 0, &p->prof, ldw, nb, alm_bstride, (long)nc*nm*ldw);
		} else {
			leg_synthesis(st, p->rs_map, tb, p->wk, alm, alm_dtype, alm_cstride, p->d_mstart.as<uint64_t>(), p->lstride,
				p->leg.as<double2>(), 0, &p->prof, ldw, nb, alm_bstride, (long)nc*nm*ldw);
			hipLaunchKernelGGL(scale_rings, dim3((unsigned)((tot+255)/256)), dim3(256), 0, st, p->leg.as<double2>(), (long)nct*nm, nr, ldw, wr);
			leg2map(p, st, p->leg.as<double2>(), ldw, map, map_dtype, map_cstride, nct, false, map_bstride, ncbw);
		}
		PXS_HIP(hipGetLastError());
		return;
	}
	if (path == ANA_CC_WEIGHTS) {
		// the weights form on an F1 grid with at least 2 lmax + 2 rings, Legendre stage on the ~lmax + 2 rings of the CC grid: the
		// synthesis there is (theta upsampling) o (Legendre on the CC grid), exactly, so its transpose applied to the weighted
		// ring spectra is the reference's adjoint_synthesis(map * weights) (curvedsky.py:852-861, 1068-1084) -- three chain stages
		// (FftChain::from_cc_adjoint, the ring weights folded into its first one) instead of the five of the interpolant (to_cc)
		const double2* wr = ring_weights(p);
		const long ldm = FftChain::pad8(nr), ldc = p->ld_cc(), ldh = p->ld_h();
		const int ncb = nb > 1 ? nc : 0;
		p->leg2.ensure(sizeof(double2)*(size_t)nct*nm*ldc);
		if (!adjoint) {
			p->leg.ensure(sizeof(double2)*(size_t)nct*nm*ldm);
			map2leg(p, st, map, map_dtype, map_cstride, nct, p->leg.as<double2>(), 1.0, ldm, map_bstride, ncb);
			p->prof.begin(st, PXS_STAGE_RESAMPLE);
			p->chain->from_cc_adjoint(st, p->tp, p->leg.as<double2>(), ldm, nr, p->mir_c, p->leg2.as<double2>(), ldc, p->ncc, nct, nm, spin, p->lmax,
				p->ph_shift.as<double2>(), p->wadj.as<double2>(), ring_weights_ext(p));
			p->prof.end(st, PXS_STAGE_RESAMPLE);
			leg_analysis(st, p->rs_cc, tb, p->wk, p->leg2.as<double2>(), alm, alm_dtype, alm_cstride, p->d_mstart.as<uint64_t>(), p->lstride, 0, &p->prof, ldc, nb, alm_bstride, (long)nc*nm*ldc);
		} else {
			p->hbuf.ensure(sizeof(double2)*(size_t)nct*nr*ldh);
			leg_synthesis(st, p->rs_cc, tb, p->wk, alm, alm_dtype, alm_cstride, p->d_mstart.as<uint64_t>(), p->lstride,
				p->leg2.as<double2>(), 0, &p->prof, ldc, nb, alm_bstride, (long)nc*nm*ldc);
			p->prof.begin(st, PXS_STAGE_RESAMPLE);
			p->chain->from_cc(st, p->tp, p->leg2.as<double2>(), ldc, p->ncc, p->hbuf.as<double2>(), ldh, nr, p->mir_c, nct, nm, spin, p->lmax,
				p->ph_up.as<double2>(), p->phase.as<double2>(), 1.0/(double)p->Ncc, wr);
			p->prof.end(st, PXS_STAGE_RESAMPLE);
			leg2map(p, st, nullptr, nr, map, map_dtype, map_cstride, nct, true, map_bstride, ncb);
		}
		return;
	}
	if (path == ANA_CHAIN && adjoint) {
		// adjoint_analysis_2d through the fused transposed chain: Legendre synthesis on the CC grid, FftChain::to_cc_adjoint straight
		// into the ring-major spectra, ring FFTs -- all maps of the call in every launch
		const long ldc = p->ld_cc(), ldh = p->ld_h();
		p->leg2.ensure(sizeof(double2)*(size_t)nct*nm*ldc); p->hbuf.ensure(sizeof(double2)*(size_t)nct*nr*ldh);
		leg_synthesis(st, p->rs_cc, tb, p->wk, alm, alm_dtype, alm_cstride, p->d_mstart.as<uint64_t>(), p->lstride,
			p->leg2.as<double2>(), 0, &p->prof, ldc, nb, alm_bstride, (long)nc*nm*ldc);
		p->prof.begin(st, PXS_STAGE_RESAMPLE);
		const AnaSet as = ana_set(p);
		p->chain->to_cc_adjoint(st, *as.tp, p->leg2.as<double2>(), ldc, p->ncc, p->hbuf.as<double2>(), ldh, nr, p->mir_c, nct, nm, spin, p->lmax,
			p->ph_shift.as<double2>(), as.sigma, as.whalf, p->phase.as<double2>(), 2.0);
		p->prof.end(st, PXS_STAGE_RESAMPLE);
		leg2map(p, st, nullptr, nr, map, map_dtype, map_cstride, nct, true, map_bstride, nb > 1 ? nc : 0);
		return;
	}
	const bool th = path == ANA_CHAIN;                  // (ANA_UNFUSED: the generic FFT engine on dense rows, one map per call)
	const long ldm = th ? FftChain::pad8(nr) : nr, ldc = th ? p->ld_cc() : p->ncc;
	const int ncb = nb > 1 ? nc : 0;
	p->leg.ensure(sizeof(double2)*(size_t)nct*nm*ldm);
	p->leg2.ensure(sizeof(double2)*(size_t)nct*nm*ldc);
	if (!adjoint) {
		map2leg(p, st, map, map_dtype, map_cstride, nct, p->leg.as<double2>(), 1.0, ldm, map_bstride, ncb);
		if (th) {
			p->prof.begin(st, PXS_STAGE_RESAMPLE);
			const AnaSet as = ana_set(p);
			p->chain->to_cc(st, *as.tp, p->leg.as<double2>(), ldm, nr, p->mir_c, p->leg2.as<double2>(), ldc, p->ncc, nct, nm, spin, p->lmax,
				p->ph_shift.as<double2>(), as.sigma, as.wcc);
			p->prof.end(st, PXS_STAGE_RESAMPLE);
		} else { PXS_REQUIRE(nb == 1, "internal: batched call on an unfused path"); const AnaSet as = ana_set(p); resample_to_cc(p, st, p->leg.as<double2>(), p->leg2.as<double2>(), nc, spin, as.M, as.sigma, as.wcc); }
		leg_analysis(st, p->rs_cc, tb, p->wk, p->leg2.as<double2>(), alm, alm_dtype, alm_cstride, p->d_mstart.as<uint64_t>(), p->lstride, 0, &p->prof, ldc, nb, alm_bstride, (long)nc*nm*ldc);
	} else {
		// adjoint_analysis_2d: the exact transpose, stage by stage in reverse
		PXS_REQUIRE(nb == 1, "internal: batched call on an unfused path");
		leg_synthesis(st, p->rs_cc, tb, p->wk, alm, alm_dtype, alm_cstride, p->d_mstart.as<uint64_t>(), p->lstride, p->leg2.as<double2>(), 0, &p->prof, ldc);
		{ const AnaSet as = ana_set(p); resample_to_cc_adjoint(p, st, p->leg2.as<double2>(), p->leg.as<double2>(), nc, spin, as.M, as.sigma, as.wcc); }
		leg2map(p, st, p->leg.as<double2>(), nr, map, map_dtype, map_cstride, nc);
	}
}

int pxs_analysis(pxs_plan* p, int spin, int adjoint, int nbatch,
                 void* map, int map_dtype, int64_t map_cstride, int64_t map_bstride,
                 void* alm, int alm_dtype, int64_t alm_cstride, int64_t alm_bstride, void* stream)
{
	PXS_TRY
	PXS_REQUIRE(p && alm && map, "pxs_analysis: null argument");
	std::lock_guard<std::mutex> plan_lock(p->call_mu);
	PXS_REQUIRE(p->is_grid, "pxs_analysis needs a grid2d plan");
	PXS_REQUIRE(map_dtype == PX_F32 || map_dtype == PX_F64, "map must be float32 or float64");
	PXS_REQUIRE(nbatch >= 1, "pxs_analysis: nbatch must be >= 1");
	if (p->lmax > grid_maxlmax(p->geometry, p->nring)) throw Error(PXS_ERR_ARG, "too few rings for analysis up to requested lmax");
	PXS_HIP(hipSetDevice(p->device));
	hipStream_t st = (hipStream_t)stream;
	const size_t aesz = alm_dtype == PX_C64 ? 8 : 16, mesz = map_dtype == PX_F32 ? 4 : 8;
	const int chunk = ana_path(p, adjoint) == ANA_UNFUSED ? 1 : batch_chunk(p, nbatch, spin == 0 ? 1 : 2);
	reserve_call(p, spin, PXS_MODE_STANDARD, false, adjoint != 0, std::min(chunk, nbatch));
	for (int b0 = 0; b0 < nbatch; b0 += chunk) {
		const int nb = std::min(chunk, nbatch - b0);
		analysis_core(p, spin, adjoint, nb, (char*)map + mesz*(size_t)b0*map_bstride, map_dtype, map_cstride, map_bstride,
			(char*)alm + aesz*(size_t)b0*alm_bstride, alm_dtype, alm_cstride, alm_bstride, st);
	}
	PXS_CATCH
}

} // extern "C"
