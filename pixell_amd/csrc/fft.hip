// Batched 1-D FFT engine for gfx950 (see fft.hpp).
//
// One workgroup (256 threads, 4 wave64) keeps T lines of n complex f64 points in LDS
// (<= 64 KiB incl. the twiddle table => 2 workgroups per CU), performs decimation-in-time
// mixed-radix passes in place after a digit-reversed scatter on load, and stores with the
// thread->element map chosen so that the unit-stride direction of global memory is the
// fast lane direction (coalesced 16-byte accesses).  Lines longer than FFT_NLOC_MAX use
// the four-step split N = n1*n2 (pass A strided + twiddle, pass B contiguous), with the
// intermediate processed in chunks of `temp_budget` bytes (see the constructor).
#include "fft.hpp"
#include "fft_dev.hpp"
#include <cmath>
#include <algorithm>
#include <functional>

namespace pxs {

struct KArgs {
	int n, nfac, T, generic, mode, forward, n1, n2;
	int ns;                          // LDS line stride in points (n padded to odd: see fill_sub)
	long N;
	const PassDesc* pass;            // device table (a by-value array indexed at run time made the compiler spill the whole
	                                 // argument struct to scratch in the larger kernels)
	const int* perm; const double2* tw;
	FastDiv dn, dT;
	FftDims d; long i_base, i_count;
	long ntile;
	FftLoad ld; FftStore st;
	double2* temp; const double2* bigtw;
	int load_inner_fast, store_inner_fast, tile_i;
};

__device__ __forceinline__ double2 read_elem(const void* p, int dtype, long off) {
	switch (dtype) {
		case PX_F32:  return make_double2((double)((const float*)p)[off], 0.0);
		case PX_F64:  return make_double2(((const double*)p)[off], 0.0);
		case PX_C64:  { float2 v = ((const float2*)p)[off]; return make_double2(v.x, v.y); }
		default:      return ((const double2*)p)[off];
	}
}
__device__ __forceinline__ void write_elem(void* p, int dtype, long off, double2 v) {
	switch (dtype) {
		case PX_F32:  ((float*)p)[off] = (float)v.x; break;
		case PX_F64:  ((double*)p)[off] = v.x; break;
		case PX_C64:  ((float2*)p)[off] = make_float2((float)v.x, (float)v.y); break;
		default:      ((double2*)p)[off] = v; break;
	}
}

// value of input element e (0 <= e < N) of line (i,o1,o2)
template<int LM> __device__ __forceinline__ double2 load_functor(const KArgs& a, long i, long o1, long o2, long e) {
	const FftLoad& ld = a.ld;
	const long base = i*a.d.is_i + o1*a.d.is_o1 + o2*a.d.is_o2;
	const long N = a.N;
	switch (LM) {      // compile-time: one instantiation of the kernels per load mode keeps the prefetch code small
	case LD_PLAIN: {
		const long se = e - ld.shift;
		if (se < 0 || (ld.ne >= 0 && se >= ld.ne)) return make_double2(0, 0);
		double2 v = read_elem(ld.ptr, ld.dtype, base + se*a.d.is_e);
		if (ld.mul) v = cmul(v, ld.mul[e]);
		return v; }
	case LD_HERM: {
		if (!ld.herm_fold) {
			// plain c2r (numpy.fft.irfft semantics): X[e] = h[e] for e < nh, conj(h[N-e]) otherwise
			const long Nh = ld.herm_n > 0 ? ld.herm_n : N;
			if (e >= Nh) return make_double2(0, 0);
			double2 v;
			if (e < ld.ne) v = read_elem(ld.ptr, ld.dtype, base + e*a.d.is_e);
			else if (Nh - e < ld.ne) v = cconj(read_elem(ld.ptr, ld.dtype, base + (Nh-e)*a.d.is_e));
			else return make_double2(0, 0);
			if (ld.mul) v = cmul(v, ld.mul[e]);
			return v;
		}
		// SHT ring synthesis: ring(x) = Re h[0] + 2 Re sum_{m>=1} h[m] e^{i m phi_x}, i.e.
		// X[e] = sum_{m = e (mod N)} h[m] + sum_{m = -e (mod N), m > 0} conj(h[m]); only Re h[0] counts.
		// Without aliasing (2*ne <= N) at most one term exists; with mmax >= N/2 the sums fold m onto the ring.
		double2 acc = make_double2(0, 0);
		for (long m = e; m < ld.ne; m += N) {
			double2 v = read_elem(ld.ptr, ld.dtype, base + m*a.d.is_e);
			if (ld.mul) v = cmul(v, ld.mul[m]);
			if (m == 0) v.y = 0;
			acc = cadd(acc, v);
		}
		for (long m = N - e; m < ld.ne; m += N) {
			double2 v = read_elem(ld.ptr, ld.dtype, base + m*a.d.is_e);
			if (ld.mul) v = cmul(v, ld.mul[m]);
			acc = cadd(acc, cconj(v));
		}
		return acc; }
	case LD_MIRROR: {
		long src = e;
		const bool odd = ((ld.par_step*(i + a.i_base) + ld.par0) & 1) != 0;
		bool neg = false;
		if (e >= ld.ne) { src = N - e - ld.mir_c; if (src < 0) src += N; neg = odd; }
		// a sample that is its own mirror image (a pole ring) cannot carry odd parity: project it out
		if (odd && (2*e + ld.mir_c) % N == 0) return make_double2(0, 0);
		double2 v = read_elem(ld.ptr, ld.dtype, base + src*a.d.is_e);
		if (ld.mul) v = cmul(v, ld.mul[src]);
		return neg ? make_double2(-v.x, -v.y) : v; }
	case LD_MIRROR_PAIR: {
		// two source lines of opposite parity share one transform: z = ext(even line) + ext(odd line).
		// direct samples: a + b; mirrored samples: (+a_even - b_odd); self-mirrored samples keep only the even line.
		const long la = 2*(i + a.i_base), lb = la + 1;
		const bool a_odd = (ld.par0 & 1) != 0;            // parity of line 2i (lines alternate parity)
		long src = e; bool mir = false;
		if (e >= ld.ne) { src = N - e - ld.mir_c; if (src < 0) src += N; mir = true; }
		const bool selfm = ((2*e + ld.mir_c) % N) == 0;
		const long ba = (2*i)*a.d.is_i + o1*a.d.is_o1 + o2*a.d.is_o2;
		double2 va = read_elem(ld.ptr, ld.dtype, ba + src*a.d.is_e);
		double2 vb = (lb < ld.pair_lines) ? read_elem(ld.ptr, ld.dtype, ba + a.d.is_i + src*a.d.is_e) : make_double2(0, 0);
		// odd-parity line: sign flip on mirrored samples, zero on self-mirrored ones
		double2& vo = a_odd ? va : vb;
		if (selfm) vo = make_double2(0, 0);
		else if (mir) { vo.x = -vo.x; vo.y = -vo.y; }
		return cadd(va, vb); }
	case LD_REAL_PAIR: {
		// two real lines (2i, 2i+1) as one complex line: z = a + i b  (half the transforms for real data)
		const long ba = (2*i)*a.d.is_i + o1*a.d.is_o1 + o2*a.d.is_o2;
		const double va = read_elem(ld.ptr, ld.dtype, ba + e*a.d.is_e).x;
		const double vb = (2*(i + a.i_base) + 1 < ld.pair_lines) ? read_elem(ld.ptr, ld.dtype, ba + a.d.is_i + e*a.d.is_e).x : 0.0;
		return make_double2(va, vb); }
	case LD_HERM_PAIR: {
		// Z = X_A + i X_B with X the Hermitian extension of the half spectra of lines (2i, 2i+1); no aliasing (2 ne <= N)
		long m; bool cj;
		if (e < ld.ne) { m = e; cj = false; }
		else if (N - e < ld.ne) { m = N - e; cj = true; }
		else return make_double2(0, 0);
		const long ba = (2*i)*a.d.is_i + o1*a.d.is_o1 + o2*a.d.is_o2;
		double2 ha = read_elem(ld.ptr, ld.dtype, ba + m*a.d.is_e);
		double2 hb = (2*(i + a.i_base) + 1 < ld.pair_lines) ? read_elem(ld.ptr, ld.dtype, ba + a.d.is_i + m*a.d.is_e) : make_double2(0, 0);
		if (m == 0) { ha.y = 0; hb.y = 0; }
		if (cj) { ha.y = -ha.y; hb.y = -hb.y; }
		return make_double2(ha.x - hb.y, ha.y + hb.x); }
	case LD_SPEC: {
		const long Ns = ld.ne;
		long k = (2*e <= N) ? e : e - N;
		long ak = k < 0 ? -k : k;
		if (ld.kmax >= 0 && ak > ld.kmax) return make_double2(0, 0);
		if (2*ak > Ns) return make_double2(0, 0);
		long si = k >= 0 ? k : Ns + k;
		double2 v = read_elem(ld.ptr, ld.dtype, base + si*a.d.is_e);
		if (2*ak == Ns && ld.nyq_half && N != Ns) { v.x *= 0.5; v.y *= 0.5; }
		if (ld.mul) { double2 ph = ld.mul[ak]; if (k < 0) ph.y = -ph.y; v = cmul(v, ph); }
		return v; }
	default: { // LD_SPEC_ADJ: adjoint of LD_SPEC(Ns=N here -> larger): source is the larger spectrum of length ld.ne
		const long Nb = ld.ne;
		long k = (2*e <= N) ? e : e - N;
		long ak = k < 0 ? -k : k;
		if (ld.kmax >= 0 && ak > ld.kmax) return make_double2(0, 0);
		double2 ph = ld.mul ? ld.mul[ak] : make_double2(1, 0);
		if (2*ak == N && ld.nyq_half && N != Nb) {
			double2 vp = read_elem(ld.ptr, ld.dtype, base + ak*a.d.is_e);
			double2 vm = read_elem(ld.ptr, ld.dtype, base + (Nb-ak)*a.d.is_e);
			double2 r = cadd(cmul(vp, cconj(ph)), cmul(vm, ph));
			return make_double2(0.5*r.x, 0.5*r.y);
		}
		long si = k >= 0 ? k : Nb + k;
		double2 v = read_elem(ld.ptr, ld.dtype, base + si*a.d.is_e);
		if (k >= 0) ph.y = -ph.y;       // conj(ph(k)), ph(-k) = conj(ph(k))
		return cmul(v, ph); }
	}
}

__device__ __forceinline__ void store_functor(const KArgs& a, long i, long o1, long o2, long e, double2 v) {
	const FftStore& st = a.st;
	if ((st.ne >= 0 && e >= st.ne) || e < st.shift) return;
	long eo = e - st.shift;
	if (st.two_sided_k >= 0) {
		if (e > st.two_sided_k && e < a.N - st.two_sided_k) return;
		if (st.compact_two_sided && e > st.two_sided_k) eo = st.two_sided_k + (a.N - e);   // (never combined with shift)
	}
	if (st.conj_out) v.y = -v.y;
	if (st.mul) v = cmul(v, st.mul[e]);
	v.x *= st.scale; v.y *= st.scale;
	if (st.real_pair) {
		const long off = (2*i)*a.d.os_i + o1*a.d.os_o1 + o2*a.d.os_o2 + eo*a.d.os_e;
		write_elem(st.ptr, st.dtype, off, make_double2(v.x, 0));
		if (2*(i + a.i_base) + 1 < st.pair_lines) write_elem(st.ptr, st.dtype, off + a.d.os_i, make_double2(v.y, 0));
		return;
	}
	const long off = i*a.d.os_i + o1*a.d.os_o1 + o2*a.d.os_o2 + eo*a.d.os_e;
	write_elem(st.ptr, st.dtype, off, v);
}

// (Tried: radix 6/8/9 butterflies (216 = 8.9.3 in 3 passes instead of 5).  Per-length gain <= 4 %, but the larger
// kernel ran every length slower (n = 200: 0.200 -> 0.238 ms, config 3 FFT stages +10 %); removed.  With the passes
// skipped altogether the kernel moves data at 3.4-3.9 TB/s versus 2.5-2.7 TB/s with them.)
template<int R, int NT> __device__ __forceinline__ void radix_pass(double2* buf, const double2* tw, const KArgs& a, const PassDesc& ps) {
	const int nb = a.n / R;
	const int total = a.T*nb;
	for (int b = threadIdx.x; b < total; b += NT) {
		const uint32_t t = fdiv(b, ps.dnb);
		const uint32_t bb = b - t*nb;
		const uint32_t blk = fdiv(bb, ps.dL);
		const uint32_t q = bb - blk*ps.L;
		const uint32_t p0 = t*a.ns + blk*ps.L*R + q;
		double2 v[R];
#pragma unroll
		for (int i = 0; i < R; i++) v[i] = buf[LPAD(p0 + i*ps.L)];
		if (ps.L > 1) {
			const int step = q*ps.tws;
#pragma unroll
			for (int i = 1; i < R; i++) v[i] = cmul(v[i], tw[i*step]);
		}
		butterfly<R>(v);
#pragma unroll
		for (int i = 0; i < R; i++) buf[LPAD(p0 + i*ps.L)] = v[i];
	}
}

// generic radix: out of place src -> dst, one thread per output point
template<int NT> __device__ __forceinline__ void generic_pass(const double2* src, double2* dst, const double2* tw, const KArgs& a, const PassDesc& ps) {
	const int R = ps.R, n = a.n;
	const int total = a.T*n;
	const int wstep = n / R;
	for (int idx = threadIdx.x; idx < total; idx += NT) {
		const uint32_t t = fdiv(idx, a.dn);
		const uint32_t j = idx - t*n;                 // output position within line
		const uint32_t LR = ps.L*R;
		const uint32_t blk = j / LR;
		const uint32_t r = j - blk*LR;
		const uint32_t ip = r / ps.L;                 // output digit i'
		const uint32_t q = r - ip*ps.L;
		const uint32_t p0 = t*a.ns + blk*LR + q;
		double2 acc = make_double2(0, 0);
		uint32_t widx = 0;                            // (i*ip) mod R
		for (int i = 0; i < R; i++) {
			double2 v = src[LPAD(p0 + i*ps.L)];
			if (ps.L > 1) v = cmul(v, tw[i*q*ps.tws]);
			acc = cadd(acc, cmul(v, tw[widx*wstep]));
			widx += ip; if (widx >= (uint32_t)R) widx -= R;
		}
		dst[LPAD(t*a.ns + j)] = acc;
	}
}

// block -> (tile, other, o1, o2).  mode 0: tile over lines i.  modes 1/2 (four-step): sub-lines are (i, s) with
// s = j2 (pass A) or k1 (pass B); tile_i selects which one is tiled.
struct TileCtx { long s0, other, o1, o2, lo; int nl; };
__device__ __forceinline__ TileCtx tile_decode(const KArgs& a, long bx) {
	TileCtx c;
	const long tile = bx % a.ntile; bx /= a.ntile;
	const long slim = (a.mode == 1) ? a.n2 : a.n1;
	c.other = 0;
	if (a.mode != 0) { const long no = a.tile_i ? slim : a.i_count; c.other = bx % no; bx /= no; }
	c.o1 = bx % a.d.n_o1; c.o2 = bx / a.d.n_o1;
	c.s0 = tile*a.T;
	const long tlim = (a.mode == 0 || a.tile_i) ? a.i_count : slim;
	c.nl = (int)min((long)a.T, tlim - c.s0);
	c.lo = c.o2*a.d.n_o1 + c.o1;                  // outer line index (four-step scratch)
	return c;
}
// input element idx of the tile -> value and LDS slot (digit-reversed); slot -1 = nothing to load
template<int LM> __device__ __forceinline__ void tile_load_one(const KArgs& a, const TileCtx& c, int idx, int total, double2& v, int& pos) {
	const int n = a.n, T = a.T;
	pos = -1; v = make_double2(0, 0);
	if (idx >= total) return;
	uint32_t t, j;
	if (a.load_inner_fast) { j = fdiv(idx, a.dT); t = idx - j*T; }
	else                   { t = fdiv(idx, a.dn); j = idx - t*n; }
	if ((int)t >= c.nl) return;
	const long il = (a.mode == 0 || a.tile_i) ? c.s0 + t : c.other;
	const long sv = (a.mode == 0) ? 0 : (a.tile_i ? c.other : c.s0 + t);
	if (a.mode == 0)      v = load_functor<LM>(a, il, c.o1, c.o2, j);
	else if (a.mode == 1) v = load_functor<LM>(a, il, c.o1, c.o2, (long)j*a.n2 + sv);
	else {
		const long p2 = sv*a.n2 + j;
		v = a.temp[a.tile_i ? (c.lo*a.N + p2)*a.i_count + il : (c.lo*a.i_count + il)*a.N + p2];
	}
	if (!a.forward && a.mode != 2) v.y = -v.y;
	pos = (int)(t*a.ns) + a.perm[j];
}
template<int NT, bool GEN> __device__ __forceinline__ double2* tile_passes(const KArgs& a, double2* cur, double2* oth, const double2* tw) {
	for (int p = 0; p < a.nfac; p++) {
		const PassDesc ps = a.pass[p];
		switch (ps.R) {
			case 2: radix_pass<2, NT>(cur, tw, a, ps); break;
			case 3: radix_pass<3, NT>(cur, tw, a, ps); break;
			case 4: radix_pass<4, NT>(cur, tw, a, ps); break;
			case 5: radix_pass<5, NT>(cur, tw, a, ps); break;
			default: if constexpr (GEN) { generic_pass<NT>(cur, oth, tw, a, ps); double2* x = cur; cur = oth; oth = x; } break;
		}
		PXS_LDS_BARRIER();
	}
	return cur;
}
template<int NT> __device__ __forceinline__ void tile_store(const KArgs& a, const TileCtx& c, const double2* cur) {
	const int n = a.n, T = a.T, total = T*n;
	for (int idx0 = threadIdx.x; idx0 < total; idx0 += 4*NT) {
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const int idx = idx0 + u*NT;
			if (idx >= total) continue;
			uint32_t t, j;
			if (a.store_inner_fast) { j = fdiv(idx, a.dT); t = idx - j*T; }
			else                    { t = fdiv(idx, a.dn); j = idx - t*n; }
			if ((int)t >= c.nl) continue;
			const long il = (a.mode == 0 || a.tile_i) ? c.s0 + t : c.other;
			const long sv = (a.mode == 0) ? 0 : (a.tile_i ? c.other : c.s0 + t);
			double2 v = cur[LPAD(t*a.ns + j)];
			if (a.mode == 1) {
				v = cmul(v, a.bigtw[(long)j*sv]);
				const long p2 = (long)j*a.n2 + sv;
				a.temp[a.tile_i ? (c.lo*a.N + p2)*a.i_count + il : (c.lo*a.i_count + il)*a.N + p2] = v;
			} else {
				if (!a.forward) v.y = -v.y;
				if (a.mode == 0) store_functor(a, il, c.o1, c.o2, j, v);
				else             store_functor(a, il, c.o1, c.o2, sv + (long)a.n1*j, v);
			}
		}
	}
}

// one tile per workgroup
// GEN: the line length has a radix other than 2,3,4,5 (direct-DFT pass, second LDS buffer).  Kept out of the common
// instantiation: these kernels are 30-50 KB of code and run measurably slower when they grow (instruction cache).
template<int LM, int NT, bool GEN> __global__ __launch_bounds__(NT) void fft_lds_kernel(const KArgs a)
{
	PXS_SHARED(double2, lds);
	const int n = a.n, T = a.T;
	double2* tw = lds;                 // [n]
	double2* bufA = lds + n;           // [T*n]
	double2* bufB = bufA + (size_t)T*a.ns + 1;    // only if generic
	const TileCtx c = tile_decode(a, blockIdx.x);
	for (int k = threadIdx.x; k < n; k += NT) tw[k] = a.tw[k];
	// ---- load ---- (LU independent global loads in flight per thread before the LDS scatter)
	const int total = T*n;
	constexpr int LU = 4;
	for (int idx0 = threadIdx.x; idx0 < total; idx0 += LU*NT) {
		double2 v[LU]; int pos[LU];
#pragma unroll
		for (int u = 0; u < LU; u++) tile_load_one<LM>(a, c, idx0 + u*NT, total, v[u], pos[u]);
#pragma unroll
		for (int u = 0; u < LU; u++) if (pos[u] >= 0) bufA[LPAD(pos[u])] = v[u];
	}
	PXS_LDS_BARRIER();
	const double2* cur = tile_passes<NT, GEN>(a, bufA, bufB, tw);
	tile_store<NT>(a, c, cur);
}

// (Tried: a persistent variant of this kernel that issues the global loads of its NEXT tile into registers before the
// passes of the current one.  The compiler needed 200-260 VGPRs for it (2 waves per SIMD), or 70-230 spills when held
// to 128; not pursued.  A second form that issued the next tile's loads only after the passes (prefetch registers live
// across the store loop only) still took 200-260 VGPRs -- the persistent loop makes every load-functor invariant live -- and
// ran 2x slower (n = 200: 0.19 -> 0.44 ms).  What the attempts left behind: the LDS-only barrier, the per-load-mode instantiation and the pass
// table in device memory, which keep the argument struct out of scratch.)
// launch one of the two kernels over nblk tiles
template<int LM> static void launch_tiles_m(const KArgs& k, long nblk, size_t sh, hipStream_t st, int nt_override) {
#ifdef PXS_HOST_SIM
	hipLaunchKernelGGL((fft_lds_kernel<LM, 1, true>), dim3((unsigned)nblk), dim3(1), sh, st, k);
#else
	// 512 threads per workgroup: LDS allows 4 workgroups of 2048 points per CU, and at < 64 VGPRs twice the waves fit
	// tiles of more than 2048 points (lines of 1025..2048 points: 64 KiB of LDS, 2 workgroups per CU) get 512 threads:
	// measured 1.39 -> 1.86 TB/s at n = 2048; for the 32 KiB tiles 512 threads were 3-8 % slower
	static const int nt_env = [] { const char* e = lab_getenv("PXS_FFT_NT"); return e ? atoi(e) : 0; }();
	const int nt = nt_override ? nt_override : nt_env ? nt_env : ((long)k.T*k.n >= 2048 ? 512 : 256);
	static const bool once = [] {
		(void)hipFuncSetAttribute((const void*)fft_lds_kernel<LM, 256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 256);
		(void)hipFuncSetAttribute((const void*)fft_lds_kernel<LM, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 256);
		(void)hipFuncSetAttribute((const void*)fft_lds_kernel<LM, 512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 256);
		(void)hipFuncSetAttribute((const void*)fft_lds_kernel<LM, 128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 256);
		return true; }();
	(void)once;
	if (k.generic)      hipLaunchKernelGGL((fft_lds_kernel<LM, 256, true>), dim3((unsigned)nblk), dim3(256), sh, st, k);
	else if (nt == 512) hipLaunchKernelGGL((fft_lds_kernel<LM, 512, false>), dim3((unsigned)nblk), dim3(512), sh, st, k);
	else if (nt == 128) hipLaunchKernelGGL((fft_lds_kernel<LM, 128, false>), dim3((unsigned)nblk), dim3(128), sh, st, k);
	else                hipLaunchKernelGGL((fft_lds_kernel<LM, 256, false>), dim3((unsigned)nblk), dim3(256), sh, st, k);
#endif
}
static void launch_tiles(const KArgs& k, long nblk, size_t sh, hipStream_t st, int nt_override = 0) {
	const int lm = k.mode == 2 ? (int)LD_PLAIN : k.ld.mode;      // pass B reads the four-step scratch, no functor
	switch (lm) {
		case LD_PLAIN:       launch_tiles_m<LD_PLAIN>(k, nblk, sh, st, nt_override); break;
		case LD_HERM:        launch_tiles_m<LD_HERM>(k, nblk, sh, st, nt_override); break;
		case LD_MIRROR:      launch_tiles_m<LD_MIRROR>(k, nblk, sh, st, nt_override); break;
		case LD_SPEC:        launch_tiles_m<LD_SPEC>(k, nblk, sh, st, nt_override); break;
		case LD_SPEC_ADJ:    launch_tiles_m<LD_SPEC_ADJ>(k, nblk, sh, st, nt_override); break;
		case LD_MIRROR_PAIR: launch_tiles_m<LD_MIRROR_PAIR>(k, nblk, sh, st, nt_override); break;
		case LD_REAL_PAIR:   launch_tiles_m<LD_REAL_PAIR>(k, nblk, sh, st, nt_override); break;
		case LD_HERM_PAIR:   launch_tiles_m<LD_HERM_PAIR>(k, nblk, sh, st, nt_override); break;
		default: throw Error(PXS_ERR_ARG, "fft: unknown load mode");
	}
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
struct FftSub {
	int n; std::vector<int> fac; bool generic = false;
	DevBuf perm, tw, d_pass;
	PassDesc pass[FFT_MAXFAC]; int nfac = 0;
};

static std::vector<int> factorize(long n) {
	std::vector<int> f;
	// Odd radices first: in the first pass (L = 1) lane b touches points b*R + i, a stride of R points of 16 bytes:
	// conflict-free over the 64 LDS banks for R = 3, 5 but 4-way for R = 4.  The last radix sets the stride n/R of the
	// digit-reversed scatter on load; 4 keeps it the least harmful for lengths that are multiples of 8.
	static const int oddfirst = [] { const char* e = lab_getenv("PXS_FFT_ODDFIRST"); return e ? atoi(e) : 1; }();
	if (oddfirst) {
		while (n % 3 == 0) { f.push_back(3); n /= 3; }
		while (n % 5 == 0) { f.push_back(5); n /= 5; }
		int twos = 0; while (n % 2 == 0) { twos++; n /= 2; }
		if (twos & 1) { f.push_back(2); twos--; }
		while (twos >= 2) { f.push_back(4); twos -= 2; }
	}
	while (n % 4 == 0) { f.push_back(4); n /= 4; }
	while (n % 2 == 0) { f.push_back(2); n /= 2; }
	while (n % 3 == 0) { f.push_back(3); n /= 3; }
	while (n % 5 == 0) { f.push_back(5); n /= 5; }
	for (long p = 7; p*p <= n; p += 2) while (n % p == 0) { f.push_back((int)p); n /= p; }
	if (n > 1) f.push_back((int)n);
	return f;
}

static void twiddles(long n, std::vector<double2>& tw) {
	tw.resize(n);
	const long double tp = 6.283185307179586476925286766559L;
	for (long k = 0; k < n; k++) {
		// exploit octant symmetry for accuracy
		long double ang = tp*(long double)k/(long double)n;
		tw[k] = make_double2((double)cosl(ang), (double)(-sinl(ang)));
	}
}

FftContext::FftContext(int device) : device_(device) {
	// Four-step scratch: lines are processed in chunks of at most this many bytes.  It used to be 192 MiB so that the
	// intermediate would stay in the 256 MiB Infinity Cache between the two passes; measured at config 3 the chunk count is
	// what matters (192 MiB 431.8 ms, 1 GiB 420.0, 4 GiB 415.9 per round trip), so: 4 GiB, never more than 1/16 of free memory.
	size_t fr = 0, tot = 0;
	if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr > 0) temp_budget = std::min<size_t>(size_t(4) << 30, std::max<size_t>(fr/16, size_t(64) << 20));
}
FftContext::~FftContext() {}

static bool split_two(long n, long& n1, long& n2) {
	// choose n1*n2 = n, both <= FFT_NLOC_MAX, as balanced as possible
	// prefer both factors multiples of 8 (column runs and row starts then fall on 128-byte lines), not more lopsided than 1:4
	long best = -1, best8 = -1;
	for (long a = 1; a*a <= n; a++) if (n % a == 0) {
		long b = n/a; if (b > FFT_NLOC_MAX) continue;
		best = a;
		if (a % 8 == 0 && b % 8 == 0 && 4*a >= b) best8 = a;
	}
	if (best < 0) return false;
	static const int align8 = [] { const char* e = lab_getenv("PXS_FFT_ALIGN8"); return e ? atoi(e) : 1; }();
	if (best8 > 0 && align8) best = best8;
	n1 = best; n2 = n/best;
	return n1 <= FFT_NLOC_MAX && n2 <= FFT_NLOC_MAX;
}

bool FftContext::supported(long n, std::string* why) {
	if (n < 1) { if (why) *why = "length < 1"; return false; }
	if (n <= FFT_NLOC_MAX) {
		if ((int)factorize(n).size() > FFT_MAXFAC) { if (why) *why = "too many factors"; return false; }
		return true;
	}
	long n1, n2;
	if (!split_two(n, n1, n2)) {
		if (why) *why = "FFT length " + std::to_string(n) + " has no factorisation n1*n2 with both factors <= " +
			std::to_string(FFT_NLOC_MAX) + " (large prime factor; Bluestein not implemented)";
		return false;
	}
	return true;
}

long FftContext::good_size(long n) {
	for (long m = std::max<long>(n, 1);; m++) {
		long r = m;
		while (r % 2 == 0) r /= 2; while (r % 3 == 0) r /= 3; while (r % 5 == 0) r /= 5;
		if (r == 1 && supported(m)) return m;
	}
}

// Fewest passes with radices from {2,3,4,5} and the composite register radices {6,8,9,10,12,15,16} (fft_dev.hpp); PXS_FFT_RADICES
// restricts the set (tuning); only those up to PXS_COMP_MAXR are compiled in (fft_dev.hpp).  Odd radices go first: in the first passes (small L) consecutive lanes are R points apart, which
// for an even R puts them on few LDS banks.
static std::vector<int> factorize_comp(long n, int maxr) {
	static const std::vector<int> allowed = [] {
		std::vector<int> a; const char* e = lab_getenv("PXS_FFT_RADICES");
		std::string s = e ? e : "16,15,12,10,9,8,7,6,5,4,3,2";
		size_t pos = 0;
		while (pos < s.size()) { size_t c = s.find(',', pos); if (c == std::string::npos) c = s.size(); int v = atoi(s.substr(pos, c-pos).c_str()); if (v >= 2 && (v <= 5 || v <= PXS_COMP_MAXR)) a.push_back(v); pos = c+1; }
		return a; }();
	std::vector<int> best, cur;
	std::function<void(long, size_t)> rec = [&](long m, size_t from) {
		if (m == 1) { if (best.empty() || cur.size() < best.size()) best = cur; return; }
		if (!best.empty() && cur.size()+1 >= best.size()) return;
		for (size_t i = from; i < allowed.size(); i++) if (m % allowed[i] == 0 && (allowed[i] == 7 ? maxr >= 9 : (allowed[i] <= 5 || allowed[i] <= maxr))) { cur.push_back(allowed[i]); rec(m/allowed[i], i); cur.pop_back(); }
	};
	rec(n, 0);
	if (best.empty()) return factorize(n);
	std::stable_sort(best.begin(), best.end(), [](int a, int b) { const bool oa = a & 1, ob = b & 1; if (oa != ob) return oa; return a > b; });
	return best;
}

std::shared_ptr<FftSub> FftContext::sub(long n, bool comp, int maxr) {
	std::lock_guard<std::mutex> g(mu_);
	static const bool comp_on = [] { const char* e = lab_getenv("PXS_FFT_COMP"); return e ? atoi(e) != 0 : true; }();
	comp = comp && comp_on;
	maxr = std::min(maxr, PXS_COMP_MAXR);
	const long key = comp ? -(n + ((long)maxr << 40)) : n;
	auto it = subs_.find(key);
	if (it != subs_.end()) return it->second;
	auto s = std::make_shared<FftSub>();
	s->n = (int)n; s->fac = comp ? factorize_comp(n, maxr) : factorize(n);
	PXS_REQUIRE((int)s->fac.size() <= FFT_MAXFAC, "FFT: too many factors");
	int L = 1; s->nfac = (int)s->fac.size();
	std::vector<int> Ls(s->nfac+1); Ls[0] = 1;
	for (int p = 0; p < s->nfac; p++) {
		int R = s->fac[p];
		const bool comp_r = comp && R <= maxr && (R == 6 || R == 8 || R == 9 || R == 10 || R == 12 || R == 15 || R == 16);
		const bool seven = comp && R == 7 && maxr >= 9;      // plain radix 7 of the chain kernels' theta stages (fft_dev.hpp)
		if (R != 2 && R != 3 && R != 4 && R != 5 && !comp_r && !seven) s->generic = true;
		PassDesc& ps = s->pass[p];
		ps.R = R; ps.L = L; ps.tws = (int)(n/((long)L*R));
		ps.dL = make_fastdiv(L); ps.dnb = make_fastdiv((uint32_t)(n/R));
		L *= R; Ls[p+1] = L;
	}
	std::vector<int> perm(n);
	for (long j = 0; j < n; j++) {
		long t = j, pos = 0;
		for (int p = s->nfac-1; p >= 0; p--) { long i = t % s->fac[p]; t /= s->fac[p]; pos += i*Ls[p]; }
		perm[j] = (int)pos;
	}
	std::vector<double2> tw; twiddles(n, tw);
	s->perm = upload(perm); s->tw = upload(tw);
	s->d_pass = upload(std::vector<PassDesc>(s->pass, s->pass + FFT_MAXFAC));
	subs_[key] = s;
	return s;
}

FftContext::SubView FftContext::view(long n, int maxr) {
	auto s = sub(n, true, maxr);
	SubView v; v.n = s->n; v.nfac = s->nfac; v.ns = s->n | 1; v.generic = s->generic ? 1 : 0;
	v.pass = s->d_pass.p; v.perm = s->perm.as<int>(); v.tw = s->tw.as<double2>();
	return v;
}

const double2* FftContext::bigtw(long n) {
	std::lock_guard<std::mutex> g(mu_);
	auto it = bigtw_.find(n);
	if (it != bigtw_.end()) return it->second.as<double2>();
	std::vector<double2> tw; twiddles(n, tw);
	bigtw_[n] = upload(tw);
	return bigtw_[n].as<double2>();
}

static void fill_sub(KArgs& k, const FftSub& s, long maxlines) {
	k.n = s.n; k.nfac = s.nfac; k.generic = s.generic;
#ifdef PXS_LAB
	{ static const int nopass = [] { const char* e = getenv("PXS_FFT_DEBUG_NOPASS"); return e ? atoi(e) : 0; }(); if (nopass) k.nfac = 0; }   // timing experiments only: wrong results
#endif
	k.pass = s.d_pass.as<PassDesc>();
	k.perm = s.perm.as<int>(); k.tw = s.tw.as<double2>();
	int bufs = s.generic ? 2 : 1;
	static const long env_pts = [] { const char* e = lab_getenv("PXS_FFT_PTS"); return e ? atol(e) : 0L; }();
	long pts = s.n <= 1024 ? FFT_LDS_PTS/2 : FFT_LDS_PTS;    // 32 KiB tiles for short lines: 4-5 workgroups per CU
	if (env_pts > 0 && s.n <= env_pts/2) pts = env_pts;
	long T = (pts - s.n)/((long)bufs*s.n);
	if (T < 1) T = 1;
	if (T > 64) T = 64;
	// Tiles that run along the contiguous memory direction (four-step column passes, adjacent lines) move T*16-byte
	// runs: keep them whole 128-byte lines.  Neighbouring tiles are handled by workgroups on different XCDs, so a
	// line shared by two tiles is fetched from HBM twice (FETCH_SIZE of this kernel was 1.6x its WRITE_SIZE).
	static const int align8 = [] { const char* e = lab_getenv("PXS_FFT_ALIGN8"); return e ? atoi(e) : 1; }();
	if (align8) {
		if (T >= 8) T -= T % 8;
		else if (T >= 5 && (long)bufs*8*s.n + s.n <= FFT_LDS_PTS) T = 8;
	}
	if (T > maxlines) T = maxlines;
	k.T = (int)T;
	k.dn = make_fastdiv(s.n); k.dT = make_fastdiv((uint32_t)T);
	// Lines of the tile are read and written across lanes (stride = line stride) whenever the tile runs along the
	// contiguous memory direction.  With 16-byte points a stride that is a multiple of 8 points (which the 128-byte
	// alignment rule above makes the normal case: 200, 216, 320) puts 16 lanes on 1-2 bank groups: 8-16-way LDS bank
	// conflicts (SQ_LDS_BANK_CONFLICT was 1.9x SQ_ACTIVE_INST_LDS).  An odd stride spreads them over all banks.
	static const int lpad = [] { const char* e = lab_getenv("PXS_FFT_LINEPAD"); return e ? atoi(e) : 1; }();
	k.ns = lpad ? (s.n | 1) : s.n;
}

static size_t lds_bytes(const KArgs& k) { const size_t pts = (size_t)k.T*k.ns + 2; return sizeof(double2)*((size_t)k.n + pts*(k.generic ? 2 : 1)); }

void FftContext::exec(hipStream_t st, long n, bool forward, const FftDims& d, const FftLoad& ld, const FftStore& stf) {
	std::string why;
	if (!supported(n, &why)) throw Error(PXS_ERR_UNSUPPORTED, why);
	if (d.n_i <= 0 || d.n_o1 <= 0 || d.n_o2 <= 0) return;
	KArgs k; memset(&k, 0, sizeof(k));
	k.N = n; k.forward = forward ? 1 : 0; k.d = d; k.ld = ld; k.st = stf; k.i_base = 0; k.i_count = d.n_i;
	if (n <= FFT_NLOC_MAX) {
		auto s = sub(n);
		fill_sub(k, *s, d.n_i);
		k.mode = 0; k.n1 = (int)n; k.n2 = 1;
		k.ntile = (d.n_i + k.T - 1)/k.T;
		// lines adjacent in memory (|stride| of the line index smaller than the element stride) => line index fastest
		k.load_inner_fast  = (std::abs(d.is_i) < std::abs(d.is_e)) ? 1 : 0;
		k.store_inner_fast = (std::abs(d.os_i) < std::abs(d.os_e)) ? 1 : 0;
		long nblk = k.ntile*d.n_o1*d.n_o2;
		size_t sh = lds_bytes(k);
		launch_tiles(k, nblk, sh, st);
		PXS_HIP(hipGetLastError());
		return;
	}
	long n1, n2; split_two(n, n1, n2);
	auto s1 = sub(n1); auto s2 = sub(n2);
	const double2* btw = bigtw(n);
	// chunk lines so that the scratch stays below temp_budget
	long lines_budget = std::max<long>(1, (long)(temp_budget/(sizeof(double2)*n)));
	DevBuf* tbuf = nullptr;
	{
		std::lock_guard<std::mutex> g(mu_);
		size_t want = sizeof(double2)*n*std::min<long>(lines_budget, d.n_i*d.n_o1*d.n_o2);
		tbuf = &temps_[st];     // one scratch per stream: transforms issued on different streams may overlap
		tbuf->ensure(want);
	}
	// iterate over o2, o1 chunks, i chunks
	long o1_per = 1, i_per = d.n_i;
	if (d.n_i <= lines_budget) { o1_per = std::min<long>(d.n_o1, std::max<long>(1, lines_budget/d.n_i)); }
	else i_per = lines_budget;
	for (long o2 = 0; o2 < d.n_o2; o2++)
	for (long o1 = 0; o1 < d.n_o1; o1 += o1_per)
	for (long i0 = 0; i0 < d.n_i; i0 += i_per) {
		long no1 = std::min(o1_per, d.n_o1 - o1), ni = std::min(i_per, d.n_i - i0);
		KArgs a = k;
		a.d.n_o1 = no1; a.d.n_o2 = 1; a.i_base = i0; a.i_count = ni;
		// shift base pointers
		auto esz = [](int dt) { return dt == PX_F32 ? 4 : dt == PX_F64 ? 8 : dt == PX_C64 ? 8 : 16; };
		const long lmul = (ld.mode == LD_MIRROR_PAIR || ld.mode == LD_REAL_PAIR || ld.mode == LD_HERM_PAIR) ? 2 : 1;   // packed input lines
		const long smul = stf.real_pair ? 2 : 1;
		a.ld.ptr = (const char*)ld.ptr + esz(ld.dtype)*(o2*d.is_o2 + o1*d.is_o1 + lmul*i0*d.is_i);
		a.st.ptr = (char*)stf.ptr + esz(stf.dtype)*(o2*d.os_o2 + o1*d.os_o1 + smul*i0*d.os_i);
		a.temp = tbuf->as<double2>(); a.bigtw = btw; a.n1 = (int)n1; a.n2 = (int)n2;
		const int tile_i = (std::abs(d.is_i) < std::abs(d.is_e)) ? 1 : 0;
		// pass A: n1-point FFTs over j1 for each (i, j2); tile over j2 (lines contiguous) or over i (lines adjacent)
		KArgs pa = a; fill_sub(pa, *s1, tile_i ? ni : n2); pa.mode = 1; pa.tile_i = tile_i;
		pa.ntile = ((tile_i ? ni : n2) + pa.T - 1)/pa.T;
		pa.load_inner_fast = 1; pa.store_inner_fast = 1;
		long nblkA = pa.ntile*(tile_i ? n2 : ni)*no1;
		launch_tiles(pa, nblkA, lds_bytes(pa), st);
		// pass B: n2-point FFTs over j2 for each (i, k1)
		KArgs pb = a; fill_sub(pb, *s2, tile_i ? ni : n1); pb.mode = 2; pb.tile_i = tile_i;
		pb.ntile = ((tile_i ? ni : n1) + pb.T - 1)/pb.T;
		pb.load_inner_fast = tile_i ? 1 : 0; pb.store_inner_fast = 1;
		long nblkB = pb.ntile*(tile_i ? n1 : ni)*no1;
		launch_tiles(pb, nblkB, lds_bytes(pb), st);
		PXS_HIP(hipGetLastError());
	}
}

} // namespace pxs
