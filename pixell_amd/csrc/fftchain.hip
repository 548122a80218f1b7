// Fused FFT chains of the SHT for gfx950.
//
// Every long transform (ring FFT of nphi points, theta FFTs of N, M, Ncc points) is a four-step transform
// X = a*b: pass 1 runs a-point transforms over the residues mod b, pass 2 b-point transforms that produce the residues
// mod a.  When two consecutive transforms share the modulus g (N = g*bN, M = g*g2, Ncc = ac*g) the residue class that
// pass 2 of one transform produces is exactly the class pass 1 of the next one consumes -- the spectrum resize between
// them maps class r to class r -- so both passes run back to back on the same lines in LDS:
//
//   analysis  (leg on the map's rings -> weighted leg on the CC grid), per pair of columns (m, m+1):
//     RA1  mirror-pair extension, FFT_g           -> Y1      RA2  FFT_bN, shift/pad to M, IFFT_g2     -> Z2
//     RA3  IFFT_g, x |sin| series, FFT_g          -> V3      RA4  FFT_g2, keep |k| <= lmax, IFFT_ac   -> U4
//     RA5  IFFT_g, separate the pair, weights     -> leg_cc
//   synthesis (leg on the CC grid -> ring spectra h[ring][m]):
//     RS1  mirror-pair extension, FFT_gs -> Y    RS2  FFT_bs, keep |k| <= lmax, shift, IFFT_aNs -> Z    RS3  IFFT_gs, separate, transpose -> h
//   ring FFTs: MA1/MA2 (two real rings per complex line; pass 2 unpacks the pair and writes leg[m][ring] directly),
//              MS1/MS2 (Hermitian pair load straight from h[ring][m]; pass 2 writes the two rings).
//
// Every intermediate array is written once and read once; one side of each access is fully contiguous, the other
// side moves runs of T*16 bytes (T = lines per tile, a multiple of 8: whole 128-byte lines, measured at copy speed
// on MI355X with tools/stride_bw.hip).  All global loads of a tile are issued before the first LDS write.
// Replaces ducc0's ring FFTs / resample_theta inside synthesis_2d / analysis_2d (pixell/curvedsky.py:907-924, 1032-1046).
#include "fftchain.hpp"
#include "fft_dev.hpp"
#include "chain_dev.hpp"
#include <algorithm>
#include <set>
#include <type_traits>
#include <cmath>

namespace pxs {

#ifndef PXS_CH_PTS
#define PXS_CH_PTS 2560    /* complex points per tile */
#endif
#ifdef PXS_HOST_SIM
static constexpr int CH_NT = 1, CH_MAXE = PXS_CH_PTS;
#else
#ifndef PXS_CH_NT
#define PXS_CH_NT 512      /* threads per workgroup: measured at config 3 (ring FFT + theta resampling, one stream) 256: 129 ms, 512: 115 ms */
#endif
static constexpr int CH_NT = PXS_CH_NT, CH_MAXE = (PXS_CH_PTS + PXS_CH_NT - 1)/PXS_CH_NT;
#endif
static constexpr int CH_TILE_PTS = PXS_CH_PTS;
static constexpr int CH_NMAX = 512;           // longest LDS sub-transform of a chain stage      // points per tile = CH_NT * CH_MAXE on the device

struct TileC { int outer, t0, nl, comp, q0; };

// Tile-blocked intermediates (an option, see blocked_on()).  A stage whose lines are the fast index of its output ("line-fast"
// store) writes runs of T*16 bytes (T = 8-16 lines: 128-256 bytes) at a row stride, and the next stage reads whole rows.  With the
// blocked layout a writer's tile -- lines [t0, t0 + w) x all n outputs -- is ONE contiguous chunk [e][line in tile] of n*w points
// at offset t0*n of its outer slab, and a reader, whose tile is Tr consecutive rows e, finds Tr*w points of every chunk contiguous
// (1-2 KB).  BlkIn describes the chunks to the reader: width Tw of the full ones, their number, the width of the last one.
struct BlkIn {
	int Tw, nBf, wL; FastDiv dTTw, dTw, dwL;      // Tw = 0: plain rows of stride ld
	// offset of (row `line` of nrows, element e of na) within the outer slab
	__device__ __forceinline__ long off(int line, int e, int nrows) const {
		const uint32_t B = fdiv((uint32_t)e, dTw);
		const int w = (int)B < nBf ? Tw : wL;
		return (long)B*Tw*nrows + (long)line*w + (e - (int)B*Tw);
	}
	// load order of a reader tile of T rows: chunk, row, element in chunk -- consecutive lanes read consecutive memory
	__device__ __forceinline__ void split(int idx, int T, uint32_t& li, uint32_t& e) const {
		uint32_t B = fdiv((uint32_t)idx, dTTw);
		if ((int)B < nBf) { const uint32_t rem = idx - B*T*Tw; li = fdiv(rem, dTw); e = B*Tw + (rem - li*Tw); }
		else { const uint32_t rem = idx - (uint32_t)nBf*T*Tw; li = fdiv(rem, dwL); e = (uint32_t)nBf*Tw + (rem - li*wL); }
	}
};

struct StageBase {
	// radix 7 compiled into the stage's kernel (MAXR = 9 stages): the theta stages, for ducc0's ring counts.  Its code costs the kernels
	// that hold it 2-4 % (same box, tools/fft2_ab.sh: enmap.fft 17.3 -> 17.7 ms, enmap.ifft 32.2 -> 33.7), so the 2-D FFT stages leave it out
	static constexpr bool R7 = true;
	// threads per workgroup.  Per-stage A/B at C3 (tools/gpu_ntlab.sh, profiles/r04b_ntlab_c3.txt): 512 wins for every stage (256: ring stages
	// +15-25 %, StResize / StSigma +5-9 %; 384: 0 to +10 %) except the transposing split StSplit<1>: from_cc 12.27 -> 11.45 ms with 256
	static constexpr int NT = CH_NT;
	// most points a tile of the stage may hold (the kernel's per-thread element count follows from it)
	static constexpr int PTS = CH_TILE_PTS;
	LdsFft fa, fb;
	BlkIn bin; int bout;       // input in the blocked layout (bin.Tw > 0); write the output blocked
	int T; int ntile;
	FastDiv dT, dna, dnb, dnt;
	// four-step twiddle of the stage's output, W_X^{(t0 + li) e} = W_X^{t0 e} * W_X^{li e}: the first factor is gathered once per tile
	// from the full table btw (into LDS), the second comes from a small table tws[e][li] shared by all tiles (coalesced reads).
	// (One gather per point from the full table made every wave instruction touch 64 cache lines.)
	const double2* btw; const double2* tws;
	// Prefetch ahead: a workgroup of a stage that loads whole rows touches the rows of the tile `pf` workgroups ahead (one dword per
	// 128-byte line, issued after its own loads, consumed by nothing until the kernel ends), so that tile's loads hit the L2 when its
	// workgroup starts -- on the same XCD: workgroups go round-robin to the 8 XCDs and pf is a multiple of 8.  Measured at C3 / C4
	// (tools/chain_lab.py, 5 repetitions, profiles/r04b_prefetch_ahead.txt): pf = 24, 48: +2 %; 96: -0.5 %; 128 ... 256: -2.5 % (to_cc
	// 30.0 -> 28.4 ms, C4 11.6 -> 10.8); 384 ... 768: -2 % (from_cc 17.1 -> 16.5); 1536: 0; 3072: +6 %.  PXS_CH_PF overrides (0: off).
	int pf;
	static constexpr int PF = 256;
	// stages that load strided runs name the 128-byte lines of a tile themselves (pfaddr), PFI per thread at most.  Only the first stage of
	// the 2-D FFTs does: for StFirst of the theta chains and the ring stages MA1 / MS1 it was measured and lost (the ring stages, held to
	// 64 VGPRs, spill with it: profiles/r04b_prefetch_ahead.txt); enmap.ifft of 21600 x 43200: 32.2 -> 30.2 ms.
	static constexpr int PFI = 0;
	__device__ __forceinline__ const void* pfaddr(const TileC&, int) const { return nullptr; }
};

// (Tried: persistent workgroups that issue the global loads of their NEXT tile into registers before the LDS passes of the current
// one.  The prefetch registers pushed the single-transform stages to 151-169 VGPRs and the two-transform stages to 256 (172 with
// 512-thread workgroups), and config 3 got slower: ring FFT 58 -> 67 ms, theta resampling 71 -> 129 ms (102 ms with 512 threads).
// One tile per workgroup, many workgroups per CU in different phases, stays.)
// (Tried in round 3, tools/chain_lab.py, profiles/r03_chain_lab_fused_passes.jsonl: fusing the first radix pass of a transform with
// the global loads (each thread fetches the R inputs of its butterflies) and the last pass with the store -- 7 instead of 10 LDS
// write sweeps and barriers in a two-transform stage.  93 VGPRs (two instead of three workgroups per CU), or 80 with spills under
// __launch_bounds__: C3 chain stages 110.6 ms unfused, 124.8-136.8 ms fused; C4 27.0 / 28.4-30.4.  Not kept.)
// S::MINW = 8 asks the compiler for <= 64 VGPRs (8 waves per SIMD = four 512-thread workgroups per CU where the LDS allows):
// the ring stages, whose passes use radices up to 8 (S::MAXR), gain 6-12 % from it; with radix 9 compiled in the same bound spills.
#ifndef PXS_HOST_SIM
#define PXS_CH_BOUNDS __launch_bounds__(NT, S::MINW)
#else
#define PXS_CH_BOUNDS
#endif
template<class S, int NT, int MAXE> __global__ PXS_CH_BOUNDS void chain_kernel(const S s)
{
	PXS_SHARED(double2, lds);
	const int na = s.fa.n, nb = s.fb.n;
	const int nlast = S::TWO ? nb : na, nslast = S::TWO ? s.fb.ns : s.fa.ns;
	// (two transforms of one length share the W_n table: StSigma at g = 320 then fits three times on a CU instead of twice)
	const bool same_tw = S::TWO && nb == na;
	double2* twa = lds; double2* twb = same_tw ? lds : lds + na; double2* twx = lds + na + (same_tw ? 0 : nb); double2* buf = twx + (S::HAS_TW ? nlast : 0);
	TileC c;
	if (!s.decode((int)blockIdx.x, c)) return;
	const bool have_tw = S::HAS_TW && s.btw != nullptr;
	const int T = s.T;
	int pf_val = 0;
	{	// ---- load: every global load of the tile is in flight before the first LDS write
		const int total = T*na;
		double2 v[MAXE]; int pos[MAXE];
#pragma unroll
		for (int u = 0; u < MAXE; u++) {
			const int idx = threadIdx.x + u*NT; pos[u] = -1;
			if (idx < total) {
				uint32_t li, e;
				if (S::LOAD_LINE_FAST) { e = fdiv(idx, s.dT); li = idx - e*T; }
				else if (s.bin.Tw > 0) s.bin.split(idx, T, li, e);
				else { li = fdiv(idx, s.dna); e = idx - li*na; }
				v[u] = s.load(c, (int)li, (int)e);
				if (S::INV_A) v[u].y = -v[u].y;
				pos[u] = (int)li*s.fa.ns + s.fa.perm[e];
			}
		}
		if ((S::LOADK == 0 || S::PFI > 0) && s.pf > 0 && (long)blockIdx.x + s.pf < (long)gridDim.x) {	// (issued after the tile's own loads: the wait before the LDS writes below does not include it)
			TileC c2;
			if (s.decode((int)blockIdx.x + s.pf, c2)) {
				if (S::LOADK == 0) {	// whole rows: one dword per 128-byte line
					const int per = (na*16 + 127) >> 7, i = threadIdx.x;
					if (i < T*per) {
						const int li = i/per, seg = i - li*per;
						const double2* r = s.row(c2, li);
						if (r) pf_val = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(r) + seg*128);
					}
				} else {	// runs of T elements at a stride: the stage names the lines, at most S::PFI per thread
#pragma unroll
					for (int it = 0; it < S::PFI; it++) {
						const void* q = s.pfaddr(c2, (int)threadIdx.x + it*NT);
						if (q) pf_val += *reinterpret_cast<const int*>(q);
					}
				}
			}
		}
		for (int k = threadIdx.x; k < na; k += NT) twa[k] = s.fa.tw[k];
		if (!same_tw) for (int k = threadIdx.x; k < nb; k += NT) twb[k] = s.fb.tw[k];
		if (have_tw) for (int k = threadIdx.x; k < nlast; k += NT) twx[k] = s.btw[(long)c.t0*k];
#pragma unroll
		for (int u = 0; u < MAXE; u++) if (pos[u] >= 0) buf[pos[u]] = v[u];
	}
	PXS_LDS_BARRIER();
	lds_fft<NT, S::MAXR, S::R7>(buf, twa, s.fa, T);
	if (S::TWO) {	// ---- second transform on the same lines: pull its inputs out of the first one's output (lines fastest across lanes)
		const int total = T*nb;
		double2 v[MAXE]; int pos[MAXE];
#pragma unroll
		for (int u = 0; u < MAXE; u++) {
			const int idx = threadIdx.x + u*NT; pos[u] = -1;
			if (idx < total) {
				const uint32_t e = fdiv(idx, s.dT), li = idx - e*T;
				{ const double2* Al = buf + li*s.fa.ns; v[u] = s.mid(c, (int)li, (int)e, [&](int k) { return Al[k]; }); }
				if (S::INV_B) v[u].y = -v[u].y;
				pos[u] = (int)li*s.fb.ns + s.fb.perm[e];
			}
		}
		PXS_LDS_BARRIER();
#pragma unroll
		for (int u = 0; u < MAXE; u++) if (pos[u] >= 0) buf[pos[u]] = v[u];
		PXS_LDS_BARRIER();
		lds_fft<NT, S::MAXR, S::R7>(buf, twb, s.fb, T);
	}
	{	// ---- store
		const int total = T*nlast;
		const FastDiv dl = S::TWO ? s.dnb : s.dna;
		for (int idx = threadIdx.x; idx < total; idx += NT) {
			uint32_t li, e;
			if (S::STORE_LINE_FAST) { e = fdiv(idx, s.dT); li = idx - e*T; } else { li = fdiv(idx, dl); e = idx - li*nlast; }
			double2 w = make_double2(1, 0);
			if (have_tw) w = cmul(twx[e], s.tws[e*T + li]);
			s.store(c, (int)li, (int)e, [&](int l2, int e2) { return buf[l2*nslast + e2]; }, w);
		}
	}
#ifndef PXS_HOST_SIM
	if (S::LOADK == 0 || S::PFI > 0) asm volatile("" :: "v"(pf_val));
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// theta chains
// ---------------------------------------------------------------------------------------------------------------
// pass 1 of the first transform of a chain: circle index j = b*j1 + j2, line = j2, a-point FFT over j1, four-step twiddle
struct StFirst : StageBase {
	static constexpr int SID = 0;
	static constexpr bool TWO = false, INV_A = false, INV_B = false, LOAD_LINE_FAST = true, STORE_LINE_FAST = true, HAS_TW = true;
	static constexpr int MAXR = 9, MINW = 1;
	PairSrc src; int b; double2* Y; long ldY; int npair; FastDiv dnp;       // outer = comp*npair + pair
	static constexpr int LOADK = 1, STOREK = 0;
	__device__ __forceinline__ const double2* row(const TileC&, int) const { return nullptr; }
	__device__ __forceinline__ bool decode(int bx, TileC& c) const {
		c.outer = fdiv(bx, dnt); c.t0 = (bx - c.outer*ntile)*T; c.nl = min(T, b - c.t0);
		c.comp = fdiv(c.outer, dnp); c.q0 = c.outer - c.comp*npair; return true; }
	__device__ __forceinline__ double2 load(const TileC& c, int li, int e) const {
		if (li >= c.nl) return make_double2(0, 0);
		return src.get(c.comp, c.q0, b*e + c.t0 + li); }
	template<class AF> __device__ __forceinline__ double2 mid(const TileC&, int, int, AF&&) const { return make_double2(0, 0); }
	template<class VF> __device__ __forceinline__ void store(const TileC& c, int li, int e, VF&& val, double2 w) const {
		if (li >= c.nl) return;
		const long o = bout ? ((long)c.outer*b + c.t0)*fa.n + (long)e*c.nl + li : ((long)c.outer*fa.n + e)*ldY + c.t0 + li;
		Y[o] = cmul(val(li, e), w); }
};

// ... of the 2-D FFTs (plain mode; lengths are 2-3-5-smooth there, split_balanced)
struct StFirst2D : StFirst {
	static constexpr bool R7 = false;
	// the tile's samples are runs of nl elements at j = b e + t0 of row q0: every 8th element names a 128-byte line
	static constexpr int PFI = 2;
	__device__ __forceinline__ const void* pfaddr(const TileC& c, int i) const {
		const int segs = (c.nl + 7) >> 3;
		const int e = i/segs; if (e >= fa.n) return nullptr;
		return src.leg + (long)c.comp*src.cstride + (long)c.q0*src.ld + (b*e + c.t0 + 8*(i - e*segs)); }
};

// pass 2 of transform X1 (forward, b1 points) + spectrum resize + pass 1 of transform X2 (backward, a2 points); shared modulus g.
// in: Y[outer][k1 < g][j2 < b1]; out: Z[outer][k1' < a2][k1 < g] (row stride ldZ).
// resize rule: signed frequency kappa of the X2 slot; zero beyond kmax (or beyond X1/2); Nyquist bin of X1 halved when nyq;
// optional phase table ph[|kappa|], conjugated for kappa < 0.
struct StResize : StageBase {
	static constexpr int SID = 1;
	static constexpr bool TWO = true, INV_A = false, INV_B = true, LOAD_LINE_FAST = false, STORE_LINE_FAST = true, HAS_TW = true;
	#ifndef PXS_RESIZE_MINW
#define PXS_RESIZE_MINW 1      /* 8 (at most 64 VGPRs, 20 bytes of scratch per lane, four workgroups per CU where the LDS allows) measured: to_cc 29.9 -> 34.1 ms at C3, 11.4 -> 13.0 at C4 */
#endif
	static constexpr int MAXR = 9, MINW = PXS_RESIZE_MINW;
#ifdef PXS_RESIZE_PTS      /* experiment: a larger tile cap (2880) so that lines of 321 ... 360 points still get 8 per tile: to_cc 11.2 -> 12.45 ms at C4, 4.3 -> 4.8 at C2 (74 VGPRs, more LDS per workgroup), nothing at C3 / C5 */
	static constexpr int PTS = PXS_RESIZE_PTS;
#endif
	const double2* Y; long ldY; double2* Z; long ldZ;
	int g, X1, X2, kmax, nyq; const double2* ph; FastDiv dg;
	int adj;      // transposed padding rule (X1 > X2): conjugate phase, and the Nyquist slot of X2 collects 1/2 of both +-X2/2 bins of X1
	static constexpr int LOADK = 0, STOREK = 0;
	__device__ __forceinline__ const double2* row(const TileC& c, int li) const { return li < c.nl ? Y + ((long)c.outer*g + c.t0 + li)*ldY : nullptr; }
	__device__ __forceinline__ bool decode(int bx, TileC& c) const {
		c.outer = fdiv(bx, dnt); c.t0 = (bx - c.outer*ntile)*T; c.nl = min(T, g - c.t0); return true; }
	__device__ __forceinline__ double2 load(const TileC& c, int li, int e) const {
		if (li >= c.nl) return make_double2(0, 0);
		return bin.Tw > 0 ? Y[(long)c.outer*g*fa.n + bin.off(c.t0 + li, e, g)] : Y[((long)c.outer*g + c.t0 + li)*ldY + e]; }
	template<class AF> __device__ __forceinline__ double2 mid(const TileC& c, int li, int e, AF&& A) const {
		if (li >= c.nl) return make_double2(0, 0);
		const int k1 = c.t0 + li;
		const int jp = g*e + k1;
		const int kap = (2*jp <= X2) ? jp : jp - X2;
		const int ak = kap < 0 ? -kap : kap;
		if ((kmax >= 0 && ak > kmax) || 2*ak > X1) return make_double2(0, 0);
		const int k = kap >= 0 ? kap : kap + X1;
		const uint32_t k2 = fdiv((uint32_t)(k - k1), dg);
		double2 v = A((int)k2);
		if (adj) {
			double2 t = ph ? ph[ak] : make_double2(1, 0);
			if (2*ak == X2 && X1 != X2) {      // both bins are in this line: X1, X2 and hence +-X2/2 are congruent mod g
				const double2 vm = A((int)fdiv((uint32_t)(X1 - ak - k1), dg));
				const double2 r = cadd(cmul(v, cconj(t)), cmul(vm, t));
				return cscale(r, 0.5);
			}
			if (kap >= 0) t.y = -t.y;
			return cmul(v, t);
		}
		if (nyq && 2*ak == X1) v = cscale(v, 0.5);
		if (ph) { double2 t = ph[ak]; if (kap < 0) t.y = -t.y; v = cmul(v, t); }
		return v; }
	template<class VF> __device__ __forceinline__ void store(const TileC& c, int li, int e, VF&& val, double2 w) const {
		if (li >= c.nl) return;
		const long o = bout ? ((long)c.outer*g + c.t0)*fb.n + (long)e*c.nl + li : ((long)c.outer*fb.n + e)*ldZ + c.t0 + li;
		Z[o] = cmul(cconj(val(li, e)), cconj(w)); }
};

// pass 2 of IFFT_M (g points), pointwise product with the |sin| series samples, pass 1 of FFT_M (g points)
// in: Z[outer][k1' < g2][r < g]; out: V[outer][k1'' < g][k1' < g2]
struct StSigma : StageBase {
	static constexpr int SID = 2;
	static constexpr bool TWO = true, INV_A = true, INV_B = false, LOAD_LINE_FAST = false, STORE_LINE_FAST = true, HAS_TW = true;
	static constexpr int MAXR = 9, MINW = 1;
	const double2* Z; long ldZ; double2* V; long ldV; int g, g2; const double2* sigma;
	static constexpr int LOADK = 0, STOREK = 0;
	__device__ __forceinline__ const double2* row(const TileC& c, int li) const { return li < c.nl ? Z + ((long)c.outer*g2 + c.t0 + li)*ldZ : nullptr; }
	__device__ __forceinline__ bool decode(int bx, TileC& c) const {
		c.outer = fdiv(bx, dnt); c.t0 = (bx - c.outer*ntile)*T; c.nl = min(T, g2 - c.t0); return true; }
	__device__ __forceinline__ double2 load(const TileC& c, int li, int e) const {
		if (li >= c.nl) return make_double2(0, 0);
		return bin.Tw > 0 ? Z[(long)c.outer*g2*fa.n + bin.off(c.t0 + li, e, g2)] : Z[((long)c.outer*g2 + c.t0 + li)*ldZ + e]; }
	template<class AF> __device__ __forceinline__ double2 mid(const TileC& c, int li, int e, AF&& A) const {
		if (li >= c.nl) return make_double2(0, 0);
		return cmul(cconj(A(e)), sigma[(c.t0 + li) + g2*e]); }
	template<class VF> __device__ __forceinline__ void store(const TileC& c, int li, int e, VF&& val, double2 w) const {
		if (li >= c.nl) return;
		const long o = bout ? ((long)c.outer*g2 + c.t0)*fb.n + (long)e*c.nl + li : ((long)c.outer*g + e)*ldV + c.t0 + li;
		V[o] = cmul(val(li, e), w); }
};

// last pass of a backward chain (IFFT over r < g for the lines k1 < a, circle index t = k1 + a*k2) + separation of the packed pair
// by reflection symmetry.
// MODE 0: a tile holds TH primary lines and their mirror lines (slots li and li + TH) of one pair;
//         out = leg[(col)*ld + t] * w[t] * scale, rings t < nr_out          (analysis: CC grid, weights)
// MODE 1: a tile holds T/2 pairs x (line, mirror line), slot = 2*pi + which;
//         out = h[t*ld + col] * conj(tab[col]) * scale                      (synthesis: ring-major rows for the ring FFT)
template<int MODE> struct StSplit : StageBase {
	static constexpr int SID = 3 + MODE;
#ifndef PXS_HOST_SIM
	static constexpr int NT = MODE == 1 ? 256 : CH_NT;
#endif
	static constexpr int PF = MODE == 1 ? 512 : 256;      // (per-stage sweep, profiles/r04b_prefetch_ahead.txt: from_cc 16.9 -> 16.3 ms at 512 ... 768)
	static constexpr bool TWO = false, INV_A = true, INV_B = false, LOAD_LINE_FAST = false, STORE_LINE_FAST = true, HAS_TW = false;
	static constexpr int MAXR = 9, MINW = 1;
	const double2* U; long ldU; int a, g, X, mir_c, nr_out, a_odd, ncol, npair;
	double2* out; long ld; const double2* w; const double2* tab; double scale; int TH; FastDiv da;
	long ocstride; int groups; FastDiv dnp, dgr;      // components of a launch: output stride; MODE 0: outer = comp*npair + pair, MODE 1: outer = comp*groups + group
	int self_half;     // self-mirrored output rings get weight 1/2 (adjoint of a mirror extension, which reads them once)
	static constexpr int LOADK = 0, STOREK = 1;
	__device__ __forceinline__ const double2* row(const TileC& c, int li) const {
		int line, pair; slot(c, li, line, pair);
		return line >= 0 ? U + (((long)c.comp*npair + pair)*a + line)*ldU : nullptr; }
	__device__ __forceinline__ int mirror_line(int k) const { int m = a - k - mir_c; if (m >= a) m -= a; if (m < 0) m += a; return m; }
	__device__ __forceinline__ bool decode(int bx, TileC& c) const {
		c.outer = fdiv(bx, dnt); c.t0 = (bx - c.outer*ntile)*(MODE == 0 ? TH : 1); c.nl = T;
		if (MODE == 0) { c.comp = fdiv(c.outer, dnp); c.q0 = c.outer - c.comp*npair; }
		else { c.comp = fdiv(c.outer, dgr); c.q0 = (c.outer - c.comp*groups)*(T/2); }
		if (MODE == 1) { if (mirror_line(c.t0) < c.t0) return false; }      // the tile of the mirror line does this one
		return true; }
	// line and pair (within its component) of LDS slot li (line < 0: unused slot)
	__device__ __forceinline__ void slot(const TileC& c, int li, int& line, int& pair) const {
		if (MODE == 0) {
			pair = c.q0;
			const int prim = c.t0 + (li < TH ? li : li - TH);
			line = -1;
			if (prim < a) {
				const int m = mirror_line(prim);
				if (li < TH) { if (prim <= m) line = prim; }                   // primary lines: the smaller of (line, mirror)
				else if (prim < m) line = m;
			}
		} else {
			pair = c.q0 + (li >> 1);
			const int m = mirror_line(c.t0);
			line = (li & 1) ? (m != c.t0 ? m : -1) : c.t0;
			if (pair >= npair) line = -1;
		}
	}
	__device__ __forceinline__ double2 load(const TileC& c, int li, int e) const {
		int line, pair; slot(c, li, line, pair);
		if (line < 0) return make_double2(0, 0);
		if (MODE == 0 && bin.Tw > 0) return U[((long)c.comp*npair + pair)*a*fa.n + bin.off(line, e, a)];
		return U[(((long)c.comp*npair + pair)*a + line)*ldU + e]; }
	template<class AF> __device__ __forceinline__ double2 mid(const TileC&, int, int, AF&&) const { return make_double2(0, 0); }
	template<class VF> __device__ __forceinline__ void store(const TileC& c, int li, int e, VF&& val, double2) const {
		int line, pair; slot(c, li, line, pair);
		if (line < 0) return;
		const int t = line + a*e;
		if (t >= nr_out) return;
		int tm = X - t - mir_c; if (tm >= X) tm -= X; if (tm < 0) tm += X;
		const double2 z = cconj(val(li, e));
		double2 ev, od;
		if (tm == t) { ev = self_half ? cscale(z, 0.5) : z; od = make_double2(0, 0); }
		else {
			const int ml = mirror_line(line);
			const int lp = (ml == line) ? li : (MODE == 0 ? (li < TH ? li + TH : li - TH) : (li ^ 1));
			const int e2 = (int)fdiv((uint32_t)(tm - ml), da);
			const double2 y = cconj(val(lp, e2));
			ev = make_double2(0.5*(z.x + y.x), 0.5*(z.y + y.y)); od = make_double2(0.5*(z.x - y.x), 0.5*(z.y - y.y));
		}
		const int ca = 2*pair;
		double2 va = a_odd ? od : ev, vb = a_odd ? ev : od;
		double2* oc = out + (long)c.comp*ocstride;
		if (MODE == 0) {
			const double f = scale*(w ? w[t].x : 1.0);
			oc[(long)ca*ld + t] = cscale(va, f);
			if (ca + 1 < ncol) oc[(long)(ca + 1)*ld + t] = cscale(vb, f);
		} else {
			const double f = scale*(w ? w[t].x : 1.0);
			oc[(long)t*ld + ca] = cscale(cmul(va, cconj(tab[ca])), f);
			if (ca + 1 < ncol) oc[(long)t*ld + ca + 1] = cscale(cmul(vb, cconj(tab[ca + 1])), f);
		}
	}
};

// ---------------------------------------------------------------------------------------------------------------
// ring FFTs
// ---------------------------------------------------------------------------------------------------------------

// MA1: two real rings as one complex line z = ring(2q) + i ring(2q+1); pixel x = b*j1 + j2, line = j2, a-point FFT over j1
struct StRingA1 : StageBase {
	static constexpr int SID = 5;
	// MA1 reads the map in runs of T reals: 16 lines make whole 128-byte lines.  Where 2560 points hold fewer (C3: 180-point lines, T = 14) the
	// tile may grow to 2880 points (profiles/r04b_tile2880_ab.txt: map2leg of the Q/U pair 14.74 -> 13.76 ms, enmap.fft 17.65 -> 16.78 ms;
	// 2880-point tiles for every stage: nothing elsewhere at C3, +2 % at C4)
	static constexpr int PTS = 2880;
	static constexpr bool TWO = false, INV_A = false, INV_B = false, LOAD_LINE_FAST = true, STORE_LINE_FAST = true, HAS_TW = true;
	static constexpr int MAXR = 8, MINW = 8;
	MapAddr m; int b, npair; double2* Y; long ldY; FastDiv dnp;
	static constexpr int LOADK = 1, STOREK = 0;
	__device__ __forceinline__ const double2* row(const TileC&, int) const { return nullptr; }
	__device__ __forceinline__ bool decode(int bx, TileC& c) const {
		c.outer = fdiv(bx, dnt); c.t0 = (bx - c.outer*ntile)*T; c.nl = min(T, b - c.t0);
		c.comp = fdiv(c.outer, dnp); c.q0 = c.outer - c.comp*npair; return true; }
	__device__ __forceinline__ double2 load(const TileC& c, int li, int e) const {
		if (li >= c.nl) return make_double2(0, 0);
		const int x = b*e + c.t0 + li;
		const int bi = (int)fdiv(c.comp, m.dncb);
		const long o = bi*m.bstride + (c.comp - bi*m.ncb)*m.cstride + m.off0 + (2L*c.q0)*m.rstride + x*m.pstride;
		const double re = rd_real(m.ptr, m.dtype, o).x;
		const double im = (2*c.q0 + 1 < m.nring) ? rd_real(m.ptr, m.dtype, o + m.rstride).x : 0.0;
		return make_double2(re, im); }
	template<class AF> __device__ __forceinline__ double2 mid(const TileC&, int, int, AF&&) const { return make_double2(0, 0); }
	template<class VF> __device__ __forceinline__ void store(const TileC& c, int li, int e, VF&& val, double2 w) const {
		if (li >= c.nl) return;
		Y[((long)c.outer*fa.n + e)*ldY + c.t0 + li] = cmul(val(li, e), w); }
};

// MA2: b-point FFT over j2 for the lines k1 and a - k1 of T/2 ring pairs; bin k = k1 + a*k2 <= mmax is unpacked with its
// partner bin nphi - k (in the mirror line) into the spectra of the two rings and written as leg[m][2q], leg[m][2q+1]
struct StRingA2 : StageBase {
	static constexpr int SID = 6;
	static constexpr int PF = 512;      // (map2leg 21.1 -> 20.75 ms against 256)
	static constexpr bool TWO = false, INV_A = false, INV_B = false, LOAD_LINE_FAST = false, STORE_LINE_FAST = true, HAS_TW = false;
	static constexpr int MAXR = 8, MINW = 8;
	const double2* Y; long ldY; int a, X, npair, groups, nring, mmax; double2* leg; long ldleg; int nm; const double2* tab; double scale; FastDiv da, dgr;
	static constexpr int LOADK = 0, STOREK = 1;
	__device__ __forceinline__ const double2* row(const TileC& c, int li) const {
		int line, q; slot(c, li, line, q);
		return line >= 0 ? Y + (((long)c.comp*npair + q)*a + line)*ldY : nullptr; }
	__device__ __forceinline__ int mirror_line(int k) const { return k == 0 ? 0 : a - k; }
	__device__ __forceinline__ bool decode(int bx, TileC& c) const {
		c.outer = fdiv(bx, dnt); c.t0 = bx - c.outer*ntile; c.nl = T;
		c.comp = fdiv(c.outer, dgr); c.q0 = (c.outer - c.comp*groups)*(T/2);
		return mirror_line(c.t0) >= c.t0; }
	__device__ __forceinline__ void slot(const TileC& c, int li, int& line, int& q) const {
		q = c.q0 + (li >> 1);
		const int ml = mirror_line(c.t0);
		line = (li & 1) ? (ml != c.t0 ? ml : -1) : c.t0;
		if (q >= npair) line = -1;
	}
	__device__ __forceinline__ double2 load(const TileC& c, int li, int e) const {
		int line, q; slot(c, li, line, q);
		if (line < 0) return make_double2(0, 0);
		return Y[(((long)c.comp*npair + q)*a + line)*ldY + e]; }
	template<class AF> __device__ __forceinline__ double2 mid(const TileC&, int, int, AF&&) const { return make_double2(0, 0); }
	template<class VF> __device__ __forceinline__ void store(const TileC& c, int li, int e, VF&& val, double2) const {
		int line, q; slot(c, li, line, q);
		if (line < 0) return;
		const int k = line + a*e;
		if (k > mmax) return;
		const double2 zp = val(li, e);
		double2 zm;
		if (k == 0) zm = zp;
		else {
			const int kp = X - k, ml = mirror_line(line);
			const int lp = (ml == line) ? li : (li ^ 1);
			zm = val(lp, (int)fdiv((uint32_t)(kp - ml), da));
		}
		// X_a = (Z[k] + conj Z[n-k])/2, X_b = -i (Z[k] - conj Z[n-k])/2
		double2 xa = make_double2(0.5*(zp.x + zm.x), 0.5*(zp.y - zm.y));
		const double2 d = make_double2(zp.x - zm.x, zp.y + zm.y);
		double2 xb = make_double2(0.5*d.y, -0.5*d.x);
		const double2 t = tab ? tab[k] : make_double2(1, 0);
		double2* o = leg + ((long)c.comp*nm + k)*ldleg + 2*q;
		o[0] = cscale(cmul(xa, t), scale);
		if (2*q + 1 < nring) o[1] = cscale(cmul(xb, t), scale);
	}
};

// MS1: Hermitian pair load from h[comp][ring][m]: bin k = b*j1 + j2, line = j2, backward a-point transform over j1
struct StRingS1 : StageBase {
	static constexpr int SID = 7;
	static constexpr bool TWO = false, INV_A = true, INV_B = false, LOAD_LINE_FAST = true, STORE_LINE_FAST = true, HAS_TW = true;
	static constexpr int MAXR = 8, MINW = 8;
	const double2* h; long ldh, hcomp; int b, X, npair, nring, mmax; double2* Y; long ldY; FastDiv dnp;      // hcomp: rows of h per component
	static constexpr int LOADK = 1, STOREK = 0;
	__device__ __forceinline__ const double2* row(const TileC&, int) const { return nullptr; }
	__device__ __forceinline__ bool decode(int bx, TileC& c) const {
		c.outer = fdiv(bx, dnt); c.t0 = (bx - c.outer*ntile)*T; c.nl = min(T, b - c.t0);
		c.comp = fdiv(c.outer, dnp); c.q0 = c.outer - c.comp*npair; return true; }
	__device__ __forceinline__ double2 load(const TileC& c, int li, int e) const {
		if (li >= c.nl) return make_double2(0, 0);
		const int k = b*e + c.t0 + li;
		int m; bool cj;
		if (k <= mmax) { m = k; cj = false; }
		else if (X - k <= mmax) { m = X - k; cj = true; }
		else return make_double2(0, 0);
		const double2* r = h + ((long)c.comp*hcomp + 2*c.q0)*ldh + m;
		double2 ha = r[0];
		double2 hb = (2*c.q0 + 1 < nring) ? r[ldh] : make_double2(0, 0);
		if (m == 0) { ha.y = 0; hb.y = 0; }
		if (cj) { ha.y = -ha.y; hb.y = -hb.y; }
		return make_double2(ha.x - hb.y, ha.y + hb.x); }
	template<class AF> __device__ __forceinline__ double2 mid(const TileC&, int, int, AF&&) const { return make_double2(0, 0); }
	template<class VF> __device__ __forceinline__ void store(const TileC& c, int li, int e, VF&& val, double2 w) const {
		if (li >= c.nl) return;
		const long o = bout ? ((long)c.outer*b + c.t0)*fa.n + (long)e*c.nl + li : ((long)c.outer*fa.n + e)*ldY + c.t0 + li;
		Y[o] = cmul(cconj(val(li, e)), cconj(w)); }
};

// MS2: backward b-point transform over j2 for the lines k1; pixel x = k1 + a*k2: real part -> ring 2q, imaginary part -> ring 2q+1
struct StRingS2 : StageBase {
	static constexpr int SID = 8;
	static constexpr bool TWO = false, INV_A = true, INV_B = false, LOAD_LINE_FAST = false, STORE_LINE_FAST = true, HAS_TW = false;
	static constexpr int MAXR = 8, MINW = 8;
	const double2* Y; long ldY; int a, npair; MapAddr m; FastDiv dnp;
	static constexpr int LOADK = 0, STOREK = 0;
	__device__ __forceinline__ const double2* row(const TileC& c, int li) const { return li < c.nl ? Y + ((long)c.outer*a + c.t0 + li)*ldY : nullptr; }
	__device__ __forceinline__ bool decode(int bx, TileC& c) const {
		c.outer = fdiv(bx, dnt); c.t0 = (bx - c.outer*ntile)*T; c.nl = min(T, a - c.t0);
		c.comp = fdiv(c.outer, dnp); c.q0 = c.outer - c.comp*npair; return true; }
	__device__ __forceinline__ double2 load(const TileC& c, int li, int e) const {
		if (li >= c.nl) return make_double2(0, 0);
		return bin.Tw > 0 ? Y[(long)c.outer*a*fa.n + bin.off(c.t0 + li, e, a)] : Y[((long)c.outer*a + c.t0 + li)*ldY + e]; }
	template<class AF> __device__ __forceinline__ double2 mid(const TileC&, int, int, AF&&) const { return make_double2(0, 0); }
	template<class VF> __device__ __forceinline__ void store(const TileC& c, int li, int e, VF&& val, double2) const {
		if (li >= c.nl) return;
		const int x = c.t0 + li + a*e;
		const double2 v = val(li, e);                     // conj of the backward transform: the imaginary part flips sign
		const int bi = (int)fdiv(c.comp, m.dncb);
		const long o = bi*m.bstride + (c.comp - bi*m.ncb)*m.cstride + m.off0 + (2L*c.q0)*m.rstride + x*m.pstride;
		wr_real(m.ptr, m.dtype, o, v.x);
		if (2*c.q0 + 1 < m.nring) wr_real(m.ptr, m.dtype, o + m.rstride, -v.y);
	}
};

// last pass of a transform of the 2-D FFTs (FftChain::fft2_real / fft2_c2c), with the transpose: b-point transform over j2 for line k1
// of T consecutive outer lines kx (columns of the spectrum, or rows of the input); writes out[ky = k1 + a k2][kx] (kx fastest: runs of T
// points; row stride ldo) and, with herm (real input: kx runs over the half spectrum), for 0 < kx < nx - kx the Hermitian image
// out[(ny - ky) % ny][nx - kx] = conj
struct StColOut : StageBase {
	static constexpr int SID = 9;
	static constexpr bool R7 = false;
	static constexpr bool TWO = false, INV_A = false, INV_B = false, LOAD_LINE_FAST = false, STORE_LINE_FAST = true, HAS_TW = false;
	static constexpr int MAXR = 9, MINW = 1;
	const double2* Y; long ldY; int a, nm, ny, nx, groups, conj_out, herm; double2* out; long ldo, ocomp; double scale; FastDiv dgr;
	static constexpr int LOADK = 0, STOREK = 0;
	__device__ __forceinline__ const double2* row(const TileC& c, int li) const { return c.q0 + li < nm ? Y + (((long)c.comp*nm + c.q0 + li)*a + c.t0)*ldY : nullptr; }
	__device__ __forceinline__ bool decode(int bx, TileC& c) const {
		c.outer = fdiv(bx, dnt); c.t0 = bx - c.outer*ntile; c.nl = T;
		c.comp = fdiv(c.outer, dgr); c.q0 = (c.outer - c.comp*groups)*T; return true; }
	__device__ __forceinline__ double2 load(const TileC& c, int li, int e) const {
		const int kx = c.q0 + li;
		if (kx >= nm) return make_double2(0, 0);
		return Y[(((long)c.comp*nm + kx)*a + c.t0)*ldY + e]; }
	template<class AF> __device__ __forceinline__ double2 mid(const TileC&, int, int, AF&&) const { return make_double2(0, 0); }
	template<class VF> __device__ __forceinline__ void store(const TileC& c, int li, int e, VF&& val, double2) const {
		const int kx = c.q0 + li;
		if (kx >= nm) return;
		const int ky = c.t0 + a*e;
		double2 v = cscale(val(li, e), scale);
		double2* oc = out + (long)c.comp*ocomp;
		// (backward transform = conjugate of the forward one of the conjugated input)
		oc[(long)ky*ldo + kx] = conj_out ? cconj(v) : v;
		if (herm && kx > 0 && 2*kx < nx) oc[(long)(ky == 0 ? 0 : ny - ky)*ldo + (nx - kx)] = conj_out ? v : cconj(v);
	}
};

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static bool smooth235(long n) { if (n < 1) return false; for (int p : {2, 3, 5}) while (n % p == 0) n /= p; return n == 1; }
bool FftChain::sub_ok(long n) { return n >= 2 && n <= CH_NMAX && smooth235(n); }

static LdsFft mk(FftContext* fc, long n, int maxr = 9) {
	LdsFft f; memset(&f, 0, sizeof(f));
	if (n <= 0) return f;
	auto v = fc->view(n, maxr);
#ifdef PXS_LAB
	static const int nofft = [] { const char* e = getenv("PXS_CH_NOFFT"); return e ? atoi(e) : 0; }();     // lab builds only: timing experiments (wrong results)
#else
	constexpr int nofft = 0;
#endif
	static const int nspad = [] { const char* e = lab_getenv("PXS_CH_NS_PAD"); return e ? atoi(e) : 0; }();     // experiments: line stride (n | 1) + pad
	f.n = v.n; f.nfac = nofft ? 0 : v.nfac; f.ns = v.ns + nspad; f.generic = v.generic; f.pass = (const PassDesc*)v.pass; f.perm = v.perm; f.tw = v.tw; f.dn = make_fastdiv((uint32_t)n);
	return f;
}

// W_X^{li e}, e < n, li < T, laid out [e][li] (see StageBase)
const double2* FftChain::small_tw(long X, int n, int T) {
#ifdef PXS_LAB
	{ static const int notw = [] { const char* e = getenv("PXS_CH_NOTW"); return e ? atoi(e) : 0; }(); if (notw) return nullptr; }     // lab builds only: timing experiments (wrong results)
#endif
	std::lock_guard<std::mutex> g(mu_);
	auto key = std::make_tuple(X, n, T);
	auto it = stw_.find(key);
	if (it != stw_.end()) return it->second.as<double2>();
	std::vector<double2> t((size_t)n*T);
	const long double tp = 6.283185307179586476925286766559L;
	for (int e = 0; e < n; e++) for (int li = 0; li < T; li++) {
		const long double ang = tp*(long double)(((long)li*e) % X)/(long double)X;
		t[(size_t)e*T + li] = make_double2((double)cosl(ang), (double)(-sinl(ang)));
	}
	stw_[key] = upload(t);
	return stw_[key].as<double2>();
}

// lines per tile: as many as fit CH_TILE_PTS, in multiples of `mult`
// tab_pts >= 0 (ring stages, which run at <= 64 VGPRs) and PXS_RING_TILE_KB = k > 0: shrink the tile until tile + tab_pts table
// entries fit k KiB of LDS (k = 39.5: four workgroups per CU).  Off by default -- measured at C3 / C4 / C2: ring FFT stages
// 47.4 / 60.6 / 2.85 ms with 39.5 KiB tiles against 46.4 / 60.8 / 2.83 ms with the full 2560-point tiles.
static int tile_lines(long n_a, long n_b, long nlines, int mult, long tab_pts = -1) {
	const long n = std::max(n_a, n_b);
	long T = CH_TILE_PTS / n;
	if (T >= mult) T -= T % mult;
	if (T < 1) T = 1;
	const long cap = ((nlines + mult - 1)/mult)*mult;
	if (T > cap) T = cap;
	static const long limit = [] { const char* e = lab_getenv("PXS_RING_TILE_KB"); return e ? (long)(atof(e)*1024) : 0L; }();
	if (tab_pts >= 0 && limit > 0)
		while (T > mult && (long)sizeof(double2)*(tab_pts + T*(n | 1) + 2) > limit) T -= mult;
	return (int)T;
}
// tiles of T consecutive lines; X > 0: the stage applies the four-step twiddle of a length-X transform to its output
template<class S> void FftChain::set_tiles(S& s, int T, long nlines, long X) {
	s.T = T; s.ntile = (int)((nlines + T - 1)/T); s.dT = make_fastdiv((uint32_t)T); s.dnt = make_fastdiv((uint32_t)s.ntile);
	s.dna = make_fastdiv((uint32_t)s.fa.n); s.dnb = make_fastdiv((uint32_t)std::max(1, s.fb.n));
	s.btw = nullptr; s.tws = nullptr;
	if (X > 0) { s.tws = small_tw(X, S::TWO ? s.fb.n : s.fa.n, T); s.btw = s.tws ? fc_->twiddle_table(X) : nullptr; }
}
// blocked intermediates: OFF by default.  Measured (tools/chain_lab.py, same box): C3 chain stages 106.9 ms with rows, 108.1 ms
// blocked; C4 26.6 / 26.6 -- the 128-byte runs of the intermediates are not what holds the chain kernels at ~3 TB/s (a bare
// load-tile / LDS / store-tile kernel on contiguous 40 KB tiles reaches 5.3 TB/s with 2, 3 or 4 workgroups per CU, and an LDS-DMA
// double-buffered persistent variant of it only 4.6: tools/dma_skel.hip).  PXS_CH_BLOCKED=1 turns the layout on for experiments.
static bool blocked_on() { static const bool on = [] { const char* e = lab_getenv("PXS_CH_BLOCKED"); return e ? atoi(e) != 0 : false; }(); return on; }
// reader side of a blocked intermediate: the writer made chunks of Tw lines out of na, the reader's tile has T rows
static BlkIn mk_blk(int Tw, long na, int T) {
	BlkIn b; memset(&b, 0, sizeof(b));
	if (!blocked_on()) return b;
	b.Tw = Tw; b.nBf = (int)(na/Tw); b.wL = (int)(na - (long)b.nBf*Tw);
	b.dTTw = make_fastdiv((uint32_t)(T*Tw)); b.dTw = make_fastdiv((uint32_t)Tw); b.dwL = make_fastdiv((uint32_t)std::max(b.wL, 1));
	return b;
}
// ---- second-generation kernel: transform plans and launch -------------------------------------------------------------------

static const int F2_RADICES[] = {2, 3, 4, 5, 6, 8, 9};
// n as a product of at most three register radices: fewest passes, then the smallest largest radix; ascending order (the first
// pass works on the longest rows, which wastes the least padding).  maxfirst: cap on the first radix (passes that pull their inputs

template<class S> static void launch_stage(const S& s, long nblk, hipStream_t st) {
	if (nblk <= 0) return;
	PXS_REQUIRE((long)s.T*std::max(s.fa.n, s.fb.n) <= S::PTS, "internal: chain tile too large");
	PXS_REQUIRE(nblk < (1L << 31), "internal: chain grid too large");
	size_t sh = sizeof(double2)*((size_t)s.fa.n + (S::TWO && s.fb.n == s.fa.n ? 0 : s.fb.n) + (S::HAS_TW ? std::max(s.fa.n, s.fb.n) : 0) + (size_t)s.T*std::max(s.fa.ns, s.fb.ns) + 2);
	{ static const size_t pad = [] { const char* e = lab_getenv("PXS_CH_LDS_PAD"); return e ? (size_t)atol(e) : (size_t)0; }(); sh += pad; }   // occupancy experiments
	{ static const int pf = [] {      // PXS_CH_PF: every stage; PXS_CH_PF_SID<k>: the stage with S::SID = k
		const std::string name = "PXS_CH_PF_SID" + std::to_string(S::SID);
		const char* e1 = lab_getenv(name.c_str()); const char* e = lab_getenv("PXS_CH_PF");
		return e1 ? atoi(e1) : (e ? atoi(e) : S::PF); }(); const_cast<S&>(s).pf = pf; }
	if (getenv("PXS_CHAIN_VERBOSE")) {
		static std::mutex mu; static std::set<std::tuple<int, int, int, int>> seen; std::lock_guard<std::mutex> g(mu);
		if (seen.insert(std::make_tuple(S::SID, s.fa.n, s.fb.n, s.T)).second)
			fprintf(stderr, "[pxsht] chain stage %d: na=%d nb=%d T=%d LDS %.1f KiB -> %d WG/CU by LDS, %ld workgroups\n", S::SID, s.fa.n, s.fb.n, s.T, sh/1024.0, (int)((160*1024)/sh), nblk);
	}
#ifndef PXS_HOST_SIM
#ifdef PXS_CH_NT2      /* experiment builds: stages whose bit SID is set in PXS_CH_NT2_MASK run with PXS_CH_NT2 threads per workgroup */
	static const long mask2 = [] { const char* e = lab_getenv("PXS_CH_NT2_MASK"); return e ? strtol(e, nullptr, 0) : 0L; }();
	if ((mask2 >> S::SID) & 1) {
		constexpr int NT2 = PXS_CH_NT2, MAXE2 = (S::PTS + NT2 - 1)/NT2;
		static const bool once2 = [] { (void)hipFuncSetAttribute((const void*)chain_kernel<S, NT2, MAXE2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 256); return true; }();
		(void)once2;
		hipLaunchKernelGGL((chain_kernel<S, NT2, MAXE2>), dim3((unsigned)nblk), dim3(NT2), sh, st, s);
		return;
	}
#endif
	static const bool once = [] { (void)hipFuncSetAttribute((const void*)chain_kernel<S, S::NT, (S::PTS + S::NT - 1)/S::NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 256); return true; }();
	(void)once;
#endif
	hipLaunchKernelGGL((chain_kernel<S, S::NT, (S::PTS + S::NT - 1)/S::NT>), dim3((unsigned)nblk), dim3(S::NT), sh, st, s);
}

// lines per tile of a stage
template<class S> int FftChain::tile_lines_for(long n_a, long n_b, long nlines, int mult, long tab_pts) {
	int T = tile_lines(n_a, n_b, nlines, mult, tab_pts);
	// a stage with a larger cap takes it only to reach `mult` lines (whole 128-byte lines on its strided side)
	if (S::PTS > CH_TILE_PTS && T < mult && (long)mult*std::max(n_a, n_b) <= S::PTS && nlines >= mult) T = mult;
	return T;
}
template<class S> void FftChain::launch_any(S& s, long nblk, hipStream_t st) { launch_stage(s, nblk, st); }

// balanced split n = a*b with both factors usable
static bool split_balanced(long n, Split& s, long amax = 320) {
	long best = 0; double bestc = 1e300;
	for (long a = 2; a <= 1024 && a <= n/2; a++) if (n % a == 0) {
		const long b = n/a;
		if (!FftChain::sub_ok(a) || !FftChain::sub_ok(b)) continue;
		double c = std::fabs(std::log((double)a/(double)b));
		if (a > amax) c += 10;
		if (b % 8) c += 0.3;
		if (c < bestc) { bestc = c; best = a; }
	}
	if (!best) return false;
	s.a = best; s.b = n/best; return true;
}

bool FftChain::plan_rings(long nphi) {
	nphi_ = nphi; ra_ = Split(); rs_ = Split();
	Split s;
	if (!split_balanced(nphi, s)) return false;
	// analysis: pass 1 tiles are 16 lines (128-byte runs of f64) x a points -> the smaller factor first;
	// synthesis: pass 2 tiles are 16 lines x b points -> the smaller factor last
	ra_.a = std::min(s.a, s.b); ra_.b = nphi/ra_.a;
	rs_.b = std::min(s.a, s.b); rs_.a = nphi/rs_.b;
	// The map side of the ring stages moves 8-byte reals in runs of T lines.  Synthesis (MS2 WRITES the map: a partial 128-byte
	// line is a read-modify-write): the first factor a a multiple of 16 and T a multiple of 16 (b <= 160) make every run whole,
	// aligned lines; the largest such a <= 320 (8 lines of a points still fill a tile in MS1).  Analysis (MA1 reads the map): b the
	// largest multiple of 16 up to 256.  Measured with tools/chain_lab.py (ms, h2map / map2leg of one component): C3 43200 =
	// 240 x 180 -> 320 x 135: 8.13 -> 7.04 (288 x 150: 7.48, 400 x 108: 7.33); C4 10800 = 120 x 90 -> 240 x 45: 3.88 -> 3.45 per 8
	// maps (80 x 135: 3.52, 144 x 75: 3.56); analysis 90 x 120 -> 45 x 240: 3.78 -> 3.45 (C3 stays 180 x 240; 150 x 288: 7.77 vs 7.54).
	for (long a = 320; a >= 16; a -= 16) if (nphi % a == 0 && nphi/a <= 160 && nphi/a >= 2 && sub_ok(a) && sub_ok(nphi/a)) { rs_.a = a; rs_.b = nphi/a; break; }
	for (long b = 256; b >= 16; b -= 16) if (nphi % b == 0 && nphi/b >= 2 && nphi/b <= 320 && sub_ok(b) && sub_ok(nphi/b)) { ra_.b = b; ra_.a = nphi/b; break; }
	{	// experiments: force the first factor of the analysis / synthesis split
		const char* ea = lab_getenv("PXS_RING_A_ANA"); const char* es = lab_getenv("PXS_RING_A_SYN");
		if (ea && atol(ea) > 1 && nphi % atol(ea) == 0 && sub_ok(atol(ea)) && sub_ok(nphi/atol(ea))) { ra_.a = atol(ea); ra_.b = nphi/ra_.a; }
		if (es && atol(es) > 1 && nphi % atol(es) == 0 && sub_ok(atol(es)) && sub_ok(nphi/atol(es))) { rs_.a = atol(es); rs_.b = nphi/rs_.a; }
	}
	return true;
}

static long smooth_at_least(long n, bool even_product_with = false, long g = 1) {
	for (long m = std::max<long>(n, 2);; m++) if (smooth235(m) && (!even_product_with || ((m*g) % 2 == 0))) return m;
}

// lengths the theta stages (StFirst, StResize, StSigma, StSplit: lds_fft with MAXR = 9) transform: radix 7 next to 2, 3, 5
static bool smooth2357(long n) { if (n < 1) return false; for (int p : {2, 3, 5, 7}) while (n % p == 0) n /= p; return n == 1; }
static bool sub_ok7(long n) { return n >= 2 && n <= CH_NMAX && smooth2357(n); }
// ducc0's good_size_complex: the smallest 2^a 3^b 5^c 7^d 11^e >= n.  Its analysis_2d / synthesis_2d run the Legendre stage on a CC grid of
// good_size_complex(lmax + 1) + 1 rings, i.e. a circle of N_cc = 2 good_size_complex(lmax + 1) points.
bool FftChain::sub_ok_theta(long n) { return sub_ok7(n); }
long FftChain::ducc_ncc(int lmax) {
	for (long n = std::max<long>(lmax + 1, 1);; n++) { long r = n; for (int p : {2, 3, 5, 7, 11}) while (r % p == 0) r /= p; if (r == 1) return 2*n; }
}

// How full the tiles of a stage with lines of n points are (tiles hold a multiple of 8 lines, at most CH_TILE_PTS points); lines too long
// for 8 of them (strided side in runs below 128 bytes) count as a quarter less.  Sweeps of the shared moduli with tools/gpu_sweep.sh
// (profiles/r04b_theta_modulus_sweep.txt) follow sum(points of a stage / fill) to ~5 %: C3 analysis 29.8 ms (g = 160, 144, 288, 320) against
// 34-40 (g = 180, 192, 240: half-empty tiles; 480: five-line tiles), C5 3.6-3.75 (72, 96, 288) against 3.9-4.0 (108, 144, 216, 432).
static double tile_fill(long n) {
	long T = CH_TILE_PTS / n; if (T >= 8) T -= T % 8; if (T < 1) T = 1;
	return (double)(T*n)/(double)CH_TILE_PTS*(T >= 8 ? 1.0 : 0.8);
}
// among moduli whose tiles fill alike: 64 ... 160 (16 or more lines per tile in the g-point stages) before larger or much smaller ones
static double modulus_pref(long g) { return 1.0 + 0.02*std::max(0.0, std::log2((double)g/160.0)) + 0.04*std::max(0.0, std::log2(64.0/(double)g)); }
ThetaPlan FftChain::plan_theta(long N, int lmax) {
	ThetaPlan best; double bestc = 1e300;
	static const long gforce = [] { const char* e = lab_getenv("PXS_THETA_G"); return e ? atol(e) : 0L; }();     // experiments: force the shared modulus
	static const bool ducc_size = [] { const char* e = lab_getenv("PXS_THETA_DUCC_NCC"); return e ? atoi(e) != 0 : true; }();     // 0: the planner's own N_cc only (rounds 1-3)
	const long Nd = ducc_ncc(lmax);
	auto consider = [&](long g, long ac) {
		ThetaPlan t; t.N = N; t.g = g; t.bN = N/g; t.ac = ac; t.Ncc = g*ac;
		t.g2 = smooth_at_least((N + 2L*lmax + 2 + g - 1)/g); t.M = g*t.g2;
		if ((t.Ncc & 1) || !sub_ok(t.g2) || !sub_ok7(t.ac)) return;
		// synthesis split: gs divides both Ncc and N; RS1 and RS3 run gs-point lines, RS2 lines of max(bs, aNs) points
		long gs_best = 0; double gsc = 1e300;
		static const long gsforce = [] { const char* e = lab_getenv("PXS_THETA_GS"); return e ? atol(e) : 0L; }();     // experiments: force the modulus of the synthesis chain
		for (long gs = 2; gs <= 1024; gs++) if (t.Ncc % gs == 0 && N % gs == 0 && sub_ok7(gs) && sub_ok7(t.Ncc/gs) && sub_ok7(N/gs) && (!gsforce || gs == gsforce)) {
			const double c = ((2.0*t.Ncc + 1.5*N)/tile_fill(gs) + (double)(t.Ncc + N)/tile_fill(std::max(t.Ncc/gs, N/gs)))*modulus_pref(gs);
			if (c < gsc) { gsc = c; gs_best = gs; }
		}
		if (!gs_best) return;
		t.gs = gs_best; t.bs = t.Ncc/gs_best; t.aNs = N/gs_best;
		// cost: a CC ring costs far more (Legendre stage) than a point of FFT traffic; then the points the stages of the default analysis
		// (the fine-CC form: middle circle of 2 N_cc points) and of the synthesis move, each over the fill of its tiles, with a slight
		// preference among equals for moduli of 64 ... 160 (modulus_pref).
		// ducc0's own N_cc wins whenever some g realises it: the fine-CC form of the analysis then cuts the theta spectrum of a map
		// that is not band-limited where ducc0 does (sht.hip, ana_set), and it is the smallest size ducc0 considers good.
		const double lmin = 2.0*lmax + 2, M2 = 2.0*t.Ncc;
		const double ana = (2.0*N + 2.0*M2 + 1.5*t.Ncc)/tile_fill(g) + (N + M2)/tile_fill(std::max(t.bN, 2*t.ac)) + (M2 + t.Ncc)/tile_fill(2*t.ac);
		double c = 60.0*(t.Ncc - lmin)/lmin + (ana*modulus_pref(g) + gsc)/(3.0*N + 4.0*M2 + 2.5*t.Ncc + 3.0*t.Ncc + 2.5*N);
		if (ducc_size && t.Ncc == Nd) c -= 1000.0;
		if (c < bestc) { bestc = c; best = t; best.ok = true; }
	};
	for (long g = 2; g <= 1024 && g <= N/2; g++) if (N % g == 0 && sub_ok7(g) && sub_ok7(N/g) && (!gforce || g == gforce)) {
		consider(g, smooth_at_least((2L*lmax + 2 + g - 1)/g, true, g));
		if (ducc_size && Nd % g == 0) consider(g, Nd/g);
	}
	return best;
}


// ring pairs per pass of the ring-FFT stages: PXS_RING_CHUNK_MB > 0 bounds the intermediate of a pass (experiment: keeping it in
// the 256 MB memory-side cache between the two kernels of a pass); 0 = all pairs at once.  Measured at C3 (ring FFT ms per round
// trip): all at once 47.3, 2048 MB 46.7, 512 MB 47.6, 200 MB 50.9, 128 MB 49.3, 64 MB 55.4 -- the cache does not pay for the
// shorter launches.
static long ring_chunk(long npair, long bytes_per_pair, int mult) {
	static const long mb = [] { const char* e = getenv("PXS_RING_CHUNK_MB"); return e ? atol(e) : 0L; }();
	if (mb <= 0) return npair;
	long q = std::max<long>(mult, ((mb << 20)/std::max<long>(bytes_per_pair, 1))/mult*mult);
	return std::min(q, npair);
}

void FftChain::map2leg(hipStream_t st, const MapDesc& m, int nc, int mmax, double2* leg, long ldleg, const double2* tab, double scale) {
	PXS_REQUIRE(rings_ok() && m.nphi == nphi_, "internal: ring chain not planned");
	const long npair_all = (m.nring + 1)/2, a = ra_.a, b = ra_.b, ldY = pad8(b);
	int T2 = tile_lines_for<StRingA2>(b, 0, 2*npair_all, 8, b); if (T2 < 2) T2 = 2; T2 -= T2 % 2;
	const long qchunk = ring_chunk(npair_all, (long)sizeof(double2)*nc*a*ldY, T2/2);
	s1_.ensure(sizeof(double2)*(size_t)nc*qchunk*a*ldY);
	for (long q_lo = 0; q_lo < npair_all; q_lo += qchunk) {
		const long npair = std::min(qchunk, npair_all - q_lo);
		const int nring = m.nring - (int)(2*q_lo);           // rings from the first pair of this pass on (the odd-last-ring tests are local)
		{	StRingA1 s; memset(&s, 0, sizeof(s));
			s.fa = mk(fc_, a, 8); s.fb = mk(fc_, 0, 8);
			s.m = map_addr(m); s.m.off0 += 2*q_lo*m.ring_stride; s.m.nring = nring;
			s.b = (int)b; s.npair = (int)npair; s.Y = s1_.as<double2>(); s.ldY = ldY; s.dnp = make_fastdiv((uint32_t)npair);
			set_tiles(s, tile_lines_for<StRingA1>(a, 0, b, 16, 2*a), b, nphi_);
			launch_any(s, (long)nc*npair*s.ntile, st);
		}
		{	StRingA2 s; memset(&s, 0, sizeof(s));
			s.fa = mk(fc_, b, 8); s.fb = mk(fc_, 0, 8);
			const int T = T2;
			set_tiles(s, T, a*T, 0);          // one tile per line: ntile = a
			s.Y = s1_.as<double2>(); s.ldY = ldY; s.a = (int)a; s.X = (int)nphi_; s.npair = (int)npair; s.nring = nring; s.mmax = mmax;
			s.groups = (int)((npair + T/2 - 1)/(T/2)); s.da = make_fastdiv((uint32_t)a); s.dgr = make_fastdiv((uint32_t)s.groups);
			s.leg = leg + 2*q_lo; s.ldleg = ldleg; s.nm = mmax + 1; s.tab = tab; s.scale = scale;
			launch_any(s, (long)nc*s.groups*a, st);
		}
	}
	PXS_HIP(hipGetLastError());
}

void FftChain::h2map(hipStream_t st, const double2* h, long ldh, const MapDesc& m, int nc, int mmax, long hcomp) {
	PXS_REQUIRE(rings_ok() && m.nphi == nphi_, "internal: ring chain not planned");
	if (line_h2map(st, h, ldh, m, nc, mmax, hcomp)) return;      // (ring lengths compiled into ringline.hip: one kernel, no intermediate)
	const long npair_all = (m.nring + 1)/2, a = rs_.a, b = rs_.b, ldY = pad8(b);
	const long qchunk = ring_chunk(npair_all, (long)sizeof(double2)*nc*a*ldY, 1);
	const int T1 = tile_lines_for<StRingS1>(a, 0, b, 8, 2*a), T2 = tile_lines_for<StRingS2>(b, 0, a, 16, b);
	s1_.ensure(sizeof(double2)*(size_t)nc*qchunk*a*ldY);
	for (long q_lo = 0; q_lo < npair_all; q_lo += qchunk) {
		const long npair = std::min(qchunk, npair_all - q_lo);
		const int nring = m.nring - (int)(2*q_lo);
		{	StRingS1 s; memset(&s, 0, sizeof(s));
			s.fa = mk(fc_, a, 8); s.fb = mk(fc_, 0, 8);
			s.h = h + 2*q_lo*ldh; s.ldh = ldh; s.hcomp = hcomp > 0 ? hcomp : m.nring; s.b = (int)b; s.X = (int)nphi_; s.npair = (int)npair; s.nring = nring; s.mmax = mmax;
			s.Y = s1_.as<double2>(); s.ldY = ldY; s.dnp = make_fastdiv((uint32_t)npair);
			set_tiles(s, T1, b, nphi_); s.bout = blocked_on();
			launch_any(s, (long)nc*npair*s.ntile, st);
		}
		{	StRingS2 s; memset(&s, 0, sizeof(s));
			s.fa = mk(fc_, b, 8); s.fb = mk(fc_, 0, 8);
			s.Y = s1_.as<double2>(); s.ldY = ldY; s.a = (int)a; s.npair = (int)npair; s.m = map_addr(m); s.m.off0 += 2*q_lo*m.ring_stride; s.m.nring = nring;
			s.dnp = make_fastdiv((uint32_t)npair);
			set_tiles(s, T2, a, 0); s.bin = mk_blk(T1, b, T2);
			launch_any(s, (long)nc*npair*s.ntile, st);
		}
	}
	PXS_HIP(hipGetLastError());
}

// components per pass of a theta chain: all of them unless the two ping-pong buffers would exceed PXS_CHAIN_SCRATCH_GB (12) each
static int theta_comp_chunk(int nc, size_t need1_per_comp, size_t need2_per_comp) {
	static const size_t budget = [] { const char* e = getenv("PXS_CHAIN_SCRATCH_GB"); return (size_t)(e ? atol(e) : 12) << 30; }();
	const size_t per = sizeof(double2)*std::max(need1_per_comp, need2_per_comp);
	return (int)std::max<size_t>(1, std::min<size_t>((size_t)nc, budget/std::max<size_t>(per, 1)));
}
void FftChain::theta_scratch(const ThetaPlan& tp, int nm, int nc, int kind, size_t& b1, size_t& b2) {
	const size_t npair = (size_t)(nm + 1)/2;
	size_t n1, n2;
	if (kind == 0)      { n1 = npair*std::max(tp.g*pad8(tp.bN), tp.g*pad8(tp.g2)); n2 = npair*std::max(tp.g2*pad8(tp.g), tp.ac*pad8(tp.g)); }   // to_cc
	else if (kind == 1) { n1 = npair*tp.g*pad8(tp.bN); n2 = npair*tp.ac*pad8(tp.g); }                                                    // from_cc_adjoint
	else if (kind == 2) { n1 = npair*tp.gs*pad8(tp.bs); n2 = npair*tp.aNs*pad8(tp.gs); }                                                  // from_cc
	else                { n1 = npair*std::max(tp.g*pad8(tp.ac), tp.g*pad8(tp.g2)); n2 = npair*std::max(tp.g2*pad8(tp.g), tp.bN*pad8(tp.g)); }   // to_cc_adjoint
	const int cc = theta_comp_chunk(nc, n1, n2);
	b1 = sizeof(double2)*n1*cc; b2 = sizeof(double2)*n2*cc;
}
void FftChain::ring_scratch(long nring, int nc, bool analysis, size_t& b1, int mmax) const {
	if (!analysis && mmax >= 0 && line_h2map_takes(nphi_, mmax)) { b1 = 0; return; }
	const long npair = (nring + 1)/2, a = analysis ? ra_.a : rs_.a, b = analysis ? ra_.b : rs_.b;
	b1 = sizeof(double2)*(size_t)nc*npair*a*pad8(b);       // (ring_chunk only shrinks it)
}

// All components of a call go through each stage in ONE launch (outer index = component * npair + pair; batched maps are
// components here): 64 maps of 5400 rings are 5 launches of ~150 000 workgroups instead of 320 of ~10 000 with their tails.
void FftChain::to_cc(hipStream_t st, const ThetaPlan& tp, const double2* leg, long ldleg, int nr, int mir_c, double2* leg_cc, long ldcc, int ncc,
                     int nc, int nm, int spin, int lmax, const double2* ph_shift, const double2* sigma, const double2* wcc)
{
	if (line_analysis(st, tp, true, leg, ldleg, nr, mir_c, leg_cc, ldcc, ncc, nc, nm, spin, lmax, ph_shift, sigma, wcc, nullptr)) return;
	const long npair = (nm + 1)/2;
	const long g = tp.g, bN = tp.bN, g2 = tp.g2, ac = tp.ac;
	const long ldY1 = pad8(bN), ldZ2 = pad8(g), ldV3 = pad8(g2), ldU4 = pad8(g);
	const size_t need1 = (size_t)npair*std::max(g*ldY1, g*ldV3), need2 = (size_t)npair*std::max(g2*ldZ2, ac*ldU4);
	const int cchunk = theta_comp_chunk(nc, need1, need2);
	const int T1 = tile_lines_for<StFirst>(g, 0, bN, 8), T2 = tile_lines_for<StResize>(bN, g2, g, 8), T3 = tile_lines_for<StSigma>(g, g, g2, 8), T4 = tile_lines_for<StResize>(g2, ac, g, 8);
	s1_.ensure(sizeof(double2)*need1*cchunk); s2_.ensure(sizeof(double2)*need2*cchunk);
	for (int c0 = 0; c0 < nc; c0 += cchunk) {
		const long ncl = std::min(cchunk, nc - c0);
		{	StFirst s; memset(&s, 0, sizeof(s));
			s.fa = mk(fc_, g); s.fb = mk(fc_, 0);
			s.src.leg = leg + (size_t)c0*nm*ldleg; s.src.cstride = (long)nm*ldleg; s.src.ld = ldleg; s.src.nr = nr; s.src.N = (int)tp.N; s.src.mir_c = mir_c; s.src.a_odd = spin & 1; s.src.ncol = nm;
			s.b = (int)bN; s.Y = s1_.as<double2>(); s.ldY = ldY1; s.npair = (int)npair; s.dnp = make_fastdiv((uint32_t)npair);
			set_tiles(s, T1, bN, tp.N); s.bout = blocked_on();
			launch_any(s, ncl*npair*s.ntile, st);
		}
		{	StResize s; memset(&s, 0, sizeof(s));
			s.fa = mk(fc_, bN); s.fb = mk(fc_, g2);
			s.Y = s1_.as<double2>(); s.ldY = ldY1; s.Z = s2_.as<double2>(); s.ldZ = ldZ2; s.g = (int)g; s.X1 = (int)tp.N; s.X2 = (int)tp.M;
			// M > N (the interpolant form; the fine-CC form on grids below ~2 lmax rings): zero padding, Nyquist bin of N split;
			// M <= N (the fine-CC form on larger grids): low pass, |k| < M/2 kept
			s.kmax = tp.M > tp.N ? -1 : (int)(tp.M/2 - 1); s.nyq = tp.M > tp.N ? 1 : 0;
			s.ph = ph_shift; s.dg = make_fastdiv((uint32_t)g);
#ifdef PXS_LAB
			{ static const int noph = [] { const char* e = getenv("PXS_CH_NOPH"); return e ? atoi(e) : 0; }(); if (noph) s.ph = nullptr; }     // timing experiments only (wrong results)
#endif
			set_tiles(s, T2, g, tp.M); s.bin = mk_blk(T1, bN, T2); s.bout = blocked_on();
			launch_any(s, ncl*npair*s.ntile, st);
		}
		{	StSigma s; memset(&s, 0, sizeof(s));
			s.fa = mk(fc_, g); s.fb = mk(fc_, g);
			s.Z = s2_.as<double2>(); s.ldZ = ldZ2; s.V = s1_.as<double2>(); s.ldV = ldV3; s.g = (int)g; s.g2 = (int)g2; s.sigma = sigma;
			set_tiles(s, T3, g2, tp.M); s.bin = mk_blk(T2, g, T3); s.bout = blocked_on();
			launch_any(s, ncl*npair*s.ntile, st);
		}
		{	StResize s; memset(&s, 0, sizeof(s));
			s.fa = mk(fc_, g2); s.fb = mk(fc_, ac);
			s.Y = s1_.as<double2>(); s.ldY = ldV3; s.Z = s2_.as<double2>(); s.ldZ = ldU4; s.g = (int)g; s.X1 = (int)tp.M; s.X2 = (int)tp.Ncc; s.kmax = lmax; s.nyq = 0;
			s.ph = nullptr; s.dg = make_fastdiv((uint32_t)g);
			set_tiles(s, T4, g, tp.Ncc); s.bin = mk_blk(T3, g2, T4); s.bout = blocked_on();
			launch_any(s, ncl*npair*s.ntile, st);
		}
		{	StSplit<0> s; memset(&s, 0, sizeof(s));
			s.fa = mk(fc_, g); s.fb = mk(fc_, 0);
			int T = tile_lines_for<StSplit<0>>(g, 0, 2*ac, 8); if (T < 2) T = 2; T -= T % 2;
			const int TH = T/2;
			set_tiles(s, T, (long)((ac/2 + 1 + TH - 1)/TH)*T, 0);
			s.TH = TH;
			s.U = s2_.as<double2>(); s.ldU = ldU4; s.a = (int)ac; s.g = (int)g; s.X = (int)tp.Ncc; s.mir_c = 0; s.nr_out = ncc; s.a_odd = spin & 1; s.ncol = nm; s.npair = (int)npair;
			s.out = leg_cc + (size_t)c0*nm*ldcc; s.ocstride = (long)nm*ldcc; s.dnp = make_fastdiv((uint32_t)npair); s.groups = 1; s.dgr = make_fastdiv(1);
			s.ld = ldcc; s.w = wcc; s.tab = nullptr; s.scale = 1.0; s.da = make_fastdiv((uint32_t)ac);
			s.bin = mk_blk(T4, g, T);
			launch_any(s, ncl*npair*s.ntile, st);
		}
	}
	PXS_HIP(hipGetLastError());
}

// exact transpose of from_cc (theta upsampling CC grid -> map rings): leg on the map's rings -> leg on the CC grid, for grids
// without self-mirrored rings (F1).  With U = S F_N^-1 D(e^{ik theta0}) Pad F_Ncc E (E: mirror extension of the CC samples,
// S: sampling at the rings), U^T = E^T F_Ncc^-1 Trunc D(e^{-ik theta0}) F_N S^T up to the scalar; E^T A S^T = R A E' for any
// operator A that commutes with the reflection (E': mirror extension of the ring samples, R: restriction to the CC rings) except
// at the CC pole rings, which E' counts twice: weight 1/2 there (w).  So: RA1, the resize N -> N_cc directly, RA5.
void FftChain::from_cc_adjoint(hipStream_t st, const ThetaPlan& tp, const double2* leg, long ldleg, int nr, int mir_c, double2* leg_cc, long ldcc, int ncc,
                               int nc, int nm, int spin, int lmax, const double2* ph_shift, const double2* w, const double2* wring)
{
	if (line_analysis(st, tp, false, leg, ldleg, nr, mir_c, leg_cc, ldcc, ncc, nc, nm, spin, lmax, ph_shift, nullptr, w, wring)) return;
	const long npair = (nm + 1)/2;
	const long g = tp.g, bN = tp.bN, ac = tp.ac;
	const long ldY1 = pad8(bN), ldU = pad8(g);
	const size_t need1 = (size_t)npair*g*ldY1, need2 = (size_t)npair*ac*ldU;
	const int cchunk = theta_comp_chunk(nc, need1, need2);
	const int T1 = tile_lines_for<StFirst>(g, 0, bN, 8), T2 = tile_lines_for<StResize>(bN, ac, g, 8);
	s1_.ensure(sizeof(double2)*need1*cchunk); s2_.ensure(sizeof(double2)*need2*cchunk);
	for (int c0 = 0; c0 < nc; c0 += cchunk) {
		const long ncl = std::min(cchunk, nc - c0);
		{	StFirst s; memset(&s, 0, sizeof(s));
			s.fa = mk(fc_, g); s.fb = mk(fc_, 0);
			s.src.leg = leg + (size_t)c0*nm*ldleg; s.src.cstride = (long)nm*ldleg; s.src.ld = ldleg; s.src.nr = nr; s.src.N = (int)tp.N; s.src.mir_c = mir_c; s.src.a_odd = spin & 1; s.src.ncol = nm;
			s.src.w = wring;      // (quadrature weights of the map's rings: the weights form of the analysis)
			s.b = (int)bN; s.Y = s1_.as<double2>(); s.ldY = ldY1; s.npair = (int)npair; s.dnp = make_fastdiv((uint32_t)npair);
			set_tiles(s, T1, bN, tp.N); s.bout = blocked_on();
			launch_any(s, ncl*npair*s.ntile, st);
		}
		{	StResize s; memset(&s, 0, sizeof(s));
			s.fa = mk(fc_, bN); s.fb = mk(fc_, ac);
			s.Y = s1_.as<double2>(); s.ldY = ldY1; s.Z = s2_.as<double2>(); s.ldZ = ldU; s.g = (int)g; s.X1 = (int)tp.N; s.X2 = (int)tp.Ncc; s.kmax = lmax; s.nyq = 0;
			s.ph = ph_shift; s.dg = make_fastdiv((uint32_t)g);
			set_tiles(s, T2, g, tp.Ncc); s.bin = mk_blk(T1, bN, T2); s.bout = blocked_on();
			launch_any(s, ncl*npair*s.ntile, st);
		}
		{	StSplit<0> s; memset(&s, 0, sizeof(s));
			s.fa = mk(fc_, g); s.fb = mk(fc_, 0);
			int T = tile_lines_for<StSplit<0>>(g, 0, 2*ac, 8); if (T < 2) T = 2; T -= T % 2;
			const int TH = T/2;
			set_tiles(s, T, (long)((ac/2 + 1 + TH - 1)/TH)*T, 0);
			s.TH = TH;
			s.U = s2_.as<double2>(); s.ldU = ldU; s.a = (int)ac; s.g = (int)g; s.X = (int)tp.Ncc; s.mir_c = 0; s.nr_out = ncc; s.a_odd = spin & 1; s.ncol = nm; s.npair = (int)npair;
			s.out = leg_cc + (size_t)c0*nm*ldcc; s.ocstride = (long)nm*ldcc; s.dnp = make_fastdiv((uint32_t)npair); s.groups = 1; s.dgr = make_fastdiv(1);
			s.ld = ldcc; s.w = w; s.tab = nullptr; s.scale = 1.0; s.da = make_fastdiv((uint32_t)ac);
			s.bin = mk_blk(T2, g, T);
			launch_any(s, ncl*npair*s.ntile, st);
		}
	}
	PXS_HIP(hipGetLastError());
}

// exact adjoint of to_cc (adjoint_analysis_2d): leg on the CC grid -> ring spectra h[c][ring][m] on the map's rings.
// With to_cc = W R F_Ncc^-1 T F_M Sigma F_M^-1 P F_N E' (E': parity mirror extension of the ring samples, P: phase, Nyquist split and
// zero padding N -> M, T: truncation to |k| <= lmax, R: restriction to the CC rings, W: weights) the adjoint is
// E'^T F_N^-1 P^H F_M Sigma F_M^-1 T^T F_Ncc R^T W.  R^T zero-extends, but E'^T keeps only the part of the result with the column's
// parity (times 2), every operator in between commutes with the reflection, so the zero extension may be replaced by the parity
// mirror extension of W x with weight 1/2 off the poles: the chain then has the shape of from_cc -- pairs of columns of opposite
// parity packed into one sequence and separated by reflection symmetry at the end, written ring-major for the ring FFT.
void FftChain::to_cc_adjoint(hipStream_t st, const ThetaPlan& tp, const double2* leg_cc, long ldcc, int ncc, double2* h, long ldh, int nr, int mir_c,
                             int nc, int nm, int spin, int lmax, const double2* ph_shift, const double2* sigma, const double2* whalf, const double2* tab, double scale)
{
	const long npair = (nm + 1)/2;
	const long g = tp.g, bN = tp.bN, g2 = tp.g2, ac = tp.ac;
	const long ldY = pad8(ac), ldZ = pad8(g), ldV = pad8(g2);
	const size_t need1 = (size_t)npair*std::max(g*ldY, g*ldV), need2 = (size_t)npair*std::max(g2*ldZ, bN*ldZ);
	const int cchunk = theta_comp_chunk(nc, need1, need2);
	const int T1 = tile_lines_for<StFirst>(g, 0, ac, 8), T2 = tile_lines_for<StResize>(ac, g2, g, 8), T3 = tile_lines_for<StSigma>(g, g, g2, 8), T4 = tile_lines_for<StResize>(g2, bN, g, 8);
	s1_.ensure(sizeof(double2)*need1*cchunk); s2_.ensure(sizeof(double2)*need2*cchunk);
	for (int c0 = 0; c0 < nc; c0 += cchunk) {
		const long ncl = std::min(cchunk, nc - c0);
		{	StFirst s; memset(&s, 0, sizeof(s));      // weighted mirror-pair extension on the CC circle, pass 1 of FFT_Ncc
			s.fa = mk(fc_, g); s.fb = mk(fc_, 0);
			s.src.leg = leg_cc + (size_t)c0*nm*ldcc; s.src.cstride = (long)nm*ldcc; s.src.ld = ldcc; s.src.nr = ncc; s.src.N = (int)tp.Ncc; s.src.mir_c = 0; s.src.a_odd = spin & 1; s.src.ncol = nm;
			s.src.w = whalf;
			s.b = (int)ac; s.Y = s1_.as<double2>(); s.ldY = ldY; s.npair = (int)npair; s.dnp = make_fastdiv((uint32_t)npair);
			set_tiles(s, T1, ac, tp.Ncc); s.bout = blocked_on();
			launch_any(s, ncl*npair*s.ntile, st);
		}
		{	StResize s; memset(&s, 0, sizeof(s));     // pass 2 of FFT_Ncc, |k| <= lmax embedded in the M spectrum, pass 1 of IFFT_M
			s.fa = mk(fc_, ac); s.fb = mk(fc_, g2);
			s.Y = s1_.as<double2>(); s.ldY = ldY; s.Z = s2_.as<double2>(); s.ldZ = ldZ; s.g = (int)g; s.X1 = (int)tp.Ncc; s.X2 = (int)tp.M; s.kmax = lmax; s.nyq = 0;
			s.ph = nullptr; s.dg = make_fastdiv((uint32_t)g);
			set_tiles(s, T2, g, tp.M); s.bin = mk_blk(T1, ac, T2); s.bout = blocked_on();
			launch_any(s, ncl*npair*s.ntile, st);
		}
		{	StSigma s; memset(&s, 0, sizeof(s));      // pass 2 of IFFT_M, x |sin| series, pass 1 of FFT_M
			s.fa = mk(fc_, g); s.fb = mk(fc_, g);
			s.Z = s2_.as<double2>(); s.ldZ = ldZ; s.V = s1_.as<double2>(); s.ldV = ldV; s.g = (int)g; s.g2 = (int)g2; s.sigma = sigma;
			set_tiles(s, T3, g2, tp.M); s.bin = mk_blk(T2, g, T3); s.bout = blocked_on();
			launch_any(s, ncl*npair*s.ntile, st);
		}
		{	StResize s; memset(&s, 0, sizeof(s));     // pass 2 of FFT_M, transposed padding M -> N (conjugate phase, Nyquist bins combined), pass 1 of IFFT_N
			s.fa = mk(fc_, g2); s.fb = mk(fc_, bN);
			s.Y = s1_.as<double2>(); s.ldY = ldV; s.Z = s2_.as<double2>(); s.ldZ = ldZ; s.g = (int)g; s.X1 = (int)tp.M; s.X2 = (int)tp.N; s.kmax = tp.M > tp.N ? -1 : (int)(tp.M/2 - 1); s.nyq = 0; s.adj = 1;      // (M <= N: transpose of the low pass = zero padding of |k| < M/2)
			s.ph = ph_shift; s.dg = make_fastdiv((uint32_t)g);
			set_tiles(s, T4, g, tp.N); s.bin = mk_blk(T3, g2, T4);      // (plain rows out: the transposing split takes one line of many pairs)
			launch_any(s, ncl*npair*s.ntile, st);
		}
		{	StSplit<1> s; memset(&s, 0, sizeof(s));   // pass 2 of IFFT_N, the two parities apart, rings of the map, ring-major
			s.fa = mk(fc_, g); s.fb = mk(fc_, 0);
			int T = tile_lines_for<StSplit<1>>(g, 0, 2*npair, 8); if (T < 2) T = 2; T -= T % 2;
			set_tiles(s, T, bN*T, 0);         // one tile per line: ntile = bN
			s.TH = 0;
			s.U = s2_.as<double2>(); s.ldU = ldZ; s.a = (int)bN; s.g = (int)g; s.X = (int)tp.N; s.mir_c = mir_c; s.nr_out = nr; s.a_odd = spin & 1; s.ncol = nm; s.npair = (int)npair;
			const long groups = (npair + T/2 - 1)/(T/2);
			s.out = h + (size_t)c0*nr*ldh; s.ocstride = (long)nr*ldh; s.dnp = make_fastdiv((uint32_t)npair); s.groups = (int)groups; s.dgr = make_fastdiv((uint32_t)groups);
			s.ld = ldh; s.w = nullptr; s.tab = tab; s.scale = scale; s.da = make_fastdiv((uint32_t)bN); s.self_half = 1;
			launch_any(s, ncl*groups*bN, st);
		}
	}
	PXS_HIP(hipGetLastError());
}

void FftChain::from_cc(hipStream_t st, const ThetaPlan& tp, const double2* leg_cc, long ldcc, int ncc, double2* h, long ldh, int nr, int mir_c,
                       int nc, int nm, int spin, int lmax, const double2* ph_up, const double2* tab, double scale, const double2* wring)
{
	const long npair = (nm + 1)/2;
	const long gs = tp.gs, bs = tp.bs, aN = tp.aNs;
	const long ldY = pad8(bs), ldZ = pad8(gs);
	const size_t need1 = (size_t)npair*gs*ldY, need2 = (size_t)npair*aN*ldZ;
	const int cchunk = theta_comp_chunk(nc, need1, need2);
	const int T1 = tile_lines_for<StFirst>(gs, 0, bs, 8), T2 = tile_lines_for<StResize>(bs, aN, gs, 8);
	s1_.ensure(sizeof(double2)*need1*cchunk); s2_.ensure(sizeof(double2)*need2*cchunk);
	for (int c0 = 0; c0 < nc; c0 += cchunk) {
		const long ncl = std::min(cchunk, nc - c0);
		{	StFirst s; memset(&s, 0, sizeof(s));
			s.fa = mk(fc_, gs); s.fb = mk(fc_, 0);
			s.src.leg = leg_cc + (size_t)c0*nm*ldcc; s.src.cstride = (long)nm*ldcc; s.src.ld = ldcc; s.src.nr = ncc; s.src.N = (int)tp.Ncc; s.src.mir_c = 0; s.src.a_odd = spin & 1; s.src.ncol = nm;
			s.b = (int)bs; s.Y = s1_.as<double2>(); s.ldY = ldY; s.npair = (int)npair; s.dnp = make_fastdiv((uint32_t)npair);
			set_tiles(s, T1, bs, tp.Ncc); s.bout = blocked_on();
			launch_any(s, ncl*npair*s.ntile, st);
		}
		{	StResize s; memset(&s, 0, sizeof(s));
			s.fa = mk(fc_, bs); s.fb = mk(fc_, aN);
			s.Y = s1_.as<double2>(); s.ldY = ldY; s.Z = s2_.as<double2>(); s.ldZ = ldZ; s.g = (int)gs; s.X1 = (int)tp.Ncc; s.X2 = (int)tp.N; s.kmax = lmax; s.nyq = 0;
			s.ph = ph_up; s.dg = make_fastdiv((uint32_t)gs);
			set_tiles(s, T2, gs, tp.N); s.bin = mk_blk(T1, bs, T2);      // (plain rows out, see to_cc_adjoint)
			launch_any(s, ncl*npair*s.ntile, st);
		}
		{	StSplit<1> s; memset(&s, 0, sizeof(s));
			s.fa = mk(fc_, gs); s.fb = mk(fc_, 0);
			int T = tile_lines_for<StSplit<1>>(gs, 0, 2*npair, 8); if (T < 2) T = 2; T -= T % 2;
			set_tiles(s, T, aN*T, 0);         // one tile per line: ntile = aN
			s.TH = 0;
			s.U = s2_.as<double2>(); s.ldU = ldZ; s.a = (int)aN; s.g = (int)gs; s.X = (int)tp.N; s.mir_c = mir_c; s.nr_out = nr; s.a_odd = spin & 1; s.ncol = nm; s.npair = (int)npair;
			const long groups = (npair + T/2 - 1)/(T/2);
			s.out = h + (size_t)c0*nr*ldh; s.ocstride = (long)nr*ldh; s.dnp = make_fastdiv((uint32_t)npair); s.groups = (int)groups; s.dgr = make_fastdiv((uint32_t)groups);
			s.ld = ldh; s.w = wring; s.tab = tab; s.scale = scale; s.da = make_fastdiv((uint32_t)aN);
			launch_any(s, ncl*groups*aN, st);
		}
	}
	PXS_HIP(hipGetLastError());
}

// 2-D FFT of real maps [npre][ny][nx] -> complex [npre][ny][nx] (enmap.fft of a map, pixell/enmap.py:1307-1323), through the
// chain stages: MA1 / MA2 transform two real rows per complex line and leave the half spectrum transposed, F[kx <= nx/2][y]; the
// column transforms then run along contiguous rows (StFirst in plain mode, StColOut), and the last pass writes out[ky][kx] and its
// Hermitian image.  Traffic: 7 x (8 bytes per pixel) + the complex output = 4.5 N x 16 bytes against 7.5 N x 16 for the same
// transform as a complex one over both axes.  false: no usable factorisation (the caller takes the generic engine).
bool FftChain::fft2_real(hipStream_t st, const void* in, int in_dtype, double2* out, long npre, long ny, long nx, bool forward, double scale) {
	Split sy;
	if (nx < 4 || ny < 4 || !split_balanced(ny, sy) || (nphi_ != nx && !plan_rings(nx))) return false;
	const long nm = nx/2 + 1, ldF = pad8(ny), a = std::min(sy.a, sy.b), b = ny/a, ldY = pad8(b);
	if (getenv("PXS_CHAIN_VERBOSE")) fprintf(stderr, "[pxsht] fft2_real %ld x %ld x %ld: rows %s, columns %ld x %ld\n", npre, ny, nx, describe().c_str(), a, b);
	PXS_REQUIRE(npre*nm < (1L << 31)/std::max<long>(a, b), "fft2_real: too many lines");
	// F (half spectrum, transposed) lives in s2_, the four-step intermediates of both axes in s1_
	s2_.ensure(sizeof(double2)*(size_t)npre*nm*ldF);
	MapDesc m; m.ptr = in; m.dtype = in_dtype; m.cstride = ny*nx; m.ring_off0 = 0; m.ring_stride = nx; m.pix_stride = 1; m.nring = (int)ny; m.nphi = nx;
	map2leg(st, m, (int)npre, (int)(nm - 1), s2_.as<double2>(), ldF, nullptr, 1.0);
	s1_.ensure(sizeof(double2)*(size_t)npre*nm*a*ldY);
	{	StFirst2D s; memset(&s, 0, sizeof(s));
		s.fa = mk(fc_, a); s.fb = mk(fc_, 0);
		s.src.leg = s2_.as<double2>(); s.src.cstride = nm*ldF; s.src.ld = ldF; s.src.nr = (int)ny; s.src.N = (int)ny; s.src.ncol = (int)nm; s.src.plain = 1;
		s.b = (int)b; s.Y = s1_.as<double2>(); s.ldY = ldY; s.npair = (int)nm; s.dnp = make_fastdiv((uint32_t)nm);
		set_tiles(s, tile_lines_for<StFirst2D>(a, 0, b, 8), b, ny);
		launch_any(s, npre*nm*s.ntile, st);
	}
	{	StColOut s; memset(&s, 0, sizeof(s));
		s.fa = mk(fc_, b); s.fb = mk(fc_, 0);
		int T = tile_lines_for<StColOut>(b, 0, nm, 8);
		set_tiles(s, T, a*T, 0);          // one tile per line: ntile = a
		s.Y = s1_.as<double2>(); s.ldY = ldY; s.a = (int)a; s.nm = (int)nm; s.ny = (int)ny; s.nx = (int)nx; s.conj_out = forward ? 0 : 1; s.out = out; s.scale = scale;
		s.herm = 1; s.ldo = nx; s.ocomp = ny*nx;
		s.groups = (int)((nm + T - 1)/T); s.dgr = make_fastdiv((uint32_t)s.groups);
		launch_any(s, npre*s.groups*a, st);
	}
	PXS_HIP(hipGetLastError());
	return true;
}

// 2-D complex FFT [npre][ny][nx] -> [npre][ny][nx] (enmap.ifft, enmap.fft of complex maps) with the same stage kinds: rows (StFirst
// plain over the lines of a row, StColOut into the transposed F[kx][y]), then columns along contiguous rows of F (StFirst plain,
// StColOut into out[ky][kx]).  Backward: conj on the way in and out.  false: no usable factorisation.
bool FftChain::fft2_c2c(hipStream_t st, const double2* in, double2* out, long npre, long ny, long nx, bool forward, double scale) {
	Split sx, sy;
	if (nx < 4 || ny < 4 || !split_balanced(nx, sx) || !split_balanced(ny, sy)) return false;
	const long ax = std::min(sx.a, sx.b), bx = nx/ax, ay = std::min(sy.a, sy.b), by = ny/ay, ldF = pad8(ny);
	if (npre*std::max(nx, ny) >= (1L << 31)/512) return false;
	if (getenv("PXS_CHAIN_VERBOSE")) fprintf(stderr, "[pxsht] fft2_c2c %ld x %ld x %ld: rows %ld x %ld, columns %ld x %ld\n", npre, ny, nx, ax, bx, ay, by);
	s1_.ensure(sizeof(double2)*(size_t)npre*std::max(ny*ax*pad8(bx), nx*ay*pad8(by)));
	s2_.ensure(sizeof(double2)*(size_t)npre*nx*ldF);
	auto first = [&](const double2* src, long nlines, long ld, long n, long a, long b, int conj) {
		StFirst2D s; memset(&s, 0, sizeof(s));
		s.fa = mk(fc_, a); s.fb = mk(fc_, 0);
		s.src.leg = src; s.src.cstride = nlines*ld; s.src.ld = ld; s.src.nr = (int)n; s.src.N = (int)n; s.src.ncol = (int)nlines; s.src.plain = 1; s.src.conj = conj;
		s.b = (int)b; s.Y = s1_.as<double2>(); s.ldY = pad8(b); s.npair = (int)nlines; s.dnp = make_fastdiv((uint32_t)nlines);
		set_tiles(s, tile_lines_for<StFirst2D>(a, 0, b, 8), b, n);
		launch_any(s, npre*nlines*s.ntile, st);
	};
	auto second = [&](double2* dst, long nlines, long ldo, long ocomp, long a, long b, int conj, double sc) {
		StColOut s; memset(&s, 0, sizeof(s));
		s.fa = mk(fc_, b); s.fb = mk(fc_, 0);
		int T = tile_lines_for<StColOut>(b, 0, nlines, 8);
		set_tiles(s, T, a*T, 0);
		s.Y = s1_.as<double2>(); s.ldY = pad8(b); s.a = (int)a; s.nm = (int)nlines; s.ny = 0; s.nx = 0; s.conj_out = conj; s.herm = 0; s.out = dst; s.ldo = ldo; s.ocomp = ocomp; s.scale = sc;
		s.groups = (int)((nlines + T - 1)/T); s.dgr = make_fastdiv((uint32_t)s.groups);
		launch_any(s, npre*s.groups*a, st);
	};
	first(in, ny, nx, nx, ax, bx, forward ? 0 : 1);                           // rows y: lines of nx points
	second(s2_.as<double2>(), ny, ldF, nx*ldF, ax, bx, 0, 1.0);               // -> F[kx][y]
	first(s2_.as<double2>(), nx, ldF, ny, ay, by, 0);                         // columns kx: contiguous lines of ny points
	second(out, nx, nx, ny*nx, ay, by, forward ? 0 : 1, scale);               // -> out[ky][kx]
	PXS_HIP(hipGetLastError());
	return true;
}

} // namespace pxs
