// Second-generation LDS FFT core of the chain kernels (fftchain.hip): decimation in frequency with LARGE register radices.
//
// A line of n = R0 * R1 [* R2] points is transformed in at most three passes; a thread does one radix-R butterfly (R up to 25)
// entirely in registers with compile-time twiddles.  Against the first-generation core (fft_dev.hpp: DIT, radices <= 9, 3-4
// passes per transform, digit-reversed input) this
//   * halves the LDS sweeps (LDS stores run at ~79 B/clk/CU on gfx950, a third of the load rate) and the barriers,
//   * takes its input in NATURAL order, so a tile can be brought in by LDS-DMA (global_load_lds writes a wave's 64 x 16 bytes
//     linearly; a digit-reversed placement would need the VGPR detour),
//   * hands the results of the last pass to the caller in registers (fused with the global store), and
//   * pads the layout (row stride rs of the n/R0-point blocks, line stride ns) so that every pass is free of bank conflicts.
// Layout of a line in LDS: point e sits at slot (e / M1) * rs + e % M1 with M1 = n / R0 (slots of 16 bytes).
#pragma once
#include "fft_dev.hpp"

namespace pxs {

// ---- radix-R DFT in registers, natural order in and out -------------------------------------------------------------
template<> struct RadixTw<18> { static __device__ __forceinline__ double2 w(int j) { return make_double2(PXS_RTW18[2*j], PXS_RTW18[2*j+1]); } };
template<> struct RadixTw<20> { static __device__ __forceinline__ double2 w(int j) { return make_double2(PXS_RTW20[2*j], PXS_RTW20[2*j+1]); } };
template<> struct RadixTw<24> { static __device__ __forceinline__ double2 w(int j) { return make_double2(PXS_RTW24[2*j], PXS_RTW24[2*j+1]); } };
template<> struct RadixTw<25> { static __device__ __forceinline__ double2 w(int j) { return make_double2(PXS_RTW25[2*j], PXS_RTW25[2*j+1]); } };

// a * W_R^j; j is a compile-time value wherever this is called (fully unrolled loops), so the tests fold away
template<int R> __device__ __forceinline__ double2 mul_wr(double2 a, int j) {
	if (j == 0) return a;
	if (4*j == R) return make_double2(a.y, -a.x);          // -i
	if (2*j == R) return make_double2(-a.x, -a.y);         // -1
	if (4*j == 3*R) return make_double2(-a.y, a.x);        // +i
	return cmul(a, RadixTw<R>::w(j));
}

template<int R> struct RadixSplit { static constexpr int A = 1, B = R; };
template<> struct RadixSplit<6>  { static constexpr int A = 3, B = 2; };
template<> struct RadixSplit<8>  { static constexpr int A = 4, B = 2; };
template<> struct RadixSplit<9>  { static constexpr int A = 3, B = 3; };
template<> struct RadixSplit<10> { static constexpr int A = 5, B = 2; };
template<> struct RadixSplit<12> { static constexpr int A = 4, B = 3; };
template<> struct RadixSplit<15> { static constexpr int A = 5, B = 3; };
template<> struct RadixSplit<16> { static constexpr int A = 4, B = 4; };
template<> struct RadixSplit<18> { static constexpr int A = 3, B = 6; };
template<> struct RadixSplit<20> { static constexpr int A = 5, B = 4; };
template<> struct RadixSplit<24> { static constexpr int A = 4, B = 6; };
template<> struct RadixSplit<25> { static constexpr int A = 5, B = 5; };

// forward DFT of v[0..R-1] in place, natural order.  Composite R = A*B: B transforms of A points over stride B, the twiddles
// W_R^{n2 k1}, A transforms of B points; X[k1 + A k2].
template<int R> __device__ __forceinline__ void dft_reg(double2* v) {
	if constexpr (R == 1) { }
	else if constexpr (R <= 5) butterfly<R>(v);
	else {
		constexpr int A = RadixSplit<R>::A, B = RadixSplit<R>::B;
		static_assert(A > 1 && A*B == R, "radix not supported");
		double2 u[R];
#pragma unroll
		for (int n2 = 0; n2 < B; n2++) {
			double2 t[A];
#pragma unroll
			for (int n1 = 0; n1 < A; n1++) t[n1] = v[B*n1 + n2];
			dft_reg<A>(t);
#pragma unroll
			for (int k1 = 0; k1 < A; k1++) u[k1*B + n2] = mul_wr<R>(t[k1], n2*k1);
		}
#pragma unroll
		for (int k1 = 0; k1 < A; k1++) {
			dft_reg<B>(u + k1*B);
#pragma unroll
			for (int k2 = 0; k2 < B; k2++) v[k1 + A*k2] = u[k1*B + k2];
		}
	}
}

// ---- transform descriptor -------------------------------------------------------------------------------------------
static constexpr int F2_MAXR = 9;        // largest radix the chain kernels dispatch to (10 ... 25 exist as codelets; see chain2_kernel)
static constexpr int F2_MAXR_FIRST = 6;  // ... in a pass that takes its inputs from global memory or through a stage's mid()
struct Fft2 {
	int n, np;            // points, passes (1..3)
	int R0, R1, R2;       // radices in DIF order (unused ones = 1)
	int M1, M2;           // M1 = n / R0, M2 = M1 / R1 (np = 3)
	int rs, ns;           // slot stride of the M1-point blocks, of the lines
	FastDiv dM1, dM2, dR0, dR1, drs, dns;
	const double2* tw;    // W_n^k, k < n (global; the kernels stage it in LDS)
	// slot of point e (input order) and of frequency k (after the last pass) within a line
	__device__ __forceinline__ int slot_in(int e) const { const uint32_t b = fdiv((uint32_t)e, dM1); return (int)(b*rs + (e - b*M1)); }
	__device__ __forceinline__ int slot_out(int k) const {
		if (np == 1) return k;
		const uint32_t q = fdiv((uint32_t)k, dR0), k1 = k - q*R0;
		if (np == 2) return (int)(k1*rs + q);
		const uint32_t k3 = fdiv(q, dR1), k2 = q - k3*R1;
		return (int)(k1*rs + k2*M2 + k3);
	}
};

// Task = one butterfly of one line of one pass (the pass loop itself is f2_pass in fftchain.hip).
struct F2Task { int li, base, istride, step, k0, kstride; };      // twiddle index of output i: i*step; frequency of output i (last pass): k0 + i*kstride

__device__ __forceinline__ void f2_decode(const Fft2& f, int q, int tl, F2Task& t) {
	if (q == 0) { t.base = tl; t.istride = f.np == 1 ? 1 : f.rs; t.step = tl; t.k0 = 0; t.kstride = 1; if (f.np == 1) { t.base = 0; t.step = 0; } return; }
	if (f.np == 2) { t.base = tl*f.rs; t.istride = 1; t.step = 0; t.k0 = tl; t.kstride = f.R0; return; }
	if (q == 1) { const uint32_t k1 = fdiv((uint32_t)tl, f.dM2), r = tl - k1*f.M2; t.base = (int)(k1*f.rs + r); t.istride = f.M2; t.step = (int)(f.R0*r); t.k0 = 0; t.kstride = 1; return; }
	{ const uint32_t k1 = fdiv((uint32_t)tl, f.dR1), k2 = tl - k1*f.R1; t.base = (int)(k1*f.rs + k2*f.M2); t.istride = 1; t.step = 0; t.k0 = (int)(k1 + f.R0*k2); t.kstride = f.R0*f.R1; }
}
__device__ __forceinline__ int f2_radix(const Fft2& f, int q) { return q == 0 ? f.R0 : (q == 1 ? f.R1 : f.R2); }
__device__ __forceinline__ int f2_tasks(const Fft2& f, int q) { return f.n / f2_radix(f, q); }       // per line

} // namespace pxs
