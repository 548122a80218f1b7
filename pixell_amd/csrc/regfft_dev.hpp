// Register-resident line FFT for gfx950: one workgroup of NT threads holds a whole line of n complex128 points in its
// registers (PMAX points per thread at most) and transforms it with Stockham decimation-in-frequency passes whose butterflies
// (radix 2 ... 16) run entirely in registers.  Between two passes the line goes through the LDS ONE COMPONENT AT A TIME
// (re, then im): the LDS holds n doubles, so a line of 16 128 points (258 KB) fits a CU -- 129 KB in the LDS during an exchange,
// the other component in registers.  Used by the single-kernel theta resampling engine (thetaline.hip); host model of the same
// index arithmetic: tools/regfft_model.py.
//
// Pass p (radix R, s = product of the earlier radices, nb = n / R butterflies, butterfly b = q + s*pp, q = b mod s):
//   reads   a_k = x[b + nb*k]                            -- the same pattern for every pass: thread t holds butterflies t + NT*i
//   writes  y[q + s*(R*pp + j)] = W_n^{(b - q) j} * sum_k a_k W_R^{jk}
// After the last pass (s = nb) the outputs sit in natural order, output j of butterfly b being X[b + nb*j]: a transform that
// follows with the same first radix can start from the registers without an exchange (the pointwise step of the theta chain).
//
// The radix sequence of a transform is a TEMPLATE parameter (RfSeq<16, 16, 9, 7>): lengths, strides, butterflies per thread and every
// register index are compile-time constants and a transform is straight-line code.  What the compiler needs for the line to STAY in
// registers was found the hard way (hipcc 7.2; each step measured in spill counts at the 128 registers of a 1024-thread workgroup):
//  * a first form chose the radix of each pass at run time (a switch over 12 radices inside the pass loop): the register array
//    stayed in scratch memory (SimplifyCFG merged the array accesses of the cases into pointer phis before SROA had split the array);
//    with that pass disabled and every index a template constant (sfor, not `#pragma unroll`) the array was split, and the allocator
//    spilled 1100-2700 times -- each further case of the switch added several hundred: C4's theta resampling 621 ms against the 86 of
//    the stage chain;
//  * no per-lane branch around the work of a register slot (lanes without work compute on what they hold and nothing reads it):
//    with a branch per slot 2400 spills where the select form had 150;
//  * the thread index passes through an opaque move at the start of every phase, or everything derived from it (the LDS addresses of
//    every slot of every pass) is hoisted out of the line loop into registers that are then spilled.
#pragma once
#include "fft_dev.hpp"
#include <type_traits>

namespace pxs {

static constexpr int RF_TWL = 128;         // two-level twiddles: W_n^e = lo[e mod 128] * hi[e / 128]

// compile-time loop: f(std::integral_constant<int, I>) for I = A ... B-1
template<int A, int B, class F> __device__ __forceinline__ void sfor(F&& f) {
	if constexpr (A < B) { f(std::integral_constant<int, A>()); sfor<A + 1, B>(static_cast<F&&>(f)); }
}
#define RF_INL __attribute__((always_inline))
#define RF_IDX(C) decltype(C)::value

#ifdef PXS_HOST_SIM
#define RF_OPAQUE(x) do {} while (0)
#define RF_PIN2(a) do {} while (0)
#define RF_FENCE() do {} while (0)
#else
#define RF_OPAQUE(x) asm volatile("" : "+v"(x))
#define RF_PIN2(a) asm volatile("" : "+v"((a).x), "+v"((a).y))
#define RF_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
// a scheduling fence every PXS_RF_GROUP register slots: left alone the scheduler issues the reads of ALL slots of a phase before the
// first use (up to 84 registers next to the 84 of the line); the other waves of the workgroup cover the latency of a group
#ifndef PXS_RF_GROUP
#define PXS_RF_GROUP 6
#endif
#ifndef PXS_RF_LEAN_MIN
#define PXS_RF_LEAN_MIN 6      /* composite radices from here on take rf_bfly_lean */
#endif
#define RF_FENCE_SLOT(c) do { if (((c) + 1) % PXS_RF_GROUP == 0) RF_FENCE(); } while (0)
#define RF_BARRIER() PXS_LDS_BARRIER()      /* LDS traffic only: global stores of the previous line may still be in flight */
// between the LDS phases of ONE wave on its own stretch of the LDS: nothing on the GPU (the LDS runs a wave's operations in order);
// the simulator's lanes are OS threads and meet here
#ifdef PXS_HOST_SIM
#define RF_WAVE_SYNC() pxsim::wave_sync()
#else
#define RF_WAVE_SYNC() do {} while (0)
#endif

// radix sequence of a transform of N = product points
template<int... Rs> struct RfSeq {
	static constexpr int NP = sizeof...(Rs);
	static constexpr int N = (Rs * ... * 1);
	static constexpr int rad(int p) { constexpr int r[NP] = {Rs...}; return r[p]; }
	static constexpr int stride(int p) { int s = 1; for (int q = 0; q < p; q++) s *= rad(q); return s; }      // product of the earlier radices
};
// pass P of sequence S on NT threads
template<class S, int P, int NT> struct RfPassT {
	static constexpr int R = S::rad(P), s = S::stride(P), n = S::N, nb = n/R, K = (nb + NT - 1)/NT, slots = K*R;
	static constexpr bool twiddled = s < nb;
};

// R-point butterfly on a local array; output j is left in position slot(j) (the composite radices of fft_dev.hpp permute)
template<int R> struct RfB;
#define PXS_RFB_PRIM(RR) template<> struct RfB<RR> { \
	static __device__ __forceinline__ void run(double2* a) { butterfly<RR>(a); } \
	static constexpr __host__ __device__ int slot(int j) { return j; } };
#define PXS_RFB_COMP(AA, BB) template<> struct RfB<AA*BB> { \
	static __device__ __forceinline__ void run(double2* a) { butterfly_comp<AA, BB>(a); } \
	static constexpr __host__ __device__ int slot(int j) { return BB*(j % AA) + j/AA; } };
PXS_RFB_PRIM(2) PXS_RFB_PRIM(3) PXS_RFB_PRIM(4) PXS_RFB_PRIM(5) PXS_RFB_PRIM(7)
PXS_RFB_COMP(3, 2) PXS_RFB_COMP(4, 2) PXS_RFB_COMP(3, 3) PXS_RFB_COMP(5, 2) PXS_RFB_COMP(4, 3) PXS_RFB_COMP(5, 3) PXS_RFB_COMP(4, 4)

// Composite butterfly of A*B points + the Stockham twiddles W^j on output j (w1 = W; TWID: multiply at all), written for REGISTERS.
// Under hipcc's scheduler the generic form (butterfly_comp<A, B> of fft_dev.hpp, then a loop over the outputs) has the independent
// sub-butterflies of a stage and all twiddle powers in flight at once: for 16 points 64 registers of data and ~70 more of temporaries,
// 70-190 spills per transform at the 128 registers of a 1024-thread workgroup.  Here the outputs of every sub-butterfly and the inputs
// of the next one pass through empty asm statements ("+v" operands: the values are redefined there, so what depends on them cannot
// start earlier, and volatile statements keep their order), and the twiddles advance in steps of W^A inside each output group with
// only W, W^A, the group's first power and the running one alive.  Output j ends in a[B (j mod A) + j / A], the layout of
// butterfly_comp<A, B>.  Spills of a transform of 16 128 points as 16 16 9 7: 147 -> 25.
// RF_PIN2 on a[OFF + STRIDE i], i < CNT (a function, not a lambda: an asm operand may not name a variable captured by a nested lambda)
template<int I, int CNT, int STRIDE, int OFF, int LEN> __device__ __forceinline__ void rf_pin_range(double2 (&a)[LEN]) {
	if constexpr (I < CNT) { RF_PIN2(a[OFF + STRIDE*I]); rf_pin_range<I + 1, CNT, STRIDE, OFF, LEN>(a); }
}
__device__ __forceinline__ void rf_pin_two(double2& x, double2& y) { RF_PIN2(x); RF_PIN2(y); }
template<int A, int B, bool TWID> __device__ __forceinline__ void rf_bfly_lean(double2 (&a)[A*B], const double2 w1) {
	// stage 1: radix A over (n2, n2 + B, ...), then W_{AB}^{n2 k1}
	sfor<0, B>([&](auto N2) RF_INL {
		constexpr int n2 = RF_IDX(N2);
		double2 t[A];
		sfor<0, A>([&](auto N1) RF_INL { t[RF_IDX(N1)] = a[B*RF_IDX(N1) + n2]; });
		butterfly<A>(t);
		sfor<0, A>([&](auto K1) RF_INL {
			constexpr int k1 = RF_IDX(K1), m = n2*k1;
			if constexpr (m == 0) a[B*k1 + n2] = t[k1];
			else if constexpr (4*m == A*B) a[B*k1 + n2] = mulmi(t[k1]);      // W^{AB/4} = -i
			else a[B*k1 + n2] = cmul(t[k1], RadixTw<A*B>::w(m));
		});
		if constexpr (n2 + 1 < B) { rf_pin_range<0, A, B, n2, A*B>(a); rf_pin_range<0, A, B, n2 + 1, A*B>(a); }
	});
	// stage 2: radix B over (B k1 ... B k1 + B - 1): X[k1 + A k2] in a[B k1 + k2]; twiddle W^{k1 + A k2}
	double2 wA = w1, start = make_double2(1, 0);      // W^A; W^{k1}, the first power of group k1
	if constexpr (TWID) sfor<1, A>([&](auto) RF_INL { wA = cmul(wA, w1); });
	sfor<0, A>([&](auto K1) RF_INL {
		constexpr int k1 = RF_IDX(K1);
		butterfly<B>(&a[B*k1]);
		if constexpr (TWID) {
			if constexpr (k1 == 1) start = w1; else if constexpr (k1 > 1) start = cmul(start, w1);
			double2 t = start;
			if constexpr (k1 > 0) a[B*k1] = cmul(a[B*k1], t);
			sfor<1, B>([&](auto K2) RF_INL {
				constexpr int k2 = RF_IDX(K2);
				if constexpr (k1 == 0 && k2 == 1) t = wA; else t = cmul(t, wA);
				a[B*k1 + k2] = cmul(a[B*k1 + k2], t);
			});
		}
		if constexpr (k1 + 1 < A) {
			rf_pin_range<0, B, 1, B*k1, A*B>(a); rf_pin_range<0, B, 1, B*(k1 + 1), A*B>(a);
			if constexpr (TWID) rf_pin_two(wA, start);
		}
	});
}
template<int R> struct RfLean { static constexpr bool has = false; static constexpr int A = 1, B = 1; };
#define PXS_RFLEAN(AA, BB) template<> struct RfLean<AA*BB> { static constexpr bool has = true; static constexpr int A = AA, B = BB; };
PXS_RFLEAN(3, 2) PXS_RFLEAN(4, 2) PXS_RFLEAN(3, 3) PXS_RFLEAN(5, 2) PXS_RFLEAN(4, 3) PXS_RFLEAN(5, 3) PXS_RFLEAN(4, 4)

template<int NT, int PMAX> struct RegFft {
	using Regs = double2 (&)[PMAX];
	// LDS word of line index i: the index itself.  Measured on the C4 configuration (tools/tl_lab.sh, shader clocks per line): no padding
	// 265 000, one padding word per 16 -- the first form -- 296 000, the words of each block of 16 rotated by the block's number 292 000.
	// A bank model (ds_write_b64: 16 lanes x 16 banks, ds_read_b64: 32 x 32) gives the plain layout 3x the conflict-free write cycles and
	// the padded ones 1.25x, but with the plain layout every LDS address of a phase is one base register plus a compile-time offset
	// (base + s*j for the writes of a butterfly, tid + constant for the reads), and the address arithmetic of the other two -- ten integer
	// operations per point and exchange -- costs more than the conflicts.
#if defined(PXS_LAB_TL_PAD16)      /* lab builds: the other two layouts (the line area is sized for n + 32 words: pad16 on small lines only) */
	static __device__ __forceinline__ int pad(int i) { return i + (i >> 4); }
#elif defined(PXS_LAB_TL_ROT16)
	static __device__ __forceinline__ int pad(int i) { return (i & ~15) | ((i + (i >> 4)) & 15); }
#else
	static __device__ __forceinline__ int pad(int i) { return i; }
#endif

	// read pattern of pass PS: slot c = i*R + k holds x[tid + NT*i + nb*k] (threads with tid + NT*i >= nb: no element; they get f(0)).
	// Every slot of the pass is overwritten for every lane, so that the compiler sees the old contents die.
	template<class PS, class F> static __device__ __forceinline__ void fill(Regs v, int tid, F&& f) {
		static_assert(PS::slots <= PMAX, "pass needs more register slots than the kernel has");
		RF_OPAQUE(tid);
		sfor<0, PS::slots>([&](auto C) RF_INL {
			constexpr int c = RF_IDX(C), i = c/PS::R, k = c % PS::R;
			const bool ok = (i + 1)*NT <= PS::nb || tid < PS::nb - NT*i;
			v[c] = f(ok ? tid + (NT*i + PS::nb*k) : 0);
			RF_FENCE_SLOT(c);
		});
	}
	template<class PS, int COMP> static __device__ __forceinline__ void read_comp(Regs v, int tid, const double* line) {
		static_assert(PS::slots <= PMAX, "pass needs more register slots than the kernel has");
		RF_OPAQUE(tid);
		sfor<0, PS::slots>([&](auto C) RF_INL {
			constexpr int c = RF_IDX(C), i = c/PS::R, k = c % PS::R;
			const bool ok = (i + 1)*NT <= PS::nb || tid < PS::nb - NT*i;
			const double x = line[pad(ok ? tid + (NT*i + PS::nb*k) : 0)];
			if (COMP) v[c].y = x; else v[c].x = x;
			RF_FENCE_SLOT(c);
		});
	}
	// outputs of pass PS (output j of butterfly i in slot i*R + j) to their Stockham positions; COMP: 0 = re, 1 = im into a line of doubles
	template<class PS, int COMP> static __device__ __forceinline__ void write_comp(Regs v, int tid, double* line) {
		RF_OPAQUE(tid);
		sfor<0, PS::K>([&](auto I) RF_INL {
			constexpr int i = RF_IDX(I);
			const int b = tid + NT*i;
			if ((i + 1)*NT <= PS::nb || b < PS::nb) {
				const int base = PS::R*b - (PS::R - 1)*(b % PS::s);
				sfor<0, PS::R>([&](auto J) RF_INL { constexpr int j = RF_IDX(J); line[pad(base + PS::s*j)] = COMP ? v[i*PS::R + j].y : v[i*PS::R + j].x; });
			}
		});
	}
	// ... both components at once (the line area must hold n complex points)
	template<class PS> static __device__ __forceinline__ void write_c128(Regs v, int tid, double2* line) {
		RF_OPAQUE(tid);
		sfor<0, PS::K>([&](auto I) RF_INL {
			constexpr int i = RF_IDX(I);
			const int b = tid + NT*i;
			if ((i + 1)*NT <= PS::nb || b < PS::nb) {
				const int base = PS::R*b - (PS::R - 1)*(b % PS::s);
				sfor<0, PS::R>([&](auto J) RF_INL { constexpr int j = RF_IDX(J); line[pad(base + PS::s*j)] = v[i*PS::R + j]; });
			}
		});
	}
	// the butterflies of a pass and their Stockham twiddles (tw: this length's two-level table in the LDS)
	template<class PS> static __device__ __forceinline__ void compute(Regs v, int tid, const double2* tw) {
		constexpr int R = PS::R;
#ifdef PXS_LAB_TL_NOCOMP      /* lab builds: timing without the butterflies (wrong results) */
		return;
#endif
		RF_OPAQUE(tid);
		sfor<0, PS::K>([&](auto I) RF_INL {
			constexpr int i = RF_IDX(I);
			const int b = min(tid + NT*i, PS::nb - 1);      // (lanes past the last butterfly compute on what they hold; nothing reads it)
			double2 a[R];
			sfor<0, R>([&](auto J) RF_INL { constexpr int j = RF_IDX(J); a[j] = v[i*R + j]; });
			if constexpr (RfLean<R>::has && R >= PXS_RF_LEAN_MIN) {      // the composite radices: the register-lean form (above)
				double2 w1 = make_double2(1, 0);
				if constexpr (PS::twiddled) { const int e = b - b % PS::s; w1 = cmul(tw[e & (RF_TWL - 1)], tw[RF_TWL + (e >> 7)]); }
				rf_bfly_lean<RfLean<R>::A, RfLean<R>::B, PS::twiddled>(a, w1);
			} else {
			RfB<R>::run(a);
			if constexpr (PS::twiddled) {
				const int e = b - b % PS::s;
				const double2 w1 = cmul(tw[e & (RF_TWL - 1)], tw[RF_TWL + (e >> 7)]);
				if constexpr (PS::slots > 16) {      // a full register file (three radix-7 butterflies per thread): ONE chain of powers, tied to the outputs
					double2 wj = w1;
					sfor<1, R>([&](auto J) RF_INL {
						constexpr int j = RF_IDX(J);
						a[RfB<R>::slot(j)] = cmul(a[RfB<R>::slot(j)], wj);
						if constexpr (j + 1 < R) { wj = cmul(wj, w1); rf_pin_two(a[RfB<R>::slot(j)], wj); }
					});
				} else {
				// W^j, j = 1 ... R-1, as two chains (odd and even powers) that advance by W^2
				const double2 w2 = cmul(w1, w1);
				double2 wo = w1, we = w2;
#pragma unroll
				for (int j = 1; j < R; j++) {
					double2& x = a[RfB<R>::slot(j)];
					if (j & 1) { x = cmul(x, wo); if (j + 2 < R) wo = cmul(wo, w2); }
					else       { x = cmul(x, we); if (j + 2 < R) we = cmul(we, w2); }
				}
				}
			}
			}
			sfor<0, R>([&](auto J) RF_INL { constexpr int j = RF_IDX(J); v[i*R + j] = a[RfB<R>::slot(j)]; RF_PIN2(v[i*R + j]); });
			RF_FENCE();      // (the butterflies of a thread one after the other: interleaved they need their temporaries K times)
		});
	}
	// pointwise step between a transform that ends with pass PS and one that starts with the same radix: output j of butterfly (tid, i)
	// becomes input j of the same butterfly; f(value, line index)
	template<class PS, class F> static __device__ __forceinline__ void pointwise(Regs v, int tid, F&& f) {
		RF_OPAQUE(tid);
		sfor<0, PS::slots>([&](auto C) RF_INL {
			constexpr int c = RF_IDX(C), i = c/PS::R, j = c % PS::R;
			const int b = min(tid + NT*i, PS::nb - 1);
			v[c] = f(v[c], b + PS::nb*j);
			RF_FENCE_SLOT(c);
		});
	}

	// the registers hold the outputs of pass PP; they take the read pattern of pass PN (of the same or of another transform of the same length)
	// SYNC = false: the line belongs to ONE wave (NT = 64, `line` = the wave's own stretch of the LDS): the LDS executes a wave's
	// operations in order and no barrier is needed.  (A four-step form built on this -- a radix-16 pass across the 16 waves, then every
	// wave transforms its row of n / 16 points on its own, the waves drifting apart -- was measured and lost: wave 0 sees its private
	// exchanges at 3 000 clocks, but the LAST wave finishes after 7 600 per pass, the price of a barrier-separated pass, and the index
	// maps of the row layout cost the boundary steps more than the passes gained: 305 000 clocks per line against 272 000.
	// tools/xchg_probe2.hip keeps the measurement.)
	template<class PP, class PN, bool SYNC = true> static __device__ __forceinline__ void exchange(Regs v, int tid, double* line) {
#ifdef PXS_LAB_TL_NOXCHG      /* lab builds: timing without the LDS exchanges between the passes (wrong results) */
		return;
#endif
		if (SYNC) RF_BARRIER(); else RF_WAVE_SYNC();
		write_comp<PP, 0>(v, tid, line);
		if (SYNC) RF_BARRIER(); else RF_WAVE_SYNC();
		read_comp<PN, 0>(v, tid, line);
		if (SYNC) RF_BARRIER(); else RF_WAVE_SYNC();
		write_comp<PP, 1>(v, tid, line);
		if (SYNC) RF_BARRIER(); else RF_WAVE_SYNC();
		read_comp<PN, 1>(v, tid, line);
	}

	// forward transform of sequence S.  In: the registers hold the line in the read pattern of pass 0.  Out: they hold the outputs of the
	// last pass.  tw: this length's twiddle table in the LDS.
	template<class S, bool SYNC = true> static __device__ __forceinline__ void run(Regs v, int tid, double* line, const double2* tw) {
		sfor<0, S::NP>([&](auto P) RF_INL {
			constexpr int p = RF_IDX(P);
			using PS = RfPassT<S, p, NT>;
			compute<PS>(v, tid, tw);
			if constexpr (p + 1 < S::NP) exchange<PS, RfPassT<S, p + 1, NT>, SYNC>(v, tid, line);
		});
	}
};

} // namespace pxs
