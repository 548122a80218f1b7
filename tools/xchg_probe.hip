// Round 6 probe: what one LDS exchange of the line FFT (regfft_dev.hpp) costs, in isolation -- a workgroup of 1024 threads runs the
// exchanges of a radix sequence back to back on register data (no butterflies), shader clocks per exchange.  Variants by -D:
//   XP_SEQ=8,8,6,6,7   the sequence;   XP_NOBAR: without the barriers (wrong results; LDS throughput alone)
// build + run on the GPU box:  hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ipixell_amd/csrc tools/xchg_probe.hip -o /tmp/xp && /tmp/xp
#include "regfft_dev.hpp"
#include <cstdio>
#include <vector>
using namespace pxs;
#ifndef XP_SEQ
#define XP_SEQ 8, 8, 6, 6, 7
#endif
#ifdef XP_NOBAR
#undef RF_BARRIER
#define RF_BARRIER() do {} while (0)
#endif
using S = RfSeq<XP_SEQ>;
constexpr int NT = 1024;
template<class SS, int P = 0> constexpr int slots_of() { if constexpr (P >= SS::NP) return 0; else { constexpr int a = RfPassT<SS, P, NT>::slots, b = slots_of<SS, P + 1>(); return a > b ? a : b; } }
constexpr int PMAX = slots_of<S>();
using F = RegFft<NT, PMAX>;

__global__ __launch_bounds__(1024) void probe(double2* out, unsigned long long* clk, int reps, int mode) {
	extern __shared__ __attribute__((aligned(16))) double2 lds[];
	double* line = (double*)lds;
	const int tid = threadIdx.x;
	double2 v[PMAX];
	sfor<0, PMAX>([&](auto C) RF_INL { v[RF_IDX(C)] = make_double2(tid + RF_IDX(C), tid - RF_IDX(C)); });
	__syncthreads();
	const unsigned long long t0 = clock64();
	for (int r = 0; r < reps; r++) {
		sfor<0, S::NP - 1>([&](auto P) RF_INL {
			constexpr int p = RF_IDX(P);
			using PS = RfPassT<S, p, NT>; using PN = RfPassT<S, p + 1, NT>;
			if (mode == 0) F::template exchange<PS, PN>(v, tid, line);
			else if (mode == 1) { RF_BARRIER(); F::template write_comp<PS, 0>(v, tid, line); RF_BARRIER(); F::template write_comp<PS, 1>(v, tid, line); }      // writes only
			else { RF_BARRIER(); F::template read_comp<PN, 0>(v, tid, line); RF_BARRIER(); F::template read_comp<PN, 1>(v, tid, line); }                        // reads only
		});
	}
	const unsigned long long t1 = clock64();
	if (tid == 0) clk[blockIdx.x] = t1 - t0;
	double2 acc = make_double2(0, 0);
	sfor<0, PMAX>([&](auto C) RF_INL { acc.x += v[RF_IDX(C)].x; acc.y += v[RF_IDX(C)].y; });
	out[blockIdx.x*NT + tid] = acc;
}

int main() {
	const int nwg = 256, reps = 50;
	double2* out; unsigned long long* clk;
	hipMalloc(&out, sizeof(double2)*nwg*NT); hipMalloc(&clk, 8*nwg);
	hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024);
	const size_t lds = sizeof(double)*(S::N + 64);
	const char* names[3] = {"exchange (write re, read re, write im, read im; 4 barriers)", "writes only (2 barriers)", "reads only (2 barriers)"};
	for (int mode = 0; mode < 3; mode++) {
		probe<<<nwg, NT, lds>>>(out, clk, reps, mode); hipDeviceSynchronize();
		std::vector<unsigned long long> h(nwg); hipMemcpy(h.data(), clk, 8*nwg, hipMemcpyDeviceToHost);
		double s = 0; for (auto x : h) s += (double)x;
		printf("n = %d, %d passes, PMAX %d, %s: %.0f clocks per exchange\n", S::N, S::NP, PMAX, names[mode], s/nwg/reps/(S::NP - 1));
	}
	return 0;
}
