// Round 6 probe 2: wave-private exchanges.  Each of the 16 waves of a workgroup owns a line of B points (64 lanes x <= 21 points) in
// its own stretch of the LDS and runs the exchanges of a radix sequence on it without any barrier; shader clocks per exchange of
// the WHOLE workgroup (16 lines of B points = as many points as one exchange of a 16 B-point line), with butterflies (mode 1)
// and without (mode 0).
#include "regfft_dev.hpp"
#include <cstdio>
#include <vector>
using namespace pxs;
#ifndef XP_SEQ
#define XP_SEQ 8, 2, 9, 7
#endif
using S = RfSeq<XP_SEQ>;
constexpr int NTW = 64;
template<class SS, int P = 0> constexpr int slots_of() { if constexpr (P >= SS::NP) return 0; else { constexpr int a = RfPassT<SS, P, NTW>::slots, b = slots_of<SS, P + 1>(); return a > b ? a : b; } }
constexpr int PMAX = slots_of<S>();
using F = RegFft<NTW, PMAX>;

__global__ __launch_bounds__(1024) void probe(double2* out, unsigned long long* clk, const double2* twg, int reps, int mode) {
	extern __shared__ __attribute__((aligned(16))) double2 lds[];
	double2* tw = lds;
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	double* line = (double*)(lds + 256) + w*(S::N + 16);
	for (int k = tid; k < 256; k += 1024) tw[k] = twg[k];
	double2 v[PMAX];
	sfor<0, PMAX>([&](auto C) RF_INL { v[RF_IDX(C)] = make_double2(tid + RF_IDX(C), tid - RF_IDX(C)); });
	__syncthreads();
	const unsigned long long t0 = clock64();
	for (int r = 0; r < reps; r++) {
		sfor<0, S::NP>([&](auto P) RF_INL {
			constexpr int p = RF_IDX(P);
			using PS = RfPassT<S, p, NTW>;
			if (mode == 1) F::template compute<PS>(v, lane, tw);
			if constexpr (p + 1 < S::NP) F::template exchange<PS, RfPassT<S, p + 1, NTW>, false>(v, lane, line);
		});
	}
	const unsigned long long t1 = clock64();
	__shared__ unsigned long long tend[16];
	if (lane == 0) tend[w] = t1 - t0;
	__syncthreads();
	if (tid == 0) { unsigned long long m = 0; for (int k = 0; k < 16; k++) m = tend[k] > m ? tend[k] : m; clk[blockIdx.x] = m; }      // the LAST wave's clock
	double2 acc = make_double2(0, 0);
	sfor<0, PMAX>([&](auto C) RF_INL { acc.x += v[RF_IDX(C)].x; acc.y += v[RF_IDX(C)].y; });
	out[blockIdx.x*1024 + tid] = acc;
}

int main() {
	const int nwg = 256, reps = 50;
	double2* out; unsigned long long* clk; double2* tw;
	(void)hipMalloc(&out, sizeof(double2)*nwg*1024); (void)hipMalloc(&clk, 8*nwg); (void)hipMalloc(&tw, 16*256); (void)hipMemset(tw, 0, 16*256);
	(void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024);
	const size_t lds = 16*256 + sizeof(double)*16*(S::N + 16);
	for (int mode = 0; mode < 2; mode++) {
		probe<<<nwg, 1024, lds>>>(out, clk, tw, reps, mode); (void)hipDeviceSynchronize();
		std::vector<unsigned long long> h(nwg); (void)hipMemcpy(h.data(), clk, 8*nwg, hipMemcpyDeviceToHost);
		double s = 0; for (auto x : h) s += (double)x;
		printf("16 waves x B = %d, %d passes, PMAX %d, %s: %.0f clocks per pass of the workgroup (the last wave's clock)\n", S::N, S::NP, PMAX, mode ? "butterflies + exchanges" : "exchanges only", s/nwg/reps/(S::NP - (mode ? 0 : 1)));
	}
	return 0;
}
