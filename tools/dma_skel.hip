// Skeleton experiment for the chain kernels (round 4, corrected): does prefetching the next tile by LDS-DMA beat keeping three one-tile
// workgroups per CU, at EQUAL LDS / VALU work per tile?  The round-3 probe could not overlap anything: hipcc put `s_waitcnt vmcnt(0)`
// straight after the prefetch (every __syncthreads() drains an LDS-DMA), and its variants did unequal work.  Here
//   * ONE dynamic __shared__ array, raw s_barrier with lgkmcnt-only waits (BAR), and the DMA waited for exactly where the tile is needed;
//   * every variant does `work` rounds of (ds_read_b128 x NE, FL fused multiply-adds per point, ds_write_b128 x NE, barrier) per tile:
//     round 0 reads the landed tile and writes the work buffer, the others work in place -- the shape of a chain stage's radix passes;
//   * variants: (A) one tile per workgroup, loads through registers, three workgroups per CU (the structure of chain_kernel);
//               (B) persistent, every wave issues its share of the next tile's DMA after round 0 and waits vmcnt(0) at the top of the next
//                   tile -- which also waits for the wave's own stores of the previous tile;
//               (C) persistent with a LOADER wave: one extra wave per workgroup issues the whole DMA and waits for it, the eight
//                   compute waves never wait on vmcnt (their stores are fire-and-forget); the loader takes part in every barrier and
//                   does its vmcnt(0) just before the top barrier of the next tile.
// Contiguous 36 KB tiles on both sides (the chain stages' strided side runs at the same speed, tools/stride_bw.hip).
// hipcc --offload-arch=gfx950 -O3 tools/dma_skel.hip -o tools/dma_skel.bin ; the .s must show no vmcnt(0) between the DMA issue and the
// ds_reads of the current tile in (B) / none at all in the compute branch of (C): tools/dma_skel.sh checks it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int PTS = 2304, NT = 512, NE = (PTS + NT - 1)/NT;        // 36 KB tiles: input buffer + work buffer = 72 KB -> two workgroups per CU
typedef double2 d2;
#ifndef FL
#define FL 6      /* FMAs per point and round (a radix-8 butterfly with twiddles is ~11 f64 operations per point) */
#endif
#define BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
__device__ __forceinline__ void dma16(const d2* g, d2* l) { __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0); }

// one round: every thread moves its NE points from src to the same slots of dst (src == dst: in place), FL FMA pairs on the way
__device__ __forceinline__ void round_(const d2* src, d2* dst, int tid, double c) {
	d2 v[NE];
#pragma unroll
	for (int u = 0; u < NE; u++) { const int idx = tid + u*NT; v[u] = idx < PTS ? src[idx] : make_double2(0, 0); }
#pragma unroll
	for (int u = 0; u < NE; u++)
#pragma unroll
		for (int f = 0; f < FL; f++) { v[u].x = fma(v[u].y, c, v[u].x); v[u].y = fma(v[u].x, -c, v[u].y); }
#pragma unroll
	for (int u = 0; u < NE; u++) { const int idx = tid + u*NT; if (idx < PTS) dst[idx] = v[u]; }
}

__global__ __launch_bounds__(256) void k_copy(const d2* __restrict__ in, d2* __restrict__ out, long n) {
	for (long i = (long)blockIdx.x*256 + threadIdx.x; i < n; i += (long)gridDim.x*256) out[i] = in[i];
}
// (A) one tile per workgroup
__global__ __launch_bounds__(NT) void k_tile(const d2* __restrict__ in, d2* __restrict__ out, int work, double c) {
	extern __shared__ d2 lds[];
	const long base = (long)blockIdx.x*PTS;
	const int tid = threadIdx.x;
	d2 v[NE];
#pragma unroll
	for (int u = 0; u < NE; u++) { const int idx = tid + u*NT; v[u] = idx < PTS ? in[base + idx] : make_double2(0, 0); }
#pragma unroll
	for (int u = 0; u < NE; u++) { const int idx = tid + u*NT; if (idx < PTS) lds[idx] = v[u]; }      // (the register detour of chain_kernel's load)
	BAR();
	for (int r = 0; r < work; r++) { round_(lds, lds, tid, c); BAR(); }
#pragma unroll
	for (int u = 0; u < NE; u++) { const int idx = tid + u*NT; if (idx < PTS) out[base + idx] = lds[idx]; }
}
// (B) persistent, all waves issue the DMA; LDS: in[PTS] | w[PTS]
__global__ __launch_bounds__(NT) void k_dma_all(const d2* __restrict__ in, d2* __restrict__ out, int ntile, int work, double c) {
	extern __shared__ d2 lds[];
	d2* A = lds; d2* W = lds + PTS;
	const int tid = threadIdx.x, lane = tid & 63;
	auto fetch = [&](int tile) { for (int i0 = tid; i0 < PTS; i0 += NT) dma16(in + (long)tile*PTS + i0, A + (i0 - lane)); };
	int t = blockIdx.x;
	if (t < ntile) fetch(t);
	for (; t < ntile; t += gridDim.x) {
		WAIT_VM0();
		BAR();
		round_(A, W, tid, c);
		BAR();
		if (t + (int)gridDim.x < ntile) fetch(t + gridDim.x);           // the input buffer is free
		for (int r = 1; r < work; r++) { round_(W, W, tid, c); BAR(); }
		const long base = (long)t*PTS;
#pragma unroll
		for (int u = 0; u < NE; u++) { const int idx = tid + u*NT; if (idx < PTS) out[base + idx] = W[idx]; }
	}
}
// (C) persistent with a loader wave (the last one): NT compute threads + 64
__global__ __launch_bounds__(NT + 64) void k_dma_loader(const d2* __restrict__ in, d2* __restrict__ out, int ntile, int work, double c) {
	extern __shared__ d2 lds[];
	d2* A = lds; d2* W = lds + PTS;
	const int tid = threadIdx.x, lane = tid & 63;
	if (tid >= NT) {      // ---- loader: the whole tile, 64 x 16 bytes per instruction
		auto fetch = [&](int tile) { for (int i0 = lane; i0 < PTS; i0 += 64) dma16(in + (long)tile*PTS + i0, A + (i0 - lane)); };
		int t = blockIdx.x;
		if (t < ntile) fetch(t);
		for (; t < ntile; t += gridDim.x) {
			WAIT_VM0();
			BAR();                                     // top: the tile has landed
			BAR();                                     // round 0 done: the input buffer is free
			if (t + (int)gridDim.x < ntile) fetch(t + gridDim.x);
			for (int r = 1; r < work; r++) BAR();
		}
		return;
	}
	for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
		BAR();
		round_(A, W, tid, c);
		BAR();
		for (int r = 1; r < work; r++) { round_(W, W, tid, c); BAR(); }
		const long base = (long)t*PTS;
#pragma unroll
		for (int u = 0; u < NE; u++) { const int idx = tid + u*NT; if (idx < PTS) out[base + idx] = W[idx]; }
	}
}
template<class F> static double timeit(F f, int reps = 5) {
	f(); CK(hipDeviceSynchronize());
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	CK(hipEventRecord(a)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
	float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms/reps;
}
int main() {
	const int ntile = 220000; const long n = (long)ntile*PTS; const double gb = 2.0*n*16/1e9;
	d2 *in, *out; CK(hipMalloc(&in, n*16)); CK(hipMalloc(&out, n*16)); CK(hipMemset(in, 0, n*16));
	for (auto k : {(const void*)k_tile, (const void*)k_dma_all, (const void*)k_dma_loader}) CK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 256));
	double ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(256*16), dim3(256), 0, 0, in, out, n); });
	printf("tiles of %d points (%.1f KB), %d FMA pairs per point and round\n", PTS, PTS*16/1024.0, FL);
	printf("plain copy                                      %7.3f ms  %5.2f TB/s\n", ms, gb/ms);
	const double c = 1e-9;
	for (int work : {0, 2, 4, 6, 8, 10}) {
		for (int wgs : {3, 2}) {
			ms = timeit([&] { hipLaunchKernelGGL(k_tile, dim3(ntile), dim3(NT), (size_t)(wgs == 3 ? 52 : 78)*1024, 0, in, out, work, c); });
			printf("work %2d  (A) tile per WG, %d WG/CU                %7.3f ms  %5.2f TB/s\n", work, wgs, ms, gb/ms);
		}
		if (work == 0) continue;
		ms = timeit([&] { hipLaunchKernelGGL(k_dma_all, dim3(256*2), dim3(NT), (size_t)2*PTS*16, 0, in, out, ntile, work, c); });
		printf("work %2d  (B) persistent, DMA by all waves, 2 WG/CU %7.3f ms  %5.2f TB/s\n", work, ms, gb/ms);
		ms = timeit([&] { hipLaunchKernelGGL(k_dma_loader, dim3(256*2), dim3(NT + 64), (size_t)2*PTS*16, 0, in, out, ntile, work, c); });
		printf("work %2d  (C) persistent, loader wave, 2 WG/CU      %7.3f ms  %5.2f TB/s\n", work, ms, gb/ms);
	}
	return 0;
}
