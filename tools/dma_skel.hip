// Skeleton experiment for the chain kernels: what does the tile-synchronous structure (load tile -> LDS -> barrier -> store) cost
// against a plain copy, and does an LDS-DMA double-buffered persistent workgroup recover it?  Contiguous 40 KB tiles both sides.
// hipcc --offload-arch=gfx950 -O3 tools/dma_skel.hip -o tools/dma_skel.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int PTS = 2560, NT = 512, NE = PTS/NT;
typedef double2 d2;

__global__ __launch_bounds__(256) void k_copy(const d2* __restrict__ in, d2* __restrict__ out, long n) {
	for (long i = (long)blockIdx.x*256 + threadIdx.x; i < n; i += (long)gridDim.x*256) out[i] = in[i];
}
// (a) one tile per workgroup: loads -> LDS -> barrier -> LDS -> stores
__global__ __launch_bounds__(NT) void k_tile(const d2* __restrict__ in, d2* __restrict__ out, int rot) {
	extern __shared__ d2 lds[];
	const long base = (long)blockIdx.x*PTS;
	d2 v[NE];
#pragma unroll
	for (int u = 0; u < NE; u++) v[u] = in[base + threadIdx.x + u*NT];
#pragma unroll
	for (int u = 0; u < NE; u++) lds[(threadIdx.x + u*NT + rot) % PTS] = v[u];
	__syncthreads();
#pragma unroll
	for (int u = 0; u < NE; u++) out[base + threadIdx.x + u*NT] = lds[threadIdx.x + u*NT];
}
// (b) persistent workgroup, two LDS buffers, next tile fetched by LDS-DMA while the current one is stored
__device__ __forceinline__ void dma16(const d2* g, d2* l) { __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0); }
__global__ __launch_bounds__(NT) void k_dma(const d2* __restrict__ in, d2* __restrict__ out, int ntile, int work) {
	extern __shared__ d2 lds[];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	auto fetch = [&](int tile, d2* b) {
		const long base = (long)tile*PTS;
#pragma unroll
		for (int u = 0; u < NE; u++) { const int idx = (u*(NT/64) + wave)*64; dma16(in + base + idx + lane, b + idx); }
	};
	int t = blockIdx.x, cur = 0;
	if (t < ntile) fetch(t, lds);
	for (; t < ntile; t += gridDim.x, cur ^= 1) {
		const int nxt = t + gridDim.x;
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		d2* bc = lds + cur*PTS;
		if (nxt < ntile) fetch(nxt, lds + (cur ^ 1)*PTS);
		const long base = (long)t*PTS;
		d2 v[NE];
#pragma unroll
		for (int u = 0; u < NE; u++) v[u] = bc[threadIdx.x + u*NT];
		for (int w = 0; w < work; w++) {	// stand-in for the LDS passes: w rounds of LDS write + barrier + read
#pragma unroll
			for (int u = 0; u < NE; u++) bc[(threadIdx.x + u*NT + 1) % PTS] = v[u];
			__syncthreads();
#pragma unroll
			for (int u = 0; u < NE; u++) v[u] = bc[threadIdx.x + u*NT];
		}
#pragma unroll
		for (int u = 0; u < NE; u++) out[base + threadIdx.x + u*NT] = v[u];
	}
}
// (a') as (a) with the same stand-in work
__global__ __launch_bounds__(NT) void k_tile_work(const d2* __restrict__ in, d2* __restrict__ out, int work) {
	extern __shared__ d2 lds[];
	const long base = (long)blockIdx.x*PTS;
	d2 v[NE];
#pragma unroll
	for (int u = 0; u < NE; u++) v[u] = in[base + threadIdx.x + u*NT];
	for (int w = 0; w <= work; w++) {
#pragma unroll
		for (int u = 0; u < NE; u++) lds[(threadIdx.x + u*NT + 1) % PTS] = v[u];
		__syncthreads();
#pragma unroll
		for (int u = 0; u < NE; u++) v[u] = lds[threadIdx.x + u*NT];
		__syncthreads();
	}
#pragma unroll
	for (int u = 0; u < NE; u++) out[base + threadIdx.x + u*NT] = v[u];
}
template<class F> static double timeit(F f, int reps = 5) {
	f(); CK(hipDeviceSynchronize());
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	CK(hipEventRecord(a)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
	float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms/reps;
}
int main() {
	const int ntile = 200000; const long n = (long)ntile*PTS; const double gb = 2.0*n*16/1e9;
	d2 *in, *out; CK(hipMalloc(&in, n*16)); CK(hipMalloc(&out, n*16)); CK(hipMemset(in, 1, n*16));
	CK(hipFuncSetAttribute((const void*)k_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 256));
	CK(hipFuncSetAttribute((const void*)k_tile, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 256));
	CK(hipFuncSetAttribute((const void*)k_tile_work, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 256));
	double ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(256*16), dim3(256), 0, 0, in, out, n); });
	printf("plain copy                      %7.3f ms  %6.2f TB/s\n", ms, gb/ms);
	for (int ldskb : {40, 50, 64, 80}) {
		ms = timeit([&] { hipLaunchKernelGGL(k_tile, dim3(ntile), dim3(NT), ldskb*1024, 0, in, out, 0); });
		printf("tile per WG, %2d KB LDS (%d WG/CU) %7.3f ms  %6.2f TB/s\n", ldskb, 160/ldskb, ms, gb/ms);
	}
	for (int work : {0, 2, 4, 8}) {
		ms = timeit([&] { hipLaunchKernelGGL(k_tile_work, dim3(ntile), dim3(NT), 50*1024, 0, in, out, work); });
		printf("tile per WG, 50 KB, work %d      %7.3f ms  %6.2f TB/s\n", work, ms, gb/ms);
	}
	for (int wgs : {1, 2}) for (int work : {0, 2, 4, 8}) {
		const int ldsb = wgs == 1 ? 2*PTS*16 + 4096 : 2*PTS*16;      // 1 or (at 80 KB) 2 workgroups per CU
		ms = timeit([&] { hipLaunchKernelGGL(k_dma, dim3(256*wgs), dim3(NT), wgs == 1 ? 100*1024 : 80*1024, 0, in, out, ntile, work); });
		(void)ldsb;
		printf("DMA double buffer, %d WG/CU, work %d %7.3f ms  %6.2f TB/s\n", wgs, work, ms, gb/ms);
	}
	return 0;
}
