// Round 6 probe 3: what a workgroup barrier costs on MI355X with 16 waves (1024 threads) per workgroup, one workgroup per CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(1024) void probe(unsigned long long* clk, double* out, int reps, int mode) {
	__shared__ double buf[16512];
	const int tid = threadIdx.x;
	double acc = tid;
	__syncthreads();
	const unsigned long long t0 = clock64();
	for (int r = 0; r < reps; r++) {
		if (mode == 0) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
		else if (mode == 1) { __syncthreads(); }
		else if (mode == 2) {	// 16 ds_write_b64 per thread (one component of a 16 384-point line), barrier, 16 ds_read_b64, barrier
#pragma unroll
			for (int k = 0; k < 16; k++) buf[tid + 1024*k] = acc + k;
			asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
			for (int k = 0; k < 16; k++) acc += buf[((tid + 17) & 1023) + 1024*k];
			asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
		} else {	// the same without barriers, each wave on its own 1024 words
			double* b = buf + (tid >> 6)*1024; const int l = tid & 63;
#pragma unroll
			for (int k = 0; k < 16; k++) b[l + 64*k] = acc + k;
#pragma unroll
			for (int k = 0; k < 16; k++) acc += b[((l + 17) & 63) + 64*k];
		}
	}
	const unsigned long long t1 = clock64();
	if (tid == 0) clk[blockIdx.x] = t1 - t0;
	out[blockIdx.x*1024 + tid] = acc;
}
int main() {
	const int nwg = 256, reps = 200;
	unsigned long long* clk; double* out; (void)hipMalloc(&clk, 8*nwg); (void)hipMalloc(&out, 8*nwg*1024);
	const char* nm[4] = {"s_waitcnt lgkmcnt(0) + s_barrier", "__syncthreads()", "16 writes, barrier, 16 reads, barrier (8-byte, conflict-free)", "16 writes, 16 reads, wave-private, no barrier"};
	for (int mode = 0; mode < 4; mode++) {
		probe<<<nwg, 1024>>>(clk, out, reps, mode); (void)hipDeviceSynchronize();
		std::vector<unsigned long long> h(nwg); (void)hipMemcpy(h.data(), clk, 8*nwg, hipMemcpyDeviceToHost);
		double s = 0; for (auto x : h) s += (double)x;
		printf("%-64s %8.0f clocks per iteration\n", nm[mode], s/nwg/reps);
	}
	return 0;
}
