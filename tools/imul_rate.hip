// issue rate of the 32-bit integer multiplies next to an add on MI355X (the index arithmetic of the chain kernels): cycles per wave instruction
// hipcc --offload-arch=gfx950 -O3 tools/imul_rate.hip -o tools/imul_rate.bin && tools/imul_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
template<int OP> __global__ __launch_bounds__(256) void k(unsigned* out, unsigned a, unsigned b, int iters) {
	unsigned x0 = threadIdx.x + a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	for (int i = 0; i < iters; i++) {
#define STEP(x) \
		if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b)); \
		else if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b)); \
		else if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(b)); \
		else if (OP == 3) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b)); \
		else if (OP == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x) : "v"(b)); \
		else asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(unsigned long long*)&x##_w) : "v"(x), "v"(b) : "vcc");
		unsigned long long x0_w = 0, x1_w = 0, x2_w = 0, x3_w = 0, x4_w = 0, x5_w = 0, x6_w = 0, x7_w = 0;
		STEP(x0) STEP(x1) STEP(x2) STEP(x3) STEP(x4) STEP(x5) STEP(x6) STEP(x7)
		x0 += (unsigned)x0_w; x1 += (unsigned)x1_w; x2 += (unsigned)x2_w; x3 += (unsigned)x3_w; x4 += (unsigned)x4_w; x5 += (unsigned)x5_w; x6 += (unsigned)x6_w; x7 += (unsigned)x7_w;
	}
	out[blockIdx.x*blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
template<int OP> void run(const char* name, unsigned* d) {
	const int iters = 20000, blocks = 256*8;      // 8 workgroups of 4 waves per CU: 8 waves per SIMD
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u, 100);
	hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u, iters); hipEventRecord(e1); hipEventSynchronize(e1);
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	const double winst = (double)blocks*4*iters*8;            // wave instructions of the measured op (the loop overhead adds ~2 per 8)
	const double per_simd = winst/(256.0*4);                  // per SIMD
	printf("%-16s %8.3f ms  %.2f cycles per wave instruction at 2.4 GHz\n", name, ms, ms*1e-3*2.4e9/per_simd);
}
int main() {
	unsigned* d; hipMalloc(&d, 256*8*256*4);
	run<0>("v_add_u32", d); run<1>("v_mul_lo_u32", d); run<2>("v_mul_hi_u32", d); run<3>("v_mul_u32_u24", d); run<4>("v_mad_u32_u24", d); run<5>("v_mad_u64_u32", d);
	return 0;
}
