// v_fma_f64 issue rate on MI355X against waves per SIMD and independent chains per wave (round 5: do the Legendre kernels' 3 waves per
// SIMD cap them?).  hipcc --offload-arch=gfx950 -O3 tools/dp_rate.hip -o tools/dp_rate.bin ; workgroups of 256 W' threads, G per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template<int NCH> __global__ __launch_bounds__(1024) void probe(int iters, double* out, long long* cyc) {
	double f[NCH];
	const double x = threadIdx.x*1e-3, y = 1.0 + blockIdx.x*1e-9;
#pragma unroll
	for (int c = 0; c < NCH; c++) f[c] = x + c;
	__syncthreads();
	const long long t0 = clock64();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 64/NCH; r++) {
#pragma unroll
			for (int c = 0; c < NCH; c++) f[c] = fma(f[c], y, x);
		}
	}
	const long long t1 = clock64();
	double s = 0;
#pragma unroll
	for (int c = 0; c < NCH; c++) s += f[c];
	out[blockIdx.x*blockDim.x + threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) cyc[blockIdx.x*16 + (threadIdx.x >> 6)] = t1 - t0;
}
template<int NCH> static void run(int wps, double* out, long long* cyc) {
	const int iters = 4000;
	// wps waves per SIMD = 4 wps waves per CU: one workgroup of 256 wps threads up to 4, two of 128 wps beyond
	const int g = wps > 4 ? 2 : 1, nt = 256*wps/g, nb = 256*g;
	hipLaunchKernelGGL(probe<NCH>, dim3(nb), dim3(nt), 0, 0, 50, out, cyc); hipDeviceSynchronize();
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0); hipLaunchKernelGGL(probe<NCH>, dim3(nb), dim3(nt), 0, 0, iters, out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	static long long h[512*16]; hipMemcpy(h, cyc, sizeof(long long)*nb*16, hipMemcpyDeviceToHost);
	double avg = 0; const int wpb = nt/64; for (int b = 0; b < nb; b++) for (int w = 0; w < wpb; w++) avg += h[b*16 + w]; avg /= (double)nb*wpb;
	const double fmas = 64.0*iters;                      // per wave
	const double tf = 2.0*64*fmas*(1024.0*wps)/(ms*1e-3)/1e12;
	printf("waves/SIMD %d chains %2d: %6.2f cycles per v_fma_f64 per wave, %5.2f per SIMD; %7.3f ms, %5.1f TFLOP/s, clock %.2f GHz\n", wps, NCH, avg/fmas, avg/fmas/wps, ms, tf, avg/(ms*1e6));
}
int main() {
	double* out; long long* cyc; hipMalloc(&out, 512*1024*8); hipMalloc(&cyc, 512*16*8);
	for (int wps = 1; wps <= 8; wps++) {
		if (wps == 7) continue;
		run<1>(wps, out, cyc); run<2>(wps, out, cyc); run<4>(wps, out, cyc); run<8>(wps, out, cyc); run<16>(wps, out, cyc);
	}
	return 0;
}
