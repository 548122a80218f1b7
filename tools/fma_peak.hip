// Sustained FP64 FMA rate of the device (what the 78.6 TFLOP/s spec figure becomes under a long all-CU FMA load):
// every lane runs NCH independent v_fma_f64 chains with wave-uniform (SGPR) multiplicands, like the Legendre kernels.
// hipcc --offload-arch=gfx950 -O3 tools/fma_peak.hip -o /tmp/fma_peak && /tmp/fma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template<int NCH> __global__ __launch_bounds__(64) void fma_kernel(double* out, const double* coef, int iters) {
	double acc[NCH];
	for (int i = 0; i < NCH; i++) acc[i] = threadIdx.x*1e-3 + i;
	const double a = coef[0], b = coef[1];      // wave-uniform
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int u = 0; u < 8; u++)
#pragma unroll
			for (int i = 0; i < NCH; i++) acc[i] = fma(acc[i], a, b);
	}
	double s = 0; for (int i = 0; i < NCH; i++) s += acc[i];
	out[blockIdx.x*64 + threadIdx.x] = s;
}
// the same with three per-lane operands, as in the analysis accumulation t = fma(lambda, data, t)
template<int NCH> __global__ __launch_bounds__(64) void fma3_kernel(double* out, const double* coef, int iters) {
	double acc[NCH], x[NCH], y[NCH];
	for (int i = 0; i < NCH; i++) { acc[i] = threadIdx.x*1e-3 + i; x[i] = 1.0 - 1e-9*(threadIdx.x + i); y[i] = 1.0 + 1e-9*(threadIdx.x - i) + coef[0]*1e-12; }
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int u = 0; u < 8; u++)
#pragma unroll
			for (int i = 0; i < NCH; i++) acc[i] = fma(x[i], y[(i + u) % NCH], acc[i]);
	}
	double s = 0; for (int i = 0; i < NCH; i++) s += acc[i];
	out[blockIdx.x*64 + threadIdx.x] = s;
}
int main() {
	const int nblk = 256*4*8, iters = 20000; constexpr int NCH = 16;
	double *out, *coef; hipMalloc(&out, sizeof(double)*nblk*64); hipMalloc(&coef, 16);
	double h[2] = {0.999999, 1e-9}; hipMemcpy(coef, h, 16, hipMemcpyHostToDevice);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int rep = 0; rep < 6; rep++) {
		hipEventRecord(e0); hipLaunchKernelGGL(fma_kernel<NCH>, dim3(nblk), dim3(64), 0, 0, out, coef, iters); hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		const double flops = 2.0*nblk*64.0*NCH*8.0*iters;
		printf("rep %d: %.2f ms, %.1f TFLOP/s FP64 FMA (%.0f%% of 78.6)\n", rep, ms, flops/ms*1e-9, flops/ms*1e-9/78.6*100);
	}
	for (int rep = 0; rep < 4; rep++) {
		hipEventRecord(e0); hipLaunchKernelGGL(fma3_kernel<NCH>, dim3(nblk), dim3(64), 0, 0, out, coef, iters); hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		const double flops = 2.0*nblk*64.0*NCH*8.0*iters;
		printf("3 VGPR operands, rep %d: %.2f ms, %.1f TFLOP/s (%.0f%% of 78.6)\n", rep, ms, flops/ms*1e-9, flops/ms*1e-9/78.6*100);
	}
	return 0;
}
