// Issue rate of v_mfma_f64_16x16x4_f64 on MI355X and what shares its pipe (round 5 probe for leg_ana_s0_mm).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/mfma_rate.bin ; one workgroup per CU, W waves per SIMD.
//   mode 0: MFMAs only (2 independent accumulators)          mode 1: v_fma_f64 only (8 independent chains)
//   mode 2: every wave alternates 8 MFMAs and 8 v_fma_f64     mode 3: half the waves do MFMAs, the other half v_fma_f64
//   mode 4: MFMAs with one DEPENDENT accumulator chain        mode 5: MFMA waves + waves doing v_add_u32 (32-bit VALU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void probe(int mode, int iters, double* out, long long* cyc) {
	const int wave = threadIdx.x >> 6;
	d4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
	double x = threadIdx.x*1e-3, y = 1.0 + blockIdx.x*1e-6;
	double f0 = x, f1 = x + 1, f2 = x + 2, f3 = x + 3, f4 = x + 4, f5 = x + 5, f6 = x + 6, f7 = x + 7;
	unsigned u0 = threadIdx.x, u1 = 1, u2 = 2, u3 = 3;
	const bool do_mfma = mode == 0 || mode == 2 || mode == 4 || ((mode == 3 || mode == 5) && (wave & 1) == 0);
	const bool do_fma = mode == 1 || mode == 2 || (mode == 3 && (wave & 1) == 1);
	const bool do_int = mode == 5 && (wave & 1) == 1;
	__syncthreads();
	const long long t0 = clock64();
	for (int it = 0; it < iters; it++) {
		if (do_mfma) {
			if (mode == 4) {
#pragma unroll
				for (int i = 0; i < 8; i++) a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
			} else {
#pragma unroll
				for (int i = 0; i < 4; i++) { a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0); }
			}
		}
		if (do_fma) {
			f0 = fma(f0, y, x); f1 = fma(f1, y, x); f2 = fma(f2, y, x); f3 = fma(f3, y, x);
			f4 = fma(f4, y, x); f5 = fma(f5, y, x); f6 = fma(f6, y, x); f7 = fma(f7, y, x);
		}
		if (do_int) {
#pragma unroll
			for (int i = 0; i < 8; i++) { u0 = u0*3u + u1; u1 += u2; u2 ^= u3; u3 += u0; }
		}
	}
	const long long t1 = clock64();
	out[blockIdx.x*blockDim.x + threadIdx.x] = a0[0] + a1[1] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + (double)(u0 + u1 + u2 + u3);
	if ((threadIdx.x & 63) == 0) cyc[blockIdx.x*16 + wave] = t1 - t0;
}
int main() {
	double* out; long long* cyc; hipMalloc(&out, 256*1024*8); hipMalloc(&cyc, 256*16*8);
	const int iters = 20000;
	for (int wps = 1; wps <= 4; wps *= 2) for (int mode = 0; mode <= 5; mode++) {
		const int nt = 256*wps;
		hipLaunchKernelGGL(probe, dim3(256), dim3(nt), 0, 0, mode, 100, out, cyc); hipDeviceSynchronize();
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		hipEventRecord(e0); hipLaunchKernelGGL(probe, dim3(256), dim3(nt), 0, 0, mode, iters, out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		long long h[256*16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
		double avg = 0; for (int b = 0; b < 256; b++) for (int w = 0; w < 4*wps; w++) avg += h[b*16 + w]; avg /= 256.0*4*wps;
		const double per_it = avg/iters;
		// per SIMD and iteration: wps waves, each (maybe) 8 MFMAs and / or 8 FMAs
		printf("waves/SIMD %d mode %d: %.1f cycles per iteration per wave (%.3f ms; clock %.2f GHz)", wps, mode, per_it, ms, avg/(ms*1e6));
		if (mode == 0 || mode == 4) printf("  -> %.1f cycles per MFMA per SIMD", per_it/(8.0*wps));
		if (mode == 1) printf("  -> %.1f cycles per v_fma_f64 per SIMD", per_it/(8.0*wps));
		if (mode == 2) printf("  -> per SIMD and (8 MFMA + 8 FMA): %.1f", per_it/wps);
		if (mode == 3 || mode == 5) printf("  -> per SIMD and pair (8 MFMA | 8 x %s): %.1f", mode == 3 ? "FMA" : "4 int ops", per_it/(wps/2.0 > 0.5 ? wps/2.0 : 1));
		printf("\n");
	}
	return 0;
}
