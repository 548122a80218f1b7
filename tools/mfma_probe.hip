// Probe the lane layouts of the f64 MFMA instructions on gfx950 (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out, int which) {
	const int lane = threadIdx.x;
	const double pw = ldexp(1.0, lane % 52);   // distinct bit per lane (mod 52 to stay exact); second probe uses lane/52 split
	if (which == 0) {        // 4x4x4_4b: A = 2^lane, B = 1
		double d = __builtin_amdgcn_mfma_f64_4x4x4f64(pw, 1.0, 0.0, 0, 0, 0);
		out[lane] = d;
	} else if (which == 1) { // A = 1, B = 2^lane
		double d = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, pw, 0.0, 0, 0, 0);
		out[lane] = d;
	} else if (which == 2) { // 16x16x4: A = 2^lane, B = 1
		d4 c = {0, 0, 0, 0};
		d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(pw, 1.0, c, 0, 0, 0);
		for (int r = 0; r < 4; r++) out[lane*4+r] = d[r];
	} else {
		d4 c = {0, 0, 0, 0};
		d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, pw, c, 0, 0, 0);
		for (int r = 0; r < 4; r++) out[lane*4+r] = d[r];
	}
}
static void decode(double v) {
	uint64_t bits = (uint64_t)llround(v);
	printf("{");
	for (int b = 0; b < 52; b++) if (bits >> b & 1) printf("%d,", b);
	printf("}");
}
int main() {
	double* d; hipMalloc(&d, 256*8); double h[256];
	for (int which = 0; which < 4; which++) {
		hipMemset(d, 0, 256*8);
		hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, which);
		hipMemcpy(h, d, 256*8, hipMemcpyDeviceToHost);
		printf("== probe %d (lanes 0..51 carry bit=lane; lanes 52..63 alias bits 0..11)\n", which);
		int per = which < 2 ? 1 : 4;
		for (int lane = 0; lane < 64; lane++) {
			printf("lane %2d:", lane);
			for (int r = 0; r < per; r++) { printf(" "); decode(h[lane*per+r]); }
			printf("\n");
		}
	}
	return 0;
}
