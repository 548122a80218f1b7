// lane layout of v_mfma_f64_4x4x4_4b_f64 and v_mfma_f64_16x16x4_f64 on gfx950, found by experiment:
// D = A*B with A = one-hot at lane la, B = one-hot at lane lb, for every (la, lb): which lane of D lights up?
// hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o tools/mfma_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe4(int la, int lb, double* out) {
	const int l = threadIdx.x;
	double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0, c = 0.0;
	double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
	out[l] = d;
}
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe16(int la, int lb, double* out) {
	const int l = threadIdx.x;
	double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0; d4 c = {0, 0, 0, 0};
	d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
	for (int r = 0; r < 4; r++) out[r*64 + l] = d[r];
}
int main() {
	double* d; hipMalloc(&d, 256*8); std::vector<double> h(256);
	printf("4x4x4_4b: (lane of A, lane of B) -> lanes of D that are 1\n");
	for (int la = 0; la < 64; la++) {
		printf("A@%2d:", la);
		for (int lb = 0; lb < 64; lb++) {
			hipLaunchKernelGGL(probe4, dim3(1), dim3(64), 0, 0, la, lb, d); hipMemcpy(h.data(), d, 64*8, hipMemcpyDeviceToHost);
			for (int l = 0; l < 64; l++) if (h[l] != 0) printf(" B@%d->D@%d", lb, l);
		}
		printf("\n");
	}
	printf("16x16x4: (lane of A, lane of B) -> (reg, lane) of D\n");
	for (int la = 0; la < 64; la += 1) {
		printf("A@%2d:", la);
		for (int lb = 0; lb < 64; lb++) {
			hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, 0, la, lb, d); hipMemcpy(h.data(), d, 256*8, hipMemcpyDeviceToHost);
			for (int i = 0; i < 256; i++) if (h[i] != 0) printf(" B@%d->D[%d]@%d", lb, i/64, i%64);
		}
		printf("\n");
	}
	return 0;
}
