// Microbenchmark (MI355X): HBM throughput of tiles made of short contiguous runs at a large stride -- the access
// pattern of the strided side of a four-step FFT pass -- as a function of the run length.
//   mode 0: contiguous read, contiguous write (copy baseline)
//   mode 1: read  nrun runs of R bytes at stride S (one tile per workgroup), write the tile contiguously
//   mode 2: read the tile contiguously, write nrun runs of R bytes at stride S
//   mode 3: both sides strided
// build: hipcc --offload-arch=gfx950 -O3 tools/stride_bw.hip -o tools/stride_bw ; run: tools/stride_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// line = N points of 16 B; tile t of a line covers points j2 in [t*T, (t+1)*T) for all j1 in [0, n1): address (j1*n2 + j2)
template<int UNR> __global__ __launch_bounds__(256) void k(const double2* __restrict__ in, double2* __restrict__ out, int n1, int n2, int T, int mode, long ntile_line)
{
	const long tile = blockIdx.x;
	const long line = tile / ntile_line, t = tile - line*ntile_line;
	const long N = (long)n1*n2;
	const double2* src = in + line*N; double2* dst = out + line*N;
	const int total = n1*T;
	for (int i0 = threadIdx.x; i0 < total; i0 += 256*UNR) {
		double2 v[UNR];
#pragma unroll
		for (int u = 0; u < UNR; u++) {
			const int i = i0 + u*256;
			if (i < total) {
				const int j1 = i / T, jj = i - j1*T;
				const long a_str = (long)j1*n2 + t*T + jj, a_con = t*(long)total + i;
				v[u] = src[(mode & 1) ? a_str : a_con];
			}
		}
#pragma unroll
		for (int u = 0; u < UNR; u++) {
			const int i = i0 + u*256;
			if (i < total) {
				const int j1 = i / T, jj = i - j1*T;
				const long a_str = (long)j1*n2 + t*T + jj, a_con = t*(long)total + i;
				dst[(mode & 2) ? a_str : a_con] = v[u];
			}
		}
	}
}

int main(int argc, char** argv) {
	const long N = 43200; const int n1 = 200, n2 = 216;
	const long lines = 12000;                     // 8.3 GB per array
	double2 *a, *b; CK(hipMalloc(&a, lines*N*16)); CK(hipMalloc(&b, lines*N*16));
	CK(hipMemset(a, 1, lines*N*16)); CK(hipMemset(b, 0, lines*N*16));
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int Ts[] = {8, 12, 24, 36, 72, 216};
	for (int mode = 0; mode < 4; mode++)
	for (int T : Ts) {
		if (mode == 0 && T != 8) continue;
		const long ntile_line = n2/T;
		const long nblk = lines*ntile_line;
		for (int unr : {4, 8}) {
			float best = 1e9;
			for (int rep = 0; rep < 3; rep++) {
				hipEventRecord(e0);
				if (unr == 4) hipLaunchKernelGGL(k<4>, dim3(nblk), dim3(256), 0, 0, a, b, n1, n2, T, mode, ntile_line);
				else          hipLaunchKernelGGL(k<8>, dim3(nblk), dim3(256), 0, 0, a, b, n1, n2, T, mode, ntile_line);
				hipEventRecord(e1); CK(hipEventSynchronize(e1));
				float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
			}
			printf("mode %d  run %5d B (T=%3d, tile %6.1f KB)  unroll %d : %7.3f ms  %6.2f TB/s (read+write)\n", mode, T*16, T, n1*T*16/1024.0, unr, best, 2.0*lines*N*16/best/1e9);
		}
	}
	return 0;
}
