// Probe (round 5): rows of the spin-0 recurrence table in double-double, on the host and on the GPU, against the long-double reference.
// hipcc --offload-arch=gfx950 -O3 tools/tables_probe.hip -o tools/tables_probe.bin   (g++ -x c++ for the host half alone)

#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#else
#define __host__
#define __device__
#define __forceinline__ inline
#define __global__
#endif
#if defined(__clang__)
#define PXS_FP_STRICT _Pragma("clang fp contract(off)")
#else
#define PXS_FP_STRICT
#endif
struct dd { double h, l; };
__host__ __device__ __forceinline__ dd dd_norm(double a, double b) {
	PXS_FP_STRICT const double s = a + b; return dd{s, b - (s - a)}; }
__host__ __device__ __forceinline__ dd dd_of(double a) {
	PXS_FP_STRICT return dd{a, 0.0}; }
__host__ __device__ __forceinline__ dd dd_add(dd a, dd b) {
	PXS_FP_STRICT
	const double s = a.h + b.h, v = s - a.h, e = (a.h - (s - v)) + (b.h - v);
	return dd_norm(s, e + (a.l + b.l));
}
__host__ __device__ __forceinline__ dd dd_neg(dd a) {
	PXS_FP_STRICT return dd{-a.h, -a.l}; }
__host__ __device__ __forceinline__ dd dd_mul(dd a, dd b) {
	PXS_FP_STRICT
	const double p = a.h*b.h, e = fma(a.h, b.h, -p);
	return dd_norm(p, e + (a.h*b.l + a.l*b.h));
}
__host__ __device__ __forceinline__ dd dd_div(dd a, dd b) {
	PXS_FP_STRICT
	const double q1 = a.h/b.h;
	dd r = dd_add(a, dd_neg(dd_mul(dd_of(q1), b)));
	const double q2 = r.h/b.h;
	r = dd_add(r, dd_neg(dd_mul(dd_of(q2), b)));
	const double q3 = r.h/b.h;
	return dd_add(dd_norm(q1, q2), dd_of(q3));
}
__host__ __device__ __forceinline__ dd dd_sqrt(dd a) {
	PXS_FP_STRICT
	if (a.h <= 0.0) return dd{0.0, 0.0};
	const double x = sqrt(a.h);
	// one Newton step in double-double: x + (a - x^2) / (2 x)
	const double p = x*x, e = fma(x, x, -p);
	const dd r = dd_add(a, dd{-p, -e});
	return dd_norm(x, r.h/(2.0*x));
}
__host__ __device__ __forceinline__ double dd_val(dd a) {
	PXS_FP_STRICT return a.h + a.l; }

typedef long double LDb;
// one m, spin 0: fills ak, ake (b), alpha
__host__ __device__ void rows_dd(int lmax, int m, dd cm, double* oa, double* ob, double* oal) {
	auto q = [&](int l) -> dd { if (l <= m) return dd{0.0, 0.0}; const double L = l, M = m; return dd_div(dd_of(L*L - M*M), dd_of(4.0*L*L - 1.0)); };
	const int nk = (lmax - m)/2 + 1;
	dd a_prev = dd{0.0, 0.0}, a_cur = dd_mul(dd_sqrt(dd_of(2.0*m + 3.0)), cm);
	dd qm1 = q(m), q0 = q(m + 1), qp1 = q(m + 2), qp2 = q(m + 3);
	for (int k = 0; k < nk; k++) {
		const dd e2 = dd_add(qp1, q0), f = dd_sqrt(dd_mul(q0, qm1)), d = dd_sqrt(dd_mul(qp1, qp2));
		const dd a_next = (k == 0) ? dd_div(a_cur, d) : dd_neg(dd_div(dd_mul(f, a_prev), d));
		const dd ak = dd_div(a_cur, dd_mul(a_next, d));
		const dd ake = dd_mul(ak, e2);
		oa[k] = dd_val(ak); ob[k] = -dd_val(ake); oal[k] = dd_val(a_cur);
		a_prev = a_cur; a_cur = a_next;
		const int lp = m + 2*k + 3;
		qm1 = qp1; q0 = qp2; qp1 = q(lp + 1); qp2 = q(lp + 2);
	}
}
void rows_ld(int lmax, int m, LDb cm, double* oa, double* ob, double* oal) {
	auto eps = [&](int l) -> LDb { if (l <= m) return 0; LDb L = l, M = m; return sqrtl((L*L-M*M)/(4*L*L-1)); };
	const int nk = (lmax - m)/2 + 1;
	LDb a_prev = 0, a_cur = sqrtl((LDb)(2*m+3))*cm;
	for (int k = 0; k < nk; k++) {
		const int lp = m + 2*k + 1;
		const LDb e2 = eps(lp+1)*eps(lp+1) + eps(lp)*eps(lp), f = eps(lp)*eps(lp-1), d = eps(lp+1)*eps(lp+2);
		const LDb a_next = (k == 0) ? a_cur/d : -f*a_prev/d;
		const LDb ak = a_cur/(a_next*d);
		oa[k] = (double)ak; ob[k] = (double)(-ak*e2); oal[k] = (double)a_cur;
		a_prev = a_cur; a_cur = a_next;
	}
}
#ifdef __HIPCC__
__global__ void kern(int lmax, int m, dd cm, double* oa, double* ob, double* oal) { if (threadIdx.x == 0) rows_dd(lmax, m, cm, oa, ob, oal); }
#endif
int main() {
	const int lmax = 4000;
	const LDb PIl = 3.141592653589793238462643383279502884L;
	std::vector<LDb> cms(lmax+1); { LDb cm = 1/sqrtl(4*PIl); for (int m = 0; m <= lmax; m++) { if (m > 0) cm = -cm*sqrtl((LDb)(2*m+1)/(LDb)(2*m)); cms[m] = cm; } }
	for (int m : {0, 7, 500, 2000}) {
		const int nk = (lmax - m)/2 + 1;
		std::vector<double> a1(nk), b1(nk), al1(nk), a2(nk), b2(nk), al2(nk), a3(nk), b3(nk), al3(nk);
		rows_ld(lmax, m, cms[m], a1.data(), b1.data(), al1.data());
		const double h = (double)cms[m]; dd cm{h, (double)(cms[m] - (LDb)h)};
		rows_dd(lmax, m, cm, a2.data(), b2.data(), al2.data());
		double e1 = 0, e2 = 0, e3 = 0;
		for (int k = 0; k < nk; k++) { e1 = std::max(e1, fabs(a1[k]-a2[k])/fabs(a1[k])); e2 = std::max(e2, fabs(b1[k]-b2[k])/fabs(b1[k])); e3 = std::max(e3, fabs(al1[k]-al2[k])/fabs(al1[k])); }
		printf("m=%d host dd vs long double: a %.2e b %.2e alpha %.2e\n", m, e1, e2, e3);
#ifdef __HIPCC__
		double *da, *db, *dal; hipMalloc(&da, nk*8); hipMalloc(&db, nk*8); hipMalloc(&dal, nk*8);
		hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, lmax, m, cm, da, db, dal); hipDeviceSynchronize();
		hipMemcpy(a3.data(), da, nk*8, hipMemcpyDeviceToHost); hipMemcpy(b3.data(), db, nk*8, hipMemcpyDeviceToHost); hipMemcpy(al3.data(), dal, nk*8, hipMemcpyDeviceToHost);
		e1 = e2 = e3 = 0;
		for (int k = 0; k < nk; k++) { e1 = std::max(e1, fabs(a1[k]-a3[k])/fabs(a1[k])); e2 = std::max(e2, fabs(b1[k]-b3[k])/fabs(b1[k])); e3 = std::max(e3, fabs(al1[k]-al3[k])/fabs(al1[k])); }
		printf("m=%d DEVICE dd vs long double: a %.2e b %.2e alpha %.2e\n", m, e1, e2, e3);
#endif
	}
	return 0;
}
