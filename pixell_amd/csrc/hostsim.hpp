// TEST-ONLY host simulator of the tiny subset of HIP the kernels in this directory use.
// Compiling the *same* .hip sources with g++ -DPXS_HOST_SIM gives libpxsht_hostsim.so, in which
// every workgroup is executed lane by lane -- each lane its own execution context (a fiber; one OS thread per lane with
// PXS_SIM_THREADS=1), barriers and wave shuffles real rendezvous between them.  It exists so that the index arithmetic of the kernels can be unit-tested in
// the GPU-less container (tests/ -m "not gpu"); it is never built by __graft_entry__.build(),
// never loaded by the product loader, and is orders of magnitude too slow to be a fallback.
#pragma once
#ifdef PXS_HOST_SIM
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <thread>
#include <vector>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <atomic>
#include <climits>
#include <unistd.h>
#include <sys/syscall.h>
#include <linux/futex.h>
#include <algorithm>

struct double2 { double x, y; };
struct float2 { float x, y; };
static inline double2 make_double2(double x, double y) { double2 r; r.x = x; r.y = y; return r; }
static inline float2 make_float2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct uint3_ { unsigned x, y, z; };

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0 };
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline const char* hipGetErrorString(hipError_t) { return "hostsim"; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? 0 : 2; }
template<class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { std::free(p); return 0; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { std::memmove(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { std::memmove(d, s, n); return 0; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return 0; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return 0; }
static inline hipError_t hipMemGetInfo(size_t* fr, size_t* tot) { *fr = size_t(2) << 30; *tot = size_t(2) << 30; return 0; }
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static const unsigned hipStreamNonBlocking = 1;
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return 0; }
typedef void* hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0; return 0; }
static const unsigned hipEventDisableTiming = 2;
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { static int dummy; *e = &dummy; return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

namespace pxsim {
// How the lanes of a workgroup run.  Default: FIBERS -- the lanes of a block are user-level contexts on ONE OS thread that hand over to each other at
// barriers and wave rendezvous (a barrier of 64 - 1024 lanes is a few microseconds instead of that many sleeps and wake-ups in the kernel: the test suite spent
// 6x its user time in futex calls), and the blocks of a grid are shared out over a few worker threads.  PXS_SIM_THREADS=1: one OS thread per lane, as before
// (real preemption between lanes; kept for cross-checks).
void fiber_yield();
bool fiber_mode();
struct Barrier {
	int n; std::atomic<int> count{0}; std::atomic<uint32_t> gen{0};
	explicit Barrier(int n_) : n(n_) {}
	void wait() {
		const uint32_t g = gen.load(std::memory_order_acquire);
		if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
			count.store(0, std::memory_order_relaxed); gen.store(g + 1, std::memory_order_release);
			if (!fiber_mode()) syscall(SYS_futex, reinterpret_cast<uint32_t*>(&gen), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
		} else if (fiber_mode()) {
			while (gen.load(std::memory_order_relaxed) == g) fiber_yield();
		} else {
			while (gen.load(std::memory_order_acquire) == g) syscall(SYS_futex, reinterpret_cast<uint32_t*>(&gen), FUTEX_WAIT_PRIVATE, g, nullptr, nullptr, 0);
		}
	}
};
struct BlockCtx {
	char* shared; Barrier* bar; std::vector<Barrier*> wbar; std::vector<uint64_t>* wslot; int nthreads;
};
extern thread_local uint3_ t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
extern thread_local BlockCtx* t_ctx;
// runs body() once per lane of one block on the calling OS thread, lanes as fibers (hostsim.cpp)
void run_block_fibers(int nt, dim3 block, const std::function<void()>& body);
int sim_workers();
template<class F> void launch(dim3 grid, dim3 block, size_t shmem, F&& body) {
	const int nt = block.x*block.y*block.z;
	const int nw = (nt + 63)/64;
	if (fiber_mode() && nt > 1) {
		const long nblk = (long)grid.x*grid.y*grid.z;
		const int W = (int)std::max<long>(1, std::min<long>(sim_workers(), nblk));
		const std::function<void()> fn = [&] { body(); };
		auto worker = [&](int w) {
			std::vector<char> sh(shmem + 64);
			Barrier bar(nt);
			std::vector<Barrier*> wb; for (int k = 0; k < nw; k++) wb.push_back(new Barrier(std::min(64, nt - 64*k)));
			std::vector<uint64_t> slots((size_t)nw*64*2);
			BlockCtx ctx{sh.data(), &bar, wb, &slots, nt};
			t_ctx = &ctx; t_blockDim = {block.x, block.y, block.z}; t_gridDim = {grid.x, grid.y, grid.z};
			for (long b = w; b < nblk; b += W) {
				t_blockIdx = {(unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x*grid.y))};
				run_block_fibers(nt, block, fn);
			}
			for (auto x : wb) delete x;
			t_ctx = nullptr;
		};
		if (W == 1) worker(0);
		else { std::vector<std::thread> th; for (int w = 0; w < W; w++) th.emplace_back(worker, w); for (auto& x : th) x.join(); }
		return;
	}
	std::vector<char> sh(shmem + 64);
	Barrier bar(nt);
	std::vector<Barrier*> wb; for (int w = 0; w < nw; w++) wb.push_back(new Barrier(std::min(64, nt - 64*w)));
	std::vector<uint64_t> slots((size_t)nw*64*2);
	BlockCtx ctx{sh.data(), &bar, wb, &slots, nt};
	// the lanes of a workgroup are OS threads created ONCE per launch; they walk over the blocks of the grid together
	// (a rendezvous between blocks: the shared-memory buffer and the barriers are reused)
	Barrier between(nt);
	auto lane = [&](int t) {
		t_threadIdx = {(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / (block.x*block.y))};
		t_blockDim = {block.x, block.y, block.z}; t_gridDim = {grid.x, grid.y, grid.z};
		t_ctx = &ctx;
		for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned by = 0; by < grid.y; by++) for (unsigned bx = 0; bx < grid.x; bx++) {
			t_blockIdx = {bx, by, bz};
			body();
			if (nt > 1) between.wait();
		}
	};
	if (nt == 1) lane(0);
	else {
		std::vector<std::thread> th;
		for (int t = 0; t < nt; t++) th.emplace_back(lane, t);
		for (auto& x : th) x.join();
	}
	for (auto b : wb) delete b;
}
static inline int lane_id() { return (t_threadIdx.x + t_threadIdx.y*t_blockDim.x) & 63; }
static inline int wave_id() { return (t_threadIdx.x + t_threadIdx.y*t_blockDim.x) >> 6; }
static inline void wave_sync() { BlockCtx* c = t_ctx; if (c->nthreads > 1) c->wbar[wave_id()]->wait(); }      // (lanes are OS threads here: what lockstep execution gives a wave for free)
static inline uint64_t xchg(uint64_t v, int src_lane) {
	BlockCtx* c = t_ctx; int w = wave_id(), l = lane_id();
	uint64_t* s = c->wslot->data() + (size_t)w*128;
	s[l] = v; c->wbar[w]->wait();
	uint64_t r = s[src_lane & 63]; c->wbar[w]->wait();
	return r;
}
}
#define threadIdx pxsim::t_threadIdx
#define blockIdx  pxsim::t_blockIdx
#define blockDim  pxsim::t_blockDim
#define gridDim   pxsim::t_gridDim
static inline void __syncthreads() { if (pxsim::t_ctx->nthreads > 1) pxsim::t_ctx->bar->wait(); }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a*b) >> 32); }
static inline double __shfl_xor(double v, int mask) { uint64_t u; std::memcpy(&u, &v, 8); u = pxsim::xchg(u, pxsim::lane_id() ^ mask); std::memcpy(&v, &u, 8); return v; }
static inline double __shfl(double v, int src) { uint64_t u; std::memcpy(&u, &v, 8); u = pxsim::xchg(u, src); std::memcpy(&v, &u, 8); return v; }
static inline double atomicAdd(double* p, double v) {   // lanes are OS threads here
	uint64_t* q = reinterpret_cast<uint64_t*>(p); uint64_t old = __atomic_load_n(q, __ATOMIC_RELAXED), nw; double o;
	do { std::memcpy(&o, &old, 8); double n = o + v; std::memcpy(&nw, &n, 8); } while (!__atomic_compare_exchange_n(q, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
	return o; }
static inline int atomicMin(int* p, int v) { int old = __atomic_load_n(p, __ATOMIC_RELAXED); while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return old; }
static inline float __fmul_rn(float a, float b) { volatile float r = a*b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a+b; return r; }
static inline int __shfl(int v, int src) { return (int)pxsim::xchg((uint64_t)(uint32_t)v, src); }
static inline unsigned long long __ballot(int pred) {
	pxsim::BlockCtx* c = pxsim::t_ctx; int w = pxsim::wave_id(), l = pxsim::lane_id();
	uint64_t* s = c->wslot->data() + (size_t)w*128 + 64;
	s[l] = pred ? 1 : 0; c->wbar[w]->wait();
	unsigned long long r = 0; int nl = std::min(64, c->nthreads - 64*w);
	for (int i = 0; i < nl; i++) r |= (unsigned long long)(s[i] & 1) << i;
	c->wbar[w]->wait();
	return r;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred) { pxsim::BlockCtx* c = pxsim::t_ctx; int w = pxsim::wave_id(); int nl = std::min(64, c->nthreads - 64*w);
	unsigned long long full = nl == 64 ? ~0ull : ((1ull << nl) - 1); return __ballot(pred) == full; }
using std::min; using std::max;
#define PXS_SHARED(type, name) type* name = reinterpret_cast<type*>(pxsim::t_ctx->shared)
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) pxsim::launch((grid), (block), (shmem), [&] { kern(__VA_ARGS__); })
#else
#include <hip/hip_runtime.h>
#define PXS_SHARED(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
#endif
