// Device-memory arena of libpxsht (see common.hpp) and its accounting: how long the library spent in hipMalloc is the first thing to
// know about a slow first call (round 5: first alm2map of a fresh process 162 ms on one box, 1248 ms on another, same code).
#include "common.hpp"
#include <chrono>
#include <map>
#include <mutex>
#include <unordered_map>

namespace pxs {
namespace {
struct Arena {
	std::mutex mu;
	std::multimap<size_t, std::pair<void*, int>> pool;      // released blocks by size: (pointer, device)
	std::unordered_map<void*, int> owner;                    // device of every live block >= MINB (noted at allocation: no runtime call when a block comes back,
	                                                         // which may be from a static destructor after the HIP runtime has shut down)
	size_t pooled = 0, live = 0, cap = size_t(48) << 30;
	double malloc_ms = 0; long nmalloc = 0, nreuse = 0; size_t malloc_bytes = 0;
	Arena() { const char* e = getenv("PXS_ARENA_GB"); if (e) cap = (size_t)atol(e) << 30; }
	static constexpr size_t MINB = size_t(32) << 20;
	void drop_all() {      // (mu held)
		for (auto& kv : pool) (void)hipFree(kv.second.first);
		pool.clear(); pooled = 0;
	}
};
Arena& arena() { static Arena* a = new Arena; return *a; }      // (never destroyed: buffers with static storage duration are released through it while the process exits)
}

void* dev_alloc(size_t n) {
	Arena& a = arena();
	std::lock_guard<std::mutex> g(a.mu);
	int dev = 0; (void)hipGetDevice(&dev);
	if (n >= Arena::MINB) {      // a released block of this size or up to a quarter more, on this device
		for (auto it = a.pool.lower_bound(n); it != a.pool.end() && it->first <= n + n/4; ++it) if (it->second.second == dev) {
			void* p = it->second.first; a.pooled -= it->first; a.live += it->first; a.pool.erase(it); a.nreuse++;
			a.owner[p] = dev;
			return p;
		}
	}
	void* p = nullptr;
	const auto t0 = std::chrono::steady_clock::now();
	hipError_t e = hipMalloc(&p, n);
	if (e != hipSuccess && !a.pool.empty()) { a.drop_all(); e = hipMalloc(&p, n); }      // out of memory: give the kept blocks back first
	a.malloc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	if (e != hipSuccess) throw Error(PXS_ERR_NOMEM, std::string("hipMalloc of ") + std::to_string(n >> 20) + " MB: " + hipGetErrorString(e));
	a.nmalloc++; a.malloc_bytes += n; a.live += n;
	if (n >= Arena::MINB) a.owner[p] = dev;
	return p;
}

void dev_free(void* p, size_t n) {
	if (!p) return;
	Arena& a = arena();
	std::lock_guard<std::mutex> g(a.mu);
	a.live -= std::min(a.live, n);
	if (n >= Arena::MINB && a.pooled + n <= a.cap) {
		// (the device the block lives on, not the thread's current one: a plan may be destroyed from a thread that has another device selected)
		int dev = 0;
		auto it = a.owner.find(p);
		if (it != a.owner.end()) { dev = it->second; a.owner.erase(it); } else (void)hipGetDevice(&dev);
		a.pool.emplace(n, std::make_pair(p, dev)); a.pooled += n;
		return;
	}
	a.owner.erase(p);
	(void)hipFree(p);
}
} // namespace pxs

extern "C" int pxs_memory(int release, double* stats) {
	pxs::Arena& a = pxs::arena();
	std::lock_guard<std::mutex> g(a.mu);
	if (release) a.drop_all();
	if (stats) { stats[0] = a.malloc_ms; stats[1] = (double)a.nmalloc; stats[2] = (double)a.malloc_bytes; stats[3] = (double)a.nreuse; stats[4] = (double)a.pooled; stats[5] = (double)a.live; }
	return 0;
}
