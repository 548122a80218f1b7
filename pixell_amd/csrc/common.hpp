// Shared host-side helpers for libpxsht (gfx950 only).
#pragma once
#include "hostsim.hpp"
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <stdexcept>

namespace pxs {

enum : int { PXS_OK = 0, PXS_ERR_ARG = -1, PXS_ERR_HIP = -2, PXS_ERR_UNSUPPORTED = -3, PXS_ERR_NOMEM = -4 };

struct Error : std::runtime_error {
	int code;
	Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string& msg);

#define PXS_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
	throw pxs::Error(pxs::PXS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
#define PXS_REQUIRE(cond, msg) do { if (!(cond)) throw pxs::Error(pxs::PXS_ERR_ARG, std::string(msg)); } while (0)

// Device memory of the library goes through one per-process arena (arena.hip): blocks of 32 MB and more that a plan releases are
// kept (up to PXS_ARENA_GB, default 48) and handed to the next plan that asks for a similar size instead of going back to the driver
// -- the scratch of a plan is tens of GB, hipMalloc maps it at a box-dependent rate, and a program that drops its plans between
// workloads (bench.py's legs, sht.clear_plans()) paid for it again each time.  pxs_memory() reports and releases.
void* dev_alloc(size_t bytes);
void dev_free(void* p, size_t bytes);

// simple owning device buffer
struct DevBuf {
	void* p = nullptr; size_t bytes = 0;
	DevBuf() {}
	explicit DevBuf(size_t n) { alloc(n); }
	DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
	DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
	DevBuf& operator=(DevBuf&& o) noexcept { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; return *this; }
	~DevBuf() { release(); }
	void alloc(size_t n) { release(); if (n) { p = dev_alloc(n); bytes = n; } }
	void ensure(size_t n) { if (n > bytes) alloc(n); }
	void release() { if (p) { dev_free(p, bytes); p = nullptr; bytes = 0; } }
	template<class T> T* as() const { return reinterpret_cast<T*>(p); }
};

template<class T> inline DevBuf upload(const std::vector<T>& v) {
	DevBuf b(v.size()*sizeof(T));
	if (!v.empty()) PXS_HIP(hipMemcpy(b.p, v.data(), v.size()*sizeof(T), hipMemcpyHostToDevice));
	return b;
}

// exact unsigned division by a small runtime constant: q = umulhi(x, mul) (valid for x*d < 2^32)
struct FastDiv { uint32_t mul; uint32_t d; };
inline FastDiv make_fastdiv(uint32_t d) {
	FastDiv f; f.d = d;
	f.mul = d <= 1 ? 0u : (uint32_t)(((1ull << 32) + d - 1) / d);
	return f;
}

enum DType : int { PX_F32 = 0, PX_F64 = 1, PX_C64 = 2, PX_C128 = 3 };

// Tuning and experiment switches (tile shapes, ring pairs per lane, planner overrides, alternative paths kept for A/B runs) are read
// from the environment in LAB builds only (-DPXS_LAB, tools/build_variants.sh); the product build compiles their defaults in.  What
// the product reads from the environment is listed in DESIGN.md ("Environment switches").
#ifdef PXS_LAB
inline const char* lab_getenv(const char* name) { return getenv(name); }
#else
inline const char* lab_getenv(const char*) { return nullptr; }
#endif

} // namespace pxs
