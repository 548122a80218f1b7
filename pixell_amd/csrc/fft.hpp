// Batched 1-D FFT engine for gfx950: LDS-resident mixed-radix passes (2,3,4,5 + generic
// odd radix), four-step decomposition for long lines, load/store functors that fuse the
// index remaps the SHT needs (Hermitian expansion, parity mirror extension, spectrum
// resize with Nyquist rule, pruned / scaled / table-multiplied stores).
// Replaces ducc0.fft.c2c/r2c/c2r as called from pixell/fft.py:45-64 and the ring / theta
// FFTs inside ducc0's SHT (not in the reference tree).
#pragma once
#include "common.hpp"
#include <map>
#include <memory>
#include <mutex>

namespace pxs {

enum LoadMode : int { LD_PLAIN = 0, LD_HERM = 1, LD_MIRROR = 2, LD_SPEC = 3, LD_SPEC_ADJ = 4, LD_MIRROR_PAIR = 5, LD_REAL_PAIR = 6, LD_HERM_PAIR = 7 };

// Line index space: a transform "line" is addressed by (i, o1, o2); element e along it.
struct FftDims {
	long n_i = 1, n_o1 = 1, n_o2 = 1;
	long is_i = 0, is_o1 = 0, is_o2 = 0, is_e = 1;   // input strides (in elements of the input dtype)
	long os_i = 0, os_o1 = 0, os_o2 = 0, os_e = 1;   // output strides
};

struct FftLoad {
	const void* ptr = nullptr; int dtype = PX_C128; int mode = LD_PLAIN;
	long ne = -1;               // PLAIN: elements >= ne read as zero (-1: all n); HERM: half-spectrum length (mmax+1);
	                            // MIRROR: number of real rings; SPEC*: source spectrum length Ns
	const double2* mul = nullptr; // PLAIN: multiplier indexed by e. SPEC: phase indexed by |k| (conjugated for k<0)
	long shift = 0;             // PLAIN: element e reads source element e - shift (zero outside 0..ne-1)
	int mir_c = 0, par0 = 0;    // MIRROR: src index for e>=ne is (-e-c) mod n, sign -1 if ((par_step*i+par0)&1)
	int par_step = 1;           // MIRROR: 1 = parity alternates with the line index (SHT m columns), 0 = same parity for every line (DCT/DST)
	long kmax = -1;             // SPEC: keep |k| <= kmax (-1: all representable)
	int nyq_half = 0;           // SPEC: source Nyquist bin (Ns even) is split 1/2,1/2 onto +-Ns/2
	long pair_lines = 0;        // MIRROR_PAIR: number of source lines (line i packs source lines 2i [parity par0] and 2i+1 [opposite parity])
	int herm_fold = 0;          // HERM: 1 = SHT ring semantics (2 Re sum over m, with aliasing folds), 0 = plain c2r
	long herm_n = 0;            // HERM, plain c2r: logical length of the Hermitian extension when it differs from the transform length
	                            // (Bluestein: the extension of n points, times mul[e], zero padded to the chirp length); 0: the transform length
};

struct FftStore {
	void* ptr = nullptr; int dtype = PX_C128;
	long ne = -1;               // store only e < ne (-1: all)
	long two_sided_k = -1;      // if >= 0: store only e <= k or e >= n-k
	const double2* mul = nullptr; // multiplier indexed by e
	double scale = 1.0;
	long shift = 0;             // bins e < shift are dropped, bin e is stored at position e - shift (ne counts bins)
	int conj_out = 0;           // conjugate the result before mul/scale
	int compact_two_sided = 0;  // with two_sided_k = k: e <= k stored at e, e >= n-k stored at k + (n - e)  (row length 2k+1)
	int real_pair = 0;          // real output dtype: Re -> line 2i, Im -> line 2i+1 (os_i is the stride between real lines)
	long pair_lines = 0;        // real_pair: number of real output lines
};

struct FftSub;   // per-length tables

class FftContext {
public:
	explicit FftContext(int device);
	~FftContext();
	// out = FFT_n(in) along e for every line; forward: e^{-2 pi i jk/n}; unnormalised.
	void exec(hipStream_t st, long n, bool forward, const FftDims& d, const FftLoad& ld, const FftStore& stf);
	static long good_size(long n);          // smallest 2^a 3^b 5^c >= n that the engine can factor
	static bool supported(long n, std::string* why = nullptr);
	// views for the fused chain kernels (fftchain.hip): LDS sub-transform tables of length n (radices 2,3,4,5 only:
	// `ok` false otherwise) and the four-step twiddle table e^{-2 pi i k/n}, k < n
	struct SubView { int n, nfac, ns, generic; const void* pass; const int* perm; const double2* tw; };
	SubView view(long n, int maxr = 1000);      // maxr: largest composite register radix the calling kernel has compiled in
	const double2* twiddle_table(long n) { return bigtw(n); }
	void release_stream(hipStream_t st) { std::lock_guard<std::mutex> g(mu_); temps_.erase(st); }   // four-step scratch of a stream that is going away
	size_t temp_budget = size_t(4) << 30;   // bytes of four-step scratch per stream (set from the free memory in the constructor)
private:
	int device_;
	std::mutex mu_;
	std::map<long, std::shared_ptr<FftSub>> subs_;
	std::map<long, DevBuf> bigtw_;
	std::map<hipStream_t, DevBuf> temps_;
	std::shared_ptr<FftSub> sub(long n, bool comp = false, int maxr = 1000);   // comp: factorisation with the composite register radices (chain kernels)
	const double2* bigtw(long n);
};

} // namespace pxs
