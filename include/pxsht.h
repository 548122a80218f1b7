/* libpxsht -- MI355X (gfx950) spherical-harmonic-transform and FFT kernels behind a C ABI.
 *
 * Drop-in boundary for the one hot path of simonsobs/pixell: the calls that
 * pixell/curvedsky.py and pixell/fft.py make into the third-party ducc0 / pyfftw / numpy
 * libraries.  The reference has no C ABI of its own for this path (its boundary is ducc0's
 * Python keyword interface), so every entry point below cites the reference call site it
 * replaces; INTEGRATION.md shows the ctypes binding a pixell maintainer would add.
 *
 * Conventions
 *  - all data pointers are DEVICE pointers (hipMalloc / torch.cuda tensors) on the plan's
 *    device; the caller owns them.  `stream` is a hipStream_t (NULL = default stream); calls
 *    are asynchronous with respect to the host.
 *  - plans own their device scratch and tables; a plan may be used by one call at a time.
 *  - return value: 0 = OK, <0 = error (pxs_last_error() returns a thread-local message).
 *    No C++ exception crosses the boundary.
 *  - dtype codes: 0 = float32, 1 = float64, 2 = complex64, 3 = complex128.  All arithmetic is
 *    float64 regardless of the I/O dtype.
 *  - alm layout: element (l,m) of component c at alm[c*alm_cstride + mstart[m] + l*lstride]
 *    (curvedsky.alm_info, curvedsky.py:409-447).
 */
#ifndef PXSHT_H
#define PXSHT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pxs_plan pxs_plan;

#define PXS_F32  0
#define PXS_F64  1
#define PXS_C64  2
#define PXS_C128 3

#define PXS_MODE_STANDARD 0
#define PXS_MODE_DERIV1   1   /* curvedsky.py:917-920: spin-1 of sqrt(l(l+1)) a_lm -> (d_theta, d_phi/sin) */

/* Plan for transforms on explicit iso-latitude rings.
 * Replaces the geometry arguments of ducc0.sht.experimental.synthesis / adjoint_synthesis
 * (curvedsky.py:936-960, 1068-1084, 328-349, 396-403; ring tables from get_ring_info, get_ring_info_healpix,
 * get_ring_info_radial, curvedsky.py:1170-1234).
 * theta[nring] colatitudes, nphi[nring] pixels per ring (>= 1), phi0[nring] azimuth of pixel 0,
 * ringstart[nring] index of pixel 0 of each ring in the flat map, pixstride: stride between pixels of a ring.
 * Rings of equal length, phase and spacing (CAR maps) take the fused ring FFTs; any other ring set (healpix, profile
 * rings, ring subsets, nphi < mmax) the general path: one batched FFT per ring length.  Equal rings that are consecutive
 * rows of a Fejer-1 grid (a declination band) run their Legendre stage on the Clenshaw-Curtis grid of that grid. */
int pxs_plan_rings(pxs_plan** plan, int nring, const double* theta, const uint64_t* nphi,
                   const double* phi0, const uint64_t* ringstart, int64_t pixstride,
                   int lmax, int mmax, const uint64_t* mstart, int64_t lstride, int device);

/* Plan for a named full-sky equiangular grid ("CC","F1","MW","MWflip","DH","F2"; curvedsky.py:1329-1342), map[ntheta][nphi].
 * Replaces the geometry arguments of ducc0.sht.experimental.{synthesis_2d, adjoint_synthesis_2d,
 * analysis_2d, adjoint_analysis_2d} (curvedsky.py:907-924, 1032-1046).
 * flip_y / flip_x fold curvedsky.map2buffer / buffer2map (curvedsky.py:1384-1411) into the kernels'
 * addressing: with flip_y the map's row 0 is the SOUTHERN-most ring, with flip_x column 0 is the
 * largest phi; phi0 is that of the flipped (ducc-orientation) map, i.e. analyse_geometry().phi0
 * (curvedsky.py:1284). */
int pxs_plan_grid2d(pxs_plan** plan, const char* geometry, int ntheta, int nphi, double phi0,
                    int flip_y, int flip_x, int lmax, int mmax, const uint64_t* mstart,
                    int64_t lstride, int device);

void pxs_plan_destroy(pxs_plan* plan);

/* alm -> map (adjoint = 0: synthesis[_2d]) or map -> alm (adjoint = 1: adjoint_synthesis[_2d]).
 * spin 0: 1 alm component, 1 map component; spin > 0: 2 and 2; mode DERIV1: 1 and 2 (spin must be 1).
 * alm_cstride / map_cstride: distance between components in elements of the respective dtype.
 * nbatch independent maps per call (the loop over pre-dimensions at curvedsky.py:763-765, 910-924): map b starts at
 * map + b*map_bstride, its alm at alm + b*alm_bstride (elements); nbatch = 1 ignores the batch strides.  The ring FFTs of a
 * batch run as one launch per pass; the plan's scratch bounds the maps per pass (PXS_BATCH_GB, default 32). */
int pxs_synthesis(pxs_plan* plan, int spin, int mode, int adjoint, int nbatch,
                  void* alm, int alm_dtype, int64_t alm_cstride, int64_t alm_bstride,
                  void* map, int map_dtype, int64_t map_cstride, int64_t map_bstride, void* stream);

/* map -> alm (adjoint = 0: analysis_2d) or alm -> map (adjoint = 1: adjoint_analysis_2d).
 * Only for grid2d plans (curvedsky.py:1018-1048; batch loop :1038-1046); how the integral over theta is taken: pxs_plan_option. */
int pxs_analysis(pxs_plan* plan, int spin, int adjoint, int nbatch,
                 void* map, int map_dtype, int64_t map_cstride, int64_t map_bstride,
                 void* alm, int alm_dtype, int64_t alm_cstride, int64_t alm_bstride, void* stream);

/* Plan options.  "analysis": how pxs_analysis integrates over theta on grid2d plans (CC, F1, MW, MWflip; DH and F2 always take ring
 * weights) -- all forms give the same alm for band-limited maps and differ at the 1e-3 level on maps that are not --
 *   2 (default) "ducc0": the route of ducc0's analysis_2d (ducc0 >= 0.36, src/ducc0/sht/sht.cc: analysis_2d -> resample_to_prepared_CC; the
 *     reference calls it at curvedsky.py:1032-1046) as restated WITHOUT access to the package or its source in the build environment: all
 *     forms agree on band-limited maps (pinned by tests); agreement with ducc0's numerics on maps that are not band-limited -- e.g. its
 *     treatment of the Nyquist bin when the spectrum is resized -- is UNPINNED until checked against an installed ducc0.  The theta-interpolant of the rings -- low-passed to
 *     |k| < N_cc where the grid's circle has at least 2 N_cc samples -- is evaluated on the Clenshaw-Curtis grid of N_cc + 1 rings,
 *     multiplied by that grid's quadrature weights and carried to the N_cc/2 + 1 rings of the Legendre stage by the transposed
 *     band-limited upsampling; N_cc = 2 good_size_complex(lmax + 1) (ducc0's) whenever the grid's circle shares a usable factor with
 *     it (pxs_plan_query "ncc_circle" tells).  A CC grid with ntheta >= 2 lmax + 2 is multiplied by its own weights directly (ducc0:
 *     need_first_resample = false).  Plans without fused theta chains run the same form through the generic FFT engine.
 *   0 "interpolant": exact quadrature of the full theta-interpolant (its product with |sin| on a circle of M > N + 2 lmax points);
 *   1 "weights": ring quadrature weights + adjoint synthesis, the reference's cyl route (curvedsky.py:852-861, 1068-1084:
 *     get_gridweights / nphi, then adjoint_synthesis), applied when ntheta >= 2 lmax + 2 (smaller grids take the default).
 * The adjoint (adjoint = 1) is the exact transpose of the chosen form.  Takes effect for the calls issued after it returns.
 * "build_tables" (value = spin): builds the recurrence tables of that spin now instead of inside the first transform that needs them.
 * "deterministic" (0 | 1; default 0, or 1 when PXS_DETERMINISTIC=1 is set as the plan is made): the Legendre analysis (pxs_analysis,
 *   adjoint synthesis) sums the contributions of the ring chunks of an m with atomic adds in arrival order -- results repeat from
 *   run to run to ~1e-14 of their size, not bit for bit (ducc0 on the CPU is bitwise repeatable).  1 selects ordered sums through
 *   per-chunk partial moments (<= 2 GB of scratch, m in passes; batched calls one map at a time): bitwise repeatable, slower.
 *   The promise is run-to-run repeatability of the SAME call.  A map transformed inside a batch of 4 or more takes the FP64-MFMA
 *   kernels (another summation order) and equals its single-map call to rounding (1e-13), not bit for bit, with or without this
 *   option -- the reference loops over the maps and is identical per map; PXS_SYN_MM_MIN=0 / PXS_ANA_MM_MIN=0 keep batches on the
 *   single-map kernels. */
int pxs_plan_option(pxs_plan* plan, const char* name, int64_t value);

/* What a plan does: "analysis_form" = the form pxs_analysis runs now (0 interpolant, 1 ring weights, 2 fine-CC form of ducc0's
 * route), "ncc_circle" = N_cc, the circle of the CC grid of the Legendre stage (0: none), "ducc_ncc_circle" = ducc0's N_cc for
 * the plan's lmax, "theta_line" = 1 if the theta resampling of pxs_analysis runs as one kernel with the line resident on a CU
 * (the circle sizes compiled into thetaline.hip; PXS_THETA_LINE=0 switches it off), 0 if as the five-stage chain: same results. */
int pxs_plan_query(const pxs_plan* plan, const char* name, int64_t* value);

/* ducc0.sht.experimental.get_gridweights (curvedsky.py:501, 855): out[ntheta], sum = 4 pi. Host memory. */
int pxs_gridweights(const char* geometry, int ntheta, double* out);

/* largest lmax analysis_2d supports on the grid (curvedsky.get_ducc_maxlmax, curvedsky.py:1349-1353) */
int pxs_grid_maxlmax(const char* geometry, int ntheta);

/* number of Legendre rings the plan iterates (R_actual of SURVEY 8d) for synthesis / analysis */
int pxs_plan_info(const pxs_plan* plan, int* nring_legendre_syn, int* nring_legendre_ana, int64_t* scratch_bytes);

/* Sizes the planner of the fused FFT chains picks for a theta circle of N samples and band limit lmax (diagnostics, tests):
 * out[10] = {ok, g, bN, g2, M, ac, Ncc, gs, bs, aNs} with N = g*bN, M = g*g2 (fine circle of the |sin| product), Ncc = ac*g
 * (Clenshaw-Curtis circle: Ncc/2+1 rings in the Legendre stage), synthesis: Ncc = gs*bs, N = aNs*gs.  See csrc/fftchain.hip. */
int pxs_debug_theta_plan(int64_t N, int lmax, int64_t* out);
/* Average milliseconds of one group of fused-chain stages of a plan, run `reps` times on scratch data (diagnostics and kernel
 * experiments, tools/chain_lab.py).  kind 0: ring FFTs map -> leg (MA1, MA2), 1: ring FFTs h -> map (MS1, MS2), 2: theta
 * resampling of the analysis (RA1-5), 3: of the synthesis (RS1-3), 4: transposed theta upsampling.  nc components in one call. */
int pxs_debug_chain(pxs_plan* plan, int kind, int nc, int spin, int reps, double* ms);

/* Optional device-side stage timers (hipEvents recorded on the launch stream around each stage):
 * used by bench.py to time the dominant kernel live.  ms[PXS_NSTAGE], counts[PXS_NSTAGE]. */
#define PXS_STAGE_LEG_SYN  0   /* Legendre synthesis kernel (alm2leg) */
#define PXS_STAGE_LEG_ANA  1   /* Legendre analysis kernel (leg2alm) */
#define PXS_STAGE_RING_FFT 2   /* ring FFTs + transposes (map2leg / leg2map) */
#define PXS_STAGE_RESAMPLE 3   /* theta resampling between the map's rings and the Clenshaw-Curtis grid */
#define PXS_NSTAGE 4
int pxs_profile(pxs_plan* plan, int enable);
int pxs_profile_read(pxs_plan* plan, double* ms, int* counts, int reset);
/* FP64 flops the Legendre kernels EXECUTED since profiling was enabled (or since the last reset): flops[0] synthesis, flops[1]
 * analysis.  Counted by the kernels themselves (steps run x ring pairs per lane x FMAs per step x 64 lanes x 2), so bench.py can
 * quote the hardware FMA utilisation next to the algorithmic flop count of SURVEY 8(d) (which credits rings and degrees the
 * kernels skip as polar-dead).  No reference counterpart. */
int pxs_profile_flops(pxs_plan* plan, double* flops, int reset);

/* N-d FFT over `naxes` axes of a strided array: the engine behind fft.engines["hip"].FFTW(a,b,axes,
 * direction) (pixell/fft.py:8-64,133-209) and enmap.fft/ifft (enmap.py:1307-1337).
 * kind 0: c2c (in/out same shape), 1: r2c (out last transformed axis n//2+1), 2: c2r,
 * 3..10: the FFTW r2r kinds REDFT00, REDFT10, REDFT01, REDFT11, RODFT00, RODFT10, RODFT01, RODFT11 (DCT-I..IV, DST-I..IV;
 *   pixell/fft.py:211-290): real in/out of the same shape, unnormalised, `forward` ignored.
 * shape[ndim] is the LOGICAL (real-space) shape; istride/ostride in elements of in/out dtype.
 * forward: e^{-i...}; unnormalised, result multiplied by `scale`. */
int pxf_fft_nd(int ndim, const int64_t* shape, const int64_t* istride, const int64_t* ostride,
               int naxes, const int* axes, int kind, int forward, double scale,
               int in_dtype, int out_dtype, const void* in, void* out, int device, void* stream);

/* alm post-processing on the device (cython/cmisc_core.c of the reference, via alm_info.alm2cl / lmul,
 * curvedsky.py:451-474, 630-712).  d_mstart: DEVICE array u64[mmax+1]; d_lmat: DEVICE f64[N][M][nl].
 * pxa_alm2cl: cl[l] = 1/(2l+1) sum_m a1_lm conj(a2_lm) (m>0 counted twice), one pair of alm per call.
 * pxa_lmatmul: out[a] = sum_b lmat[a][b][l] in[b] (N = M = 1 is almxfl); out may alias in.
 * complex64 alm use single precision arithmetic with the filter rounded to float, as the reference does. */
int pxa_alm2cl(int lmax, int mmax, const uint64_t* d_mstart, int64_t lstride, const void* alm1, const void* alm2, int alm_dtype,
               void* cl, int cl_dtype, int device, void* stream);
int pxa_lmatmul(int N, int M, int lmax, int mmax, const uint64_t* d_mstart, int64_t lstride,
                const void* alm_in, int64_t in_cstride, void* alm_out, int64_t out_cstride, int alm_dtype,
                const double* d_lmat, int nl, int device, void* stream);

/* flat-sky harmonic helpers around the 2-D map FFT (pixell/enmap.py:1358-1400 map2harm / harm2map / queb_rotmat,
 * :1959-2011 calc_ps2d, :2526-2556 lbin).  d_ly[ny], d_lx[nx]: DEVICE f64 wavenumber axes (enmap.laxes); maps [ny][nx] contiguous.
 * pxm_rotate_queb: in place (a, b) <- (c a - s b, s a + c b) with c + i s = exp(i spin atan2(+-lx, ly)); inverse_sign selects -.
 * pxm_ps2d: out = Re(a conj b).   pxm_lbin: ADDS, per bin floor(|l|/bsize) < nbin, the map value to d_sum and (when both
 * are given) |l| to d_lsum and 1 to d_hit; the caller zeroes the accumulators. */
int pxm_rotate_queb(int ny, int nx, const double* d_ly, const double* d_lx, int spin, int inverse_sign,
                    void* a, void* b, int dtype, int device, void* stream);
int pxm_ps2d(int64_t n, const void* a, const void* b, int dtype, void* out, int out_dtype, int device, void* stream);
int pxm_lbin(int ny, int nx, const double* d_ly, const double* d_lx, double bsize, int nbin,
             const void* map, int dtype, double* d_sum, double* d_lsum, double* d_hit, int device, void* stream);
/* binning by a per-pixel bin table (enmap.rbin, enmap.py:2512-2524; enmap.lbin with a transform of |l|, :2526-2531; both through _bin_helper
 * :2533-2556): ADDS map[i] (f32 | f64, n contiguous pixels) to d_sum[d_bin[i]] for 0 <= d_bin[i] < nbin; d_bin: DEVICE int32[n], made by the host
 * from the geometry alone; the caller zeroes d_sum (DEVICE f64[nbin]) */
int pxm_bin_index(int64_t n, const int32_t* d_bin, int nbin, const void* map, int dtype, double* d_sum, int device, void* stream);
/* data[i] *= vec[(i / inner) % n] on a contiguous complex array of `total` elements; d_vec: DEVICE complex128[n]
 * (the per-axis phase ramps of pixell.fft.shift, fft.py:347-368) */
int pxm_mul_axis(int64_t total, int64_t n, int64_t inner, void* data, int dtype, const void* d_vec, int device, void* stream);

/* 1 if the engine can transform this length (2,3,5-smooth or prime factors small enough) */
int pxf_fft_supported(int64_t n);
int64_t pxf_fft_good_size(int64_t n);

/* Device memory of the library.  Plans release their scratch into a per-process arena (blocks >= 32 MB, at most PXS_ARENA_GB = 48 GB
 * kept) that the next plan draws from, so that dropping and rebuilding plans does not go through hipMalloc again.  release != 0: give
 * the kept blocks back to the driver now.  stats (may be null): [0] ms spent in hipMalloc so far, [1] hipMalloc calls, [2] bytes they
 * allocated, [3] requests served from the arena, [4] bytes kept in the arena now, [5] bytes in use by plans.  No reference counterpart
 * (ducc0 allocates host memory per call). */
int pxs_memory(int release, double* stats);

const char* pxs_last_error(void);
const char* pxs_version(void);

#ifdef __cplusplus
}
#endif
#endif
