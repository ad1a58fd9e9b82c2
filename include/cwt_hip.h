/* cwt_hip.h -- C ABI of libcwt_hip.so, the MI355X (gfx950) engine behind
 * pycwt.cwt() / pycwt.icwt().
 *
 * The reference (regeirk/pycwt) has no native code; its only plug-in seam for
 * this path is the `helpers.fft` module object plus `helpers.fft_kwargs`
 * (pycwt/helpers.py:6-30), used by wavelet.cwt at pycwt/wavelet.py:91-106.
 * This library replaces what flows through that seam -- forward FFT of the
 * signal, the per-scale filter bank sqrt(2*pi*s/dt)*conj(psi_ft(s*w)), the
 * batched inverse FFT -- and the icwt column reduction (wavelet.py:169-170),
 * with hand-written HIP kernels.  Each entry point names the reference lines
 * it stands in for.  INTEGRATION.md shows the ctypes binding a pycwt
 * maintainer would add.
 *
 * Conventions: plain C types only; every function returns 0 on success and a
 * negative CWT_E* code on failure (message via cwt_last_error(), thread
 * local); no exception crosses the ABI; buffers are caller-owned; `*_dev`
 * pointers are device addresses on the plan's GPU, `*_host` pointers are host
 * addresses.  A plan is not thread-safe; distinct plans may be used from
 * distinct threads / streams.  All launches of a plan go to the stream set by
 * cwt_plan_set_stream (default: the null stream).
 *
 * Complex data are interleaved (re, im) pairs of the plan's real type:
 * precision 64 -> double/complex128, precision 32 -> float/complex64.
 */
#ifndef CWT_HIP_H
#define CWT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CWT_OK 0
#define CWT_EINVAL (-1)  /* bad argument                                   */
#define CWT_EHIP (-2)    /* a HIP runtime call failed                      */
#define CWT_ENOMEM (-3)  /* device or host allocation failed               */
#define CWT_ENODEV (-4)  /* no usable GPU                                  */

/* mother wavelets with a built-in Fourier-domain profile (pycwt/mothers.py) */
#define CWT_MORLET 0 /* psi_ft: mothers.py:26-28,  param = f0 */
#define CWT_PAUL 1   /* psi_ft: mothers.py:118-122, param = m  */
#define CWT_DOG 2    /* psi_ft: mothers.py:170-173, param = m  */

typedef struct cwt_plan cwt_plan;

/* Library identity: "hip-gfx950" for the product build. */
const char* cwt_backend(void);
/* Identity of the SOURCES this binary was built from: the first 16 hex digits of a SHA-256 over the files of pycwt_amd/csrc (the
 * translation units and headers), include/cwt_hip.h and the extra compiler flags of the build (pycwt_amd/_build.py,
 * `source_id()`).  The Python binding compares it with the tree it runs in and rebuilds -- or refuses -- on a mismatch
 * (PYCWT_AMD_ALLOW_STALE=1 overrides); bench.py prints it.  "unknown" for a build that did not go through _build.py. */
const char* cwt_build_id(void);
/* Message of the last failure on the calling thread ("" if none). */
const char* cwt_last_error(void);
/* Number of visible GPUs. */
int cwt_device_count(int* count);

/* ---- plan ---------------------------------------------------------------
 * One plan per (device, transform length, precision).  nfft is the padded
 * transform length the reference would use: the next power of two >= len(signal)
 * (helpers.py:27-30), 2 <= nfft <= 2^24.  max_rows bounds the number of scales
 * of one cwt_transform_rows call (workspace sizing).                        */
int cwt_plan_create(cwt_plan** plan, int device, int64_t nfft, int precision /* 32 | 64 */,
                    int max_rows);
int cwt_plan_destroy(cwt_plan* plan);
/* hipStream_t handle (as void*) all later launches of this plan are queued on. */
int cwt_plan_set_stream(cwt_plan* plan, void* hip_stream);
/* Tuning / test hooks; unknown keys fail with CWT_EINVAL.  (The keys of the measured-and-rejected kernel variants and of the
 * diagnostics of rounds 1-3 -- "overlap", "pass_b_prefetch", "pass_b_small", "stamps", "ols_tile", "ols_fwd_real", "sched",
 * "narrow_wave" -- are refused with a pointer to EXPERIMENTS.md: that code left the sources in round 4.)  Keys:
 *   "poly"         0 = no band-limited rows in polynomial form (k_poly_*; default 1): they then run one K-point transform
 *                  per output residue ("narrow*") or, where the call hands over the signal, as overlap-save rows
 *   "poly_degree"  preferred largest degree of those rows (default 8): a row gets the smallest interval count K' (a power
 *                  of two >= its support, >= 256, <= nfft / 64) whose degree does not exceed it
 *   "poly_min_logn" log2 of the shortest transform that uses the form (default 16)
 *   "poly_chunk_mb" the polynomial rows go through in as few chunks of about equal coefficient volume as keep that volume
 *                  below this many MiB (default 96; 0 = all rows at once): planes computed, then consumed while they still sit
 *                  in the Infinity Cache (matters from ~150 MB: fp64 Paul, round-off targets)
 *   "poly_max_logk" log2 of the largest interval count K' (8 ... 14, default 14 = the largest coefficient tile)
 *   "aols"         0 = rows clipped at the Nyquist bins stay two-pass rows (default 1: overlap-save rows on the band-passed
 *                  complex signal, k_aols_*; Morlet, Paul, and -- with the real signal at hand -- DOG of order >= 1)
 *   "aols_min_rows" ... if at least this many rows qualify (default 3: the band-passed signal costs about one two-pass row)
 *   "aols_zc"      0 = fp64 Paul rows whose filter has died out at Nyquist stay two-pass rows (default 1: they run on the
 *                  band-passed signal with the profile continued through f = 0 -- the kink there is what gives the Paul wavelet
 *                  its 1/t^(m+1) tail; costs ~1.5e3 x the arithmetic's epsilon of rounding noise, so only where the accuracy
 *                  target leaves room for it: fp64 at targets >= ~2e-12)
 *   "chunk_rows"   rows per two-pass chunk (intermediate = chunk_rows*nfft complex); 0 = default:
 *                  as many rows as fit 192 MiB, so that the intermediate stays in the Infinity Cache
 *   "narrow"       0 disables the band-limited single-pass path
 *   "narrow_max_k" largest per-row transform length of that path (power of two)
 *   "lmax"         largest single-workgroup FFT length (power of two, <= 4096)
 *   "wg_points"    complex points per workgroup of the fused kernels
 *   "narrow_terms" largest number of aliased bins per FFT input of that path at K = 1024 (1 = off, <= 16)
 *   "big_terms"    the same at K = 2048 (fp64, 16384-point workgroups; <= 8)
 *   "narrow_big"   0 = no K = 2048 single-pass rows (fp64, 16384-point workgroups)
 *   "overlap_narrow" 1 (default) = queue the band-limited rows on a side stream beside the two-pass chain
 *                  (+4 % in fp64; ignored while "profile" is on), 0 = everything on the plan's stream
 *   "band_pass_a"  0 = always run the full column FFT in pass A (no short aliased column FFTs)
 *   "pass_a_small" 0 = pass A on full-size workgroup tiles (default 1: half-size tiles, 4 per CU)
 *   "narrow_mix"   1 = the launch order of the band-limited rows alternates the light (K = 16, store bound) and the heavy
 *                  end (K = 1024, several terms) of the list, so that a CU's two tile slots hold one of each (default 1 for
 *                  precision 64: -2 % of the step; 0 for 32, where it measured +-0)
 *   "narrow_small" 0 = complex64 band-limited rows with K <= 512 on full-size tiles (default 1: half-size; the
 *                  full-size instance is compiled for 64 VGPRs, which only its K = 1024 branches meet without spills:
 *                  0 is correct but slower)
 *   "two_pass_logk" log2 of the row length K of the two-pass split N = R*K (0 = default: 1024 up to 2^21, 2048 above)
 *   "big_tiles"    0 = complex128 pass A with 4096-point columns (N >= 2^23) on 8192-point tiles (default 1: 16384)
 *   "ols"          0 = no overlap-save rows in cwt_transform / cwt_execute_host (default 1)
 *   "ols_max_halo" largest halo H (samples, multiple of 64) of an overlap-save row; 0 = a quarter of the workgroup tile
 *   "ols_big"      1 = rows with halo >= "ols_big_min_halo" (default 1536) and a block support <= 1/8 tile use blocks
 *                  of two workgroup tiles; 2 = also blocks of FOUR tiles for halos in ["ols_big4_min_halo" (2048),
 *                  "ols_big4_max_halo" (8192)]: their block spectra are one 16384-point packed transform per block
 *                  (default: 1 for precision 32, 0 for 64; 2 measured -1 ... +3 % depending on the thresholds: the rows
 *                  gain what the extra block-spectra launch costs, EXPERIMENTS.md)
 *   "ols_small_max_halo" overlap-save rows with a halo up to this many samples (multiple of 64, default 512) run on
 *                  half-size workgroup tiles -- four block transforms in flight per CU instead of two; 0 = none
 *   "ols_small_big" 0 = rows with a halo in ("ols_small_max_halo", 1024] on the default tile (default 1: on 8192-point blocks of TWO
 *                  half-size tiles each where the block support is <= 512 bins -- 256-thread workgroups, four per CU, instead of one
 *                  512-thread workgroup per block, two per CU: fp64 Morlet / DOG -0.3 ... -0.8 %, fp32 DOG -2 % of the step)
 *   "aols_long"    0 = precision 64: rows clipped at Nyquist whose kernel needs a halo of 512 ... 2048 samples stay two-pass rows (default
 *                  1: the second, 8192-point class of the rows on the band-passed signal takes them -- fp64 Paul, scales of 2 ... 5
 *                  samples: 16 of its 36 two-pass rows, -3.7 % of the step)
 *   "ols_min_logn" log2 of the shortest transform length that uses the form (default 18; tests lower it to 15)
 *   "ols_side", "ols_early" 0 = queue the block spectra / the whole overlap-save chain on the plan's own stream instead
 *                  of a side stream beside the forward FFT and the two-pass chain (defaults 1)
 *   "ols_fwd_weight" cost of one block spectrum in percent of one row's block transform (halo class grouping; 100)
 *   "host_direct"  0 = cwt_execute_host stages transforms that fit one workgroup per row through device buffers and copy
 *                  operations like the longer ones (default 1: their kernels read the signal from and write W into
 *                  page-locked host memory themselves)
 *   "queue_probe"  0 = take the side streams as the runtime made them (default 1: before its first long transform on a
 *                  caller's stream the plan measures -- two one-thread kernels, ~0.3 ms once -- whether its four streams
 *                  sit on four hardware queues, and replaces side streams that share one: streams created earlier in the
 *                  process, e.g. by other plans or the framework, otherwise cost 2-6 % of the step).  THIS ONE CALL
 *                  SYNCHRONISES the caller's stream (cwt_transform / cwt_transform_rows / cwt_transform_batch with
 *                  nfft >= 2^18, once per caller's stream: the last eight are remembered; never while the stream is being
 *                  captured into a graph); at most 16 streams are ever parked
 *   "serial_rows"  long transforms with polynomial rows: 0 = overlap-save chain, band-passed rows and polynomial rows side by
 *                  side on the plan's streams (the schedule of rounds 4-5); 1 = every kernel that WRITES W on the caller's
 *                  stream, one after the other, everything they need prepared on the side streams; 2 = also the first block
 *                  spectra on the caller's stream (their rows follow at a kernel boundary) and the forward FFT on a side
 *                  stream, on half-size tiles ("fft_aside_small"); 3 = 2 with ONE wait on the caller's stream.  Default 2 for
 *                  precision 64 (measured -1 % at config 2, -5.7 % for fp64 Paul), 0 for precision 32 (+-0 / +1.4 %)
 *   "poly_carrier" 0 = the carrier of a polynomial row is the centre bin of its band (default 1: the bin, of 15 candidates, at which
 *                  the filter-weighted degree bound is lowest -- for a lopsided filter (Paul, DOG) near its peak: half the
 *                  intervals at the same degree; fp64 Paul: coefficient planes 143 -> 73 MB)
 *   "poly_cheb"    0 = Taylor weights theta^d / d! in the interval coefficients (default 1: the weights of the Chebyshev series of
 *                  e^{i theta u} cut at the same degree and re-expanded in monomials -- error 2 (theta/2)^(D+1) / (D+1)! instead of
 *                  theta^(D+1) / (D+1)!, so fewer intervals at the same degree: planes -25 % at 2^20 x 256 Morlet scales; one table
 *                  of (D + 1) x (K' + 1) reals per (K', D) pair of the scale grid, written when the grid is first seen)
 *   "serial_s1_once" 0 = under "serial_rows" the caller's stream waits for side stream 1 once per consumer (second overlap-save
 *                  launch, band-passed rows); default 1: once, for the end of that in-order chain
 *   "coef_small"   1 = the interval coefficients of every K' in one launch of 256-thread workgroups (K' = 8192 / 16384 as 2 / 4
 *                  decimated 4096-point transforms per job); default 0: measured +5 % on the step (strided plane stores)
 *   "aols_small_b" 0 = the band-passed signal's second pass on the default tile under "serial_rows" (default 1: 4096-point
 *                  tiles, 256-thread workgroups, which find a CU beside the overlap-save rows)
 *   "graph"        1 = repeated cwt_transform calls with the same buffers and scale grid are captured into a HIP graph
 *                  and replayed (default 0: measured +-0.5 % on the step, the chain is latency bound, not launch bound)
 *   "ct"           0 = never use the compile-time specialised kernels (generic engine only)
 *   "profile"      1 = time every kernel class with HIP events (cwt_plan_timings) */
int cwt_plan_set_option(cwt_plan* plan, const char* key, int64_t value);
/* Accuracy target of the fast forms.  Every row form of the path rests on truncations of the filter, and all of them are
 * derived from this one number: the filter support (bins of psi_ft below rel_tol/10 of its largest value on the row's bins
 * count as zero: band limiting), the overlap-save halo (neglected L1 mass of the wavelet <= rel_tol/10), the "not clipped
 * at Nyquist" test of the overlap-save rows (profile at the Nyquist bins <= rel_tol of its peak) and the degree of the
 * polynomial rows (Taylor remainder of a bin, weighted by the filter's value there, <= rel_tol/10).
 * What the number bounds: the truncations are relative to the FILTER, so for a signal whose spectrum is flat (white noise:
 * max|xhat| / rms|xhat| ~ 4) the error of a row relative to its own peak, max|W - W_exact| / max|W_exact|, stays below
 * rel_tol (measured: 2e-10 at 1e-9, 5e-6 at 3e-5, N = 2^20); for a spectrum with dynamic range D = max|xhat| / rms|xhat|
 * it can reach rel_tol * D / 4 (a line 1e4 above the noise floor: 1.5e-5 at rel_tol = 1e-9, 6e-11 at 1e-16; D is meant against
 * the quietest part of the spectrum, see cwt_spectrum_range) -- it is NOT a
 * signal-independent bound.  Hence the default (rel_tol = 0) is round-off: 1e-16 for precision 64, 1e-8 for precision 32,
 * every truncation below the arithmetic's own rounding (~3e-15 / ~2e-6 against the reference), for any input.  Callers that
 * know their spectra pass a looser target and get the faster forms (bench.py: 1e-9 / 3e-5 on white noise; error and speed
 * per target: profiles/r04_tolerance_sweep.txt); callers that do not can measure D (cwt_spectrum_range) or let
 * cwt_execute_host do it per call (cwt_plan_set_auto_tolerance).  Also reachable as the option "tolerance_neglog10"
 * (integer n -> 10^-n); the environment variable CWT_TOLERANCE, read by cwt_plan_create, replaces the default of new plans.
 * The target is part of the key of the plan's cached row tables (four of them): alternating between two targets rebuilds
 * nothing. */
int cwt_plan_set_tolerance(cwt_plan* plan, double rel_tol);
/* target > 0: cwt_execute_host (the host-buffer call, which synchronises anyway) sets the plan's tolerance per call to
 * cwt_plan_auto_tolerance(target) of that call's spectrum.  0 = off (the plan's own tolerance is used).  The
 * device-resident entry points never synchronise and never do this. */
int cwt_plan_set_auto_tolerance(cwt_plan* plan, double target);
/* Of a device-resident spectrum of n bins (two small kernels + a synchronising copy): max|xhat[k]|, rms|xhat[k]| and the
 * rms of the quietest stretch of the positive half at the resolution of a row's pass band: quarter-octave windows
 * [2^b (4+q)/4, 2^b (5+q)/4) (single bins below bin 4), each pooled with its two neighbours (3/4 octave).  Every bin of
 * 1 .. n/2 - 1 is in a window: a high-passed signal's quiet low end and any notch of 3/4 octave or more are seen (round 4
 * looked at whole octaves from bin 64 up and missed both). */
int cwt_spectrum_range(cwt_plan* plan, const void* xhat_dev, int64_t n, double* max_abs, double* rms_abs, double* floor_abs);
/* The filter-relative tolerance that holds `target` relative to every row's own peak for THIS spectrum (nfft bins of the
 * plan): target * min(1, 8 / D), D = max|xhat| / floor_abs of cwt_spectrum_range, rounded down to a power of sqrt(10), never
 * below round-off (which is also the answer for a spectrum with an empty stretch or a non-finite bin).  The bound it rests
 * on -- row error <= tolerance * D / 4 -- is measured, not proved (tests/test_tolerance_emulated.py: white and red noise,
 * lines, high-passed and notched spectra); a signal whose energy in some row's band is far below the windows' floor (one
 * isolated quiet bin among loud ones) can still exceed the target on that row.  Synchronises the plan's stream. */
int cwt_plan_auto_tolerance(cwt_plan* plan, const void* xhat_dev, double target, double* rel_tol);
int cwt_plan_get_tolerance(cwt_plan* plan, double* rel_tol);
/* Block the host until everything queued by this plan has finished. */
int cwt_plan_sync(cwt_plan* plan);

/* ---- device memory helpers (so a NumPy-only host needs no other runtime) */
/* Free and total device memory in bytes (hipMemGetInfo): a host that keeps device buffers between calls bounds its pool by
 * these, not by a constant. */
int cwt_device_memory(int device, size_t* free_bytes, size_t* total_bytes);
/* Blocks until everything queued on ANY stream of the device has finished (hipDeviceSynchronize): what a host needs before it
 * hands a kept buffer to a new owner. */
int cwt_device_synchronize(int device);
int cwt_malloc(int device, void** ptr_dev, size_t bytes);
int cwt_free(int device, void* ptr_dev);
int cwt_memcpy_h2d(cwt_plan* plan, void* dst_dev, const void* src_host, size_t bytes);
int cwt_memcpy_d2h(cwt_plan* plan, void* dst_host, const void* src_dev, size_t bytes);
/* Page-locked host memory (ordinary memory to the host; the GPU reads and writes it over PCIe).  A W_host of
 * cwt_execute_host that lies in such a buffer is written by the kernels themselves when the transform fits one
 * workgroup -- no staging copy: that copy is a third of what the reference's canonical 504-point call costs here.   */
int cwt_host_malloc(void** ptr_host, size_t bytes);
int cwt_host_free(void* ptr_host);

/* ---- hot path, device resident -----------------------------------------
 * Forward transform of the real signal, zero padded at the end to nfft:
 *   xhat[k] = sum_n x[n] exp(-2*pi*i*k*n/nfft)      (unnormalised)
 * replaces `fft.fft(signal, **fft_kwargs(signal))`, wavelet.py:91.
 * x_dev: n0 reals; xhat_dev: nfft complex.                                  */
int cwt_forward_fft(cwt_plan* plan, const void* x_dev, int64_t n0, void* xhat_dev);

/* Rows of the wavelet transform for explicit scales:
 *   W[j, n] = (1/nfft) sum_k xhat[k] F_j[k] exp(+2*pi*i*k*n/nfft),  n < ncols
 *   F_j[k]  = sqrt(scales[j]*w_1*nfft) * conj(psi_ft(scales[j]*w_k)),
 *   w_k     = 2*pi*fftfreq(nfft, dt)[k]
 * replaces wavelet.py:94 (ftfreqs), :102-104 (psi_ft_bar) and :105-106 (batched
 * ifft), and the [:, :n0] trim of :123 (ncols = n0).  The filter bank is never
 * materialised.  For CWT_PAUL the profile is 0 for w <= 0 (the reference yields
 * NaN rows where exp(-f) overflows, wavelet.py:111-115 then deletes them; the
 * Python shim deletes the same rows).  W_dev: nrows x ldw complex, row-major.  */
int cwt_transform_rows(cwt_plan* plan, const void* xhat_dev, int mother, double param, double dt,
                       const double* scales_host, int nrows, void* W_dev, int64_t ldw,
                       int64_t ncols);

/* The whole device-resident transform in one call -- wavelet.py:91 (forward FFT, written to xhat_dev: nfft complex, the
 * caller needs it for the 5th return value of wavelet.py:123-124) and :94-106 (rows of W) -- for callers that still hold
 * the signal.  Same results as cwt_forward_fft + cwt_transform_rows; knowing the real signal lets rows whose wavelet is
 * compact in time (filter not clipped at the Nyquist bins, c_H*scale/dt <= a quarter of the workgroup tile) take the
 * overlap-save form: per block of P - 2H output columns one P-point transform of x[n0-H .. n0+P-H) filtered by the same
 * psi_ft sampled on the block's coarser frequency grid -- no N-point inverse transform, no intermediate in memory,
 * contiguous stores.  The neglected tail of the wavelet is below a tenth of the plan's accuracy target
 * (cwt_plan_set_tolerance) of its L1 mass.  x_dev: n0 reals.  Option "ols" = 0 turns that form off (then exactly the
 * two calls above).
 * NON-FINITE SAMPLES: the reference transforms the whole padded signal, so one NaN / inf sample makes every element of
 * W NaN (wavelet.py:91).  cwt_forward_fft + cwt_transform_rows reproduce that; cwt_transform does so only for the rows
 * that go through the spectrum -- its overlap-save rows are NaN only in the output blocks whose input window contains
 * the sample.  A caller that must match the reference on such input checks the signal (O(n0)) and uses the two-call
 * path, as the Python shim does (pycwt_amd/wavelet.py:_transform).                                                   */
/* xhat_dev may be NULL when the caller has no use for the spectrum: it is then computed into plan scratch, and not at
 * all when every row takes the overlap-save form (a rank of a scale-sharded transform that owns only such rows). */
int cwt_transform(cwt_plan* plan, const void* x_dev, int64_t n0, int mother, double param, double dt,
                  const double* scales_host, int nrows, void* xhat_dev, void* W_dev, int64_t ldw,
                  int64_t ncols);

/* The same two steps at a transform length n0 that is NOT a power of two -- what the reference computes when pyfftw is
 * installed: helpers.py:15-19 then passes n = len(signal), i.e. no zero padding and circular edges -- by Bluestein's
 * chirp-z identity on this plan's power-of-two engine.  The plan must have nfft >= 2*n0 - 1.  xhat_dev: n0 complex
 * (FFT order, xhat[k] = sum_n x[n] exp(-2*pi*i*k*n/n0)); W_dev: nrows x ldw complex, n0 columns written per row;
 * w_k = 2*pi*fftfreq(n0, dt)[k].  Any nrows (processed in slabs of at most max_rows rows).                          */
int cwt_forward_fft_n(cwt_plan* plan, const void* x_dev, int64_t n0, void* xhat_dev);
int cwt_transform_rows_n(cwt_plan* plan, const void* xhat_dev, int64_t n0, int mother, double param, double dt,
                         const double* scales_host, int nrows, void* W_dev, int64_t ldw);

/* Batch of equally long signals sharing one scale grid (BASELINE config 4): xhat_dev holds nbatch
 * spectra (signal b at xhat_dev + b*xhat_ld, e.g. written by cwt_fft_rows), W_dev is
 * nbatch x nrows x ldw: W[b, j, :] = row j of signal b.  One set of launches covers the whole batch
 * (the row table simply has nbatch*nrows entries), so short series still fill the GPU.
 * Needs nbatch*nrows <= max_rows of the plan.                                                      */
int cwt_transform_rows_batch(cwt_plan* plan, const void* xhat_dev, int nbatch, int64_t xhat_ld,
                             int mother, double param, double dt, const double* scales_host,
                             int nrows, void* W_dev, int64_t ldw, int64_t ncols);

/* The same batch from the SIGNALS (x_dev: nbatch real series of n0 samples, signal b at x_dev + b*x_ld): the forward
 * transforms (spectra to xhat_dev, nbatch x nfft complex, as cwt_fft_rows writes them) and the rows in one call.  With
 * the signals at hand the time-compact rows take the overlap-save form of cwt_transform (block spectra per signal, the
 * filter tables shared by the signals); the batch counts towards "ols_min_logn" (nfft * nbatch >= 2^18 by default,
 * nfft >= 4 tiles), so a batch of 2^16-point series replaces its two-pass rows (pycwt has no batched call: this is the
 * loop `for x in signals: cwt(x, ...)` over wavelet.py:13-124 as one launch set).                                  */
int cwt_transform_batch(cwt_plan* plan, const void* x_dev, int nbatch, int64_t x_ld, int64_t n0, int mother,
                        double param, double dt, const double* scales_host, int nrows, void* xhat_dev, void* W_dev,
                        int64_t ldw, int64_t ncols);

/* Same transform for a mother wavelet that only exists as a Python object (the reference's
 * duck-typed protocol, mothers.py): the host evaluates psi_ft_bar = sqrt(s*w1*N)*conj(psi_ft(s*w)) as
 * wavelet.py:102-104 does and hands the filter bank over explicitly.
 * table_dev: nrows x nfft complex, row j = F_j[k] in FFT order; k_lo/nband: signed-bin support of
 * each row (k_lo >= -nfft/2, k_lo + nband <= nfft/2; bins outside are taken as exactly zero).       */
int cwt_transform_rows_table(cwt_plan* plan, const void* xhat_dev, const void* table_dev,
                             const int* k_lo_host, const int* nband_host, int nrows, void* W_dev,
                             int64_t ldw, int64_t ncols);

/* ---- building blocks of the callers of cwt (next rows of the hot-path table: Morlet.smooth,
 * xwt, wct).  All device resident, queued on the plan's stream.
 *
 * Forward FFT of every row of a matrix, zero padded from ncols_in to nfft:
 *   spec[r, k] = sum_n in[r, n] exp(-2*pi*i*k*n/nfft)
 * replaces `fft.fft(W, axis=1, **fft_kwargs(W[0, :]))` of Morlet.smooth, mothers.py:90.
 * in_dev: nrows x in_ld reals (in_complex = 0) or complex (in_complex = 1); spec_dev: nrows x nfft. */
int cwt_fft_rows(cwt_plan* plan, const void* in_dev, int in_complex, int nrows, int64_t in_ld,
                 int64_t ncols_in, void* spec_dev);

/* Generalised cwt_transform_rows: every row has its own spectrum, profile scale and amplitude:
 *   W[j, n] = (1/nfft) sum_k spec[j*spec_ld + k] * amp_j * profile(a_j * sk) * exp(+2*pi*i*k*n/nfft),
 * sk = signed bin index of k (k - nfft for k >= nfft/2), profile = the real profile of `mother`
 * (exp(-(f-f0)^2/2) | f^m exp(-f) [f>0] | f^m exp(-f^2/2)).  spec_ld = 0 shares one spectrum.
 * With mother = CWT_DOG, param = 0, a_j = (s_j/dt)*2*pi/nfft, amp = 1 this is the time smoothing
 * `ifft(F * fft(W))`, F = exp(-0.5*(s/dt)^2*k^2), of mothers.py:83-93.                            */
int cwt_filter_rows(cwt_plan* plan, const void* spec_dev, int64_t spec_ld, int mother, double param,
                    const double* a_host, const double* amp_re_host, const double* amp_im_host,
                    int nrows, void* W_dev, int64_t ldw, int64_t ncols);

/* Boxcar along the scale axis, = scipy.signal.convolve2d(T, win[:, None], 'same') with zero
 * boundary (mothers.py:100-102).  in/out: nrows x ld complex, distinct buffers.                    */
int cwt_boxcar_scales(cwt_plan* plan, const void* in_dev, int nrows, int64_t ld, int64_t ncols,
                      const double* win_host, int nwin, void* out_dev);

/* Cross wavelet spectrum (pycwt/wavelet.py:399): out[j, n] = W1[j, n] * conj(W2[j, n]); nrows x ld complex of
 * the plan's precision, n < ncols; out_dev may be W1_dev.                                            */
int cwt_cross_spectrum(cwt_plan* plan, const void* W1_dev, const void* W2_dev, int nrows, int64_t ld,
                       int64_t ncols, void* out_dev);

/* Element-wise inputs of the coherence (wavelet.py:503-514), all nrows x ld:
 *   P = (|W1|^2 + i*|W2|^2)/s   (the two auto-spectra packed into one complex matrix: the smoothing
 *                                kernel is real, so one smoothing pass serves both)
 *   C = W1*conj(W2)/s ,  angle = arg(W1*conj(W2))  (reals)                                         */
int cwt_wct_products(cwt_plan* plan, const void* W1_dev, const void* W2_dev, const double* scales_host,
                     int nrows, int64_t ld, int64_t ncols, void* P_dev, void* C_dev, void* angle_dev);
/* WCT = |S12|^2 / (S1*S2) with S = S1 + i*S2 (wavelet.py:513); out: nrows x ld reals.               */
int cwt_wct_coherence(cwt_plan* plan, const void* S_dev, const void* S12_dev, int nrows, int64_t ld,
                      int64_t ncols, void* out_dev);

/* Inverse transform, TC98 eq. 11 (wavelet.py:169-170):
 *   out[n] = coeff * sum_j Re(W[j, n]) / sqrt(scales[j]),   n < ncols
 * coeff = dj*sqrt(dt)/(cdelta*psi(0)) is applied by the caller's real part; the
 * (possibly complex) psi(0) division stays in the shim.  out_dev: ncols reals. */
int cwt_icwt_reduce(cwt_plan* plan, const void* W_dev, int64_t ldw, int64_t ncols, int nrows,
                    const double* scales_host, double coeff, void* out_dev);

/* Weighted reduction over scales with explicit weights:
 *   out[n] = coeff * sum_j g(W[j, n]) * weights[j],  g = Re (power = 0) or |.|^2 (power = 1).
 * power = 1 with weights 1/s_j on the selected scales (0 elsewhere) and coeff = dj*dt/cdelta is the
 * scale-averaged power of TC98 eq. 24 (sample/simple_sample.py:87-91).  out_dev: ncols reals.       */
int cwt_reduce_scales(cwt_plan* plan, const void* W_dev, int64_t ldw, int64_t ncols, int nrows,
                      const double* weights_host, int power, double coeff, void* out_dev);
/* Global wavelet spectrum: out[j] = mean_n |W[j, n]|^2 (power.mean(axis=1), simple_sample.py:79).
 * out_dev: nrows reals.                                                                            */
int cwt_time_mean_power(cwt_plan* plan, const void* W_dev, int64_t ldw, int64_t ncols, int nrows,
                        void* out_dev);

/* Monte-Carlo significance of the coherence (pycwt/wavelet.py:609-630, the `wlc` counter):
 *   hist[j, b] += #{ n in [lo[j], hi[j]) : floor(R2[j, n] * nbins) == b },  0 <= b < nbins,
 * NaNs and values outside [0, 1) are skipped (the reference would raise on R2 == 1).  [lo[j], hi[j]) is the
 * part of row j outside the cone of influence (contiguous: the COI is a triangle).  r2_dev: nrows x ld reals of
 * the plan's precision; lo_dev / hi_dev: nrows int64 on the device; max_span >= max_j (hi[j] - lo[j]) sizes the
 * launch; hist_dev: nrows x nbins uint64 on the device, accumulated across calls (zero it before the first).  */
int cwt_coherence_histogram(cwt_plan* plan, const void* r2_dev, int64_t ld, int nrows,
                            const int64_t* lo_dev, const int64_t* hi_dev, int64_t max_span, int nbins,
                            uint64_t* hist_dev);

/* Surrogate series of the Monte-Carlo significance made on the device (the loop of pycwt/wavelet.py:609-630 draws two
 * series per iteration with NumPy on one host thread: 0.10 s of a 0.13 s iteration at two 2^20-point series).
 * cwt_random_normal: out_dev[i] = scale * N(0, 1), i < n (reals of the plan's precision): Philox4x32-10 counter-based
 * generator + Box-Muller, reproducible per (seed, offset, i) whatever the launch geometry; `offset` names the series (draw
 * 2 k and 2 k + 1 of a loop, a rank's share of the draws, ...).  NOT NumPy's sequence: results agree with the host path
 * statistically, not seed for seed (that path stays the default).
 * cwt_ar1_filter: out_dev[j] = y[tau + j], j < n, of y[i] = g y[i-1] + e[i] over e_dev[0 .. tau + n), y[-1] = 0 --
 * scipy.signal.lfilter([1, 0], [1, -g], e, axis=0)[tau:], the AR(1) surrogate helpers.py:146-173 describes (the reference
 * itself filters along the wrong axis and gets white noise; both kinds are offered, see pycwt_amd.wct_significance).
 * Segments run in parallel, each from far enough back that the forgotten history is below 1e-17. */
int cwt_random_normal(cwt_plan* plan, uint64_t seed, uint64_t offset, int64_t n, double scale, void* out_dev);
int cwt_ar1_filter(cwt_plan* plan, const void* e_dev, int64_t tau, int64_t n, double g, void* out_dev);

/* ---- host convenience: what the ctypes shim of pycwt.cwt() calls ---------
 * x_host: n0 reals of the plan's precision.  W_host: nrows x n0 complex (may be
 * NULL).  xhat_host: nfft complex (may be NULL) for the 5th return value
 * (wavelet.py:123-124).  Synchronous.  Transforms that fit one workgroup per
 * row (nfft <= 4096) read the signal and write W through page-locked host memory,
 * without copy operations (option "host_direct", cwt_host_malloc).            */
int cwt_execute_host(cwt_plan* plan, const void* x_host, int64_t n0, int mother, double param,
                     double dt, const double* scales_host, int nrows, void* W_host,
                     void* xhat_host);

/* ---- measurement --------------------------------------------------------
 * With option "profile"=1 every kernel class launched since the last call is
 * timed with HIP events on the plan's stream.  Fills up to `cap` entries:
 * names[i] (static strings), total_ms[i], launches[i]; returns the entry count
 * in *n and resets the accumulators.  Synchronises the stream.              */
int cwt_plan_timings(cwt_plan* plan, int cap, const char** names, double* total_ms,
                     int* launches, int* n);
/* Which kernel computed each row of the last transform call: codes[out_row] = kind*10000 + logK*100 + terms with
 * kind 0 = single-workgroup transform, 1 = band-limited single pass (K = 2^logK <= 1024, `terms` aliased bins per
 * input), 2 = band-limited single pass on 16384-point workgroups (K = 2048), 3 = two-pass (logK = log2 of the
 * pass-A column support class, 0 = full column), 4 = overlap-save (K = 2^logK points per aliased block FFT, `terms` =
 * workgroups per block: 1 = blocks of one workgroup tile, 2 = blocks of two), 5 = overlap-save on half-size workgroup
 * tiles (short halos; option "ols_small_max_halo").
 * *n = number of rows of the call; codes may be NULL.  The parity
 * tests and bench.py use it to report the worst row per kernel class.                                          */
int cwt_plan_row_classes(cwt_plan* plan, int* codes, int cap, int* n);
/* The same codes for a transform that has not run yet: classifies `nrows` scales exactly as cwt_transform
 * (with_signal = 1) or cwt_transform_rows (0) with these arguments would.  No transform runs; the classified table is
 * kept in the plan's cache (its upload and, for overlap-save rows, its filter tables are queued on the plan's stream), so a
 * following transform call with the same arguments finds it ready.
 * pycwt_amd.parallel uses it to cut a scale grid into cost-balanced contiguous shards.                        */
int cwt_plan_classify(cwt_plan* plan, int mother, double param, double dt, const double* scales_host, int nrows,
                      int64_t ncols, int with_signal, int* codes);
/* ---- multi-GPU hosts ------------------------------------------------------
 * Rows (scales) of one transform are independent given the signal (wavelet.py:102-106), so a host in ANY language shards
 * them over its GPUs -- one plan per GPU, every rank calling cwt_transform on its own contiguous run of `scales`, the
 * signal broadcast once with whatever collective the host uses (pycwt_amd.parallel: RCCL through torch.distributed).
 * cwt_plan_balanced_shards cuts the grid into `world` CONTIGUOUS runs of about equal estimated cost: it classifies the rows
 * as cwt_transform would and prices a run as a per-launch part per kernel class present + a per-row part (fitted to
 * measured launch durations, profiles/r03_per_class.txt; per-row parts scale with nfft, the two-pass chunk limit is the
 * plan's own).  first[r], count[r] (r < world) = rank r's rows; identical on every rank (host arithmetic only).
 * cwt_shard_codes is the same search on row-class codes (cwt_plan_row_classes / cwt_plan_classify) without a plan:
 * nscale = nfft / 2^20, chunk_rows = rows per two-pass launch pair.                                                   */
int cwt_plan_balanced_shards(cwt_plan* plan, int mother, double param, double dt, const double* scales_host, int nrows,
                             int64_t ncols, int world, int* first, int* count);
int cwt_shard_codes(const int* codes, int nrows, int precision, double nscale, int chunk_rows, int world, int* first,
                    int* count);
/* The model's estimate (microseconds per step) for a rank that owns exactly the rows with these codes. */
int cwt_shard_cost(const int* codes, int nrows, int precision, double nscale, int chunk_rows, double* cost_us);

/* How the last cwt_transform_rows / cwt_transform call split its rows: counts[0] = rows done by the single-workgroup
 * kernel, [1] = band-limited single pass with K <= 1024 and at most 4 aliased terms, [2] = two-pass, [3] = band-limited
 * single pass with K = 2048 (fp64, 16384-point workgroups), [4] = band-limited single pass with K = 1024 and 5..16
 * aliased terms, [5] = overlap-save. */
int cwt_plan_last_split(cwt_plan* plan, int counts[6]);
/* The same with counts[6] = rows clipped at the Nyquist bins that ran as overlap-save rows on the band-passed complex
 * signal (k_aols_*; before round 4 these were two-pass rows) and counts[7] = band-limited rows in polynomial form
 * (k_poly_coef + k_poly_rows; counts[1], [3], [4] then only hold the band-limited rows that did not fit that form). */
int cwt_plan_last_split8(cwt_plan* plan, int counts[8]);

#ifdef __cplusplus
}
#endif
#endif /* CWT_HIP_H */
