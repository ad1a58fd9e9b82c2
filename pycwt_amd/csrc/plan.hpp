// plan.hpp -- the plan (struct cwt_plan) and the host-side internals shared by the translation units of libcwt_hip.so:
//   plan_host.cpp    row classification (build_row_table and its searches), row-table cache, scratch buffers, host copies,
//                    shard cost model -- plain C++ against the HIP runtime API, no kernels
//   launch_impl.hpp  every kernel launch, as templates over the precision; instantiated by launch_f64.hip / launch_f32.hip
//   abi.hip          the exported C functions (include/cwt_hip.h)
// Everything here lives in namespace cwtd ("detail"); nothing is exported but the C ABI.
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "cwt_hip.h"
#include "cwt_types.hpp"

#ifndef CWT_BACKEND_NAME
#define CWT_BACKEND_NAME "hip-gfx950"
#endif

namespace cwtd {
using namespace cwt;

extern thread_local std::string g_err;
// bumped whenever a plan scratch buffer is freed and reallocated (grow, ensure_z): part of the key of a captured HIP graph,
// whose kernels have those pointers baked in (option "graph")
extern uint64_t g_scratch_gen;
int fail(int code, const std::string& msg);

#define HIPCHECK(expr)                                                                        \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(CWT_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));               \
  } while (0)

enum KernelClass { KC_FWD_SMALL, KC_FWD_A, KC_FWD_B, KC_SMALL, KC_DIRECT, KC_NARROW, KC_NARROW_MANY, KC_NARROW_BIG,
                   KC_PASS_A, KC_PASS_B, KC_ICWT, KC_ELEMENTWISE, KC_OLS_FWD, KC_OLS, KC_OLS_SMALL, KC_AOLS_PRE, KC_AOLS,
                   KC_POLY_COEF, KC_POLY, KC_COUNT };
extern const char* const kClassNames[KC_COUNT];

int ilog2(int64_t v);

struct Timed { int cls; hipEvent_t a, b; };

struct HostCopier;
HostCopier* copier_for(int device);

}  // namespace cwtd

struct cwt_plan {
  int device = 0;
  int logN = 0;
  int64_t N = 0;
  int prec = 64;
  int max_rows = 0;
  hipStream_t stream = nullptr;
  // options
  int chunk_rows = 0;      // rows per two-pass chunk; 0 = as many as fit 192 MiB of intermediate, which
                           // stays inside the 256 MiB Infinity Cache (12 rows at N = 2^20 fp64)
  int narrow = 1;
  int narrow_max_logk = 10;
  int loglmax = 12;
  int log_wg_points = 13;
  int profile = 0;
  int use_ct = 1;          // compile-time specialised kernels where the geometry matches
  int narrow_terms = 4;    // band-limited path: up to this many aliased bins per FFT input at K = 1024 (<= 16)
  int big_terms = 6;       // ... and at K = 2048 (fp64, 16384-point workgroups; <= 8)
  int pass_a_small = 1;      // pass A on half-size workgroup tiles (4 per CU instead of 2): -5 % fp64, -8 % fp32
  int narrow_small = 1;    // complex64: K <= 512 band-limited rows on half-size tiles
  int narrow_mix = 0;      // launch order of the band-limited rows alternates light and heavy rows (default: on for
                           // precision 64 -- measured -3.5 % on that kernel, -2 % on the step; +-0 / -2 % in fp32)
  int big_tiles = 1;       // complex128, R = 4096: pass A on 16384-point tiles
  int force_logk = 0;
  int narrow_big = 1;      // fp64: K = 2048 single-pass rows on 16384-point workgroups
  int overlap_narrow = 1;  // band-limited rows on a side stream beside the two-pass chain (measured: +4 % in
                           // fp64); ignored while "profile" is on so that every timed kernel runs alone
  int band_pass_a = 1;     // pass A with short aliased column FFTs for rows of moderate support
  int ols = 1;             // overlap-save rows (time-compact wavelets) when the call hands over the signal itself
  int ols_side = 1;        // their block spectra on a side stream beside the two-pass chain
  int ols_early = 1;       // cwt_transform: the whole overlap-save chain on a side stream, queued before the forward FFT
  int poly = 1;            // band-limited rows in polynomial form (k_poly_coef + k_poly_rows) where they fit
  int poly_carrier = 1;    // polynomial rows: the carrier bin chosen by the filter-weighted degree bound (0 = always the band's centre)
  int poly_cheb = 1;       // polynomial rows: e^{i theta u} cut as its Chebyshev series (Jacobi-Anger), re-expanded in monomials -- the error of
                           // degree D is 2 (theta/2)^(D+1) / (D+1)! instead of Taylor's theta^(D+1) / (D+1)!: 2^D smaller, i.e. half the
                           // intervals K' at the same degree for most rows (0 = Taylor)
  int poly_degree = 8;     // preferred largest degree: the interval count K' of a row is the smallest that needs no more
  int poly_min_logn = 16;  // shortest transform that takes the form
  int poly_max_logk = 14;  // largest log2 K' (tuning: 13 keeps the rows that need 16384 intervals out of the form)
  int coef_small = 0;      // interval coefficients of every K' in one launch of 256-thread workgroups (K' = 8192 / 16384 split in 2 / 4); measured slower (EXPERIMENTS R6.2)
  int poly_chunk_mb = 96;  // coefficient planes computed and consumed per chunk of polynomial rows (MiB; 0 = all rows at once)
  int host_direct = 1;     // cwt_execute_host, transforms that fit one workgroup: the kernels read the signal from / write W into page-locked host memory
  int graph = 0;           // cwt_transform: capture the launches of a repeated call (same buffers, same row table) into a
                           // HIP graph on its second occurrence and replay it from the third on
  int aols = 1;            // rows clipped at Nyquist as overlap-save rows on the band-passed complex signal (k_aols_*)
  int aols_zc = 1;         // Paul rows not clipped at Nyquist on the band-passed signal too, their profile continued through f = 0
  int aols_long = 1;       // complex128: clipped rows with halos of 512 ... 2048 samples in the second (8192-point) class of the band-passed rows
  int aols_min_rows = 3;   // ... if at least this many rows qualify (the band-passed signal costs about one two-pass row)
  int serial_rows = 2;     // (complex128; complex64 plans start at 0: measured +-0 ... +1.5 % there) long transforms with polynomial rows: every kernel that writes W on the caller's stream, one after the
                           // other, the preparation on the side streams (rows_launch_serial); 2 = also the first block spectra on the
                           // caller's stream (its rows follow at a kernel boundary) and the forward FFT on side stream 0
  int serial_s1_once = 1;  // serial schedule: the caller's stream waits ONCE for side stream 1 (block spectra of the longer blocks, band-passed signal
                           // and its block spectra: one in-order chain) instead of once per consumer
  hipEvent_t spectrum_ready = nullptr;   // (transient) set by cwt_transform when the forward FFT ran on side stream 0
  int fft_aside_small = 1; // serial_rows = 2: the forward FFT (on side stream 0 beside the first overlap-save rows) on half-size tiles
  int aols_small_b = 1;    // serial schedule, complex128: the band-passed signal's second pass on 4096-point tiles (256-thread workgroups)
  int fft_small = 0;       // (transient) set by cwt_transform while the forward FFT is queued on side stream 0 (serial_rows = 2)
  int ols_launched = 0;    // (transient) set by cwt_transform for rows_impl
  int ols_first_on_main = 0;   // (transient) serial_rows = 2: the block spectra of the half-size tiles were queued on the caller's stream
  int64_t ols_x_ld = 0;    // (transient) set by cwt_transform_batch: elements between the signals of the batch
  int ols_min_logn = 18;   // shortest transform that takes the form (measured: 2^18 +12 %, 2^17 -10 %, 2^16 -13 %)
  int ols_small_max_halo = 512;   // rows with a halo up to this many samples run on half-size tiles (0 = none)
  int ols_small_big = 1;   // half-size tiles: rows with a halo in (ols_small_max_halo, 1024] and a block support <= 1/8 tile
                           // on blocks of TWO half-size tiles (8192 points, two 256-thread workgroups per block) instead of the default
                           // tile (one 512-thread workgroup per 8192-point block: two per CU, the slowest row kernel of the step)
  int ols_big = 1;         // tile 8192: blocks of 2P points for rows with long halos (two workgroups per block)
  int ols_big_min_halo = 1536;   // measured: equal cost below (strided segments + twice the twiddle range against the kept fraction)
  int ols_big4_min_halo = 2048;  // ols_big = 2: rows with a halo from here on use blocks of 4P points (four workgroups per block)
  int ols_big4_max_halo = 8192;  // ... up to this halo (a quarter of the block at most)
  int ols_max_halo = 0;    // largest halo H of such a row in samples; 0 = a quarter of the workgroup tile (L >= P/2)
  double ols_fwd_weight = 1.0;   // cost of one block spectrum in units of one row's block transform (class grouping)
  // Accuracy target of a row, max|dW| / max|W| against the exact transform (cwt_plan_set_tolerance; 0 = the precision's
  // default).  The three truncations of the fast forms are derived from it (see tolerances()).
  double tolerance = 0.0;
  double auto_target = 0.0;   // > 0: cwt_execute_host derives the tolerance of each call from this target and the measured
                              // dynamic range of the call's spectrum (cwt_plan_set_auto_tolerance)
  double last_range = 0.0;    // max|xhat| / rms|xhat| of the last such call
  double* range_dev = nullptr;
  // device resources
  void* tw_all = nullptr;   // e^{2 pi i p / L} for L = 2,4,..,16384; table of L starts at L-2
  void* twn_lo = nullptr;   // e^{2 pi i i / N}, i < 2^twn_shift
  int twn_shift = 0;
  void* weights_dev = nullptr;
  void* weights_pinned[2] = {nullptr, nullptr};   // staging of upload_reals, used in turn
  hipEvent_t weights_ev[2] = {nullptr, nullptr};  // recorded after the copy out of weights_pinned[i]
  int weights_turn = 0;
  void* Z = nullptr;
  size_t z_bytes = 0;
  void* xs = nullptr;       // block spectra of the overlap-save rows
  size_t xs_bytes = 0;
  void* pcoef = nullptr;    // interval coefficients of the polynomial rows
  size_t pcoef_bytes = 0;
  void* pband = nullptr;    // their filtered bands in transform-input order
  size_t pband_bytes = 0;
  void* xm = nullptr;       // band-passed complex signal x_M of the k_aols rows (N complex)
  size_t xm_bytes = 0;
  void* xsa = nullptr;      // its block spectra (nblocks x (P + 8) complex)
  size_t xsa_bytes = 0;
  // buffers of cwt_execute_host
  void* hstage = nullptr; size_t hstage_bytes = 0;   // page-locked staging of its small calls (signal in, W and spectrum out)
  void* hx = nullptr; size_t hx_bytes = 0;
  void* hxhat = nullptr; size_t hxhat_bytes = 0;
  void* hW = nullptr; size_t hW_bytes = 0;
  // Classified row tables with their device copies.  Four slots (the accuracy target is part of the key), least recently used one rebuilt on a miss, so that
  // callers that alternate between two kinds of calls with fixed arguments (the coherence pipeline: cwt rows, then
  // the smoothing filter rows, draw after draw) build and upload each table once.  No host synchronisation on the
  // way: every slot has its own pinned staging buffer and an event that marks its last copy as done.
  struct Group { int logK; int first; int count; int nterms; };
  struct RowTable {
    std::vector<double> key;             // the call it was built from; empty = not valid
    std::vector<cwt::RowDesc> table;          // ordered: [small | narrow classes by logK | wide]
    std::vector<Group> narrow_groups;
    int n_small = 0, n_narrow = 0, n_wide = 0, wide_first = 0;
    int n_ols = 0, ols_first = 0;        // overlap-save rows (after the wide rows), sorted by halo class
    // The overlap-save rows run on up to two workgroup-tile sizes: group 0 = half-size tiles (short halos: four tiles
    // in flight per CU instead of two; measured -10...-20 % per row, profiles/r03_ols_tiles.txt), group 1 = the default tile
    struct OlsGroup {
      int logp = 13;                     // log2 of the workgroup tile
      cwt::OlsClasses cls;                    // halo classes of this group (wg_first / row_first relative to the group)
      long wgs = 0;                      // workgroups of its k_ols_ct launch
      long wgs_base = 0;                 // ... of which the classes on blocks of ONE tile come first (their block spectra are the first to exist)
      long fwd_blocks[3] = {0, 0, 0};    // blocks of P, 2P, 4P points (k_ols_fwd_r launches)
      int row_first = 0, nrows = 0;      // its rows inside [ols_first, ols_first + n_ols)
    };
    OlsGroup ols_grp[2];
    long ols_xs_elems = 0, ols_gt_elems = 0;
    int ols_nbatch = 1;                  // signals of a batched call (cwt_transform_batch): block spectra per signal,
    long ols_xs_sig = 0;                 // ols_xs_sig elements apart; the rows carry their signal's offset in spec_off
    void* gt_dev = nullptr;              // filter tables of the overlap-save rows, written when the table is built
    size_t gt_bytes = 0;
    // rows clipped at Nyquist on the band-passed complex signal (after the overlap-save rows), one halo class; the
    // table entry at aux_first is the pseudo-row whose "filter" is the mask (profile 1 on the bins [k_s, N/2))
    // band-limited rows in polynomial form (at the end of the table), grouped by K'
    int n_poly = 0, poly_first = 0;
    // ... in chunks of bounded coefficient volume (largest K' first): the planes of a chunk are computed, then consumed by
    // k_poly_rows while they still sit in the Infinity Cache -- with all rows' planes (80 - 300 MB) computed first the
    // coefficient fetches of the streaming kernel come from HBM and it loses 10 - 40 % (tests/perf/poly_chunks.py)
    struct PolyChunk {
      int row_first = 0, nrows = 0, max_logk = 8;   // rows relative to poly_first
      cwt::PolyClasses cls{};                            // (row_first of a class relative to the chunk)
      long wgs[3] = {0, 0, 0};                      // workgroups of the k_poly_coef launches on 4096- / 8192- / 16384-point tiles
      long wgs_all = 0;                             // ... of the single launch of k_poly_coef_all (option "coef_small")
    };
    std::vector<PolyChunk> poly_chunks;
    long poly_coef_elems = 0, poly_band_elems = 0;
    // tables of the economised monomial weights, one per (K', D) pair of the table: (D + 1) x (K' + 1) reals at `off` of prt_dev
    struct PolyRtab { int logK, deg; long off; };
    std::vector<PolyRtab> poly_rtabs;
    long poly_rtab_elems = 0;
    void* prt_dev = nullptr;
    size_t prt_bytes = 0;
    int n_aols = 0, aols_first = 0, aux_first = -1, aols_logp = 12;
    int aols_nbatch = 1;                 // signals of a batched call: aols_geom.nrows rows and one mask pseudo-row (aux_first + b) each
    cwt::AolsGeom aols_geom{};
    long aols_wgs = 0, aols_gt_elems = 0;
    // second class of such rows (Paul continued through f = 0, 8192-point tiles): n_aols2 of the n_aols rows, at aols2_first
    int n_aols2 = 0, aols2_first = 0;
    cwt::AolsGeom aols2_geom{};
    long aols2_wgs = 0;
    void* agt_dev = nullptr;             // their (real) filter tables
    size_t agt_bytes = 0;
    cwt::RowDesc* rows_dev = nullptr;
    cwt::RowDesc* rows_pinned = nullptr;
    hipEvent_t uploaded = nullptr;
    uint64_t used = 0;
    uint64_t build_id = 0;               // changes whenever the table is rebuilt (graphs captured over it are stale then)
  };
  RowTable slots[4];
  RowTable* rt = &slots[0];
  uint64_t tick = 0;
  int split[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // rows: single-workgroup, band-limited K <= 1024 with <= 4 terms, two-pass,
                                          // band-limited K = 2048, band-limited K = 1024 with 5..16 terms, overlap-save,
                                          // overlap-save on the band-passed complex signal, polynomial form
  // Bluestein state for transform lengths n0 that are not powers of two (this plan's N is then M >= 2 n0 - 1)
  int64_t bs_n0 = 0;
  void* bs_khat[2] = {nullptr, nullptr};   // FFT_M of the chirp kernels: [0] e^{+pi i m^2/n0} (forward), [1] conjugate
  void* bs_a = nullptr; size_t bs_a_bytes = 0;          // chirp-premultiplied rows, slab x n0
  void* bs_spec = nullptr; size_t bs_spec_bytes = 0;    // their spectra, slab x M
  void* bs_par = nullptr; size_t bs_par_bytes = 0;      // per-row a, amp_re, amp_im (doubles)
  // HIP graphs of repeated cwt_transform calls (option "graph"): key = the call's buffers + the row table's identity
  struct GraphSlot { std::vector<uint64_t> key; hipGraphExec_t exec = nullptr; int seen = 0; uint64_t used = 0; };
  GraphSlot graphs[4];
  uint64_t graph_replays = 0;
  std::vector<cwtd::Timed> timed;
  std::vector<hipEvent_t> free_events;
  hipStream_t side[2] = {nullptr, nullptr};       // side streams of the two-pass pipeline
  hipEvent_t ev_ols = nullptr;
  hipStream_t side2 = nullptr;       // third side stream: the multi-term band-limited kernels beside the one-term kernel
  // Hardware queues (ensure_distinct_queues, abi.hip): the runtime multiplexes streams onto a few hardware queues (4 by
  // default) in creation order; two of the plan's four streams on ONE queue run in submission order and lose their overlap
  int queue_probe = 1;               // option "queue_probe"
  bool queues_probed = false;
  hipStream_t probed_main = nullptr; // the caller's stream the side streams were checked against last
  std::vector<hipStream_t> probed_streams;   // ... and every caller's stream checked since the side streams last changed (<= 8)
  std::vector<hipStream_t> spacers;  // streams that collided: kept (idle) until the plan goes, so that their replacements land elsewhere
  int* probe_dev = nullptr;          // flag + result of the probe kernels
  hipEvent_t ev_probe = nullptr;
  int queue_collisions = 0;          // streams replaced so far (cwt_plan_get_option "queue_collisions")
  hipEvent_t ev_big = nullptr;
  hipEvent_t ev_fork = nullptr, ev_a[2] = {nullptr, nullptr}, ev_b[2] = {nullptr, nullptr};

  size_t esize() const { return prec == 64 ? sizeof(double) : sizeof(float); }
};

namespace cwtd {


// Default accuracy targets: every truncation of the fast forms below the arithmetic's own rounding.  The truncations are
// relative to the FILTER's peak, so the error they leave relative to a row's own peak grows with the dynamic range of the
// signal's spectrum; a caller that knows its spectra (bench.py: white noise) or measures them (cwt_spectrum_range; the
// automatic mode of cwt_execute_host, cwt_plan_set_auto_tolerance) passes a looser target and gets the faster forms.
constexpr double kDefaultTolerance64 = 1e-16, kDefaultTolerance32 = 1e-8;
// The truncations that make the fast forms possible, all derived from the one accuracy target tol of the plan:
//   support  bins whose profile is below this fraction of its peak are treated as exactly zero (band limiting);
//   halo     neglected fraction of the L1 mass of |psi| beyond the overlap-save halo (a bound on the relative error);
//   clip     a row counts as "not clipped at Nyquist" (time-compact wavelet) if its profile at the Nyquist bins is below
//            this fraction of its peak (measured error of the overlap-save form: about a tenth of the fraction).
// Each is floored where the arithmetic's own rounding takes over.
struct Tolerances { double support, halo, clip; };
Tolerances tolerances(const cwt_plan* p);

// ---- plan_host.cpp ----
int get_event(cwt_plan* p, hipEvent_t* e);
int mother_constant(int mother, double param, double* cre, double* cim);
size_t table_capacity(int max_rows);
int build_row_table(cwt_plan* p, int mother, double param, const double* a, const double* amp_re,
                    const double* amp_im, int64_t spec_ld, int nrows, const int* tab_klo = nullptr,
                    const int* tab_nband = nullptr, int rows_per_signal = 0, int64_t tab_ld = -1,
                    int64_t ols_ncols = 0, int64_t out_ncols = 0);
void set_split(cwt_plan* p);
bool serial_schedule(const cwt_plan* p, bool ols_early);
int chunk_rows_of(const cwt_plan* p);
int balanced_chunk(const cwt_plan* p, int nrows);
int two_pass_logk(const cwt_plan* p);
int check_geometry(const cwt_plan* p);
int ensure_z(cwt_plan* p, int rows);
int grow(void** buf, size_t* have, size_t need, hipStream_t s);
int copy_d2h(cwt_plan* p, void* dst_host, const void* src_dev, size_t bytes);
int ensure_distinct_queues(cwt_plan* p);          // abi.hip: the plan's four streams on four hardware queues
hipError_t create_side_stream(hipStream_t* s);
bool select_table(cwt_plan* p, const std::vector<double>& key);
int upload_row_table(cwt_plan* p, const std::vector<double>& key);
std::vector<double> call_key(double kind, std::initializer_list<double> head, std::initializer_list<std::pair<const double*, int>> arrays);

}  // namespace cwtd
