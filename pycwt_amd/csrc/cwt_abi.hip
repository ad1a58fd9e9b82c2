// cwt_abi.hip -- host side of libcwt_hip.so: plan, row classification, launches, C ABI.
// See include/cwt_hip.h for the contract and the reference lines each entry point replaces.
#include <hip/hip_runtime.h>

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
#include "cwt_kernels.hpp"

#ifndef CWT_BACKEND_NAME
#define CWT_BACKEND_NAME "hip-gfx950"
#endif

using namespace cwt;

namespace {

thread_local std::string g_err;
// bumped whenever a plan scratch buffer is freed and reallocated (grow, ensure_z): part of the key of a captured HIP graph,
// whose kernels have those pointers baked in (option "graph")
uint64_t g_scratch_gen = 0;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIPCHECK(expr)                                                                        \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(CWT_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));               \
  } while (0)

enum KernelClass { KC_FWD_SMALL, KC_FWD_A, KC_FWD_B, KC_SMALL, KC_DIRECT, KC_NARROW, KC_NARROW_MANY, KC_NARROW_BIG,
                   KC_PASS_A, KC_PASS_B, KC_ICWT, KC_ELEMENTWISE, KC_OLS_FWD, KC_OLS, KC_OLS_SMALL, KC_AOLS_PRE, KC_AOLS,
                   KC_POLY_COEF, KC_POLY, KC_COUNT };
const char* const kClassNames[KC_COUNT] = {"fwd_small", "fwd_pass_a",  "fwd_pass_b", "small",  "direct", "narrow",
                                           "narrow_many", "narrow_big", "pass_a",     "pass_b", "icwt",   "elementwise",
                                           "ols_fwd", "ols", "ols_small", "aols_pre", "aols", "poly_coef", "poly"};

int ilog2(int64_t v) {
  int l = 0;
  while ((int64_t(1) << l) < v) ++l;
  return l;
}

struct Timed { int cls; hipEvent_t a, b; };

// Device -> pageable host copies of results (the W matrix of the drop-in call is GiBs of fresh NumPy memory).  One
// hipMemcpyAsync into pageable memory runs at ~12 GB/s on this platform (single staging thread + first-touch page
// faults); page-locking the caller's array costs more than it saves.  Here the copy is cut into chunks that the DMA
// engine writes into a ring of page-locked slots (allocated once per plan) while a few worker threads memcpy the
// previous chunks into the caller's memory, each thread touching its own pages.
struct HostCopier {
  std::mutex busy;                       // one large copy at a time per device (the copier is shared by its plans)
  static constexpr int kSlots = 3;
  static constexpr size_t kChunk = size_t(32) << 20;
  int kThreads = 8;   // worker threads: CWT_COPY_THREADS, default min(32, cores / 2) -- first-touch page faults of the
                      // destination dominate, and they scale with the number of threads touching distinct pages
  void* slot[kSlots] = {nullptr, nullptr, nullptr};
  hipEvent_t ev[kSlots] = {nullptr, nullptr, nullptr};
  struct Task { char* dst; const char* src; size_t n; int slot; };
  std::vector<std::thread> workers;
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  std::deque<Task> q;
  int pending[kSlots] = {0, 0, 0};
  bool stop = false;

  bool ready() const { return slot[0] != nullptr; }
  int init() {
    const unsigned hc = std::thread::hardware_concurrency();
    kThreads = int(std::max(1u, std::min(32u, hc / 2)));
    if (const char* e = std::getenv("CWT_COPY_THREADS")) kThreads = std::max(1, std::min(256, std::atoi(e)));
    for (int i = 0; i < kSlots; ++i) {
      if (hipHostMalloc(&slot[i], kChunk) != hipSuccess || hipEventCreate(&ev[i]) != hipSuccess) return -1;
    }
    for (int t = 0; t < kThreads; ++t) workers.emplace_back([this] { run(); });
    return 0;
  }
  void run() {
    for (;;) {
      Task t;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_work.wait(lk, [this] { return stop || !q.empty(); });
        if (q.empty()) return;
        t = q.front();
        q.pop_front();
      }
      std::memcpy(t.dst, t.src, t.n);
      {
        std::lock_guard<std::mutex> lk(m);
        if (--pending[t.slot] == 0) cv_done.notify_all();
      }
    }
  }
  void wait_slot(int i) {
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [this, i] { return pending[i] == 0; });
  }
  // hands the `n` bytes that sit in slot i to the workers, in kThreads pieces
  void scatter(int i, char* dst, size_t n) {
    const size_t piece = ((n + kThreads - 1) / kThreads + 4095) & ~size_t(4095);
    std::lock_guard<std::mutex> lk(m);
    for (size_t off = 0; off < n; off += piece) {
      q.push_back({dst + off, static_cast<const char*>(slot[i]) + off, std::min(piece, n - off), i});
      ++pending[i];
    }
    cv_work.notify_all();
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv_work.notify_all();
    for (auto& w : workers) w.join();
    workers.clear();
    for (int i = 0; i < kSlots; ++i) {
      if (slot[i]) (void)hipHostFree(slot[i]);
      if (ev[i]) (void)hipEventDestroy(ev[i]);
      slot[i] = nullptr; ev[i] = nullptr;
    }
  }
};

// One copier per device for the whole process (created by the first large copy, never torn down: its worker threads and
// pinned slots are shared by every plan of the device instead of living and dying with each plan).
HostCopier* copier_for(int device) {
  static std::mutex mu;
  static std::vector<HostCopier*> all;
  std::lock_guard<std::mutex> lk(mu);
  if (device < 0) return nullptr;
  if (size_t(device) >= all.size()) all.resize(size_t(device) + 1, nullptr);
  if (!all[device]) {
    HostCopier* c = new HostCopier();
    if (c->init() != 0) {
      (void)hipGetLastError();
      c->shutdown();
      delete c;
      return nullptr;
    }
    all[device] = c;
  }
  return all[device];
}

}  // namespace

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
  int poly_degree = 8;     // preferred largest degree: the interval count K' of a row is the smallest that needs no more
  int poly_min_logn = 16;  // shortest transform that takes the form
  int poly_max_logk = 14;  // largest log2 K' (tuning: 13 keeps the rows that need 16384 intervals out of the form)
  int poly_chunk_mb = 96;  // coefficient planes computed and consumed per chunk of polynomial rows (MiB; 0 = all rows at once)
  int host_direct = 1;     // cwt_execute_host, transforms that fit one workgroup: the kernels read the signal from / write W into page-locked host memory
  int graph = 0;           // cwt_transform: capture the launches of a repeated call (same buffers, same row table) into a
                           // HIP graph on its second occurrence and replay it from the third on
  int aols = 1;            // rows clipped at Nyquist as overlap-save rows on the band-passed complex signal (k_aols_*)
  int aols_min_rows = 3;   // ... if at least this many rows qualify (the band-passed signal costs about one two-pass row)
  int ols_launched = 0;    // (transient) set by cwt_transform for rows_impl
  int64_t ols_x_ld = 0;    // (transient) set by cwt_transform_batch: elements between the signals of the batch
  int ols_min_logn = 18;   // shortest transform that takes the form (measured: 2^18 +12 %, 2^17 -10 %, 2^16 -13 %)
  int ols_small_max_halo = 512;   // rows with a halo up to this many samples run on half-size tiles (0 = none)
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
  // Classified row tables with their device copies.  Two slots, least recently used one rebuilt on a miss, so that
  // callers that alternate between two kinds of calls with fixed arguments (the coherence pipeline: cwt rows, then
  // the smoothing filter rows, draw after draw) build and upload each table once.  No host synchronisation on the
  // way: every slot has its own pinned staging buffer and an event that marks its last copy as done.
  struct Group { int logK; int first; int count; int nterms; };
  struct RowTable {
    std::vector<double> key;             // the call it was built from; empty = not valid
    std::vector<RowDesc> table;          // ordered: [small | narrow classes by logK | wide]
    std::vector<Group> narrow_groups;
    int n_small = 0, n_narrow = 0, n_wide = 0, wide_first = 0;
    int n_ols = 0, ols_first = 0;        // overlap-save rows (after the wide rows), sorted by halo class
    // The overlap-save rows run on up to two workgroup-tile sizes: group 0 = half-size tiles (short halos: four tiles
    // in flight per CU instead of two; measured -10...-20 % per row, profiles/r03_ols_tiles.txt), group 1 = the default tile
    struct OlsGroup {
      int logp = 13;                     // log2 of the workgroup tile
      OlsClasses cls;                    // halo classes of this group (wg_first / row_first relative to the group)
      long wgs = 0;                      // workgroups of its k_ols_ct launch
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
      PolyClasses cls{};                            // (row_first of a class relative to the chunk)
      long wgs[3] = {0, 0, 0};                      // workgroups of the k_poly_coef launches on 4096- / 8192- / 16384-point tiles
    };
    std::vector<PolyChunk> poly_chunks;
    long poly_coef_elems = 0, poly_band_elems = 0;
    int n_aols = 0, aols_first = 0, aux_first = -1, aols_logp = 12;
    int aols_nbatch = 1;                 // signals of a batched call: aols_geom.nrows rows and one mask pseudo-row (aux_first + b) each
    AolsGeom aols_geom{};
    long aols_wgs = 0, aols_gt_elems = 0;
    void* agt_dev = nullptr;             // their (real) filter tables
    size_t agt_bytes = 0;
    RowDesc* rows_dev = nullptr;
    RowDesc* rows_pinned = nullptr;
    hipEvent_t uploaded = nullptr;
    uint64_t used = 0;
    uint64_t build_id = 0;               // changes whenever the table is rebuilt (graphs captured over it are stale then)
  };
  RowTable slots[2];
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
  std::vector<Timed> timed;
  std::vector<hipEvent_t> free_events;
  hipStream_t side[2] = {nullptr, nullptr};       // side streams of the two-pass pipeline
  hipEvent_t ev_ols = nullptr;
  hipStream_t side2 = nullptr;       // third side stream: the multi-term band-limited kernels beside the one-term kernel
  hipEvent_t ev_big = nullptr;
  hipEvent_t ev_fork = nullptr, ev_a[2] = {nullptr, nullptr}, ev_b[2] = {nullptr, nullptr};

  size_t esize() const { return prec == 64 ? sizeof(double) : sizeof(float); }
};

namespace {

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
Tolerances tolerances(const cwt_plan* p) {
  const double t = p->tolerance > 0 ? p->tolerance : (p->prec == 64 ? kDefaultTolerance64 : kDefaultTolerance32);
  Tolerances r;
  r.support = std::max(t * 0.1, p->prec == 64 ? 1e-18 : 1e-9);
  r.halo = std::max(t * 0.1, p->prec == 64 ? 1e-17 : 5e-7);
  r.clip = std::max(t, p->prec == 64 ? 1e-16 : 1e-8);
  return r;
}

template <typename T>
const cplx<T>* tw_table(const cwt_plan* p, int logL) {
  return static_cast<const cplx<T>*>(p->tw_all) + ((size_t(1) << logL) - 2);
}

template <typename T>
TwN<T> twn_of(const cwt_plan* p) {
  TwN<T> t;
  t.shift = p->twn_shift;
  t.lo = static_cast<const cplx<T>*>(p->twn_lo);
  t.hi = tw_table<T>(p, p->logN - p->twn_shift);
  return t;
}

int get_event(cwt_plan* p, hipEvent_t* e) {
  if (!p->free_events.empty()) {
    *e = p->free_events.back();
    p->free_events.pop_back();
    return CWT_OK;
  }
  HIPCHECK(hipEventCreate(e));
  return CWT_OK;
}

// Runs `launch()` (which enqueues exactly one kernel class) and, when profiling, brackets it with
// HIP events on the plan's stream.
template <class F>
int timed_launch(cwt_plan* p, int cls, F&& launch, hipStream_t stream) {
  if (!p->profile) {
    launch();
    HIPCHECK(hipGetLastError());
    return CWT_OK;
  }
  Timed t;
  t.cls = cls;
  int rc = get_event(p, &t.a);
  if (rc) return rc;
  rc = get_event(p, &t.b);
  if (rc) return rc;
  HIPCHECK(hipEventRecord(t.a, stream));
  launch();
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipEventRecord(t.b, stream));
  p->timed.push_back(t);
  return CWT_OK;
}
template <class F>
int timed_launch(cwt_plan* p, int cls, F&& launch) {
  return timed_launch(p, cls, launch, p->stream);
}

template <typename T>
int build_tables(cwt_plan* p) {
  const long double two_pi = 6.283185307179586476925286766559L;
  std::vector<cplx<T>> all(32766);
  for (int l = 1; l <= 14; ++l) {
    const size_t L = size_t(1) << l;
    for (size_t i = 0; i < L; ++i) {
      const long double ang = two_pi * (long double)i / (long double)L;
      all[L - 2 + i] = mk<T>(T(cosl(ang)), T(sinl(ang)));
    }
  }
  HIPCHECK(hipMalloc(&p->tw_all, all.size() * sizeof(cplx<T>)));
  HIPCHECK(hipMemcpy(p->tw_all, all.data(), all.size() * sizeof(cplx<T>), hipMemcpyHostToDevice));
  p->twn_shift = p->logN / 2;
  if (p->logN - p->twn_shift > 12) p->twn_shift = p->logN - 12;
  const size_t nlo = size_t(1) << p->twn_shift;
  std::vector<cplx<T>> lo(nlo);
  for (size_t i = 0; i < nlo; ++i) {
    const long double ang = two_pi * (long double)i / (long double)p->N;
    lo[i] = mk<T>(T(cosl(ang)), T(sinl(ang)));
  }
  HIPCHECK(hipMalloc(&p->twn_lo, nlo * sizeof(cplx<T>)));
  HIPCHECK(hipMemcpy(p->twn_lo, lo.data(), nlo * sizeof(cplx<T>), hipMemcpyHostToDevice));
  return CWT_OK;
}

// ---- filter support (band) of one row ------------------------------------------------------
// Bins whose profile is below eps * (peak of the profile) are treated as exactly zero; eps is far
// below the arithmetic's own rounding (1e-18 for fp64, 1e-9 for fp32).
double solve_decreasing(double lo, double target, double (*h)(double, double), double m) {
  // find f > lo with h(f, m) = target, h decreasing beyond lo
  double hi = lo + 1.0;
  while (h(hi, m) > target) hi = lo + 2.0 * (hi - lo);
  for (int it = 0; it < 200; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (h(mid, m) > target) lo = mid; else hi = mid;
  }
  return hi;
}
double h_paul(double f, double m) { return (m > 0 ? m * std::log(f) : 0.0) - f; }
double h_dog(double f, double m) { return (m > 0 ? m * std::log(f) : 0.0) - 0.5 * f * f; }

void profile_support(int mother, double p, double eps, double* f_lo, double* f_hi) {
  const double le = std::log(eps);
  if (mother == MOTHER_MORLET) {
    const double xc = std::sqrt(-2.0 * le);
    *f_lo = p - xc;
    *f_hi = p + xc;
  } else if (mother == MOTHER_PAUL) {
    const double peak = p > 0 ? h_paul(p, p) : 0.0;
    *f_lo = 0.0;
    *f_hi = solve_decreasing(p > 0 ? p : 0.0, peak + le, h_paul, p);
  } else {
    const double fp = std::sqrt(p > 0 ? p : 0.0);
    const double peak = p > 0 ? h_dog(fp, p) : 0.0;
    *f_hi = solve_decreasing(fp, peak + le, h_dog, p);
    *f_lo = -*f_hi;
  }
}

// log(profile(f) / peak of the profile) for the built-in mothers (-inf where the profile is 0)
double profile_log_rel(int mother, double p, double f) {
  const double ninf = -std::numeric_limits<double>::infinity();
  if (mother == MOTHER_MORLET) return -0.5 * (f - p) * (f - p);
  if (mother == MOTHER_PAUL) return f > 0 ? h_paul(f, p) - (p > 0 ? h_paul(p, p) : 0.0) : ninf;
  const double fp = std::sqrt(p > 0 ? p : 0.0);
  if (p > 0 && f == 0) return ninf;
  return h_dog(std::fabs(f), p) - (p > 0 ? h_dog(fp, p) : 0.0);
}
double profile_peak_f(int mother, double p) {
  return mother == MOTHER_MORLET ? p : mother == MOTHER_PAUL ? p : std::sqrt(p > 0 ? p : 0.0);
}

// Overlap-save rows: the wavelet of scale s is treated as zero beyond |t| > c_H * s, c_H chosen so that the neglected
// tail carries less than eps of the L1 mass of |psi| (the bound on the relative error of any output sample):
//   Morlet, DOG m: |psi(eta)| = |He_m(eta)| exp(-eta^2/2) (m = 0 for Morlet) -- numerical quadrature;
//   Paul m:        |psi(eta)| = (1 + eta^2)^(-(m+1)/2)   -- tail <= c^-m / m, total sqrt(pi) Gamma(m/2) / (2 Gamma((m+1)/2)).
double time_halo_factor(int mother, double param, double eps) {
  if (mother == MOTHER_PAUL) {
    const double m = param;
    const double total = 0.5 * std::sqrt(3.14159265358979323846) * std::tgamma(0.5 * m) / std::tgamma(0.5 * (m + 1.0));
    return std::pow(eps * m * total, -1.0 / m);
  }
  const int m = mother == MOTHER_DOG ? int(std::lround(param)) : 0;
  const double h = 1e-3;
  const int n = 60000;
  std::vector<double> g(n);
  double total = 0;
  for (int i = 0; i < n; ++i) {
    const double eta = (i + 0.5) * h;
    double h0 = 1.0, h1 = eta;                                  // probabilists' Hermite polynomials
    for (int k = 1; k < m; ++k) { const double h2 = eta * h1 - k * h0; h0 = h1; h1 = h2; }
    const double he = m == 0 ? 1.0 : h1;
    g[i] = std::fabs(he) * std::exp(-0.5 * eta * eta);
    total += g[i];
  }
  double tail = 0;
  for (int i = n - 1; i >= 0; --i) {
    tail += g[i];
    if (tail > eps * total) return (i + 1) * h;
  }
  return h;
}

int two_pass_logk(const cwt_plan* p);
int check_geometry(const cwt_plan* p);
int grow(void** buf, size_t* have, size_t need, hipStream_t s);


// ---- rows clipped at Nyquist: overlap-save on the band-passed complex signal (k_aols_*) ------------------------
// in-place radix-2 inverse DFT (e^{+2 pi i k n / n}, unnormalised) of a power-of-two length; host helper of aols_halo
void host_ifft(std::vector<std::complex<double>>& v) {
  const size_t n = v.size();
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(v[i], v[j]);
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const double ang = 6.283185307179586476925 / double(len);
    const std::complex<double> wl(std::cos(ang), std::sin(ang));
    for (size_t i = 0; i < n; i += len) {
      std::complex<double> w(1.0, 0.0);
      for (size_t k = 0; k < len / 2; ++k) {
        const std::complex<double> a = v[i + k], b = v[i + k + len / 2] * w;
        v[i + k] = a + b;
        v[i + k + len / 2] = a - b;
        w *= wl;
      }
    }
  }
}

double host_profile(int mother, double p, double f) {
  if (mother == MOTHER_MORLET) return std::exp(-0.5 * (f - p) * (f - p));
  if (mother == MOTHER_PAUL) return f > 0 ? std::exp(p * std::log(f) - f) : 0.0;
  return std::pow(f, p) * std::exp(-0.5 * f * f);
}

// the window of k_aols_gtab (aols_window in cwt_kernels.hpp), on the host
double host_aols_window(const AolsGeom& g, double f) {
  if (f > 0.5) {
    const double hw = 0.5 * (g.f_s + 0.5), c = 0.5 + hw;
    return 0.5 * std::erfc(g.z * (f - c) / hw);
  }
  if (f < g.f1_lo) {
    const double hw = 0.5 * (g.f1_lo - g.f_s), c = g.f_s + hw;
    return hw > 0 ? 0.5 * std::erfc(g.z * (c - f) / hw) : 0.0;
  }
  return 1.0;
}

// Smallest halo H (multiple of 64, <= hmax) beyond which the kernel e = IFFT(E), E(f) = G(aN f) u(f), carries less than
// eps of its L1 mass -- the bound on the relative error of an output sample, as for the overlap-save rows on the real
// signal -- or 0 if there is none or if the row does not qualify (see below).  e is evaluated numerically on a 4 hmax-point grid (its wrap-around beyond 2 hmax
// samples is far below eps for every row that passes).
int aols_halo(int mother, double param, double aN, const AolsGeom& g, double eps, int hmax) {
  const int n = 4 * hmax;
  std::vector<std::complex<double>> e(size_t(n), std::complex<double>(0.0, 0.0));
  const int k0 = int(std::ceil(g.f_s * n));
  double in_band = 0, beyond = 0;
  for (int q = 0; q < n; ++q) {
    const int kappa = k0 + (((q - k0) % n) + n) % n;
    const double f = double(kappa) / double(n);
    const double v = host_profile(mother, param, aN * f) * host_aols_window(g, f);
    e[size_t(q)] = v;
    (f <= 0.5 ? in_band : beyond) = std::max(f <= 0.5 ? in_band : beyond, std::fabs(v));
  }
  // A profile that keeps RISING past Nyquist (its peak lies beyond pi / dt: scales below the mother's own Nyquist limit)
  // would make the tapered continuation larger than the filter itself: exact arithmetic never sees it (x_M has nothing
  // there), rounding noise of the block spectra does, amplified by that ratio.  Such rows keep the N-point transform.
  if (!(beyond <= 2.0 * in_band)) return 0;
  host_ifft(e);
  std::vector<double> ring(size_t(n / 2) + 1, 0.0);                 // |e| by distance from t = 0
  double total = 0;
  for (int t = 0; t < n; ++t) {
    const double v = std::abs(e[size_t(t)]);
    ring[size_t(std::min(t, n - t))] += v;
    total += v;
  }
  if (!(total > 0)) return 0;
  double tail = 0;
  int best = 0;
  for (int d = n / 2; d > 0; --d) {                                 // tail = mass at distance >= d
    tail += ring[size_t(d)];
    if (tail > eps * total) break;
    if ((d - 1) % 64 == 0 && d - 1 >= 64 && d - 1 <= hmax) best = d - 1;   // halo H = d - 1 neglects distances > H
  }
  return best;
}

// z with erfc(z) / 2 = tail
double erfc_arg(double tail) {
  double lo = 0, hi = 10;
  for (int it = 0; it < 100; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (0.5 * std::erfc(mid) > tail) lo = mid; else hi = mid;
  }
  return hi;
}


// Degree of the polynomial form of a band-limited row (k_poly_*) on K' = 2^logk intervals: the smallest even D >= 2 with
//   F(kappa) / F_max * |theta_kappa|^(D+1) / (D+1)!  <=  eps   on the band,  theta = pi kappa / K'
// (F_max = the filter's largest value on the row's bins, log_best its log relative to the profile's peak) -- the truncated
// Taylor terms are bounded like the bins beyond the support threshold.  The band is sampled at <= 257 points.
int poly_degree_for(int mother, double param, double a, int kc, int k_lo, int nband, int logk, double log_best, double eps) {
  const int npts = std::min(nband, 257);
  const double tscale = 3.14159265358979323846 / double(1 << logk);
  std::vector<double> term(static_cast<size_t>(npts), 0.0), th(static_cast<size_t>(npts), 0.0);
  for (int i = 0; i < npts; ++i) {
    const int k = k_lo + (npts > 1 ? int((long(nband - 1) * i) / (npts - 1)) : 0);
    const double lg = profile_log_rel(mother, param, a * double(k)) - log_best;
    term[size_t(i)] = std::isfinite(lg) ? std::exp(std::min(lg, 0.0)) : 0.0;
    th[size_t(i)] = std::fabs(double(k - kc) + (k >= kc ? 1.0 : -1.0)) * tscale;   // + 1: the sampling skips neighbours
  }
  for (int d = 0; d <= POLY_MAX_DEGREE + 1; ++d) {
    double worst = 0;
    for (int i = 0; i < npts; ++i) {
      term[size_t(i)] *= th[size_t(i)] / double(d + 1);
      worst = std::max(worst, term[size_t(i)]);
    }
    if (worst <= eps && d >= 2 && (d & 1) == 0) return d;
  }
  return POLY_MAX_DEGREE + 2;
}

// Mother constant conj(c) with psi_ft(f) = c * profile(f)  (mothers.py:26-28, 118-122, 170-173)
int mother_constant(int mother, double param, double* cre, double* cim) {
  const double pi = 3.14159265358979323846;
  const int m = int(std::lround(param));
  *cre = 1.0; *cim = 0.0;
  if (mother == MOTHER_MORLET) {
    *cre = std::pow(pi, -0.25);
  } else if (mother == MOTHER_PAUL) {
    if (m < 1 || double(m) != param) return fail(CWT_EINVAL, "Paul order m must be an integer >= 1");
    *cre = std::pow(2.0, m) / std::sqrt(double(m) * std::tgamma(2.0 * m));  // (2m-1)! = Gamma(2m)
  } else if (mother == MOTHER_DOG) {
    if (m < 0 || double(m) != param) return fail(CWT_EINVAL, "DOG order m must be an integer >= 0");
    const double g = 1.0 / std::sqrt(std::tgamma(m + 0.5));
    // conj(-(i^m)): m%4 = 0 -> -1, 1 -> +i, 2 -> +1, 3 -> -i
    const double tr[4] = {-1, 0, 1, 0}, ti[4] = {0, 1, 0, -1};
    *cre = tr[m & 3] * g;
    *cim = ti[m & 3] * g;
  } else {
    return fail(CWT_EINVAL, "unknown mother id");
  }
  return CWT_OK;
}

// Entries of a plan's row tables: max_rows rows + the pseudo-rows some forms add.
size_t table_capacity(int max_rows) { return size_t(max_rows) + size_t(max_rows) / 3 + 4; }

// Row table for W[j,:] = IFFT_N( spec_j[k] * (amp_j * profile(a_j * signed_bin(k))) ), spec_j = spec + j*spec_ld.
// a_j = profile argument per bin, amp_j = complex amplitude WITHOUT the 1/N of the inverse FFT.
// ols_ncols > 0: the caller also has the real signal (cwt_transform): time-compact rows may take the overlap-save
// form, their output blocks covering ols_ncols columns.
int build_row_table(cwt_plan* p, int mother, double param, const double* a, const double* amp_re,
                    const double* amp_im, int64_t spec_ld, int nrows, const int* tab_klo = nullptr,
                    const int* tab_nband = nullptr, int rows_per_signal = 0, int64_t tab_ld = -1,
                    int64_t ols_ncols = 0, int64_t out_ncols = 0) {
  const int64_t N = p->N;
  double f_lo = 0, f_hi = 0;
  if (mother < MOTHER_MORLET || mother > MOTHER_TABLE) return fail(CWT_EINVAL, "unknown mother id");
  const Tolerances tol = tolerances(p);
  if (mother != MOTHER_TABLE) profile_support(mother, param, tol.support, &f_lo, &f_hi);

  const int logP = std::min(p->log_wg_points, p->logN);
  const bool use_small = p->logN <= p->loglmax;
  const int narrow_cap = std::min(p->narrow_max_logk, logP - 1);
  // pass A specialised for narrow column supports (default geometry only)
  const int two_pass_logr = p->logN - two_pass_logk(p);
  const bool band_pass_a = p->use_ct && p->band_pass_a && logP == (p->prec == 64 ? 13 : 14);
  // the multi-term form exists only in the compile-time kernel for K = 1024 at the default geometry
  const bool multi_ok = p->use_ct && p->narrow_terms > 1 && narrow_cap >= 10 &&
                        logP == (p->prec == 64 ? 13 : 14);
  // K = 2048 single-pass rows: fp64 only, N >= 2^14 (a 16384-point workgroup tile must fit the row)
  const bool big_ok = p->use_ct && p->narrow_big && p->prec == 64 && narrow_cap >= 10 && logP == 13 &&
                      p->logN >= 14;
  // overlap-save rows: default geometry, at least 4 workgroup tiles per row, built-in mothers, one shared spectrum
  // workgroup tile of the overlap-save rows: 8192 points (512 threads)
  const int ols_logp = 13;
  // half-size tiles for short halos (only beside the default 8192-point tile)
  const int ols_logp_s = (p->ols_small_max_halo > 0 && ols_logp == 13) ? 12 : 0;
  // a batch of signals (cwt_transform_batch: rows_per_signal > 0 with the signals at hand) has nbatch times the blocks
  // of one signal to fill the GPU with, so the form pays from shorter transforms: the threshold counts the batch
  const int ols_nbatch = (rows_per_signal > 0 && ols_ncols > 0) ? std::max(1, nrows / rows_per_signal) : 1;
  const bool ols_batch_ok = true;
  const bool ols_layout = rows_per_signal > 0 ? (ols_batch_ok && ols_ncols > 0 && nrows % rows_per_signal == 0) : spec_ld == 0;
  const bool ols_ok = p->ols && ols_ncols > 0 && p->use_ct && logP == (p->prec == 64 ? 13 : 14) &&
                      p->logN + ilog2(ols_nbatch) >= p->ols_min_logn && p->logN >= ols_logp + 2 &&
                      mother != MOTHER_TABLE && ols_layout && !use_small;
  const int ols_P = 1 << ols_logp;
  const int ols_hmax = p->ols_max_halo > 0 ? std::min(p->ols_max_halo, ols_P / 4) : ols_P / 4;
  const bool ols_big = ols_ok && p->ols_big && ols_logp == 13 && p->logN >= ols_logp + 3;   // blocks of 2P points
  const bool ols_big4 = ols_big && p->ols_big >= 2 && p->logN >= ols_logp + 4;               // ... and of 4P points
  const double ols_ch = ols_ok ? time_halo_factor(mother, param, tol.halo) : 0.0;
  // "not clipped at Nyquist": the profile at the Nyquist bins is below this fraction of its peak (the jump there is what
  // gives the sampled wavelet its slow 1/t tail; measured error of the form ~ a tenth of that fraction)
  // rows clipped at Nyquist as overlap-save rows on the band-passed complex signal (k_aols_*): needs the spectrum only;
  // Morlet and Paul (a real mother constant and nothing to keep on the masked-out bins), one shared spectrum
  // DOG (order >= 1): also, but only when the call hands over the REAL signal (its negative bins are the mirror image then)
  // a batch (rows_per_signal > 0: the same rows for every signal): one mask pseudo-row and one set of block spectra per
  // signal; like the overlap-save rows the form pays from shorter transforms there, the threshold counts the batch
  const int aols_nbatch = rows_per_signal > 0 ? std::max(1, nrows / rows_per_signal) : 1;
  const bool aols_layout = rows_per_signal > 0 ? (nrows % rows_per_signal == 0 && size_t(nrows) + size_t(aols_nbatch) <= table_capacity(p->max_rows))
                                               : spec_ld == 0;
  const bool aols_ok = p->ols && p->aols && out_ncols > 0 && p->use_ct && logP == (p->prec == 64 ? 13 : 14) &&
                       p->logN + ilog2(aols_nbatch) >= p->ols_min_logn && p->logN >= 15 && aols_layout && !use_small &&
                       (mother == MOTHER_MORLET || mother == MOTHER_PAUL ||
                        (mother == MOTHER_DOG && param >= 1 && ols_ncols > 0));
  double fc_lo = 0, fc_hi = 0;
  if (ols_ok || aols_ok) profile_support(mother, param, tol.clip, &fc_lo, &fc_hi);
  std::vector<RowDesc> narrow_rows, wide_rows, small_rows, poly_rows;
  std::vector<char> wide_clipped;
  const bool poly_ok = p->poly && p->use_ct && logP == (p->prec == 64 ? 13 : 14) && mother != MOTHER_TABLE &&
                       p->logN >= std::max(POLY_LOGP, p->poly_min_logn) && !use_small && rows_per_signal == 0;   // (not for batches yet)
  struct OlsRow { RowDesc rd; int grp, lb, h64; };
  std::vector<OlsRow> ols_rows;
  for (int j = 0; j < nrows; ++j) {
    if (!(a[j] > 0) || !std::isfinite(a[j])) return fail(CWT_EINVAL, "scales must be positive and finite");
    RowDesc rd;
    rd.a = a[j];
    rd.amp_re = amp_re[j] / double(N);
    rd.amp_im = amp_im[j] / double(N);
    // batched signals: row j belongs to signal j / rows_per_signal, whose spectrum starts at spec_ld * that
    rd.spec_off = rows_per_signal ? long(spec_ld) * (j / rows_per_signal) : long(spec_ld) * j;
    rd.tab_off = (tab_ld < 0 ? long(N) : long(tab_ld)) * j;       // tab_ld = 0: every row uses the same table
    rd.aux_off = 0;
    rd.nyq_re = rd.nyq_im = 0.0;
    double row_lo = f_lo, row_hi = f_hi, row_best = 0.0;     // row_best: log of the filter's largest value on the row's bins / its peak
    if (mother != MOTHER_TABLE) {
      // The support threshold is meant relative to the largest value the filter takes ON THE ROW'S BINS.  Where the bins
      // are coarser than the profile (a >~ 1: the largest scales) that is far below the profile's own peak: the
      // threshold follows it, or the row would lose the few bins that carry all of its (tiny) energy.
      const double kc = profile_peak_f(mother, param) / rd.a;
      double best = -std::numeric_limits<double>::infinity();
      for (double k : {std::floor(kc), std::ceil(kc), -std::floor(kc), -std::ceil(kc)}) {
        if (mother == MOTHER_PAUL) k = std::max(k, 1.0);
        if (mother == MOTHER_MORLET && k < 0) continue;
        k = std::min(std::max(k, -double(N / 2)), double(N / 2 - 1));
        best = std::max(best, profile_log_rel(mother, param, rd.a * k));
      }
      if (std::isfinite(best) && best < std::log(0.25)) {
        const double eps_row = std::max(tol.support * std::exp(best), 1e-300);
        profile_support(mother, param, eps_row, &row_lo, &row_hi);
        row_best = best;
      }
    }
    double klo = std::ceil(row_lo / rd.a), khi = std::floor(row_hi / rd.a);
    if (mother == MOTHER_PAUL) klo = std::max(klo, 1.0);
    const bool vanishes = std::ceil(fc_lo / rd.a) > -double(N / 2) &&                 // F_j vanishes at the Nyquist bins
                          std::floor(fc_hi / rd.a) < double(N / 2 - 1);
    const bool unclipped = ols_ok && vanishes;
    klo = std::max(klo, -double(N / 2));
    khi = std::min(khi, double(N / 2 - 1));
    if (mother == MOTHER_TABLE) { klo = tab_klo[j]; khi = klo + tab_nband[j] - 1; }
    if (khi < klo - 1) khi = klo - 1;
    if (klo < -double(N / 2) || khi > double(N / 2 - 1)) return fail(CWT_EINVAL, "filter support outside [-N/2, N/2)");
    rd.k_lo = int(klo);
    rd.nband = khi >= klo ? int(khi - klo + 1) : 0;
    if (rd.nband == 0) rd.k_lo = 0;
    rd.out_row = j;
    rd.logK = 0;
    rd.nterms = 1;
    if (use_small) {
      small_rows.push_back(rd);
    } else {
      const int need = std::max(4, ilog2(std::max(rd.nband, 1)));
      // Support wider than 1024 bins: several aliased terms per FFT input, at K = 1024 (8192-point tiles) or, fp64
      // only, K = 2048 (16384-point tiles, one workgroup per CU).  Measured us per row at N = 2^20 (tools/
      // terms_sweep.py; two-pass: 9.1 fp64, 5.0 fp32): fp64 K = 1024: 4.3 / 4.9 / 6.1 / 7.2 / 8.5 for 2 / 3 / 4 / 6 / 8
      // terms, K = 2048: 5.0 / 5.5 / 6.2 / 6.8 / 7.6 / 8.2 / 9.4 for 1 / 2 / 3 / 4 / 5 / 6 / 8; fp32 K = 1024: 2.9 /
      // 3.2 / 3.5 / 4.3 / 4.8 / 5.3 for 2 / 3 / 4 / 6 / 8 / 10.  Hence: K = 1024 up to 3 terms, K = 2048 beyond.
      const int t1 = (rd.nband + 1023) >> 10, t2 = (rd.nband + 2047) >> 11;
      const bool k1_ok = p->narrow && multi_ok && t1 <= p->narrow_terms;
      const bool k2_ok = p->narrow && big_ok && t2 <= p->big_terms;
      // overlap-save form (see k_ols_ct): halo H = c_H * (scale in samples), a multiple of 64 so that whole
      // wavefronts fall inside or outside the kept part of a block.  Block length P_b = P, or 2P (fp64) where that
      // keeps a larger fraction of every block transform and the stores stay >= 128-byte segments (K <= P/8).
      int halo = 0, lb = ols_logp, grp = 1;
      if (ols_ok && unclipped && rd.nband > 0) {
        const double s_samples = rd.a * double(N) / 6.283185307179586476925;
        const double hh = std::ceil(ols_ch * s_samples / 64.0) * 64.0;
        const double cap = ols_big4 ? std::max(double(std::min(p->ols_big4_max_halo, 4 * ols_hmax)), 2.0 * ols_hmax)
                                    : double(ols_hmax) * (ols_big ? 2.0 : 1.0);
        if (hh <= cap) halo = std::max(64, int(hh));
      }
      RowDesc od = rd;
      if (halo) {
        // the same filter sampled on the block's coarser frequency grid: bin k' of a P_b-point block is bin k' N / P_b.
        // K-point block FFTs, K >= the support; the band start is moved down to a multiple of K/16 (the bins added
        // lie below the support threshold) so that the aliased index wraps at the same slot in every thread
        auto describe = [&](int logb, int logp_tile, RowDesc& o) {
          const int Pb = 1 << logb;
          const double ab = rd.a * double(N >> logb);
          double kl = std::ceil(f_lo / ab), kh = std::floor(f_hi / ab);
          if (mother == MOTHER_PAUL) kl = std::max(kl, 1.0);
          kl = std::max(kl, -double(Pb / 2));
          kh = std::min(kh, double(Pb / 2 - 1));
          o.a = ab;
          o.amp_re = amp_re[j] / double(Pb);
          o.amp_im = amp_im[j] / double(Pb);
          o.k_lo = int(kl);
          o.nband = kh >= kl ? int(kh - kl + 1) : 0;
          if (o.nband == 0) o.k_lo = 0;
          for (o.logK = std::max(4, ilog2(std::max(o.nband, 1))); o.logK < logp_tile; ++o.logK) {
            const int nt = 1 << (o.logK - 4);
            const int lo = o.k_lo - (((o.k_lo % nt) + nt) % nt);
            if (o.nband + (o.k_lo - lo) <= (1 << o.logK) && lo >= -(Pb / 2)) {
              o.nband += o.k_lo - lo;
              o.k_lo = lo;
              break;
            }
          }
        };
        RowDesc big = rd;
        bool big_fits = false;
        if (ols_logp_s && halo <= p->ols_small_max_halo) {
          describe(ols_logp_s, ols_logp_s, od);
          lb = ols_logp_s; grp = 0;
        } else {
          int big_lb = 0;
          if (ols_big4 && halo >= p->ols_big4_min_halo) {       // blocks of 4P points: the stores stay >= 128-byte segments
            describe(ols_logp + 2, ols_logp, big);             // while K <= P/8, as for 2P
            if (big.logK <= ols_logp - 3) { big_fits = true; big_lb = ols_logp + 2; }
          }
          if (!big_fits && ols_big && halo >= p->ols_big_min_halo && halo <= 2 * ols_hmax) {
            big = rd;
            describe(ols_logp + 1, ols_logp, big);
            if (big.logK <= ols_logp - 3) { big_fits = true; big_lb = ols_logp + 1; }
          }
          if (big_fits) { od = big; lb = big_lb; }
          else if (halo <= ols_hmax) describe(ols_logp, ols_logp, od);
          else halo = 0;
        }
      }
      // polynomial form (k_poly_coef / k_poly_rows): K' >= the support intervals of R = N / K' >= 64 samples (128 in
      // complex64: a lane stores two outputs), degree D from the filter-weighted truncation rule
      int poly_logk = 0, poly_deg = 0;
      if (poly_ok && rd.nband > 0) {
        const int lk_max = std::min(p->poly_max_logk, p->logN - POLY_MIN_LOGR);
        const int kc = rd.k_lo + (rd.nband >> 1);
        for (int lk = std::max(8, ilog2(rd.nband)); lk <= lk_max; ++lk) {
          const int deg = poly_degree_for(mother, param, rd.a, kc, rd.k_lo, rd.nband, lk, row_best, tol.support);
          if (deg > POLY_MAX_DEGREE) continue;
          poly_logk = lk; poly_deg = deg;
          if (deg <= p->poly_degree) break;
        }
      }
      if (poly_logk) {
        rd.logK = poly_logk;
        rd.nterms = poly_deg;
        poly_rows.push_back(rd);
      } else if (p->narrow && need <= narrow_cap) {
        rd.logK = need;
        narrow_rows.push_back(rd);
      } else if (halo) {
        ols_rows.push_back({od, grp, lb, halo / 64});
      } else if (k1_ok && (!k2_ok || t1 <= 3)) {
        rd.logK = 10;                                   // k_narrow_ct_all (<= 4 terms) / k_narrow_ct_many
        rd.nterms = t1;
        narrow_rows.push_back(rd);
      } else if (k2_ok) {
        rd.logK = 11;                                   // k_narrow_ct_big
        rd.nterms = t2;
        narrow_rows.push_back(rd);
      } else {
        // pass A class: how many bins k1 of a column can be non-zero (see pass_a_band_body)
        const int span = (rd.nband >> two_pass_logk(p)) + 2;
        const int cls = span <= 16 ? 4 : span <= 64 ? 6 : span <= 256 ? 8 : 0;
        rd.logK = (band_pass_a && cls && cls < two_pass_logr) ? cls : 0;   // only if shorter than the column
        wide_rows.push_back(rd);
        wide_clipped.push_back(aols_ok && !vanishes && rd.nband > 0 &&
                               (mother == MOTHER_DOG ? (amp_re[j] == 0.0) != (amp_im[j] == 0.0) : amp_im[j] == 0.0));
      }
    }
  }
  // Rows clipped at Nyquist (so far two-pass rows) that can run as overlap-save rows on the band-passed complex signal:
  // one mask and one window for all of them (from the smallest scale), the halo of each from its kernel, one halo class.
  std::vector<RowDesc> aols_rows;
  AolsGeom ag{};
  int aols_logp = 12, aols_ks = 1;
  if (aols_ok) {
    double a_min = 0;
    for (size_t i = 0; i < wide_rows.size(); ++i)
      if (wide_clipped[i] && (a_min == 0 || wide_rows[i].a < a_min)) a_min = wide_rows[i].a;
    bool geom_ok = a_min > 0;
    if (geom_ok) {
      ag.z = erfc_arg(std::max(tol.halo * 0.1, 1e-19));
      if (mother == MOTHER_MORLET) {
        // the filter of the smallest scale is above the support threshold from f1_lo on (negative: Morlet's Gaussian is
        // not gated at f = 0, mothers.py:26-28); below it a taper of 1/32 cycle per sample, then the mask ends
        double s_lo, s_hi;
        profile_support(mother, param, tol.support, &s_lo, &s_hi);
        ag.f1_lo = std::min(s_lo / (a_min * double(N)), 0.0);
        ag.f_s = ag.f1_lo - 1.0 / 32.0;
        if (0.5 + ag.f_s < 0.12) ag.f_s = ag.f1_lo - 1.0 / 128.0;
        geom_ok = 0.5 + ag.f_s >= 0.10;                   // room for the taper above Nyquist
        aols_ks = int(std::ceil(ag.f_s * double(N)));
      } else if (mother == MOTHER_DOG) {
        // two-sided profile, smooth through f = 0: the mask is the positive bins 1 .. N/2 - 1 (the negative ones are their
        // mirror image, added by the kernel's epilogue), the window continues the profile below 0 and tapers it there
        // (a quarter cycle each side: the profile is NOT small there, so the tapers must be as gentle as the one above Nyquist)
        ag.f1_lo = 0.0;
        ag.f_s = -0.25;
        aols_ks = 1;
      } else {                                             // Paul: Heaviside -- the mask starts at bin 1
        ag.f1_lo = ag.f_s = 1.0 / double(N);
        aols_ks = 1;
      }
    }
    std::vector<int> halos(wide_rows.size(), 0);
    int hmax_seen = 0, cnt = 0;
    const int rps = rows_per_signal > 0 ? rows_per_signal : nrows;
    if (geom_ok) {
      const double eps = std::max(tol.halo, p->prec == 64 ? 2e-14 : 5e-7);
      std::vector<int> halo_of_scale(size_t(rps), -1);     // (a numeric tail search each: once per scale, not per signal)
      for (size_t i = 0; i < wide_rows.size(); ++i) {
        if (!wide_clipped[i]) continue;
        int& h = halo_of_scale[size_t(wide_rows[i].out_row % rps)];
        if (h < 0) h = aols_halo(mother, param, wide_rows[i].a * double(N), ag, eps, 512);
        halos[i] = h;
        if (halos[i]) { ++cnt; hmax_seen = std::max(hmax_seen, halos[i]); }
      }
    }
    if (geom_ok && cnt % aols_nbatch == 0 && cnt / aols_nbatch >= std::max(1, p->aols_min_rows)) {
      aols_logp = 12;                                      // 4096-point tiles: four block transforms in flight per CU
      const int P = 1 << aols_logp, L = P - 2 * hmax_seen;
      ag.halo = hmax_seen;
      ag.nrows = cnt / aols_nbatch;                         // per signal
      ag.nblocks = int((out_ncols + L - 1) / L);
      ag.ksp = int(std::ceil(ag.f_s * double(P)));
      std::vector<RowDesc> keep;
      long toff = 0;
      std::vector<long> tab_of_scale(size_t(rps), -1);     // one filter table per scale, shared by the signals
      for (size_t i = 0; i < wide_rows.size(); ++i) {
        if (!halos[i]) { keep.push_back(wide_rows[i]); continue; }
        RowDesc o = wide_rows[i];
        o.a = wide_rows[i].a * double(N >> aols_logp);     // profile argument per block bin
        o.amp_re = amp_re[o.out_row] / double(P);          // 1/P of the block's inverse transform (x_M carries its own 1/N)
        o.amp_im = 0.0;
        o.k_lo = ag.ksp; o.nband = P;
        o.logK = aols_logp; o.nterms = 1;
        o.nyq_re = o.nyq_im = 0.0;
        if (mother == MOTHER_DOG) {
          const int mm = int(std::lround(param));
          const bool odd = (mm & 1) != 0;
          o.nterms = odd ? 3 : 2;                           // W = 2 Re y | -2 Im y (table scale = the non-zero part of amp)
          if (odd) o.amp_re = amp_im[o.out_row] / double(P);
          const double pn = host_profile(mother, param, wide_rows[i].a * double(N / 2)) * (odd ? -1.0 : 1.0) / double(N);
          o.nyq_re = amp_re[o.out_row] * pn;                // F_j at the Nyquist bin (w = -pi / dt, wavelet.py:94) / N
          o.nyq_im = amp_im[o.out_row] * pn;
        }
        long& t = tab_of_scale[size_t(o.out_row % rps)];
        if (t < 0) { t = toff; toff += P; }
        o.tab_off = t;                                      // (spec_off stays the offset of the row's signal in the spectra)
        aols_rows.push_back(o);
      }
      wide_rows.swap(keep);
    }
  }
  // launch classes, in table order: 0 = k_narrow_ct_all (K <= 1024, <= 4 terms), 1 = k_narrow_ct_many (K = 1024,
  // 5..16 terms), 2 = k_narrow_ct_big (K = 2048)
  auto group_key = [](const RowDesc& x) {
    const int cls = x.logK == 11 ? 2 : (x.nterms > 4 ? 1 : 0);
    return cls * 100000 + x.logK + 100 * x.nterms;
  };
  std::stable_sort(narrow_rows.begin(), narrow_rows.end(),
                   [&](const RowDesc& x, const RowDesc& y) { return group_key(x) < group_key(y); });
  if (p->narrow_mix && p->use_ct && logP == (p->prec == 64 ? 13 : 14)) {   // (the generic kernels launch per (K, terms) group)
    // Launch order inside k_narrow_ct_all: the rows are sorted light (K = 16: store bound) to heavy (K = 1024 with three
    // terms: the longest compute phase); consecutive rows share the CUs, so alternate the two ends of the list -- a CU's two
    // tile slots then hold one store-heavy and one compute-heavy tile instead of two of a kind.  (Complex64: the rows
    // that run on half-size tiles, K <= 512 with one term, stay a block of their own at the front.)
    auto zigzag = [&](size_t lo, size_t hi) {
      std::vector<RowDesc> tmp(narrow_rows.begin() + lo, narrow_rows.begin() + hi);
      size_t a = 0, b = tmp.size();
      for (size_t i = lo; i < hi; ++i) narrow_rows[i] = ((i - lo) & 1) ? tmp[--b] : tmp[a++];
    };
    size_t n0 = 0;
    while (n0 < narrow_rows.size() && group_key(narrow_rows[n0]) < 100000) ++n0;        // class 0: k_narrow_ct_all
    size_t nh = 0;
    if (p->prec == 32 && p->narrow_small)
      while (nh < n0 && narrow_rows[nh].logK <= 9 && narrow_rows[nh].nterms == 1) ++nh;
    if (nh > 1) zigzag(0, nh);
    if (n0 - nh > 1) zigzag(nh, n0);
  }
  p->rt->table.clear();
  p->rt->narrow_groups.clear();
  p->rt->table.insert(p->rt->table.end(), small_rows.begin(), small_rows.end());
  for (size_t i = 0; i < narrow_rows.size(); ++i) {
    const int nt = narrow_rows[i].nterms;
    if (p->rt->narrow_groups.empty() || p->rt->narrow_groups.back().logK != narrow_rows[i].logK ||
        p->rt->narrow_groups.back().nterms != nt)
      p->rt->narrow_groups.push_back({narrow_rows[i].logK, int(p->rt->table.size()), 0, nt});
    p->rt->narrow_groups.back().count++;
    p->rt->table.push_back(narrow_rows[i]);
  }
  p->rt->wide_first = int(p->rt->table.size());
  p->rt->table.insert(p->rt->table.end(), wide_rows.begin(), wide_rows.end());
  p->rt->n_small = int(small_rows.size());
  p->rt->n_narrow = int(narrow_rows.size());
  p->rt->n_wide = int(wide_rows.size());
  // Overlap-save rows, grouped into at most OLS_MAX_CLASSES halo classes.  A class of rows i..j (sorted by halo) runs
  // at the largest halo H_j: every block transform yields P - 2 H_j columns, and the class pays one block spectrum per
  // block on top of its rows -> cost (rows + w) * P / (P - 2 H_j); dynamic programme over the distinct halos.
  p->rt->ols_first = int(p->rt->table.size());
  p->rt->n_ols = int(ols_rows.size());
  p->rt->ols_xs_elems = p->rt->ols_gt_elems = 0;
  for (int g = 0; g < 2; ++g) {
    auto& G = p->rt->ols_grp[g];
    G.logp = g == 0 ? (ols_logp_s ? ols_logp_s : ols_logp) : ols_logp;
    G.cls.n = 0; G.wgs = 0; G.fwd_blocks[0] = G.fwd_blocks[1] = G.fwd_blocks[2] = 0; G.row_first = G.nrows = 0;
    for (int i = 0; i < OLS_MAX_CLASSES; ++i) G.cls.wg_first[i] = 0x7fffffff;
  }
  if (!ols_rows.empty()) {
    // by tile group, then block length, then halo
    // (batch: then scale by scale, the signals of a scale in order -- k_ols_ct indexes a class's rows that way)
    const int rps = rows_per_signal > 0 ? rows_per_signal : 1 << 30;
    std::stable_sort(ols_rows.begin(), ols_rows.end(), [rps](const OlsRow& x, const OlsRow& y) {
      if (x.grp != y.grp) return x.grp < y.grp;
      if (x.lb != y.lb) return x.lb < y.lb;
      if (x.h64 != y.h64) return x.h64 < y.h64;
      return x.rd.out_row % rps < y.rd.out_row % rps;       // stable: equal scales stay in signal order
    });
    long xs = 0;
    int row0 = 0;                                                  // index into ols_rows
    for (int g = 0; g < 2; ++g) {
      auto& grp = p->rt->ols_grp[g];
      OlsClasses& oc = grp.cls;
      grp.row_first = row0;
      long wg = 0;
      for (int lb = grp.logp; lb <= grp.logp + 2; ++lb) {
        int nr = 0;
        while (row0 + nr < int(ols_rows.size()) && ols_rows[row0 + nr].grp == g && ols_rows[row0 + nr].lb == lb) ++nr;
        if (!nr) continue;
        const int Pb = 1 << lb, G = 1 << (lb - grp.logp);
        std::vector<int> hv, cnt;                                   // distinct halos (units of 64) and their row counts
        for (int i = row0; i < row0 + nr; ++i) {
          if (hv.empty() || hv.back() != ols_rows[i].h64) { hv.push_back(ols_rows[i].h64); cnt.push_back(0); }
          cnt.back()++;
        }
        const int nd = int(hv.size()), KC = (lb == grp.logp || !ols_big4) ? OLS_MAX_CLASSES / 2 : OLS_MAX_CLASSES / 4;
        std::vector<int> pre(nd + 1, 0);
        for (int i = 0; i < nd; ++i) pre[i + 1] = pre[i] + cnt[i];
        auto cost = [&](int i, int j) {                             // distinct halos i..j-1 as one class
          return (double(pre[j] - pre[i]) + p->ols_fwd_weight * ols_nbatch) * double(Pb) / double(Pb - 128 * hv[j - 1]);
        };
        const double inf = 1e300;
        std::vector<std::vector<double>> dp(KC + 1, std::vector<double>(nd + 1, inf));
        std::vector<std::vector<int>> from(KC + 1, std::vector<int>(nd + 1, -1));
        dp[0][0] = 0;
        for (int k = 1; k <= KC; ++k)
          for (int j = 1; j <= nd; ++j)
            for (int i = 0; i < j; ++i)
              if (dp[k - 1][i] < inf && dp[k - 1][i] + cost(i, j) < dp[k][j]) { dp[k][j] = dp[k - 1][i] + cost(i, j); from[k][j] = i; }
        int bestk = 1;
        for (int k = 2; k <= KC; ++k) if (dp[k][nd] < dp[bestk][nd]) bestk = k;
        std::vector<int> cuts;                                      // class boundaries in distinct-halo indices
        for (int k = bestk, j = nd; k >= 1; --k) { cuts.push_back(j); j = from[k][j]; }
        std::reverse(cuts.begin(), cuts.end());
        int lo_d = 0;
        long blk = 0;
        const long stride = (Pb / 2) + 8;
        for (size_t ci = 0; ci < cuts.size(); ++ci) {
          const int hi_d = cuts[ci], H = 64 * hv[hi_d - 1], L = Pb - 2 * H;
          OlsClass& k = oc.c[oc.n++];
          k.halo = H;
          k.logb = lb;
          k.nsig = ols_nbatch;
          k.nblocks = int((ols_ncols + L - 1) / L);
          k.nrows = pre[hi_d] - pre[lo_d];
          k.row_first = row0 - grp.row_first + pre[lo_d];
          k.wg_first = int(wg);
          k.blk_first = int(blk);
          k.xs_off = xs;
          // the 8 XCDs share the (signal, block) pairs: nblocks alone can be as few as 17 (N = 2^16), which would leave
          // one XCD with 3 blocks and seven with 2 + an idle pass (measured: +40 % on that kernel)
          wg += ((long(k.nblocks) * ols_nbatch + 7) / 8) * 8 * (k.nrows / ols_nbatch) * G;
          blk += k.nblocks;
          xs += long(k.nblocks) * stride;
          lo_d = hi_d;
        }
        grp.fwd_blocks[lb - grp.logp] = blk;
        row0 += nr;
      }
      grp.nrows = row0 - grp.row_first;
      grp.wgs = wg;
      for (int i = 0; i < OLS_MAX_CLASSES; ++i) oc.wg_first[i] = i < oc.n ? oc.c[i].wg_first : 0x7fffffff;
    }
    long gt_off = 0;                                            // filter tables: 2^logK entries per row (k_ols_gtab)
    // batch: the table depends on the scale only (one per scale, shared by the signals); the block spectra are per
    // signal, xs elements apart -- an overlap-save row reads its spectra at xs_dev + spec_off + class offset
    std::vector<long> tab_of_scale(rows_per_signal > 0 ? rows_per_signal : 0, -1);
    for (auto& r : ols_rows) {
      r.rd.nterms = 1 << (r.lb - p->rt->ols_grp[r.grp].logp);   // nterms = workgroups per block
      if (rows_per_signal > 0) {
        long& t = tab_of_scale[r.rd.out_row % rows_per_signal];
        if (t < 0) { t = gt_off; gt_off += 1L << r.rd.logK; }
        r.rd.tab_off = t;
        r.rd.spec_off = long(r.rd.out_row / rows_per_signal) * xs;
      } else {
        r.rd.tab_off = gt_off;
        gt_off += 1L << r.rd.logK;
        r.rd.spec_off = 0;
      }
      p->rt->table.push_back(r.rd);
    }
    p->rt->ols_gt_elems = gt_off;
    p->rt->ols_xs_sig = xs;
    p->rt->ols_xs_elems = xs * ols_nbatch;
  }
  p->rt->ols_nbatch = ols_nbatch;
  p->rt->aols_first = int(p->rt->table.size());
  p->rt->n_aols = int(aols_rows.size());
  p->rt->aux_first = -1;
  p->rt->aols_gt_elems = 0;
  if (!aols_rows.empty()) {
    p->rt->table.insert(p->rt->table.end(), aols_rows.begin(), aols_rows.end());
    p->rt->aols_logp = aols_logp;
    p->rt->aols_geom = ag;
    p->rt->aols_wgs = long((ag.nblocks + 7) / 8) * 8 * ag.nrows;
    p->rt->aols_gt_elems = long(ag.nrows) << aols_logp;
    p->rt->aols_nbatch = aols_nbatch;
    RowDesc m{};                                          // (zero-initialised: no Nyquist term) the mask as a row: profile 1 (DOG m = 0 at a = 0) on [k_s, N/2)
    m.a = 0.0; m.amp_re = 1.0 / double(N); m.amp_im = 0.0;
    m.k_lo = aols_ks; m.nband = int(N / 2) - aols_ks;
    m.logK = 0; m.nterms = 1; m.tab_off = 0;
    p->rt->aux_first = int(p->rt->table.size());
    for (int b = 0; b < aols_nbatch; ++b) {               // one per signal
      m.out_row = b;
      m.spec_off = rows_per_signal > 0 ? long(spec_ld) * b : 0;
      p->rt->table.push_back(m);
    }
  }
  // polynomial rows: by K', then by degree; coefficient offsets; the workgroups of k_poly_coef per class
  p->rt->poly_first = int(p->rt->table.size());
  p->rt->n_poly = int(poly_rows.size());
  p->rt->poly_chunks.clear();
  p->rt->poly_coef_elems = 0;
  if (!poly_rows.empty()) {
    // largest K' first (their planes are the bulk and their k_poly_coef tiles the slowest to get going), then by degree
    std::stable_sort(poly_rows.begin(), poly_rows.end(), [](const RowDesc& x, const RowDesc& y) {
      return x.logK != y.logK ? x.logK > y.logK : x.nterms < y.nterms;
    });
    const size_t esz = p->esize() * 2;
    size_t cap = ~size_t(0);
    if (p->poly_chunk_mb > 0) {                             // as few chunks as the limit allows, of about equal volume
      size_t total = 0;
      for (const RowDesc& r : poly_rows) total += (size_t(r.nterms) + 1) * (size_t(1) << r.logK) * esz;
      const size_t limit = size_t(p->poly_chunk_mb) << 20, n = (total + limit - 1) / limit;
      cap = n > 1 ? (total + n - 1) / n : ~size_t(0);
    }
    long off = 0, boff = 0;
    size_t vol = 0;
    for (size_t i = 0; i < poly_rows.size(); ++i) {
      RowDesc& r = poly_rows[i];
      r.tab_off = off;                                      // planes: (D + 1) K' complex
      off += (long(r.nterms) + 1) << r.logK;
      r.aux_off = boff;                                     // band: K' complex
      boff += 1L << r.logK;
      const size_t bytes = (size_t(r.nterms) + 1) * (size_t(1) << r.logK) * esz;
      if (p->rt->poly_chunks.empty() || vol + bytes / 2 > cap) {
        p->rt->poly_chunks.emplace_back();
        p->rt->poly_chunks.back().row_first = int(i);
        vol = 0;
      }
      vol += bytes;
      auto& ch = p->rt->poly_chunks.back();
      ch.nrows++;
      ch.max_logk = std::max(ch.max_logk, r.logK);
      PolyClasses& pc = ch.cls;
      if (pc.n == 0 || pc.c[pc.n - 1].logK != r.logK) {
        if (pc.n == POLY_MAX_CLASSES) return fail(CWT_EINVAL, "too many polynomial-row classes");
        pc.c[pc.n++] = PolyClass{r.logK, int(i) - ch.row_first, 0, 0, 0};
      }
      PolyClass& c = pc.c[pc.n - 1];
      c.nrows++;
      c.ndeg = std::max(c.ndeg, r.nterms + 1);
    }
    for (auto& ch : p->rt->poly_chunks)
      for (int i = 0; i < ch.cls.n; ++i) {                  // per tile size (launch): classes in table order
        PolyClass& c = ch.cls.c[i];
        const int tile = std::max(12, c.logK);                // log2 of the workgroup tile
        const long tb = 1L << (tile - c.logK);
        long& wg = ch.wgs[tile - 12];
        c.wg_first = int(wg);
        wg += (long(c.nrows) * c.ndeg + tb - 1) / tb;
      }
    p->rt->poly_coef_elems = off;
    p->rt->poly_band_elems = boff;
    p->rt->table.insert(p->rt->table.end(), poly_rows.begin(), poly_rows.end());
  }
  return CWT_OK;
}

// log2 of the row length K of the two-pass factorisation N = R*K
void set_split(cwt_plan* p) {
  int n_big = 0, n_many = 0;
  for (const auto& g : p->rt->narrow_groups) {
    if (g.logK == 11) n_big += g.count;
    else if (g.nterms > 4) n_many += g.count;
  }
  p->split[0] = p->rt->n_small; p->split[1] = p->rt->n_narrow - n_big - n_many; p->split[2] = p->rt->n_wide;
  p->split[3] = n_big; p->split[4] = n_many; p->split[5] = p->rt->n_ols; p->split[6] = p->rt->n_aols;
  p->split[7] = p->rt->n_poly;
}

int chunk_rows_of(const cwt_plan* p) {
  if (p->chunk_rows > 0) return p->chunk_rows;
  const size_t row_bytes = size_t(p->N) * 2 * p->esize();
  return int(std::max<size_t>(1, (size_t(192) << 20) / row_bytes));
}

// Rows per two-pass launch for `nrows` rows: as few launches as the chunk limit allows, of equal size (102 rows at a
// limit of 12 -> 9 launches of 11-12 rows instead of 8 x 12 + 6; 13 rows -> 7 + 6 instead of 12 + 1).
int balanced_chunk(const cwt_plan* p, int nrows) {
  const int limit = std::max(1, std::min(chunk_rows_of(p), nrows));
  const int nchunks = (nrows + limit - 1) / limit;
  return (nrows + nchunks - 1) / nchunks;
}

// N = R*K.  K = 1024 up to N = 2^21, K = 2048 at 2^22 and 2^23 (measured: 155 vs 117 GS/s at 2^22 against
// K = 1024, 106 vs 60 at 2^23 against K = 4096: 32-byte store tiles in pass B hurt more than in pass A),
// K = 4096 at 2^24 (forced by the 4096-point workgroup FFT limit).
int two_pass_logk(const cwt_plan* p) {
  int lk = std::min(10, p->logN - 4);
  if (p->logN >= 22) lk = 11;
  if (p->force_logk) lk = p->force_logk;
  lk = std::max(lk, p->logN - p->loglmax);
  lk = std::min(lk, p->loglmax);
  return lk;
}

// The tuning options can describe geometries the kernels do not support (they exist for tests); refuse them.
int check_geometry(const cwt_plan* p) {
  if (p->logN <= p->loglmax) return CWT_OK;                     // single-workgroup transform: nothing to check
  if (p->logN > 2 * p->loglmax) return fail(CWT_EINVAL, "nfft exceeds lmax^2 (two-pass limit)");
  const int logK = two_pass_logk(p), logR = p->logN - logK, logP = std::min(p->log_wg_points, p->logN);
  if (logK < 4 || logR < 4) return fail(CWT_EINVAL, "two-pass transform needs both factors >= 16: raise lmax");
  if (logP < logK || logP < logR)
    return fail(CWT_EINVAL, "wg_points must be at least as large as both two-pass factors");
  return CWT_OK;
}

int ensure_z(cwt_plan* p, int rows) {
  const size_t need = size_t(rows) * size_t(p->N) * 2 * p->esize();
  if (p->z_bytes >= need) return CWT_OK;
  ++g_scratch_gen;
  if (p->Z) { HIPCHECK(hipStreamSynchronize(p->stream)); HIPCHECK(hipFree(p->Z)); p->Z = nullptr; p->z_bytes = 0; }
  if (hipMalloc(&p->Z, need) != hipSuccess) return fail(CWT_ENOMEM, "cannot allocate two-pass workspace");
  p->z_bytes = need;
  return CWT_OK;
}


// ---- compile-time specialised kernels for the default geometry --------------------------------
template <typename T> constexpr int default_logp() { return sizeof(T) == 8 ? 13 : 14; }

// all band-limited rows in one launch (k_narrow_ct_all); false if the geometry is not the default one
template <typename T>
bool narrow_ct_all_applies(const cwt_plan* p) {
  if (!p->use_ct || std::min(p->log_wg_points, p->logN) != default_logp<T>()) return false;
  for (const auto& g : p->rt->narrow_groups) {
    if (g.logK == 11 && sizeof(T) == 8 && g.nterms >= 1 && g.nterms <= 8) continue;      // k_narrow_ct_big
    if (g.logK < 4 || g.logK > 10 || g.nterms < 1 || g.nterms > 16 || (g.nterms > 1 && g.logK != 10)) return false;
  }
  return true;
}

constexpr int kMaxGridY = 32768;   // rows per launch (gridDim.y is limited to 65535)

// rows of the two compile-time band-limited kernels: the row table is sorted by class, groups with
// K <= 1024 first, then (fp64 only) the K = 2048 groups
void narrow_class_counts(const cwt_plan* p, int* n_small_k, int* n_big, int* n_many = nullptr) {
  int many = 0;
  *n_small_k = *n_big = 0;
  for (const auto& g : p->rt->narrow_groups) (g.logK == 11 ? *n_big : g.nterms > 4 ? many : *n_small_k) += g.count;
  if (n_many) *n_many = many;
}

template <typename T>
void launch_narrow_ct_many(cwt_plan* p, const cplx<T>* xhat, const Mother& mo, cplx<T>* W, int64_t ldw,
                           int64_t ncols) {
  constexpr int LOGP = default_logp<T>();
  const int first = p->rt->narrow_groups.front().first;
  int n_small_k, n_big, n_many;
  narrow_class_counts(p, &n_small_k, &n_big, &n_many);
  for (int r0 = 0; r0 < n_many; r0 += kMaxGridY)
    hipLaunchKernelGGL((k_narrow_ct_many<T, LOGP>), dim3(1u << (p->logN - LOGP), std::min(kMaxGridY, n_many - r0)),
                       dim3(1 << (LOGP - 4)), (size_t(1) << LOGP) * sizeof(T), p->stream, xhat,
                       p->rt->rows_dev + first + n_small_k + r0, mo, static_cast<const cplx<T>*>(p->tw_all),
                       twn_of<T>(p), p->logN, W, long(ldw), long(ncols));
}

template <typename T>
void launch_narrow_ct_all(cwt_plan* p, const cplx<T>* xhat, const Mother& mo, cplx<T>* W, int64_t ldw,
                          int64_t ncols) {
  constexpr int LOGP = default_logp<T>();
  const int first = p->rt->narrow_groups.front().first;
  int n_small_k, n_big;
  narrow_class_counts(p, &n_small_k, &n_big);
  int n_wave = 0;
  // complex64 only: rows with K <= 512 (sorted first) on half-size workgroup tiles (store segments stay
  // >= 128 B): -6 % on this kernel; complex128 measured +7 %
  int n_half = 0;
  constexpr bool kHalfTiles64 = false;
  if constexpr (sizeof(T) == 4 || kHalfTiles64) {
    if (p->narrow_small && (sizeof(T) == 4 || p->narrow_small == 2) && p->logN >= LOGP)
      for (const auto& g : p->rt->narrow_groups) if (g.logK <= 9 && g.nterms == 1) n_half += g.count;
    n_half = std::max(n_half, n_wave);
    for (int r0 = n_wave; r0 < n_half; r0 += kMaxGridY)
      hipLaunchKernelGGL((k_narrow_ct_all<T, LOGP - 1>), dim3(1u << (p->logN - LOGP + 1), std::min(kMaxGridY, n_half - r0)),
                         dim3(1 << (LOGP - 5)), (size_t(1) << (LOGP - 1)) * sizeof(T), p->stream, xhat,
                         p->rt->rows_dev + first + r0, mo, static_cast<const cplx<T>*>(p->tw_all), twn_of<T>(p),
                         p->logN, W, long(ldw), long(ncols));
  }
  for (int r0 = std::max(n_half, n_wave); r0 < n_small_k; r0 += kMaxGridY)
    hipLaunchKernelGGL((k_narrow_ct_all<T, LOGP>), dim3(1u << (p->logN - LOGP), std::min(kMaxGridY, n_small_k - r0)),
                       dim3(1 << (LOGP - 4)), (size_t(1) << LOGP) * sizeof(T), p->stream, xhat,
                       p->rt->rows_dev + first + r0, mo, static_cast<const cplx<T>*>(p->tw_all), twn_of<T>(p),
                       p->logN, W, long(ldw), long(ncols));
}

template <typename T>
void launch_narrow_ct_big(cwt_plan* p, const cplx<T>* xhat, const Mother& mo, cplx<T>* W, int64_t ldw,
                          int64_t ncols) {
  if constexpr (sizeof(T) == 8) {
    const int first = p->rt->narrow_groups.front().first;
    int n_small_k, n_big, n_many;
    narrow_class_counts(p, &n_small_k, &n_big, &n_many);
    for (int r0 = 0; r0 < n_big; r0 += kMaxGridY)
      hipLaunchKernelGGL((k_narrow_ct_big<T>), dim3(1u << (p->logN - 14), std::min(kMaxGridY, n_big - r0)), dim3(1024),
                         (size_t(1) << 14) * sizeof(T), p->stream, xhat, p->rt->rows_dev + first + n_small_k + n_many + r0, mo,
                         static_cast<const cplx<T>*>(p->tw_all), twn_of<T>(p), p->logN, W, long(ldw), long(ncols));
  }
}

// Large transforms (N >= 2^23, complex128): a 4096-point column FFT leaves only 2 columns per 8192-point tile, i.e.
// 32-byte memory segments in pass A.  16384-point tiles (1024 threads, 128 KiB of LDS, one workgroup per CU)
// double them: pass A -33 %, forward FFT's pass A -63 % at N = 2^23.  (Pass B measured slower on such tiles.)
template <typename F>
void allow_big_lds(F kernel) {
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          144 * 1024) != hipSuccess)
    (void)hipGetLastError();
}

template <typename T, int LOGR, int MODE>
void launch_pass_a_ct(cwt_plan* p, const void* in, const RowDesc* rows, int cnt, const Mother& mo, long n0,
                      long in_ld, cplx<T>* Z, hipStream_t st) {
  constexpr int LOGP = default_logp<T>();
  if constexpr (sizeof(T) == 8 && LOGR == 12) {
    if (p->big_tiles) {
      constexpr int LP = 14;
      static const bool once = (allow_big_lds(&k_pass_a_ct<T, LOGR, LP, MODE>), true);
      (void)once;
      hipLaunchKernelGGL((k_pass_a_ct<T, LOGR, LP, MODE>), dim3(1u << (p->logN - LP), cnt), dim3(1 << (LP - 4)),
                         (size_t(1) << LP) * sizeof(T), st, in, rows, mo, tw_table<T>(p, LOGR), twn_of<T>(p),
                         p->logN, n0, in_ld, Z);
      return;
    }
  }
  hipLaunchKernelGGL((k_pass_a_ct<T, LOGR, LOGP, MODE>), dim3(1u << (p->logN - LOGP), cnt),
                     dim3(1 << (LOGP - 4)), (size_t(1) << LOGP) * sizeof(T), st, in, rows, mo,
                     tw_table<T>(p, LOGR), twn_of<T>(p), p->logN, n0, in_ld, Z);
}

template <typename T, int LOGR, int LP>
void launch_pass_a_ct_rows_lp(cwt_plan* p, const void* in, const RowDesc* rows, int cnt, const Mother& mo,
                              cplx<T>* Z, hipStream_t st) {
  const dim3 grid(1u << (p->logN - LP), cnt), block(1 << (LP - 4));
  const size_t lds = (size_t(1) << LP) * sizeof(T);
  if constexpr (LP == 14) {
    static const bool once = (allow_big_lds(&k_pass_a_ct_rows<T, LOGR, LP>), true);
    (void)once;
  }
  hipLaunchKernelGGL((k_pass_a_ct_rows<T, LOGR, LP>), grid, block, lds, st, static_cast<const cplx<T>*>(in),
                       rows, mo, static_cast<const cplx<T>*>(p->tw_all), twn_of<T>(p), p->logN, Z);
}

template <typename T, int LOGR>
void launch_pass_a_ct_rows(cwt_plan* p, const void* in, const RowDesc* rows, int cnt, const Mother& mo,
                           cplx<T>* Z, hipStream_t st) {
  constexpr int LOGP = default_logp<T>();
  if constexpr (LOGR <= 10) {
    if (p->pass_a_small) return launch_pass_a_ct_rows_lp<T, LOGR, LOGP - 1>(p, in, rows, cnt, mo, Z, st);
  }
  if constexpr (sizeof(T) == 8 && LOGR == 12) {
    if (p->big_tiles) return launch_pass_a_ct_rows_lp<T, LOGR, 14>(p, in, rows, cnt, mo, Z, st);
  }
  launch_pass_a_ct_rows_lp<T, LOGR, LOGP>(p, in, rows, cnt, mo, Z, st);
}

// Compile-time pass A for every column length R = 2^4 .. 2^12 (i.e. every N the two-pass path handles).
template <typename T, int MODE>
bool try_pass_a_ct(cwt_plan* p, int logR, const void* in, const RowDesc* rows, int cnt, const Mother& mo,
                   long n0, long in_ld, cplx<T>* Z, hipStream_t st) {
  if (!p->use_ct || std::min(p->log_wg_points, p->logN) != default_logp<T>()) return false;
#define CWT_CASE(LR)                                                                                  \
  case LR:                                                                                            \
    if constexpr (MODE == IN_SPECTRUM) launch_pass_a_ct_rows<T, LR>(p, in, rows, cnt, mo, Z, st);    \
    else launch_pass_a_ct<T, LR, MODE>(p, in, rows, cnt, mo, n0, in_ld, Z, st);                       \
    return true;
  switch (logR) {
    CWT_CASE(4) CWT_CASE(5) CWT_CASE(6) CWT_CASE(7) CWT_CASE(8) CWT_CASE(9) CWT_CASE(10) CWT_CASE(11) CWT_CASE(12)
    default: return false;
  }
#undef CWT_CASE
}

template <typename T, int LOGK, int LP, bool CONJ>
void launch_pass_b_ct_lp(cwt_plan* p, const RowDesc* rows, int cnt, cplx<T>* W, int64_t ldw, int64_t ncols,
                         const cplx<T>* Z, hipStream_t st) {
  const size_t lds = ((size_t(1) << LP) + (size_t(1) << (LP - 4))) * sizeof(T);
  const dim3 grid(1u << (p->logN - LP), cnt), block(1 << (LP - 4));
  hipLaunchKernelGGL((k_pass_b_ct<T, LOGK, LP, CONJ>), grid, block, lds, st, Z, rows, tw_table<T>(p, LOGK),
                     twn_of<T>(p), p->logN, W, long(ldw), long(ncols));
}

template <typename T, int LOGK, bool CONJ>
void launch_pass_b_ct(cwt_plan* p, const RowDesc* rows, int cnt, cplx<T>* W, int64_t ldw, int64_t ncols,
                      const cplx<T>* Z, hipStream_t st) {
  constexpr int LOGP = default_logp<T>();
  launch_pass_b_ct_lp<T, LOGK, LOGP, CONJ>(p, rows, cnt, W, ldw, ncols, Z, st);
}

// Compile-time pass B for row lengths K = 2^9 .. 2^12 (K = 1024 for every N from 2^14 to 2^22).
template <typename T, bool CONJ>
bool try_pass_b_ct(cwt_plan* p, int logK, const RowDesc* rows, int cnt, cplx<T>* W, int64_t ldw,
                   int64_t ncols, const cplx<T>* Z, hipStream_t st) {
  if (!p->use_ct || std::min(p->log_wg_points, p->logN) != default_logp<T>()) return false;
  switch (logK) {
    case 9: launch_pass_b_ct<T, 9, CONJ>(p, rows, cnt, W, ldw, ncols, Z, st); return true;
    case 10: launch_pass_b_ct<T, 10, CONJ>(p, rows, cnt, W, ldw, ncols, Z, st); return true;
    case 11: launch_pass_b_ct<T, 11, CONJ>(p, rows, cnt, W, ldw, ncols, Z, st); return true;
    case 12: launch_pass_b_ct<T, 12, CONJ>(p, rows, cnt, W, ldw, ncols, Z, st); return true;
    default: return false;
  }
}

// Forward FFT of nrows rows (real, or complex for MODE = IN_CPLX), each zero padded from n0 to N:
// out[r, k] = sum_n in[r, n] e^{-2 pi i k n / N}, computed as conj(inverse(conj(in))).
template <typename T, int MODE>
int fft_rows_impl(cwt_plan* p, const void* in_dev, int64_t in_ld, int nrows, int64_t n0, void* out_dev) {
  const int logN = p->logN;
  if (int rc = check_geometry(p)) return rc;
  const Mother mo{MOTHER_MORLET, 0, 0.0, nullptr};
  cplx<T>* out = static_cast<cplx<T>*>(out_dev);
  if (logN <= 3) {
    const int total = nrows << logN;
    return timed_launch(p, KC_FWD_SMALL, [&] {
      hipLaunchKernelGGL((k_direct<T, MODE>), dim3((total + 63) / 64), dim3(64), 0, p->stream, in_dev,
                         (const RowDesc*)nullptr, nrows, mo, logN, long(n0), long(in_ld), out, long(p->N),
                         long(p->N));
    });
  }
  if (logN <= p->loglmax) {
    const int logTB = nrows > 1 ? std::max(0, std::min(12, p->log_wg_points) - logN) : 0;
    const int TB = 1 << logTB;
    const int threads = TB << (logN - 4);
    const size_t lds = (size_t(TB) << logN) * sizeof(T);
    return timed_launch(p, KC_FWD_SMALL, [&] {
      hipLaunchKernelGGL((k_small<T, MODE>), dim3((nrows + TB - 1) / TB), dim3(threads), lds, p->stream,
                         in_dev, (const RowDesc*)nullptr, nrows, mo, tw_table<T>(p, logN), logN, logTB,
                         long(n0), long(in_ld), out, long(p->N), long(p->N));
    });
  }
  const int logK = two_pass_logk(p), logR = logN - logK;
  const int logP = std::min(p->log_wg_points, logN);
  const int chunk = balanced_chunk(p, nrows);
  int rc = ensure_z(p, chunk);
  if (rc) return rc;
  const size_t lds = (size_t(1) << logP) * sizeof(T);
  const int threads = 1 << (logP - 4);
  const size_t esz = (MODE == IN_REAL ? 1 : 2) * sizeof(T);
  for (int first = 0; first < nrows; first += chunk) {
    const int cnt = std::min(chunk, nrows - first);
    const void* in = static_cast<const char*>(in_dev) + size_t(first) * size_t(in_ld) * esz;
    cplx<T>* o = out + size_t(first) * size_t(p->N);
    rc = timed_launch(p, KC_FWD_A, [&] {
      if (try_pass_a_ct<T, MODE>(p, logR, in, nullptr, cnt, mo, long(n0), long(in_ld),
                                 static_cast<cplx<T>*>(p->Z), p->stream)) return;
      hipLaunchKernelGGL((k_pass_a<T, MODE>), dim3(1u << (logN - logP), cnt), dim3(threads), lds, p->stream,
                         in, (const RowDesc*)nullptr, mo, tw_table<T>(p, logR), twn_of<T>(p), logN, logK,
                         logP - logR, long(n0), long(in_ld), static_cast<cplx<T>*>(p->Z));
    });
    if (rc) return rc;
    rc = timed_launch(p, KC_FWD_B, [&] {
      if (try_pass_b_ct<T, true>(p, logK, nullptr, cnt, o, p->N, p->N, static_cast<const cplx<T>*>(p->Z),
                                 p->stream)) return;
      hipLaunchKernelGGL((k_pass_b<T, true>), dim3(1u << (logN - logP), cnt), dim3(threads), lds, p->stream,
                         static_cast<const cplx<T>*>(p->Z), (const RowDesc*)nullptr, tw_table<T>(p, logK),
                         twn_of<T>(p), logN, logK, logP - logK, o, long(p->N), long(p->N));
    });
    if (rc) return rc;
  }
  return CWT_OK;
}

// Overlap-save rows of the current row table: block spectra of the real signal x_dev (k_ols_fwd_r; block length
// 2^(LOGM + 1)) ...
template <typename T, int LOGM>
int launch_ols_fwd_r(cwt_plan* p, const void* x_dev, int64_t n0, long blocks, const OlsClasses& cls, hipStream_t st) {
  static const bool once = (allow_big_lds(&k_ols_fwd_r<T, LOGM>), true);
  (void)once;
  const size_t lds = ((size_t(1) << LOGM) + (size_t(1) << (LOGM - 4))) * sizeof(T);
  return timed_launch(p, KC_OLS_FWD, [&] {
    hipLaunchKernelGGL((k_ols_fwd_r<T, LOGM>), dim3(unsigned(blocks), unsigned(p->rt->ols_nbatch)), dim3(1 << (LOGM - 4)),
                       lds, st, static_cast<const T*>(x_dev), long(n0), p->logN, cls,
                       static_cast<const cplx<T>*>(p->tw_all), twn_of<T>(p), static_cast<cplx<T>*>(p->xs),
                       long(p->ols_x_ld), p->rt->ols_xs_sig);
  }, st);
}
template <typename T>
int launch_ols_fwd(cwt_plan* p, const void* x_dev, int64_t n0, hipStream_t st) {
  int rc = CWT_OK;
  {
    for (int g = 0; g < 2 && !rc; ++g) {
      const auto& G = p->rt->ols_grp[g];
      if (!G.nrows) continue;
      for (int d = 0; d < 3 && !rc; ++d) {
        if (!G.fwd_blocks[d]) continue;
        switch (G.logp + d) {                                   // log2 of the block length
          case 12: rc = launch_ols_fwd_r<T, 11>(p, x_dev, n0, G.fwd_blocks[d], G.cls, st); break;
          case 13: rc = launch_ols_fwd_r<T, 12>(p, x_dev, n0, G.fwd_blocks[d], G.cls, st); break;
          case 14: rc = launch_ols_fwd_r<T, 13>(p, x_dev, n0, G.fwd_blocks[d], G.cls, st); break;
          case 15: rc = launch_ols_fwd_r<T, 14>(p, x_dev, n0, G.fwd_blocks[d], G.cls, st); break;
          default: return fail(CWT_EINVAL, "overlap-save block length");
        }
      }
    }
    return rc;
  }
}
// ... and the rows themselves (k_ols_ct)
template <typename T, int LOGP>
int launch_ols_rows_p(cwt_plan* p, int g, cplx<T>* W, int64_t ldw, int64_t ncols, hipStream_t st) {
  const cwt_plan::RowTable* rt = p->rt;
  const auto& G = rt->ols_grp[g];
  static const bool once = (allow_big_lds(&k_ols_ct<T, LOGP>), true);
  (void)once;
  const size_t lds = ((size_t(1) << LOGP) + (size_t(1) << (LOGP - 4))) * sizeof(T);
  return timed_launch(p, g == 0 ? KC_OLS_SMALL : KC_OLS, [&] {
    hipLaunchKernelGGL((k_ols_ct<T, LOGP>), dim3(unsigned(G.wgs)), dim3(1 << (LOGP - 4)), lds, st,
                       static_cast<const cplx<T>*>(p->xs), rt->rows_dev + rt->ols_first + G.row_first,
                       static_cast<const cplx<T>*>(rt->gt_dev), static_cast<const cplx<T>*>(p->tw_all), twn_of<T>(p),
                       p->logN, G.cls, W, long(ldw), long(ncols));
  }, st);
}
template <typename T>
int launch_ols_rows(cwt_plan* p, cplx<T>* W, int64_t ldw, int64_t ncols, hipStream_t st) {
  int rc = CWT_OK;
  for (int g = 0; g < 2 && !rc; ++g) {        // the half-size tiles first (by far the longer launch since the rows with long
                                              // halos went to the polynomial form), then the default tile's rows
    const auto& G = p->rt->ols_grp[g];
    if (!G.nrows) continue;
    switch (G.logp) {
      case 12: rc = launch_ols_rows_p<T, 12>(p, g, W, ldw, ncols, st); break;
      case 13: rc = launch_ols_rows_p<T, 13>(p, g, W, ldw, ncols, st); break;
      default: return fail(CWT_EINVAL, "overlap-save tile size");
    }
  }
  return rc;
}

// Rows clipped at Nyquist (k_aols_*): band-passed complex signal x_M = IFFT_N(xhat mask) through the two-pass kernels
// (the mask is the pseudo-row at aux_first: profile 1), its block spectra, then every (block, row) pair.
template <typename T, int LOGP>
int launch_aols_p(cwt_plan* p, const void* xhat_dev, cplx<T>* W, int64_t ldw, int64_t ncols, hipStream_t st) {
  const cwt_plan::RowTable* rt = p->rt;
  const AolsGeom& g = rt->aols_geom;
  constexpr int P = 1 << LOGP;
  // a batch goes through in chunks of signals: the band-passed signals and their block spectra of one chunk stay in the
  // Infinity Cache between the four kernels (2 x 16 N + ~18 N bytes per signal)
  const int nb = rt->aols_nbatch;
  const int chunk = std::max(1, std::min(nb, std::min(balanced_chunk(p, nb), int((size_t(96) << 20) / (size_t(p->N) * sizeof(cplx<T>))))));
  int rc = grow(&p->xm, &p->xm_bytes, size_t(chunk) * size_t(p->N) * sizeof(cplx<T>), st);
  if (!rc) rc = grow(&p->xsa, &p->xsa_bytes, size_t(chunk) * size_t(g.nblocks) * size_t(P + 8) * sizeof(cplx<T>), st);
  if (!rc) rc = ensure_z(p, chunk);
  if (rc) return rc;
  const int logK = two_pass_logk(p), logR = p->logN - logK;
  Mother one;
  one.kind = MOTHER_DOG; one.m = 0; one.p = 0.0; one.table = nullptr;       // profile(0 * k) = 1
  cplx<T>* Z = static_cast<cplx<T>*>(p->Z);
  cplx<T>* xm = static_cast<cplx<T>*>(p->xm);
  static const bool once = (allow_big_lds(&k_aols_fwd<T, LOGP>), allow_big_lds(&k_aols_rows<T, LOGP>), true);
  (void)once;
  const size_t lds = ((size_t(1) << LOGP) + (size_t(1) << (LOGP - 4))) * sizeof(T);
  for (int b0 = 0; b0 < nb; b0 += chunk) {
    const int cnt = std::min(chunk, nb - b0);
    bool ok = true;
    rc = timed_launch(p, KC_AOLS_PRE, [&] {
      ok = try_pass_a_ct<T, IN_SPECTRUM>(p, logR, xhat_dev, rt->rows_dev + rt->aux_first + b0, cnt, one, 0L, 0L, Z, st);
    }, st);
    if (!rc && !ok) rc = fail(CWT_EINVAL, "k_aols rows need the default geometry");
    if (!rc) rc = timed_launch(p, KC_AOLS_PRE, [&] {
      ok = try_pass_b_ct<T, false>(p, logK, nullptr, cnt, xm, p->N, p->N, Z, st);
    }, st);
    if (!rc && !ok) rc = fail(CWT_EINVAL, "k_aols rows need the default geometry");
    if (!rc) rc = timed_launch(p, KC_AOLS_PRE, [&] {
      hipLaunchKernelGGL((k_aols_fwd<T, LOGP>), dim3(unsigned(g.nblocks), unsigned(cnt)), dim3(1 << (LOGP - 4)), lds, st, xm,
                         p->logN, g.halo, static_cast<const cplx<T>*>(p->tw_all), static_cast<cplx<T>*>(p->xsa));
    }, st);
    if (!rc) rc = timed_launch(p, KC_AOLS, [&] {
      hipLaunchKernelGGL((k_aols_rows<T, LOGP>), dim3(unsigned(rt->aols_wgs), unsigned(cnt)), dim3(1 << (LOGP - 4)), lds, st,
                         static_cast<const cplx<T>*>(p->xsa), rt->rows_dev + rt->aols_first + long(b0) * g.nrows,
                         static_cast<const T*>(rt->agt_dev), static_cast<const cplx<T>*>(p->tw_all), g,
                         static_cast<const cplx<T>*>(xhat_dev), long(p->N >> 1), W, long(ldw), long(ncols));
    }, st);
    if (rc) return rc;
  }
  return CWT_OK;
}
template <typename T>
int launch_aols(cwt_plan* p, const void* xhat_dev, cplx<T>* W, int64_t ldw, int64_t ncols, hipStream_t st) {
  switch (p->rt->aols_logp) {
    case 12: return launch_aols_p<T, 12>(p, xhat_dev, W, ldw, ncols, st);
    default: return fail(CWT_EINVAL, "k_aols tile size");
  }
}

// Band-limited rows in polynomial form: the filtered bands and the interval coefficients (k_poly_band, k_poly_coef) ...
// st2 != nullptr: the 8192- and 4096-point tiles on that second stream beside the 16384-point ones (three independent,
// latency-bound launches of one round of workgroups each: 30 + 19 + 20 us back to back), joined into st again.
template <typename T>
int launch_poly_coef(cwt_plan* p, const cplx<T>* xhat, const Mother& mo, int chunk, hipStream_t st, hipStream_t st2 = nullptr) {
  const cwt_plan::RowTable* rt = p->rt;
  const auto& ch = rt->poly_chunks[size_t(chunk)];
  int rc = grow(&p->pcoef, &p->pcoef_bytes, size_t(rt->poly_coef_elems) * sizeof(cplx<T>), st);
  if (!rc) rc = grow(&p->pband, &p->pband_bytes, size_t(rt->poly_band_elems) * sizeof(cplx<T>), st);
  if (rc) return rc;
  static const bool once = (allow_big_lds(&k_poly_coef<T, 13>), allow_big_lds(&k_poly_coef<T, 14>), true);
  (void)once;
  const RowDesc* rows = rt->rows_dev + rt->poly_first + ch.row_first;
  cplx<T>* coef = static_cast<cplx<T>*>(p->pcoef);
  cplx<T>* band = static_cast<cplx<T>*>(p->pband);
  rc = timed_launch(p, KC_POLY_COEF, [&] {
    for (int r0 = 0; r0 < ch.nrows; r0 += kMaxGridY)
      hipLaunchKernelGGL((k_poly_band<T>), dim3(1u << (ch.max_logk - 8), std::min(kMaxGridY, ch.nrows - r0)), dim3(256), 0, st,
                         xhat, rows + r0, mo, twn_of<T>(p), p->logN, band);
  }, st);
  if (rc) return rc;
  // largest tiles first: the 16384-point workgroups take a whole CU each and should find the chip as empty as it gets
  const cplx<T>* tw = static_cast<const cplx<T>*>(p->tw_all);
  auto lds_of = [](int lp) { return ((size_t(1) << lp) + (size_t(1) << (lp - 4))) * sizeof(T); };
  const bool split = st2 && ch.wgs[2] && (ch.wgs[1] || ch.wgs[0]);
  hipStream_t s2 = split ? st2 : st;
  if (split) {
    HIPCHECK(hipEventRecord(p->ev_big, st));             // the bands are ready
    HIPCHECK(hipStreamWaitEvent(st2, p->ev_big, 0));
  }
  if (!rc && ch.wgs[2]) rc = timed_launch(p, KC_POLY_COEF, [&] {
    hipLaunchKernelGGL((k_poly_coef<T, 14>), dim3(unsigned(ch.wgs[2])), dim3(1024), lds_of(14), st,
                       static_cast<const cplx<T>*>(band), rows, tw, ch.cls, coef); }, st);
  if (!rc && ch.wgs[1]) rc = timed_launch(p, KC_POLY_COEF, [&] {
    hipLaunchKernelGGL((k_poly_coef<T, 13>), dim3(unsigned(ch.wgs[1])), dim3(512), lds_of(13), s2,
                       static_cast<const cplx<T>*>(band), rows, tw, ch.cls, coef); }, s2);
  if (!rc && ch.wgs[0]) rc = timed_launch(p, KC_POLY_COEF, [&] {
    hipLaunchKernelGGL((k_poly_coef<T, 12>), dim3(unsigned(ch.wgs[0])), dim3(256), lds_of(12), s2,
                       static_cast<const cplx<T>*>(band), rows, tw, ch.cls, coef); }, s2);
  if (!rc && split) {
    HIPCHECK(hipEventRecord(p->ev_big, st2));
    HIPCHECK(hipStreamWaitEvent(st, p->ev_big, 0));
  }
  return rc;
}
// ... then the streaming kernel (k_poly_rows) over the rows of the chunk
template <typename T>
int launch_poly_rows(cwt_plan* p, int chunk, cplx<T>* W, int64_t ldw, int64_t ncols, hipStream_t st) {
  const cwt_plan::RowTable* rt = p->rt;
  const auto& ch = rt->poly_chunks[size_t(chunk)];
  const RowDesc* rows = rt->rows_dev + rt->poly_first + ch.row_first;
  const cplx<T>* coef = static_cast<const cplx<T>*>(p->pcoef);
  const int64_t per_wg = 256 * (sizeof(T) == 8 ? 1 : 2) * POLY_PASSES;
  // LDS: the coefficient sets of the intervals one workgroup touches (shortest interval 2^POLY_MIN_LOGR samples)
  const size_t lds2 = size_t((per_wg >> POLY_MIN_LOGR) + 2) * (POLY_MAX_DEGREE + 1) * sizeof(cplx<T>);
  return timed_launch(p, KC_POLY, [&] {
    for (int r0 = 0; r0 < ch.nrows; r0 += kMaxGridY)
      hipLaunchKernelGGL((k_poly_rows<T>), dim3(unsigned((ncols + per_wg - 1) / per_wg), std::min(kMaxGridY, ch.nrows - r0)),
                         dim3(256), lds2, st, rows + r0, coef, twn_of<T>(p), p->logN, W, long(ldw), long(ncols));
  }, st);
}

// Restores the plan's stream when a scope that redirected launches to a side stream is left on any path.
struct StreamGuard {
  cwt_plan* p;
  hipStream_t keep;
  explicit StreamGuard(cwt_plan* plan) : p(plan), keep(plan->stream) {}
  ~StreamGuard() { p->stream = keep; }
};

template <typename T>
int rows_launch(cwt_plan* p, const void* xhat_dev, const Mother& mo, int nrows, void* W_dev, int64_t ldw,
                int64_t ncols, const void* x_dev, int64_t n0);

// Queues every row of the current row table.  On an error after work was forked to the side streams the side streams
// are drained before returning, so that no kernel still reads the row table, the block spectra or the filter tables when
// the caller (or the next call) frees or rebuilds them.
template <typename T>
int rows_impl(cwt_plan* p, const void* xhat_dev, const Mother& mo, int nrows, void* W_dev, int64_t ldw,
              int64_t ncols, const void* x_dev = nullptr, int64_t n0 = 0) {
  const int rc = rows_launch<T>(p, xhat_dev, mo, nrows, W_dev, ldw, ncols, x_dev, n0);
  if (rc) {
    const std::string msg = g_err;                       // the drain below must not overwrite the message
    for (hipStream_t s : {p->side[0], p->side[1], p->side2}) if (s) (void)hipStreamSynchronize(s);
    (void)hipGetLastError();
    g_err = msg;
  }
  return rc;
}

template <typename T>
int rows_launch(cwt_plan* p, const void* xhat_dev, const Mother& mo, int nrows, void* W_dev, int64_t ldw,
                int64_t ncols, const void* x_dev, int64_t n0) {
  const int logN = p->logN;
  const cplx<T>* xhat = static_cast<const cplx<T>*>(xhat_dev);
  cplx<T>* W = static_cast<cplx<T>*>(W_dev);
  int rc = check_geometry(p);
  if (rc) return rc;
  if (p->rt->n_small) {
    if (logN <= 3) {
      const int total = nrows << logN;
      return timed_launch(p, KC_DIRECT, [&] {
        hipLaunchKernelGGL((k_direct<T, IN_SPECTRUM>), dim3((total + 63) / 64), dim3(64), 0, p->stream,
                           xhat_dev, p->rt->rows_dev, nrows, mo, logN, 0L, 0L, W, long(ldw), long(ncols));
      });
    }
    // several rows per workgroup: aim at 4096 points (256 threads)
    const int logTB = std::max(0, std::min(12, p->log_wg_points) - logN);
    const int TB = 1 << logTB;
    const int threads = TB << (logN - 4);
    const size_t lds = (size_t(TB) << logN) * sizeof(T);
    return timed_launch(p, KC_SMALL, [&] {
      hipLaunchKernelGGL((k_small<T, IN_SPECTRUM>), dim3((nrows + TB - 1) / TB), dim3(threads), lds,
                         p->stream, xhat_dev, p->rt->rows_dev, nrows, mo, tw_table<T>(p, logN), logN, logTB,
                         0L, 0L, W, long(ldw), long(ncols));
    });
  }
  const int logP = std::min(p->log_wg_points, logN);
  const int threads = 1 << (logP - 4);
  const size_t lds = (size_t(1) << logP) * sizeof(T);
  if (p->rt->n_ols && !x_dev) return fail(CWT_EINVAL, "overlap-save rows need the signal");
  // (short transforms run their kernels back to back: at N = 2^16 / 2^17 the events and waits of the side streams cost
  // more than the overlap returns -- measured 0.149 against 0.129 ms and 0.226 against 0.204 ms per 256-row transform)
  const bool side_narrow = p->overlap_narrow && !p->profile && (p->rt->n_wide || p->rt->n_ols || p->rt->n_aols) &&
                           (p->rt->n_narrow || p->rt->n_poly) && logN >= 18;
  // block spectra of the overlap-save rows: beside the two-pass chain on side stream 1 (they only need the signal)
  const bool ols_early = p->rt->n_ols && p->ols_launched;       // already queued on side stream 1 by cwt_transform
  const bool ols_side = p->rt->n_ols && !ols_early && p->ols_side && !p->profile && p->rt->n_wide;
  if (p->rt->n_ols && !ols_early) {
    rc = grow(&p->xs, &p->xs_bytes, size_t(p->rt->ols_xs_elems) * sizeof(cplx<T>), p->stream);
    if (rc) return rc;
  }
  if (side_narrow || ols_side) HIPCHECK(hipEventRecord(p->ev_fork, p->stream));
  if (side_narrow) HIPCHECK(hipStreamWaitEvent(p->side[0], p->ev_fork, 0));   // starts after the spectrum exists
  if (ols_side) {
    HIPCHECK(hipStreamWaitEvent(p->side[1], p->ev_fork, 0));
    rc = launch_ols_fwd<T>(p, x_dev, n0, p->side[1]);
    if (rc) return rc;
    HIPCHECK(hipEventRecord(p->ev_ols, p->side[1]));
  }
  // Polynomial rows, first half: bands + interval coefficients.  Short, latency-bound launches of LARGE workgroups (a
  // 16384-point transform fills a CU) on which the biggest kernel of the step (k_poly_rows) waits: they are queued before
  // the overlap-save rows.  What starved them in the first build (k_poly_coef 337 us instead of 67, k_poly_rows alone at
  // the end of the step) were the 512-thread / 68-KB workgroups of the 8192-point overlap-save tiles launched first; with
  // the 4096-point tiles first the coefficient workgroups find their slots, and holding the overlap-save rows back until
  // the coefficients are done only leaves the chip idle: 0.916 against 0.898 ms at config 2 (EXPERIMENTS.md I.4).
  const bool poly_on_side = p->rt->n_poly && side_narrow;
  if (p->rt->n_poly) {
    rc = launch_poly_coef<T>(p, xhat, mo, 0, poly_on_side ? p->side[0] : p->stream, poly_on_side ? p->side2 : nullptr);
    if (rc) return rc;
  }
  if (ols_early) {                     // block spectra already queued on side stream 1 by cwt_transform
    rc = launch_ols_rows<T>(p, W, ldw, ncols, p->side[1]);
    if (rc) return rc;
    HIPCHECK(hipEventRecord(p->ev_ols, p->side[1]));
  }
  if (p->rt->n_wide) {                 // two-pass rows, chunk by chunk on the plan's stream (one intermediate buffer)
    const int logK = two_pass_logk(p), logR = logN - logK;
    const int chunk = balanced_chunk(p, p->rt->n_wide);
    const int nchunks = (p->rt->n_wide + chunk - 1) / chunk;
    rc = ensure_z(p, chunk);
    if (rc) return rc;
    cplx<T>* Z = static_cast<cplx<T>*>(p->Z);
    for (int c = 0; c < nchunks; ++c) {
      const int first = c * chunk, cnt = std::min(chunk, p->rt->n_wide - first);
      const RowDesc* rows = p->rt->rows_dev + p->rt->wide_first + first;
      rc = timed_launch(p, KC_PASS_A, [&] {
        if (try_pass_a_ct<T, IN_SPECTRUM>(p, logR, xhat_dev, rows, cnt, mo, 0L, 0L, Z, p->stream)) return;
        hipLaunchKernelGGL((k_pass_a<T, IN_SPECTRUM>), dim3(1u << (logN - logP), cnt), dim3(threads), lds, p->stream,
                           xhat_dev, rows, mo, tw_table<T>(p, logR), twn_of<T>(p), logN, logK, logP - logR, 0L, 0L, Z);
      });
      if (rc) return rc;
      rc = timed_launch(p, KC_PASS_B, [&] {
        if (try_pass_b_ct<T, false>(p, logK, rows, cnt, W, ldw, ncols, Z, p->stream)) return;
        hipLaunchKernelGGL((k_pass_b<T, false>), dim3(1u << (logN - logP), cnt), dim3(threads), lds, p->stream,
                           static_cast<const cplx<T>*>(Z), rows, tw_table<T>(p, logK), twn_of<T>(p), logN, logK,
                           logP - logK, W, long(ldw), long(ncols));
      });
      if (rc) return rc;
    }
  }
  if (p->rt->n_aols) {                 // after the two-pass chain: both use the intermediate buffer
    rc = launch_aols<T>(p, xhat_dev, W, ldw, ncols, p->stream);
    if (rc) return rc;
  }
  if (ols_early) {
    HIPCHECK(hipStreamWaitEvent(p->stream, p->ev_ols, 0));
  } else if (p->rt->n_ols) {
    if (ols_side) HIPCHECK(hipStreamWaitEvent(p->stream, p->ev_ols, 0));
    else rc = launch_ols_fwd<T>(p, x_dev, n0, p->stream);
    if (!rc) rc = launch_ols_rows<T>(p, W, ldw, ncols, p->stream);
    if (rc) return rc;
  }
  // band-limited rows: on a side stream beside the two-pass chain (fills its kernel boundaries and
  // tails) when "overlap_narrow" is set, else on the plan's own stream
  bool narrow_on_side = false;
  if (p->rt->n_poly) {                 // second half, on the same stream as the first (joined below when that is a side stream)
    narrow_on_side = poly_on_side;
    hipStream_t ps = poly_on_side ? p->side[0] : p->stream;
    const int nchunks = int(p->rt->poly_chunks.size());
    for (int c = 0; c < nchunks && !rc; ++c) {             // chunk c's rows, then chunk c + 1's coefficients, on one stream
      rc = launch_poly_rows<T>(p, c, W, ldw, ncols, ps);
      if (!rc && c + 1 < nchunks) rc = launch_poly_coef<T>(p, xhat, mo, c + 1, ps, poly_on_side ? p->side2 : nullptr);
    }
    if (rc) return rc;
    if (narrow_on_side) HIPCHECK(hipEventRecord(p->ev_a[0], p->side[0]));
  }
  if (p->rt->n_narrow) {
    if (narrow_ct_all_applies<T>(p)) {
      StreamGuard guard(p);                                 // p->stream is redirected below; restored on every path
      hipStream_t keep = p->stream;
      narrow_on_side = side_narrow;
      if (narrow_on_side) p->stream = p->side[0];
      int n_small_k, n_big, n_many;
      narrow_class_counts(p, &n_small_k, &n_big, &n_many);
      rc = CWT_OK;
      if (n_small_k) rc = timed_launch(p, KC_NARROW, [&] { launch_narrow_ct_all<T>(p, xhat, mo, W, ldw, ncols); });
      // the multi-term kernels (few rows, long workgroups) on a stream of their own: at small row counts (a rank's
      // share of 8) they would otherwise run alone at the end of the step
      const bool big_on_side2 = narrow_on_side && n_small_k && (n_many || n_big);
      hipStream_t sbig = p->side2;
      if (big_on_side2) {
        HIPCHECK(hipStreamWaitEvent(sbig, p->ev_fork, 0));
        p->stream = sbig;
      }
      if (!rc && n_many) rc = timed_launch(p, KC_NARROW_MANY, [&] { launch_narrow_ct_many<T>(p, xhat, mo, W, ldw, ncols); });
      if (!rc && n_big) rc = timed_launch(p, KC_NARROW_BIG, [&] { launch_narrow_ct_big<T>(p, xhat, mo, W, ldw, ncols); });
      p->stream = keep;
      if (rc) return rc;
      if (big_on_side2) {
        HIPCHECK(hipEventRecord(p->ev_big, sbig));
        HIPCHECK(hipStreamWaitEvent(p->side[0], p->ev_big, 0));          // joined through side stream 0
      }
      if (narrow_on_side) HIPCHECK(hipEventRecord(p->ev_a[0], p->side[0]));
    } else {
      for (const auto& g : p->rt->narrow_groups) {
        rc = timed_launch(p, KC_NARROW, [&] {
          for (int r0 = 0; r0 < g.count; r0 += kMaxGridY)
            hipLaunchKernelGGL((k_narrow<T>), dim3(1u << (logN - logP), std::min(kMaxGridY, g.count - r0)),
                               dim3(threads), lds, p->stream, xhat, p->rt->rows_dev + g.first + r0, mo,
                               tw_table<T>(p, g.logK), twn_of<T>(p), logN, g.logK, logP - g.logK, W, long(ldw),
                               long(ncols));
        });
        if (rc) return rc;
      }
    }
  }
  if (narrow_on_side) HIPCHECK(hipStreamWaitEvent(p->stream, p->ev_a[0], 0));
  return CWT_OK;
}

template <typename T>
int set_func_attrs() {
  // Workgroups use up to wg_points*sizeof(T) = 128 KiB of dynamic LDS; above 64 KiB HIP wants the
  // opt-in attribute.  A refusal is not fatal here: a launch that really needs it reports the error.
  const int big = 128 * 1024;
  const void* fns[] = {reinterpret_cast<const void*>(&k_small<T, IN_REAL>),
                       reinterpret_cast<const void*>(&k_small<T, IN_SPECTRUM>),
                       reinterpret_cast<const void*>(&k_small<T, IN_CPLX>),
                       reinterpret_cast<const void*>(&k_pass_a<T, IN_CPLX>),
                       reinterpret_cast<const void*>(&k_narrow<T>),
                       reinterpret_cast<const void*>(&k_pass_a<T, IN_REAL>),
                       reinterpret_cast<const void*>(&k_pass_a<T, IN_SPECTRUM>),
                       reinterpret_cast<const void*>(&k_pass_b<T, true>),
                       reinterpret_cast<const void*>(&k_pass_b<T, false>)};
  for (const void* f : fns)
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, big) != hipSuccess)
      (void)hipGetLastError();
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_narrow_ct_big<double>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, big) != hipSuccess)
    (void)hipGetLastError();
  return CWT_OK;
}

// Device -> host copy on the plan's stream, synchronous.  Small copies go straight through hipMemcpyAsync; large ones
// through the pinned ring of HostCopier (see there).
int copy_d2h(cwt_plan* p, void* dst_host, const void* src_dev, size_t bytes) {
  if (bytes < HostCopier::kChunk + HostCopier::kChunk / 2) {
    HIPCHECK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, p->stream));
    HIPCHECK(hipStreamSynchronize(p->stream));
    return CWT_OK;
  }
  HostCopier* c = copier_for(p->device);
  if (!c) {
    HIPCHECK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, p->stream));   // no pinned memory left
    HIPCHECK(hipStreamSynchronize(p->stream));
    return CWT_OK;
  }
  std::lock_guard<std::mutex> one_copy(c->busy);
  const size_t chunk = HostCopier::kChunk;
  const size_t nchunks = (bytes + chunk - 1) / chunk;
  char* dst = static_cast<char*>(dst_host);
  const char* src = static_cast<const char*>(src_dev);
  hipError_t err = hipSuccess;
  for (size_t i = 0; i <= nchunks && err == hipSuccess; ++i) {
    if (i < nchunks) {                                   // DMA of chunk i into its slot (after the slot's last scatter)
      const int sl = int(i % HostCopier::kSlots);
      c->wait_slot(sl);
      const size_t n = std::min(chunk, bytes - i * chunk);
      err = hipMemcpyAsync(c->slot[sl], src + i * chunk, n, hipMemcpyDeviceToHost, p->stream);
      if (err == hipSuccess) err = hipEventRecord(c->ev[sl], p->stream);
    }
    if (i > 0 && err == hipSuccess) {                    // chunk i-1 has landed: hand it to the workers
      const int sl = int((i - 1) % HostCopier::kSlots);
      err = hipEventSynchronize(c->ev[sl]);
      if (err == hipSuccess) c->scatter(sl, dst + (i - 1) * chunk, std::min(chunk, bytes - (i - 1) * chunk));
    }
  }
  for (int sl = 0; sl < HostCopier::kSlots; ++sl) c->wait_slot(sl);
  if (err != hipSuccess) { (void)hipStreamSynchronize(p->stream); return fail(CWT_EHIP, std::string("device -> host copy: ") + hipGetErrorString(err)); }
  return CWT_OK;
}

// Side streams (band-limited rows, overlap-save chain, coefficients of the polynomial rows beside the plan's stream) at the
// default priority: all queues are served alike.  Rounds 1-2 created them at the lowest priority (+1 % on the fp64 step at
// sustained clocks); a high priority for the coefficient stream measured +-0 in round 4 (EXPERIMENTS.md).
hipError_t create_side_stream(hipStream_t* s) { return hipStreamCreateWithFlags(s, hipStreamNonBlocking); }

int grow(void** buf, size_t* have, size_t need, hipStream_t s) {
  if (*have >= need) return CWT_OK;
  ++g_scratch_gen;
  if (*buf) { HIPCHECK(hipStreamSynchronize(s)); HIPCHECK(hipFree(*buf)); *buf = nullptr; *have = 0; }
  if (hipMalloc(buf, need) != hipSuccess) return fail(CWT_ENOMEM, "device allocation failed");
  *have = need;
  return CWT_OK;
}

}  // namespace

// =============================================================================================
extern "C" {

const char* cwt_backend(void) { return CWT_BACKEND_NAME; }
const char* cwt_last_error(void) { return g_err.c_str(); }

int cwt_device_count(int* count) {
  if (!count) return fail(CWT_EINVAL, "count is NULL");
  *count = 0;
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) { *count = 0; return fail(CWT_ENODEV, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
  return CWT_OK;
}

int cwt_plan_create(cwt_plan** plan, int device, int64_t nfft, int precision, int max_rows) {
  if (!plan) return fail(CWT_EINVAL, "plan is NULL");
  *plan = nullptr;
  if (precision != 32 && precision != 64) return fail(CWT_EINVAL, "precision must be 32 or 64");
  if (nfft < 2 || nfft > (int64_t(1) << 24) || (nfft & (nfft - 1)))
    return fail(CWT_EINVAL, "nfft must be a power of two in [2, 2^24]");
  if (max_rows < 1) return fail(CWT_EINVAL, "max_rows must be >= 1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(CWT_ENODEV, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(CWT_EINVAL, "device index out of range");
  HIPCHECK(hipSetDevice(device));
  cwt_plan* p = new cwt_plan();
  p->device = device;
  p->N = nfft;
  p->logN = ilog2(nfft);
  p->prec = precision;
  p->max_rows = max_rows;
  p->log_wg_points = precision == 64 ? 13 : 14;
  p->narrow_terms = precision == 64 ? 4 : 8;      // see the cost table in build_row_table
  if (const char* e = std::getenv("CWT_TOLERANCE")) {   // default accuracy target of plans created from here on
    const double t = std::atof(e);
    if (t > 0 && t <= 1e-2) p->tolerance = t;
  }
  int rc = precision == 64 ? build_tables<double>(p) : build_tables<float>(p);
  if (!rc) rc = precision == 64 ? set_func_attrs<double>() : set_func_attrs<float>();
  for (int i = 0; i < 2 && !rc; ++i) {
    if (create_side_stream(&p->side[i]) != hipSuccess ||
        hipEventCreate(&p->ev_a[i]) != hipSuccess || hipEventCreate(&p->ev_b[i]) != hipSuccess)
      rc = fail(CWT_EHIP, "cannot create side streams/events");
  }
  if (!rc && hipEventCreate(&p->ev_fork) != hipSuccess) rc = fail(CWT_EHIP, "cannot create event");
  if (!rc && hipEventCreate(&p->ev_ols) != hipSuccess) rc = fail(CWT_EHIP, "cannot create event");
  if (!rc && (create_side_stream(&p->side2) != hipSuccess || hipEventCreate(&p->ev_big) != hipSuccess))
    rc = fail(CWT_EHIP, "cannot create side streams/events");
  p->narrow_mix = precision == 64;
  p->ols_big = precision == 32;                   // measured: +2.5 % (fp32 DOG), +-0 at one GPU and -3 % per rank of 8 in fp64
  for (auto& t : p->slots) {
    // (+ max_rows / 3 + 4: pseudo-rows -- the mask of the k_aols rows, one per signal of a batch)
    if (!rc && hipMalloc(reinterpret_cast<void**>(&t.rows_dev), table_capacity(max_rows) * sizeof(RowDesc)) != hipSuccess)
      rc = fail(CWT_ENOMEM, "row table allocation failed");
    if (!rc && hipHostMalloc(reinterpret_cast<void**>(&t.rows_pinned), table_capacity(max_rows) * sizeof(RowDesc)) != hipSuccess)
      rc = fail(CWT_ENOMEM, "pinned row table allocation failed");
    if (!rc && hipEventCreate(&t.uploaded) != hipSuccess) rc = fail(CWT_EHIP, "cannot create event");
  }
  if (!rc && hipMalloc(&p->weights_dev, size_t(max_rows) * sizeof(double)) != hipSuccess)
    rc = fail(CWT_ENOMEM, "weights allocation failed");
  for (int i = 0; i < 2; ++i) {
    if (!rc && hipHostMalloc(&p->weights_pinned[i], size_t(max_rows) * sizeof(double)) != hipSuccess)
      rc = fail(CWT_ENOMEM, "pinned weights allocation failed");
    if (!rc && hipEventCreate(&p->weights_ev[i]) != hipSuccess) rc = fail(CWT_EHIP, "cannot create event");
  }
  if (rc) { cwt_plan_destroy(p); return rc; }
  *plan = p;
  return CWT_OK;
}

int cwt_plan_destroy(cwt_plan* p) {
  if (!p) return CWT_OK;
  (void)hipSetDevice(p->device);
  (void)hipStreamSynchronize(p->stream);
  for (int i = 0; i < 2; ++i) {
    if (p->side[i]) { (void)hipStreamSynchronize(p->side[i]); (void)hipStreamDestroy(p->side[i]); }
    if (p->ev_a[i]) (void)hipEventDestroy(p->ev_a[i]);
    if (p->ev_b[i]) (void)hipEventDestroy(p->ev_b[i]);
  }
  if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
  if (p->ev_ols) (void)hipEventDestroy(p->ev_ols);
  if (p->side2) { (void)hipStreamSynchronize(p->side2); (void)hipStreamDestroy(p->side2); }
  if (p->ev_big) (void)hipEventDestroy(p->ev_big);
  for (auto& g : p->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
  for (auto& t : p->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
  for (auto e : p->free_events) (void)hipEventDestroy(e);
  void* bufs[] = {p->tw_all, p->twn_lo, p->weights_dev, p->Z, p->xs, p->xm, p->xsa, p->pcoef, p->pband, p->range_dev, p->hx, p->hxhat, p->hW,
                  p->bs_khat[0], p->bs_khat[1], p->bs_a, p->bs_spec, p->bs_par};
  for (void* b : bufs) if (b) (void)hipFree(b);
  for (auto& t : p->slots) {
    if (t.gt_dev) (void)hipFree(t.gt_dev);
    if (t.agt_dev) (void)hipFree(t.agt_dev);
    if (t.rows_dev) (void)hipFree(t.rows_dev);
    if (t.rows_pinned) (void)hipHostFree(t.rows_pinned);
    if (t.uploaded) (void)hipEventDestroy(t.uploaded);
  }
  if (p->hstage) (void)hipHostFree(p->hstage);
  for (int i = 0; i < 2; ++i) {
    if (p->weights_pinned[i]) (void)hipHostFree(p->weights_pinned[i]);
    if (p->weights_ev[i]) (void)hipEventDestroy(p->weights_ev[i]);
  }
  delete p;
  return CWT_OK;
}

int cwt_plan_set_stream(cwt_plan* p, void* hip_stream) {
  if (!p) return fail(CWT_EINVAL, "plan is NULL");
  HIPCHECK(hipStreamSynchronize(p->stream));
  p->stream = static_cast<hipStream_t>(hip_stream);
  return CWT_OK;
}

int cwt_plan_set_option(cwt_plan* p, const char* key, int64_t value) {
  if (!p || !key) return fail(CWT_EINVAL, "plan/key is NULL");
  const std::string k(key);
  auto pow2 = [](int64_t v) { return v > 0 && (v & (v - 1)) == 0; };
  for (const char* gone : {"overlap", "pass_b_prefetch", "pass_b_small", "stamps", "ols_tile", "ols_fwd_real", "sched", "narrow_wave"})
    if (k == gone)
      return fail(CWT_EINVAL, "option " + k + " belonged to a measured-and-rejected variant or a diagnostic that left the sources in "
                              "round 4 (EXPERIMENTS.md names the commit that still has it)");
  for (auto& t : p->slots) t.key.clear();   // the classification depends on the options
  struct Restore {   // a rejected geometry leaves every geometry-affecting field as it was
    cwt_plan* p; int lmax, wg, logk, nmax;
    ~Restore() {
      if (check_geometry(p) != CWT_OK) { p->loglmax = lmax; p->log_wg_points = wg; p->force_logk = logk; p->narrow_max_logk = nmax; }
    }
  } restore{p, p->loglmax, p->log_wg_points, p->force_logk, p->narrow_max_logk};
  if (k == "chunk_rows") { if (value < 0) return fail(CWT_EINVAL, "chunk_rows >= 0"); p->chunk_rows = int(value); }
  else if (k == "narrow") p->narrow = value != 0;
  else if (k == "narrow_max_k") { if (!pow2(value) || value < 16 || value > 4096) return fail(CWT_EINVAL, "narrow_max_k: power of two in [16,4096]"); p->narrow_max_logk = ilog2(value); }
  else if (k == "lmax") { if (!pow2(value) || value < 16 || value > 4096) return fail(CWT_EINVAL, "lmax: power of two in [16,4096]"); p->loglmax = ilog2(value); }
  else if (k == "wg_points") { if (!pow2(value) || value < 256 || value > 16384) return fail(CWT_EINVAL, "wg_points: power of two in [256,16384]"); p->log_wg_points = ilog2(value); }
  else if (k == "profile") p->profile = value != 0;
  else if (k == "ct") p->use_ct = value != 0;
  else if (k == "band_pass_a") p->band_pass_a = value != 0;
  else if (k == "overlap_narrow") p->overlap_narrow = value != 0;
  else if (k == "narrow_big") p->narrow_big = value != 0;
  else if (k == "narrow_mix") p->narrow_mix = value != 0;
  else if (k == "two_pass_logk") { if (value < 0 || value > 12) return fail(CWT_EINVAL, "two_pass_logk in [0,12] (0 = default)"); p->force_logk = int(value); }
  else if (k == "big_tiles") p->big_tiles = value != 0;
  else if (k == "narrow_small") p->narrow_small = value != 0;
  else if (k == "pass_a_small") p->pass_a_small = value != 0;
  else if (k == "narrow_terms") { if (value < 1 || value > 16) return fail(CWT_EINVAL, "narrow_terms in [1,16]"); p->narrow_terms = int(value); }
  else if (k == "ols") p->ols = value != 0;
  else if (k == "graph") p->graph = value != 0;
  else if (k == "host_direct") p->host_direct = value != 0;
  else if (k == "aols") p->aols = value != 0;
  else if (k == "poly") p->poly = value != 0;
  else if (k == "poly_degree") { if (value < 2 || value > POLY_MAX_DEGREE) return fail(CWT_EINVAL, "poly_degree in [2, 24]"); p->poly_degree = int(value); }
  else if (k == "poly_chunk_mb") { if (value < 0 || value > 4096) return fail(CWT_EINVAL, "poly_chunk_mb in [0, 4096] (0 = one chunk)"); p->poly_chunk_mb = int(value); }
  else if (k == "poly_max_logk") { if (value < 8 || value > 14) return fail(CWT_EINVAL, "poly_max_logk in [8, 14]"); p->poly_max_logk = int(value); }
  else if (k == "poly_min_logn") { if (value < 14 || value > 24) return fail(CWT_EINVAL, "poly_min_logn in [14, 24]"); p->poly_min_logn = int(value); }
  else if (k == "aols_min_rows") { if (value < 1 || value > 65536) return fail(CWT_EINVAL, "aols_min_rows >= 1"); p->aols_min_rows = int(value); }
  else if (k == "ols_side") p->ols_side = value != 0;
  else if (k == "ols_big") { if (value < 0 || value > 2) return fail(CWT_EINVAL, "ols_big: 0, 1 (blocks of two tiles) or 2 (also of four)"); p->ols_big = int(value); }
  else if (k == "ols_big4_max_halo") { if (value < 2048 || value > 8192 || (value & 63)) return fail(CWT_EINVAL, "ols_big4_max_halo: multiple of 64 in [2048, 8192]"); p->ols_big4_max_halo = int(value); }
  else if (k == "ols_big4_min_halo") { if (value < 64 || value > 8192) return fail(CWT_EINVAL, "ols_big4_min_halo in [64, 8192]"); p->ols_big4_min_halo = int(value); }
  else if (k == "ols_min_logn") { if (value < 15 || value > 24) return fail(CWT_EINVAL, "ols_min_logn in [15, 24]"); p->ols_min_logn = int(value); }
  else if (k == "ols_small_max_halo") { if (value < 0 || value > 1024 || (value & 63)) return fail(CWT_EINVAL, "ols_small_max_halo: multiple of 64 in [0, 1024]"); p->ols_small_max_halo = int(value); }
  else if (k == "ols_big_min_halo") { if (value < 64 || value > 8192) return fail(CWT_EINVAL, "ols_big_min_halo in [64, 8192]"); p->ols_big_min_halo = int(value); }
  else if (k == "ols_early") p->ols_early = value != 0;
  else if (k == "ols_max_halo") { if (value < 0 || value > 4096 || (value & 63)) return fail(CWT_EINVAL, "ols_max_halo: multiple of 64 in [0, 4096]"); p->ols_max_halo = int(value); }
  else if (k == "ols_fwd_weight") { if (value < 0 || value > 1000) return fail(CWT_EINVAL, "ols_fwd_weight: percent of a row, 0..1000"); p->ols_fwd_weight = double(value) / 100.0; }
  else if (k == "tolerance_neglog10") {   // integer alias of cwt_plan_set_tolerance for option sweeps: 10^-value; 0 = default
    if (value < 0 || value > 18) return fail(CWT_EINVAL, "tolerance_neglog10 in [0, 18]");
    p->tolerance = value ? std::pow(10.0, -double(value)) : 0.0;
  }
  else if (k == "big_terms") { if (value < 1 || value > 8) return fail(CWT_EINVAL, "big_terms in [1,8]"); p->big_terms = int(value); }
  else return fail(CWT_EINVAL, "unknown option " + k);
  return check_geometry(p);
}

int cwt_plan_set_tolerance(cwt_plan* p, double rel_tol) {
  if (!p) return fail(CWT_EINVAL, "plan is NULL");
  if (!(rel_tol >= 0) || rel_tol > 1e-2) return fail(CWT_EINVAL, "tolerance must be in [0, 1e-2] (0 = default)");
  for (auto& t : p->slots) t.key.clear();   // the classification depends on it
  p->tolerance = rel_tol;
  return CWT_OK;
}

int cwt_plan_set_auto_tolerance(cwt_plan* p, double target) {
  if (!p) return fail(CWT_EINVAL, "plan is NULL");
  if (!(target >= 0) || target > 1e-2) return fail(CWT_EINVAL, "target must be in [0, 1e-2] (0 = off)");
  p->auto_target = target;
  return CWT_OK;
}

int cwt_spectrum_range(cwt_plan* p, const void* xhat_dev, int64_t n, double* max_abs, double* rms_abs, double* floor_abs) {
  if (!p || !xhat_dev || !max_abs || !rms_abs || !floor_abs) return fail(CWT_EINVAL, "NULL argument");
  if (n < 1) return fail(CWT_EINVAL, "n must be >= 1");
  HIPCHECK(hipSetDevice(p->device));
  constexpr int kOut = SPECTRUM_SLOTS, kGroupsMax = 512;
  if (!p->range_dev && hipMalloc(reinterpret_cast<void**>(&p->range_dev), size_t(kGroupsMax + 1) * kOut * sizeof(double)) != hipSuccess)
    return fail(CWT_ENOMEM, "device allocation failed");
  // slices of at least 4096 bins, at most two workgroups per CU
  const int groups = int(std::max<int64_t>(1, std::min<int64_t>(kGroupsMax, n / 4096)));
  double* part = p->range_dev + kOut;
  if (p->prec == 64)
    hipLaunchKernelGGL((k_spectrum_range<double>), dim3(groups), dim3(256), (256 + kOut + 2) * sizeof(double), p->stream,
                       static_cast<const double2*>(xhat_dev), long(n), part);
  else
    hipLaunchKernelGGL((k_spectrum_range<float>), dim3(groups), dim3(256), (256 + kOut + 2) * sizeof(double), p->stream,
                       static_cast<const float2*>(xhat_dev), long(n), part);
  hipLaunchKernelGGL(k_spectrum_fold, dim3(1), dim3(192), 0, p->stream, part, groups, p->range_dev);
  HIPCHECK(hipGetLastError());
  double h[kOut] = {0};
  HIPCHECK(hipMemcpyAsync(h, p->range_dev, sizeof(h), hipMemcpyDeviceToHost, p->stream));
  HIPCHECK(hipStreamSynchronize(p->stream));
  *max_abs = std::sqrt(h[0]);
  *rms_abs = std::sqrt(h[1] / double(n));
  // The quietest stretch of the positive half at the resolution of a row's pass band: quarter-octave windows (single bins
  // below bin 4), each pooled with its two neighbours (3/4 octave ~ the 1-sigma band of the narrowest built-in filter).
  // Every bin from 1 to n/2 - 1 belongs to a window, so neither a quiet low end (a high-passed signal) nor a notch of
  // 3/4 octave or more escapes; a narrower notch does not take a row's energy away.
  std::vector<double> e, cnt;
  for (int w = 0; w < SPECTRUM_WINDOWS; ++w) {
    const int64_t lo = spectrum_window_lo(w), hi = std::min<int64_t>(spectrum_window_lo(w + 1), n / 2);
    if (hi <= lo) continue;
    e.push_back(h[2 + w]);
    cnt.push_back(double(hi - lo));
  }
  double fl = -1;
  for (size_t i = 0; i < e.size(); ++i) {
    double es = e[i], cs = cnt[i];
    if (i > 0) { es += e[i - 1]; cs += cnt[i - 1]; }
    if (i + 1 < e.size()) { es += e[i + 1]; cs += cnt[i + 1]; }
    const double r = std::sqrt(es / cs);
    if (fl < 0 || r < fl || r != r) fl = r;
  }
  *floor_abs = fl >= 0 ? fl : *rms_abs;
  return CWT_OK;
}

// The filter-relative tolerance that keeps `target` relative to every row's own peak for a spectrum of dynamic range
// D = max|xhat| / floor (cwt_spectrum_range): the truncation error of a row can reach tolerance * D / 4 (cwt_hip.h); white
// noise has D ~ 5 ... 7, up to ~20 when one of the few-bin windows at the low end happens to be quiet (which then costs half
// a decade of tolerance, not accuracy); a power of sqrt(10) (calls with like spectra share one cached row table), never
// looser than the target, never below round-off.  A spectrum with an empty stretch or a non-finite bin: round-off.
static double auto_tolerance_of(const cwt_plan* p, double target, double mx, double fl) {
  const double round_off = p->prec == 64 ? kDefaultTolerance64 : kDefaultTolerance32;
  double tol = target;
  if (!(fl > 0) || !std::isfinite(mx)) return round_off;
  const double excess = (mx / fl) / 8.0;
  if (excess > 1.0) tol = std::pow(10.0, 0.5 * std::floor(2.0 * std::log10(target / excess)));
  return std::max(tol, round_off);
}

int cwt_plan_auto_tolerance(cwt_plan* p, const void* xhat_dev, double target, double* rel_tol) {
  if (!p || !xhat_dev || !rel_tol) return fail(CWT_EINVAL, "NULL argument");
  if (!(target > 0) || target > 1e-2) return fail(CWT_EINVAL, "target must be in (0, 1e-2]");
  double mx = 0, rms = 0, fl = 0;
  const int rc = cwt_spectrum_range(p, xhat_dev, p->N, &mx, &rms, &fl);
  if (rc) return rc;
  p->last_range = fl > 0 ? mx / fl : std::numeric_limits<double>::infinity();
  *rel_tol = auto_tolerance_of(p, target, mx, fl);
  return CWT_OK;
}

int cwt_plan_get_tolerance(cwt_plan* p, double* rel_tol) {
  if (!p || !rel_tol) return fail(CWT_EINVAL, "NULL argument");
  *rel_tol = p->tolerance > 0 ? p->tolerance : (p->prec == 64 ? kDefaultTolerance64 : kDefaultTolerance32);
  return CWT_OK;
}

int cwt_plan_sync(cwt_plan* p) {
  if (!p) return fail(CWT_EINVAL, "plan is NULL");
  HIPCHECK(hipStreamSynchronize(p->stream));
  return CWT_OK;
}

int cwt_malloc(int device, void** ptr, size_t bytes) {
  if (!ptr) return fail(CWT_EINVAL, "ptr is NULL");
  HIPCHECK(hipSetDevice(device));
  if (hipMalloc(ptr, bytes) != hipSuccess) return fail(CWT_ENOMEM, "hipMalloc failed");
  return CWT_OK;
}
int cwt_free(int device, void* ptr) {
  HIPCHECK(hipSetDevice(device));
  HIPCHECK(hipFree(ptr));
  return CWT_OK;
}
int cwt_memcpy_h2d(cwt_plan* p, void* dst, const void* src, size_t bytes) {
  if (!p) return fail(CWT_EINVAL, "plan is NULL");
  HIPCHECK(hipSetDevice(p->device));
  HIPCHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, p->stream));
  HIPCHECK(hipStreamSynchronize(p->stream));
  return CWT_OK;
}
int cwt_memcpy_d2h(cwt_plan* p, void* dst, const void* src, size_t bytes) {
  if (!p) return fail(CWT_EINVAL, "plan is NULL");
  HIPCHECK(hipSetDevice(p->device));
  return copy_d2h(p, dst, src, bytes);
}

int cwt_forward_fft(cwt_plan* p, const void* x_dev, int64_t n0, void* xhat_dev) {
  if (!p || !x_dev || !xhat_dev) return fail(CWT_EINVAL, "NULL argument");
  if (n0 < 1 || n0 > p->N) return fail(CWT_EINVAL, "n0 must be in [1, nfft]");
  HIPCHECK(hipSetDevice(p->device));
  return p->prec == 64 ? fft_rows_impl<double, IN_REAL>(p, x_dev, 0, 1, n0, xhat_dev)
                       : fft_rows_impl<float, IN_REAL>(p, x_dev, 0, 1, n0, xhat_dev);
}

extern "C++" {
namespace {
// Makes the slot built from `key` current and returns true, or picks the least recently used slot for a rebuild
// (returns false; the caller builds p->rt->table and calls upload_row_table).  An empty key never matches.
bool select_table(cwt_plan* p, const std::vector<double>& key) {
  ++p->tick;
  if (!key.empty())
    for (auto& t : p->slots)
      if (t.key == key) { p->rt = &t; t.used = p->tick; return true; }
  cwt_plan::RowTable* lru = &p->slots[0];
  for (auto& t : p->slots) if (t.used < lru->used) lru = &t;
  lru->key.clear();
  lru->used = p->tick;
  p->rt = lru;
  return false;
}

// Copies the freshly built row table of the current slot to the device through the slot's pinned staging buffer and
// marks the slot as built from `key`.  The only wait is for the slot's previous copy (an event that completed long
// ago unless rebuilds come back to back); the stream is never synchronised.
int upload_row_table(cwt_plan* p, const std::vector<double>& key) {
  cwt_plan::RowTable* t = p->rt;
  HIPCHECK(hipEventSynchronize(t->uploaded));
  std::memcpy(t->rows_pinned, t->table.data(), t->table.size() * sizeof(RowDesc));
  HIPCHECK(hipMemcpyAsync(t->rows_dev, t->rows_pinned, t->table.size() * sizeof(RowDesc), hipMemcpyHostToDevice,
                          p->stream));
  HIPCHECK(hipEventRecord(t->uploaded, p->stream));
  t->key = key;
  t->build_id = ++p->tick;
  return CWT_OK;
}

std::vector<double> call_key(double kind, std::initializer_list<double> head, std::initializer_list<std::pair<const double*, int>> arrays) {
  std::vector<double> k{kind};
  k.insert(k.end(), head.begin(), head.end());
  for (const auto& a : arrays) k.insert(k.end(), a.first, a.first + a.second);
  return k;
}
}  // namespace
}  // extern "C++"

extern "C++" {
namespace {
Mother mother_of(int mother, double param) {
  Mother mo;
  mo.kind = mother; mo.m = int(std::lround(param)); mo.p = param; mo.table = nullptr;
  return mo;
}

// Filter tables of the overlap-save rows of the freshly uploaded row table (k_ols_gtab), on the plan's stream.
template <typename T>
int fill_ols_tables(cwt_plan* p, const Mother& mo) {
  cwt_plan::RowTable* t = p->rt;
  int rc = grow(&t->gt_dev, &t->gt_bytes, size_t(t->ols_gt_elems) * sizeof(cplx<T>), p->stream);
  if (rc) return rc;
  cplx<T>* gt = static_cast<cplx<T>*>(t->gt_dev);
  for (int g = 0; g < 2; ++g) {             // one launch per tile size: a row's K = P table is indexed by signed bins
    const auto& G = t->ols_grp[g];
    if (!G.nrows) continue;
    int maxk = 16;
    for (int i = 0; i < G.nrows; ++i) maxk = std::max(maxk, 1 << t->table[t->ols_first + G.row_first + i].logK);
    const dim3 grid((maxk + 255) / 256, G.nrows), block(256);
    const RowDesc* rows = t->rows_dev + t->ols_first + G.row_first;
    if (mo.kind == MOTHER_MORLET) hipLaunchKernelGGL((k_ols_gtab<T, MOTHER_MORLET>), grid, block, 0, p->stream, rows, mo, G.logp, gt);
    else if (mo.kind == MOTHER_PAUL) hipLaunchKernelGGL((k_ols_gtab<T, MOTHER_PAUL>), grid, block, 0, p->stream, rows, mo, G.logp, gt);
    else hipLaunchKernelGGL((k_ols_gtab<T, MOTHER_DOG>), grid, block, 0, p->stream, rows, mo, G.logp, gt);
  }
  HIPCHECK(hipGetLastError());
  return CWT_OK;
}

// Filter tables of the rows on the band-passed complex signal (k_aols_gtab), on the plan's stream.
template <typename T>
int fill_aols_tables(cwt_plan* p, const Mother& mo) {
  cwt_plan::RowTable* t = p->rt;
  int rc = grow(&t->agt_dev, &t->agt_bytes, size_t(t->aols_gt_elems) * sizeof(T), p->stream);
  if (rc) return rc;
  const int P = 1 << t->aols_logp;
  const dim3 grid(P / 256, t->aols_geom.nrows), block(256);     // (a batch: the tables of the first signal's rows serve all)
  const RowDesc* rows = t->rows_dev + t->aols_first;
  T* gt = static_cast<T*>(t->agt_dev);
  if (mo.kind == MOTHER_MORLET) hipLaunchKernelGGL((k_aols_gtab<T, MOTHER_MORLET>), grid, block, 0, p->stream, rows, mo, t->aols_logp, t->aols_geom, gt);
  else if (mo.kind == MOTHER_PAUL) hipLaunchKernelGGL((k_aols_gtab<T, MOTHER_PAUL>), grid, block, 0, p->stream, rows, mo, t->aols_logp, t->aols_geom, gt);
  else hipLaunchKernelGGL((k_aols_gtab<T, MOTHER_DOG>), grid, block, 0, p->stream, rows, mo, t->aols_logp, t->aols_geom, gt);
  HIPCHECK(hipGetLastError());
  return CWT_OK;
}

int prepare_rows_table(cwt_plan* p, bool have_signal, int mother, double param, double dt, const double* scales,
                       int nrows, int64_t ldw, int64_t ncols) {
  if (nrows < 1 || nrows > p->max_rows) return fail(CWT_EINVAL, "nrows must be in [1, max_rows]");
  if (ncols < 1 || ncols > p->N || ldw < ncols) return fail(CWT_EINVAL, "need 1 <= ncols <= nfft and ldw >= ncols");
  if (!(dt > 0) || !std::isfinite(dt)) return fail(CWT_EINVAL, "dt must be positive");
  const std::vector<double> key = call_key(0, {double(mother), param, dt, double(nrows), have_signal ? 1.0 : 0.0, double(ncols)},
                                           {{scales, nrows}});
  if (!select_table(p, key)) {
    double cre, cim;
    int rc = mother_constant(mother, param, &cre, &cim);
    if (rc) return rc;
    const double w1 = 2.0 * 3.14159265358979323846 * (1.0 / (double(p->N) * dt));  // ftfreqs[1], wavelet.py:94
    std::vector<double> a(nrows), ar(nrows), ai(nrows);
    for (int j = 0; j < nrows; ++j) {
      if (!(scales[j] > 0) || !std::isfinite(scales[j])) return fail(CWT_EINVAL, "scales must be positive and finite");
      a[j] = scales[j] * w1;
      const double norm = std::sqrt(scales[j] * w1 * double(p->N));                 // wavelet.py:102
      ar[j] = norm * cre;
      ai[j] = norm * cim;
    }
    rc = build_row_table(p, mother, param, a.data(), ar.data(), ai.data(), 0, nrows, nullptr, nullptr, 0, -1,
                         have_signal ? ncols : 0, ncols);
    if (!rc) rc = upload_row_table(p, key);
    if (!rc && p->rt->n_ols)
      rc = p->prec == 64 ? fill_ols_tables<double>(p, mother_of(mother, param)) : fill_ols_tables<float>(p, mother_of(mother, param));
    if (!rc && p->rt->n_aols)
      rc = p->prec == 64 ? fill_aols_tables<double>(p, mother_of(mother, param)) : fill_aols_tables<float>(p, mother_of(mother, param));
    if (rc) { p->rt->key.clear(); return rc; }
  }
  set_split(p);
  return CWT_OK;
}

// Rows of W from the spectrum xhat_dev; x_dev != NULL: the real signal the spectrum came from (n0 samples), which lets
// time-compact rows take the overlap-save form.
int transform_rows_common(cwt_plan* p, const void* xhat_dev, const void* x_dev, int64_t n0, int mother, double param,
                          double dt, const double* scales, int nrows, void* W_dev, int64_t ldw, int64_t ncols) {
  int rc = prepare_rows_table(p, x_dev != nullptr, mother, param, dt, scales, nrows, ldw, ncols);
  if (rc) return rc;
  const Mother mo = mother_of(mother, param);
  return p->prec == 64 ? rows_impl<double>(p, xhat_dev, mo, nrows, W_dev, ldw, ncols, x_dev, n0)
                       : rows_impl<float>(p, xhat_dev, mo, nrows, W_dev, ldw, ncols, x_dev, n0);
}

// The overlap-save rows need the signal only: cwt_transform queues them on side stream 1 BEFORE the forward FFT, so
// that they run beside it and beside the two-pass chain; rows_impl then skips them and joins the stream at its end.
template <typename T>
int launch_ols_early(cwt_plan* p, const void* x_dev, int64_t n0, void* W_dev, int64_t ldw, int64_t ncols) {
  int rc = grow(&p->xs, &p->xs_bytes, size_t(p->rt->ols_xs_elems) * sizeof(cplx<T>), p->stream);
  if (rc) return rc;
  HIPCHECK(hipEventRecord(p->ev_fork, p->stream));        // after the previous call's work and the row-table upload
  HIPCHECK(hipStreamWaitEvent(p->side[1], p->ev_fork, 0));
  rc = launch_ols_fwd<T>(p, x_dev, n0, p->side[1]);     // (the rows follow in rows_launch, behind the coefficients of the
  if (rc) return rc;                                    // polynomial rows)
  (void)W_dev; (void)ldw; (void)ncols;
  p->ols_launched = 1;
  return CWT_OK;
}
}  // namespace
}  // extern "C++"

int cwt_transform_rows(cwt_plan* p, const void* xhat_dev, int mother, double param, double dt,
                       const double* scales, int nrows, void* W_dev, int64_t ldw, int64_t ncols) {
  if (!p || !xhat_dev || !scales || !W_dev) return fail(CWT_EINVAL, "NULL argument");
  HIPCHECK(hipSetDevice(p->device));
  return transform_rows_common(p, xhat_dev, nullptr, 0, mother, param, dt, scales, nrows, W_dev, ldw, ncols);
}

int cwt_transform(cwt_plan* p, const void* x_dev, int64_t n0, int mother, double param, double dt,
                  const double* scales, int nrows, void* xhat_dev, void* W_dev, int64_t ldw, int64_t ncols) {
  if (!p || !x_dev || !scales || !W_dev) return fail(CWT_EINVAL, "NULL argument");
  if (n0 < 1 || n0 > p->N) return fail(CWT_EINVAL, "n0 must be in [1, nfft]");
  HIPCHECK(hipSetDevice(p->device));
  int rc = prepare_rows_table(p, true, mother, param, dt, scales, nrows, ldw, ncols);
  if (rc) return rc;
  // the caller does not want the spectrum: computed (into plan scratch) only if some row needs it
  const bool only_ols = !xhat_dev && p->rt->n_ols == nrows;      // every row is an overlap-save row on the real signal
  if (!xhat_dev && !only_ols) {
    rc = grow(&p->hxhat, &p->hxhat_bytes, size_t(p->N) * 2 * p->esize(), p->stream);
    if (rc) return rc;
    xhat_dev = p->hxhat;
  }
  const Mother mo = mother_of(mother, param);
  auto enqueue = [&]() -> int {
    p->ols_launched = 0;
    int r = CWT_OK;
    if (only_ols)
      return p->prec == 64 ? rows_impl<double>(p, nullptr, mo, nrows, W_dev, ldw, ncols, x_dev, n0)
                           : rows_impl<float>(p, nullptr, mo, nrows, W_dev, ldw, ncols, x_dev, n0);
    if (p->rt->n_ols && p->ols_early && !p->profile) {
      r = p->prec == 64 ? launch_ols_early<double>(p, x_dev, n0, W_dev, ldw, ncols)
                        : launch_ols_early<float>(p, x_dev, n0, W_dev, ldw, ncols);
      if (r) return r;
    }
    r = p->prec == 64 ? fft_rows_impl<double, IN_REAL>(p, x_dev, 0, 1, n0, xhat_dev)
                      : fft_rows_impl<float, IN_REAL>(p, x_dev, 0, 1, n0, xhat_dev);
    if (!r) r = p->prec == 64 ? rows_impl<double>(p, xhat_dev, mo, nrows, W_dev, ldw, ncols, x_dev, n0)
                              : rows_impl<float>(p, xhat_dev, mo, nrows, W_dev, ldw, ncols, x_dev, n0);
    p->ols_launched = 0;
    return r;
  };
  if (!p->graph || p->profile) return enqueue();
  // Option "graph": the same call (buffers, shapes, row table) for the second time is captured into a HIP graph -- the side
  // streams join the capture through the events that fork and join them -- and replayed from then on: one launch instead
  // of 10-20 launches and as many event operations per transform.
  const std::vector<uint64_t> gkey = {uint64_t(reinterpret_cast<uintptr_t>(x_dev)), uint64_t(n0),
                                      uint64_t(reinterpret_cast<uintptr_t>(xhat_dev)), uint64_t(reinterpret_cast<uintptr_t>(W_dev)),
                                      uint64_t(ldw), uint64_t(ncols), uint64_t(reinterpret_cast<uintptr_t>(p->rt)), p->rt->build_id,
                                      uint64_t(reinterpret_cast<uintptr_t>(p->stream)), g_scratch_gen};
  cwt_plan::GraphSlot* slot = nullptr;
  for (auto& g : p->graphs) if (g.key == gkey) slot = &g;
  if (slot && slot->exec) {
    slot->used = ++p->tick;
    ++p->graph_replays;
    HIPCHECK(hipGraphLaunch(slot->exec, p->stream));
    return CWT_OK;
  }
  if (!slot) {                                            // first occurrence: remember it (least recently used slot), run plainly
    slot = &p->graphs[0];
    for (auto& g : p->graphs) if (g.used < slot->used) slot = &g;
    if (slot->exec) { HIPCHECK(hipStreamSynchronize(p->stream)); (void)hipGraphExecDestroy(slot->exec); slot->exec = nullptr; }
    slot->key = gkey; slot->seen = 1; slot->used = ++p->tick;
    return enqueue();
  }
  slot->used = ++p->tick;                                 // second occurrence: every buffer has its size, nothing allocates
  if (hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    p->graph = 0;                                         // no capture on this runtime: plain launches from now on
    return enqueue();
  }
  rc = enqueue();
  hipGraph_t graph = nullptr;
  const hipError_t ec = hipStreamEndCapture(p->stream, &graph);
  if (rc || ec != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    p->graph = 0;
    return rc ? rc : enqueue();
  }
  const hipError_t ei = hipGraphInstantiate(&slot->exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ei != hipSuccess) { slot->exec = nullptr; (void)hipGetLastError(); p->graph = 0; return enqueue(); }
  HIPCHECK(hipGraphLaunch(slot->exec, p->stream));
  return CWT_OK;
}

int cwt_transform_rows_batch(cwt_plan* p, const void* xhat_dev, int nbatch, int64_t xhat_ld, int mother,
                             double param, double dt, const double* scales, int nrows, void* W_dev,
                             int64_t ldw, int64_t ncols) {
  if (!p || !xhat_dev || !scales || !W_dev) return fail(CWT_EINVAL, "NULL argument");
  if (nbatch < 1 || nrows < 1 || int64_t(nbatch) * nrows > p->max_rows)
    return fail(CWT_EINVAL, "need nbatch*nrows <= max_rows");
  if (xhat_ld < p->N) return fail(CWT_EINVAL, "xhat_ld must be >= nfft");
  if (ncols < 1 || ncols > p->N || ldw < ncols) return fail(CWT_EINVAL, "need 1 <= ncols <= nfft and ldw >= ncols");
  if (!(dt > 0) || !std::isfinite(dt)) return fail(CWT_EINVAL, "dt must be positive");
  HIPCHECK(hipSetDevice(p->device));
  const int total = nbatch * nrows;
  const std::vector<double> key = call_key(2, {double(mother), param, dt, double(nbatch), double(xhat_ld), double(nrows)},
                                           {{scales, nrows}});
  if (!select_table(p, key)) {
    double cre, cim;
    int rc = mother_constant(mother, param, &cre, &cim);
    if (rc) return rc;
    const double w1 = 2.0 * 3.14159265358979323846 * (1.0 / (double(p->N) * dt));
    std::vector<double> a(total), ar(total), ai(total);
    for (int j = 0; j < total; ++j) {
      const double s = scales[j % nrows];
      if (!(s > 0) || !std::isfinite(s)) return fail(CWT_EINVAL, "scales must be positive and finite");
      a[j] = s * w1;
      const double norm = std::sqrt(s * w1 * double(p->N));
      ar[j] = norm * cre;
      ai[j] = norm * cim;
    }
    // W is treated as one (nbatch*nrows) x ldw matrix: row b*nrows + j = scale j of signal b
    rc = build_row_table(p, mother, param, a.data(), ar.data(), ai.data(), xhat_ld, total, nullptr, nullptr, nrows);
    if (!rc) rc = upload_row_table(p, key);
    if (rc) return rc;
  }
  set_split(p);
  Mother mo;
  mo.kind = mother; mo.m = int(std::lround(param)); mo.p = param; mo.table = nullptr;
  return p->prec == 64 ? rows_impl<double>(p, xhat_dev, mo, total, W_dev, ldw, ncols)
                       : rows_impl<float>(p, xhat_dev, mo, total, W_dev, ldw, ncols);
}

int cwt_transform_batch(cwt_plan* p, const void* x_dev, int nbatch, int64_t x_ld, int64_t n0, int mother,
                        double param, double dt, const double* scales, int nrows, void* xhat_dev, void* W_dev,
                        int64_t ldw, int64_t ncols) {
  if (!p || !x_dev || !scales || !xhat_dev || !W_dev) return fail(CWT_EINVAL, "NULL argument");
  if (nbatch < 1 || nrows < 1 || int64_t(nbatch) * nrows > p->max_rows)
    return fail(CWT_EINVAL, "need nbatch*nrows <= max_rows");
  if (n0 < 1 || n0 > p->N || x_ld < n0) return fail(CWT_EINVAL, "need 1 <= n0 <= nfft and x_ld >= n0");
  if (ncols < 1 || ncols > p->N || ldw < ncols) return fail(CWT_EINVAL, "need 1 <= ncols <= nfft and ldw >= ncols");
  if (!(dt > 0) || !std::isfinite(dt)) return fail(CWT_EINVAL, "dt must be positive");
  HIPCHECK(hipSetDevice(p->device));
  const int total = nbatch * nrows;
  const std::vector<double> key = call_key(3, {double(mother), param, dt, double(nbatch), double(nrows), double(ncols)},
                                           {{scales, nrows}});
  if (!select_table(p, key)) {
    double cre, cim;
    int rc = mother_constant(mother, param, &cre, &cim);
    if (rc) return rc;
    const double w1 = 2.0 * 3.14159265358979323846 * (1.0 / (double(p->N) * dt));
    std::vector<double> a(total), ar(total), ai(total);
    for (int j = 0; j < total; ++j) {
      const double s = scales[j % nrows];
      if (!(s > 0) || !std::isfinite(s)) return fail(CWT_EINVAL, "scales must be positive and finite");
      a[j] = s * w1;
      const double norm = std::sqrt(s * w1 * double(p->N));
      ar[j] = norm * cre;
      ai[j] = norm * cim;
    }
    // as cwt_transform_rows_batch, with the signals at hand: time-compact rows may take the overlap-save form
    rc = build_row_table(p, mother, param, a.data(), ar.data(), ai.data(), p->N, total, nullptr, nullptr, nrows, -1, ncols, ncols);
    if (!rc) rc = upload_row_table(p, key);
    if (!rc && p->rt->n_ols)
      rc = p->prec == 64 ? fill_ols_tables<double>(p, mother_of(mother, param)) : fill_ols_tables<float>(p, mother_of(mother, param));
    if (!rc && p->rt->n_aols)
      rc = p->prec == 64 ? fill_aols_tables<double>(p, mother_of(mother, param)) : fill_aols_tables<float>(p, mother_of(mother, param));
    if (rc) { p->rt->key.clear(); return rc; }
  }
  set_split(p);
  const Mother mo = mother_of(mother, param);
  int rc = p->prec == 64 ? fft_rows_impl<double, IN_REAL>(p, x_dev, x_ld, nbatch, n0, xhat_dev)
                         : fft_rows_impl<float, IN_REAL>(p, x_dev, x_ld, nbatch, n0, xhat_dev);
  if (rc) return rc;
  p->ols_launched = 0;
  p->ols_x_ld = x_ld;
  rc = p->prec == 64 ? rows_impl<double>(p, xhat_dev, mo, total, W_dev, ldw, ncols, x_dev, n0)
                     : rows_impl<float>(p, xhat_dev, mo, total, W_dev, ldw, ncols, x_dev, n0);
  p->ols_x_ld = 0;
  return rc;
}

int cwt_transform_rows_table(cwt_plan* p, const void* xhat_dev, const void* table_dev, const int* k_lo,
                             const int* nband, int nrows, void* W_dev, int64_t ldw, int64_t ncols) {
  if (!p || !xhat_dev || !table_dev || !k_lo || !nband || !W_dev) return fail(CWT_EINVAL, "NULL argument");
  if (nrows < 1 || nrows > p->max_rows) return fail(CWT_EINVAL, "nrows must be in [1, max_rows]");
  if (ncols < 1 || ncols > p->N || ldw < ncols) return fail(CWT_EINVAL, "need 1 <= ncols <= nfft and ldw >= ncols");
  HIPCHECK(hipSetDevice(p->device));
  select_table(p, {});                                   // explicit filter banks are not cached
  std::vector<double> one(nrows, 1.0), zero(nrows, 0.0);
  int rc = build_row_table(p, MOTHER_TABLE, 0.0, one.data(), one.data(), zero.data(), 0, nrows, k_lo, nband);
  if (!rc) rc = upload_row_table(p, {});
  if (rc) return rc;
  set_split(p);
  Mother mo;
  mo.kind = MOTHER_TABLE; mo.m = 0; mo.p = 0; mo.table = table_dev;
  return p->prec == 64 ? rows_impl<double>(p, xhat_dev, mo, nrows, W_dev, ldw, ncols)
                       : rows_impl<float>(p, xhat_dev, mo, nrows, W_dev, ldw, ncols);
}

int cwt_fft_rows(cwt_plan* p, const void* in_dev, int in_complex, int nrows, int64_t in_ld, int64_t ncols_in,
                 void* spec_dev) {
  if (!p || !in_dev || !spec_dev) return fail(CWT_EINVAL, "NULL argument");
  if (nrows < 1) return fail(CWT_EINVAL, "nrows must be >= 1");
  if (ncols_in < 1 || ncols_in > p->N || in_ld < ncols_in) return fail(CWT_EINVAL, "need 1 <= ncols_in <= nfft and in_ld >= ncols_in");
  HIPCHECK(hipSetDevice(p->device));
  if (p->prec == 64)
    return in_complex ? fft_rows_impl<double, IN_CPLX>(p, in_dev, in_ld, nrows, ncols_in, spec_dev)
                      : fft_rows_impl<double, IN_REAL>(p, in_dev, in_ld, nrows, ncols_in, spec_dev);
  return in_complex ? fft_rows_impl<float, IN_CPLX>(p, in_dev, in_ld, nrows, ncols_in, spec_dev)
                    : fft_rows_impl<float, IN_REAL>(p, in_dev, in_ld, nrows, ncols_in, spec_dev);
}

int cwt_filter_rows(cwt_plan* p, const void* spec_dev, int64_t spec_ld, int mother, double param,
                    const double* a, const double* amp_re, const double* amp_im, int nrows, void* W_dev,
                    int64_t ldw, int64_t ncols) {
  if (!p || !spec_dev || !a || !amp_re || !amp_im || !W_dev) return fail(CWT_EINVAL, "NULL argument");
  if (nrows < 1 || nrows > p->max_rows) return fail(CWT_EINVAL, "nrows must be in [1, max_rows]");
  if (ncols < 1 || ncols > p->N || ldw < ncols) return fail(CWT_EINVAL, "need 1 <= ncols <= nfft and ldw >= ncols");
  if (spec_ld != 0 && spec_ld < p->N) return fail(CWT_EINVAL, "spec_ld must be 0 (shared) or >= nfft");
  HIPCHECK(hipSetDevice(p->device));
  const std::vector<double> key = call_key(1, {double(mother), param, double(spec_ld), double(nrows)},
                                           {{a, nrows}, {amp_re, nrows}, {amp_im, nrows}});
  if (!select_table(p, key)) {
    double cre, cim;
    int rc = mother_constant(mother, param, &cre, &cim);   // validates mother / order only
    if (!rc) rc = build_row_table(p, mother, param, a, amp_re, amp_im, spec_ld, nrows);
    if (!rc) rc = upload_row_table(p, key);
    if (rc) return rc;
  }
  set_split(p);
  Mother mo;
  mo.kind = mother; mo.m = int(std::lround(param)); mo.p = param; mo.table = nullptr;
  return p->prec == 64 ? rows_impl<double>(p, spec_dev, mo, nrows, W_dev, ldw, ncols)
                       : rows_impl<float>(p, spec_dev, mo, nrows, W_dev, ldw, ncols);
}

extern "C++" {
namespace {
template <typename T>
int upload_reals(cwt_plan* p, const double* v, int n) {          // -> p->weights_dev as T[n]
  // two staging buffers used in turn; the only wait is for the copy that left this buffer two calls ago
  const int i = p->weights_turn;
  p->weights_turn ^= 1;
  HIPCHECK(hipEventSynchronize(p->weights_ev[i]));
  for (int j = 0; j < n; ++j) {
    if (sizeof(T) == 8) static_cast<double*>(p->weights_pinned[i])[j] = v[j];
    else static_cast<float*>(p->weights_pinned[i])[j] = float(v[j]);
  }
  HIPCHECK(hipMemcpyAsync(p->weights_dev, p->weights_pinned[i], size_t(n) * sizeof(T), hipMemcpyHostToDevice, p->stream));
  HIPCHECK(hipEventRecord(p->weights_ev[i], p->stream));
  return CWT_OK;
}

template <typename T>
int wct_products_impl(cwt_plan* p, const void* W1, const void* W2, const double* scales, int nrows, int64_t ld,
                      int64_t ncols, void* P, void* C, void* A) {
  std::vector<double> inv(nrows);
  for (int j = 0; j < nrows; ++j) inv[j] = 1.0 / scales[j];
  int rc = upload_reals<T>(p, inv.data(), nrows);
  if (rc) return rc;
  return timed_launch(p, KC_ELEMENTWISE, [&] {
    hipLaunchKernelGGL((k_wct_products<T>), dim3(unsigned((ncols + 255) / 256), nrows), dim3(256), 0, p->stream,
                       static_cast<const cplx<T>*>(W1), static_cast<const cplx<T>*>(W2),
                       static_cast<const T*>(p->weights_dev), long(ld), long(ncols), static_cast<cplx<T>*>(P),
                       static_cast<cplx<T>*>(C), static_cast<T*>(A));
  });
}

template <typename T>
int boxcar_impl(cwt_plan* p, const void* in, int nrows, int64_t ld, int64_t ncols, const double* win, int nwin,
                void* out) {
  int rc = upload_reals<T>(p, win, nwin);
  if (rc) return rc;
  const size_t ring_bytes = size_t(nwin) * 256 * sizeof(cplx<T>);
  if (nwin > 1 && ring_bytes <= 64 * 1024) {       // sliding window over 32-row strips (see the kernel)
    const int RB = 32;
    return timed_launch(p, KC_ELEMENTWISE, [&] {
      hipLaunchKernelGGL((k_boxcar_scales_ring<T>), dim3(unsigned((ncols + 255) / 256), unsigned((nrows + RB - 1) / RB)),
                         dim3(256), ring_bytes, p->stream, static_cast<const cplx<T>*>(in), nrows, long(ld),
                         long(ncols), static_cast<const T*>(p->weights_dev), nwin, static_cast<cplx<T>*>(out), RB);
    });
  }
  return timed_launch(p, KC_ELEMENTWISE, [&] {
    hipLaunchKernelGGL((k_boxcar_scales<T>), dim3(unsigned((ncols + 255) / 256), nrows), dim3(256), 0, p->stream,
                       static_cast<const cplx<T>*>(in), nrows, long(ld), long(ncols),
                       static_cast<const T*>(p->weights_dev), nwin, static_cast<cplx<T>*>(out));
  });
}

template <typename T>
int coherence_impl(cwt_plan* p, const void* S, const void* S12, int nrows, int64_t ld, int64_t ncols, void* out) {
  return timed_launch(p, KC_ELEMENTWISE, [&] {
    hipLaunchKernelGGL((k_wct_coherence<T>), dim3(unsigned((ncols + 255) / 256), nrows), dim3(256), 0, p->stream,
                       static_cast<const cplx<T>*>(S), static_cast<const cplx<T>*>(S12), long(ld), long(ncols),
                       static_cast<T*>(out));
  });
}
}  // namespace
}  // extern "C++"

int cwt_wct_products(cwt_plan* p, const void* W1_dev, const void* W2_dev, const double* scales, int nrows,
                     int64_t ld, int64_t ncols, void* P_dev, void* C_dev, void* angle_dev) {
  if (!p || !W1_dev || !W2_dev || !scales || !P_dev || !C_dev || !angle_dev) return fail(CWT_EINVAL, "NULL argument");
  if (nrows < 1 || nrows > p->max_rows || ncols < 1 || ld < ncols) return fail(CWT_EINVAL, "bad shape");
  HIPCHECK(hipSetDevice(p->device));
  return p->prec == 64 ? wct_products_impl<double>(p, W1_dev, W2_dev, scales, nrows, ld, ncols, P_dev, C_dev, angle_dev)
                       : wct_products_impl<float>(p, W1_dev, W2_dev, scales, nrows, ld, ncols, P_dev, C_dev, angle_dev);
}

int cwt_cross_spectrum(cwt_plan* p, const void* W1_dev, const void* W2_dev, int nrows, int64_t ld, int64_t ncols,
                       void* out_dev) {
  if (!p || !W1_dev || !W2_dev || !out_dev) return fail(CWT_EINVAL, "NULL argument");
  if (nrows < 1 || nrows > 65535 || ncols < 1 || ld < ncols) return fail(CWT_EINVAL, "bad shape");
  HIPCHECK(hipSetDevice(p->device));
  const dim3 grid(unsigned((ncols + 255) / 256), unsigned(nrows));
  return timed_launch(p, KC_ELEMENTWISE, [&] {
    if (p->prec == 64)
      hipLaunchKernelGGL((k_cross_spectrum<double>), grid, dim3(256), 0, p->stream, static_cast<const double2*>(W1_dev),
                         static_cast<const double2*>(W2_dev), long(ld), long(ncols), static_cast<double2*>(out_dev));
    else
      hipLaunchKernelGGL((k_cross_spectrum<float>), grid, dim3(256), 0, p->stream, static_cast<const float2*>(W1_dev),
                         static_cast<const float2*>(W2_dev), long(ld), long(ncols), static_cast<float2*>(out_dev));
  });
}

int cwt_boxcar_scales(cwt_plan* p, const void* in_dev, int nrows, int64_t ld, int64_t ncols, const double* win,
                      int nwin, void* out_dev) {
  if (!p || !in_dev || !win || !out_dev) return fail(CWT_EINVAL, "NULL argument");
  if (nrows < 1 || ncols < 1 || ld < ncols || nwin < 1 || nwin > p->max_rows) return fail(CWT_EINVAL, "bad shape");
  if (in_dev == out_dev) return fail(CWT_EINVAL, "boxcar cannot run in place");
  HIPCHECK(hipSetDevice(p->device));
  return p->prec == 64 ? boxcar_impl<double>(p, in_dev, nrows, ld, ncols, win, nwin, out_dev)
                       : boxcar_impl<float>(p, in_dev, nrows, ld, ncols, win, nwin, out_dev);
}

int cwt_wct_coherence(cwt_plan* p, const void* S_dev, const void* S12_dev, int nrows, int64_t ld, int64_t ncols,
                      void* out_dev) {
  if (!p || !S_dev || !S12_dev || !out_dev) return fail(CWT_EINVAL, "NULL argument");
  if (nrows < 1 || ncols < 1 || ld < ncols) return fail(CWT_EINVAL, "bad shape");
  HIPCHECK(hipSetDevice(p->device));
  return p->prec == 64 ? coherence_impl<double>(p, S_dev, S12_dev, nrows, ld, ncols, out_dev)
                       : coherence_impl<float>(p, S_dev, S12_dev, nrows, ld, ncols, out_dev);
}

extern "C++" {
namespace {
template <typename T, bool POWER>
int reduce_scales_impl(cwt_plan* p, const void* W_dev, int64_t ldw, int64_t ncols, int nrows,
                       const double* weights, double coeff, void* out_dev) {
  int rc = upload_reals<T>(p, weights, nrows);
  if (rc) return rc;
  const unsigned blocks = unsigned((ncols + 255) / 256);
  return timed_launch(p, KC_ICWT, [&] {
    hipLaunchKernelGGL((k_icwt<T, POWER>), dim3(blocks), dim3(256), 0, p->stream,
                       static_cast<const cplx<T>*>(W_dev), long(ldw), long(ncols), nrows,
                       static_cast<const T*>(p->weights_dev), T(coeff), static_cast<T*>(out_dev));
  });
}
}  // namespace
}  // extern "C++"

int cwt_reduce_scales(cwt_plan* p, const void* W_dev, int64_t ldw, int64_t ncols, int nrows,
                      const double* weights, int power, double coeff, void* out_dev) {
  if (!p || !W_dev || !weights || !out_dev) return fail(CWT_EINVAL, "NULL argument");
  if (nrows < 1 || nrows > p->max_rows) return fail(CWT_EINVAL, "nrows must be in [1, max_rows]");
  if (ncols < 1 || ldw < ncols) return fail(CWT_EINVAL, "need ncols >= 1 and ldw >= ncols");
  HIPCHECK(hipSetDevice(p->device));
  if (p->prec == 64)
    return power ? reduce_scales_impl<double, true>(p, W_dev, ldw, ncols, nrows, weights, coeff, out_dev)
                 : reduce_scales_impl<double, false>(p, W_dev, ldw, ncols, nrows, weights, coeff, out_dev);
  return power ? reduce_scales_impl<float, true>(p, W_dev, ldw, ncols, nrows, weights, coeff, out_dev)
               : reduce_scales_impl<float, false>(p, W_dev, ldw, ncols, nrows, weights, coeff, out_dev);
}

int cwt_icwt_reduce(cwt_plan* p, const void* W_dev, int64_t ldw, int64_t ncols, int nrows,
                    const double* scales, double coeff, void* out_dev) {
  if (!p || !scales) return fail(CWT_EINVAL, "NULL argument");
  if (nrows < 1 || nrows > p->max_rows) return fail(CWT_EINVAL, "nrows must be in [1, max_rows]");
  std::vector<double> w(nrows);
  for (int j = 0; j < nrows; ++j) {
    if (!(scales[j] > 0)) return fail(CWT_EINVAL, "scales must be positive");
    w[j] = 1.0 / std::sqrt(scales[j]);
  }
  return cwt_reduce_scales(p, W_dev, ldw, ncols, nrows, w.data(), 0, coeff, out_dev);
}

int cwt_time_mean_power(cwt_plan* p, const void* W_dev, int64_t ldw, int64_t ncols, int nrows, void* out_dev) {
  if (!p || !W_dev || !out_dev) return fail(CWT_EINVAL, "NULL argument");
  if (nrows < 1 || ncols < 1 || ldw < ncols) return fail(CWT_EINVAL, "bad shape");
  HIPCHECK(hipSetDevice(p->device));
  if (p->prec == 64)
    return timed_launch(p, KC_ICWT, [&] {
      hipLaunchKernelGGL((k_time_mean<double>), dim3(nrows), dim3(256), 256 * sizeof(double), p->stream,
                         static_cast<const double2*>(W_dev), long(ldw), long(ncols), static_cast<double*>(out_dev));
    });
  return timed_launch(p, KC_ICWT, [&] {
    hipLaunchKernelGGL((k_time_mean<float>), dim3(nrows), dim3(256), 256 * sizeof(double), p->stream,
                       static_cast<const float2*>(W_dev), long(ldw), long(ncols), static_cast<float*>(out_dev));
  });
}

int cwt_coherence_histogram(cwt_plan* p, const void* r2_dev, int64_t ld, int nrows, const int64_t* lo_dev,
                            const int64_t* hi_dev, int64_t max_span, int nbins, uint64_t* hist_dev) {
  if (!p || !r2_dev || !lo_dev || !hi_dev || !hist_dev) return fail(CWT_EINVAL, "NULL argument");
  if (nrows < 1 || ld < 1 || nbins < 1 || nbins > 16384 || max_span < 0) return fail(CWT_EINVAL, "bad shape");
  if (max_span == 0) return CWT_OK;
  HIPCHECK(hipSetDevice(p->device));
  static_assert(sizeof(long) == sizeof(int64_t) && sizeof(unsigned long long) == sizeof(uint64_t), "LP64 expected");
  const unsigned gx = unsigned(std::min<int64_t>(512, (max_span + 4095) / 4096));   // >= 16 columns per thread
  const size_t lds = size_t(nbins) * sizeof(unsigned);
  return timed_launch(p, KC_ELEMENTWISE, [&] {
    if (p->prec == 64)
      hipLaunchKernelGGL((k_coherence_hist<double>), dim3(gx, nrows), dim3(256), lds, p->stream,
                         static_cast<const double*>(r2_dev), long(ld), reinterpret_cast<const long*>(lo_dev),
                         reinterpret_cast<const long*>(hi_dev), nbins, reinterpret_cast<unsigned long long*>(hist_dev));
    else
      hipLaunchKernelGGL((k_coherence_hist<float>), dim3(gx, nrows), dim3(256), lds, p->stream,
                         static_cast<const float*>(r2_dev), long(ld), reinterpret_cast<const long*>(lo_dev),
                         reinterpret_cast<const long*>(hi_dev), nbins, reinterpret_cast<unsigned long long*>(hist_dev));
  });
}

extern "C++" {
namespace {
// Chirp-kernel spectra for length n0 on this plan (N = M >= 2 n0 - 1), cached per n0.
template <typename T>
int bluestein_prepare(cwt_plan* p, int64_t n0) {
  if (n0 < 1 || 2 * n0 - 1 > p->N) return fail(CWT_EINVAL, "this plan's nfft must be >= 2*n0 - 1 for a length-n0 transform");
  if (p->bs_n0 == n0) return CWT_OK;
  p->bs_n0 = 0;
  const size_t bytes = size_t(p->N) * sizeof(cplx<T>);
  int rc = grow(&p->bs_a, &p->bs_a_bytes, bytes, p->stream);      // staging for the kernel in the time domain
  for (int i = 0; i < 2 && !rc; ++i) {
    if (!p->bs_khat[i] && hipMalloc(&p->bs_khat[i], bytes) != hipSuccess) return fail(CWT_ENOMEM, "chirp table allocation failed");
    const unsigned blocks = unsigned((p->N + 255) / 256);
    hipLaunchKernelGGL((k_chirp_kernel<T>), dim3(blocks), dim3(256), 0, p->stream, long(n0), long(p->N), i == 0 ? +1 : -1,
                       static_cast<cplx<T>*>(p->bs_a));
    HIPCHECK(hipGetLastError());
    rc = fft_rows_impl<T, IN_CPLX>(p, p->bs_a, p->N, 1, p->N, p->bs_khat[i]);
  }
  if (rc) return rc;
  p->bs_n0 = n0;
  return CWT_OK;
}

// out[j, 0..n0) = IFFT_M( spec[j, :] * khat[which] ), rows x ldo; the table inverse of the engine with one shared table
template <typename T>
int bluestein_convolve(cwt_plan* p, const void* spec, int nrows, int which, void* out, int64_t ldo, int64_t n0) {
  select_table(p, {});
  std::vector<double> one(nrows, 1.0), zero(nrows, 0.0);
  std::vector<int> klo(nrows, int(-(p->N / 2))), nb(nrows, int(p->N));
  int rc = build_row_table(p, MOTHER_TABLE, 0.0, one.data(), one.data(), zero.data(), p->N, nrows, klo.data(), nb.data(),
                           0, 0);
  if (!rc) rc = upload_row_table(p, {});
  if (rc) return rc;
  set_split(p);
  Mother mo;
  mo.kind = MOTHER_TABLE; mo.m = 0; mo.p = 0; mo.table = p->bs_khat[which];
  return rows_impl<T>(p, spec, mo, nrows, out, ldo, n0);
}

template <typename T>
int forward_fft_n_impl(cwt_plan* p, const void* x_dev, int64_t n0, void* xhat_dev) {
  int rc = bluestein_prepare<T>(p, n0);
  if (!rc) rc = grow(&p->bs_a, &p->bs_a_bytes, size_t(p->N) * sizeof(cplx<T>), p->stream);
  if (!rc) rc = grow(&p->bs_spec, &p->bs_spec_bytes, size_t(p->N) * sizeof(cplx<T>), p->stream);
  if (rc) return rc;
  const dim3 grid(unsigned((n0 + 255) / 256), 1);
  // a[n] = x[n] conj(c[n]);  xhat[k] = conj(c[k]) * (a conv c)[k]
  hipLaunchKernelGGL((k_chirp_mul<T, IN_REAL>), grid, dim3(256), 0, p->stream, x_dev, long(n0), long(n0), -1, 1.0,
                     static_cast<cplx<T>*>(p->bs_a), long(n0));
  HIPCHECK(hipGetLastError());
  rc = fft_rows_impl<T, IN_CPLX>(p, p->bs_a, n0, 1, n0, p->bs_spec);
  if (!rc) rc = bluestein_convolve<T>(p, p->bs_spec, 1, 0, xhat_dev, n0, n0);
  if (rc) return rc;
  hipLaunchKernelGGL((k_chirp_mul<T, IN_CPLX>), grid, dim3(256), 0, p->stream, xhat_dev, long(n0), long(n0), -1, 1.0,
                     static_cast<cplx<T>*>(xhat_dev), long(n0));
  HIPCHECK(hipGetLastError());
  return CWT_OK;
}

template <typename T>
int transform_rows_n_impl(cwt_plan* p, const void* xhat_dev, int64_t n0, int mother, double param, double dt,
                          const double* scales, int nrows, void* W_dev, int64_t ldw) {
  double cre, cim;
  int rc = mother_constant(mother, param, &cre, &cim);
  if (!rc) rc = bluestein_prepare<T>(p, n0);
  if (rc) return rc;
  const double w1 = 2.0 * 3.14159265358979323846 * (1.0 / (double(n0) * dt));    // ftfreqs[1] at length n0
  std::vector<double> par(size_t(3) * nrows);
  for (int j = 0; j < nrows; ++j) {
    if (!(scales[j] > 0) || !std::isfinite(scales[j])) return fail(CWT_EINVAL, "scales must be positive and finite");
    const double norm = std::sqrt(scales[j] * w1 * double(n0));                   // wavelet.py:102
    par[j] = scales[j] * w1;
    par[nrows + j] = norm * cre;
    par[2 * size_t(nrows) + j] = norm * cim;
  }
  rc = grow(&p->bs_par, &p->bs_par_bytes, par.size() * sizeof(double), p->stream);
  if (rc) return rc;
  HIPCHECK(hipMemcpyAsync(p->bs_par, par.data(), par.size() * sizeof(double), hipMemcpyHostToDevice, p->stream));
  HIPCHECK(hipStreamSynchronize(p->stream));            // `par` is pageable and dies with this frame
  const double* dpar = static_cast<const double*>(p->bs_par);
  const int slab = int(std::max<size_t>(1, std::min<size_t>(size_t(std::min(nrows, p->max_rows)),
                                                           (size_t(1) << 31) / (size_t(p->N) * sizeof(cplx<T>)))));
  rc = grow(&p->bs_a, &p->bs_a_bytes, std::max(size_t(p->N), size_t(slab) * size_t(n0)) * sizeof(cplx<T>), p->stream);
  if (!rc) rc = grow(&p->bs_spec, &p->bs_spec_bytes, size_t(slab) * size_t(p->N) * sizeof(cplx<T>), p->stream);
  if (rc) return rc;
  Mother mo;
  mo.kind = mother; mo.m = int(std::lround(param)); mo.p = param; mo.table = nullptr;
  for (int first = 0; first < nrows; first += slab) {
    const int cnt = std::min(slab, nrows - first);
    const dim3 grid(unsigned((n0 + 255) / 256), unsigned(cnt));
    hipLaunchKernelGGL((k_bluestein_band<T>), grid, dim3(256), 0, p->stream, static_cast<const cplx<T>*>(xhat_dev),
                       dpar + first, dpar + nrows + first, dpar + 2 * size_t(nrows) + first, mo, long(n0),
                       static_cast<cplx<T>*>(p->bs_a), long(n0));
    HIPCHECK(hipGetLastError());
    rc = fft_rows_impl<T, IN_CPLX>(p, p->bs_a, n0, cnt, n0, p->bs_spec);
    cplx<T>* Wslab = static_cast<cplx<T>*>(W_dev) + size_t(first) * size_t(ldw);
    if (!rc) rc = bluestein_convolve<T>(p, p->bs_spec, cnt, 1, Wslab, ldw, n0);
    if (rc) return rc;
    // W[j, n] = c[n] / n0 * conv[n]
    hipLaunchKernelGGL((k_chirp_mul<T, IN_CPLX>), grid, dim3(256), 0, p->stream, static_cast<const void*>(Wslab), long(ldw),
                       long(n0), +1, 1.0 / double(n0), Wslab, long(ldw));
    HIPCHECK(hipGetLastError());
  }
  return CWT_OK;
}
}  // namespace
}  // extern "C++"

int cwt_forward_fft_n(cwt_plan* p, const void* x_dev, int64_t n0, void* xhat_dev) {
  if (!p || !x_dev || !xhat_dev) return fail(CWT_EINVAL, "NULL argument");
  HIPCHECK(hipSetDevice(p->device));
  return p->prec == 64 ? forward_fft_n_impl<double>(p, x_dev, n0, xhat_dev) : forward_fft_n_impl<float>(p, x_dev, n0, xhat_dev);
}

int cwt_transform_rows_n(cwt_plan* p, const void* xhat_dev, int64_t n0, int mother, double param, double dt,
                         const double* scales, int nrows, void* W_dev, int64_t ldw) {
  if (!p || !xhat_dev || !scales || !W_dev) return fail(CWT_EINVAL, "NULL argument");
  if (nrows < 1) return fail(CWT_EINVAL, "nrows must be >= 1");
  if (ldw < n0) return fail(CWT_EINVAL, "ldw must be >= n0");
  if (!(dt > 0) || !std::isfinite(dt)) return fail(CWT_EINVAL, "dt must be positive");
  if (mother < MOTHER_MORLET || mother > MOTHER_DOG) return fail(CWT_EINVAL, "unknown mother id");
  HIPCHECK(hipSetDevice(p->device));
  return p->prec == 64 ? transform_rows_n_impl<double>(p, xhat_dev, n0, mother, param, dt, scales, nrows, W_dev, ldw)
                       : transform_rows_n_impl<float>(p, xhat_dev, n0, mother, param, dt, scales, nrows, W_dev, ldw);
}

// Page-locked host buffers handed out by cwt_host_malloc (start -> bytes): cwt_execute_host lets the kernels of a short
// transform write W straight into such a buffer.
extern "C++" {
namespace {
std::mutex g_pinned_mutex;
std::map<uintptr_t, size_t> g_pinned;

bool is_pinned(const void* ptr, size_t bytes) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
  std::lock_guard<std::mutex> lock(g_pinned_mutex);
  auto it = g_pinned.upper_bound(a);
  if (it == g_pinned.begin()) return false;
  --it;
  return a >= it->first && a + bytes <= it->first + it->second;
}
}  // namespace
}  // extern "C++"

int cwt_host_malloc(void** ptr_host, size_t bytes) {
  if (!ptr_host || !bytes) return fail(CWT_EINVAL, "NULL argument or zero size");
  void* q = nullptr;
  // (portable + mapped: a buffer serves the plans of every device of the process, whichever was current when it was made)
  if (hipHostMalloc(&q, bytes, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return fail(CWT_ENOMEM, "page-locked allocation failed"); }
  { std::lock_guard<std::mutex> lock(g_pinned_mutex); g_pinned[reinterpret_cast<uintptr_t>(q)] = bytes; }
  *ptr_host = q;
  return CWT_OK;
}

int cwt_host_free(void* ptr_host) {
  if (!ptr_host) return CWT_OK;
  {
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    auto it = g_pinned.find(reinterpret_cast<uintptr_t>(ptr_host));
    if (it == g_pinned.end()) return fail(CWT_EINVAL, "not a cwt_host_malloc buffer");
    g_pinned.erase(it);
  }
  HIPCHECK(hipHostFree(ptr_host));
  return CWT_OK;
}

int cwt_execute_host(cwt_plan* p, const void* x_host, int64_t n0, int mother, double param, double dt,
                     const double* scales, int nrows, void* W_host, void* xhat_host) {
  if (!p || !x_host || !scales) return fail(CWT_EINVAL, "NULL argument");
  if (n0 < 1 || n0 > p->N) return fail(CWT_EINVAL, "n0 must be in [1, nfft]");
  HIPCHECK(hipSetDevice(p->device));
  const size_t es = p->esize();
  // A transform that fits one workgroup per row (the reference's canonical 504-point call: 4 KB in, 0.8 MB out) is all
  // latency, and copy operations are the larger part of it.  Here it has none: the forward FFT reads the signal from the
  // plan's page-locked staging buffer, the row kernel writes W over PCIe itself -- into W_host when that is a
  // cwt_host_malloc buffer, else into the staging buffer (then one memcpy) -- and only the spectrum (nfft values) is
  // copied.  45 us against 83 at 504 x 97, fp64 [measured, profiles/r04_latency.txt; tools/microbench/host_latency.cpp].
  if (W_host && p->host_direct && !p->profile && p->logN > 3 && p->logN <= p->loglmax) {
    const size_t in_b = (size_t(n0) * es + 255) & ~size_t(255), xh_b = size_t(p->N) * 2 * es;
    const size_t w_b = size_t(nrows) * size_t(n0) * 2 * es;
    const bool w_direct = is_pinned(W_host, w_b);
    if (in_b + xh_b + (w_direct ? 0 : w_b) <= (size_t(4) << 20)) {
      if (p->hstage_bytes < (size_t(4) << 20)) {
        if (hipHostMalloc(&p->hstage, size_t(4) << 20) != hipSuccess) return fail(CWT_ENOMEM, "pinned staging allocation failed");
        p->hstage_bytes = size_t(4) << 20;
      }
      if (p->auto_target > 0) {                            // (round-off costs such transforms nothing: no need to look)
        const double floor_tol = p->prec == 64 ? kDefaultTolerance64 : kDefaultTolerance32;
        if (floor_tol != p->tolerance) { for (auto& t : p->slots) t.key.clear(); p->tolerance = floor_tol; }
      }
      int rc = grow(&p->hxhat, &p->hxhat_bytes, xh_b, p->stream);
      if (rc) return rc;
      char* stage = static_cast<char*>(p->hstage);
      std::memcpy(stage, x_host, size_t(n0) * es);
      void* W_out = w_direct ? W_host : stage + in_b + xh_b;
      rc = cwt_transform(p, stage, n0, mother, param, dt, scales, nrows, p->hxhat, W_out, n0, n0);
      if (rc) return rc;
      if (xhat_host) HIPCHECK(hipMemcpyAsync(stage + in_b, p->hxhat, xh_b, hipMemcpyDeviceToHost, p->stream));
      HIPCHECK(hipStreamSynchronize(p->stream));
      if (xhat_host) std::memcpy(xhat_host, stage + in_b, xh_b);
      if (!w_direct) std::memcpy(W_host, W_out, w_b);
      return CWT_OK;
    }
  }
  int rc = grow(&p->hx, &p->hx_bytes, size_t(n0) * es, p->stream);
  if (!rc) rc = grow(&p->hxhat, &p->hxhat_bytes, size_t(p->N) * 2 * es, p->stream);
  if (!rc && W_host) rc = grow(&p->hW, &p->hW_bytes, size_t(nrows) * size_t(n0) * 2 * es, p->stream);
  if (rc) return rc;
  // Small calls (the reference's canonical 504-point series: 4 KB in, 0.7 MB out) are all latency: a copy to or from pageable
  // memory makes the runtime stage and synchronise on its own, once per copy.  They go through ONE page-locked buffer of the
  // plan instead -- memcpy in, three asynchronous copies, one synchronisation, memcpy out.
  const size_t in_b = size_t(n0) * es, xh_b = xhat_host ? size_t(p->N) * 2 * es : 0;
  const size_t w_b = W_host ? size_t(nrows) * size_t(n0) * 2 * es : 0;
  const bool staged = in_b + xh_b + w_b <= (size_t(4) << 20);
  char* stage = nullptr;
  if (staged) {
    if (p->hstage_bytes < (size_t(4) << 20)) {
      if (hipHostMalloc(&p->hstage, size_t(4) << 20) != hipSuccess) return fail(CWT_ENOMEM, "pinned staging allocation failed");
      p->hstage_bytes = size_t(4) << 20;
    }
    stage = static_cast<char*>(p->hstage);
    std::memcpy(stage, x_host, in_b);
    HIPCHECK(hipMemcpyAsync(p->hx, stage, in_b, hipMemcpyHostToDevice, p->stream));
  } else {
    HIPCHECK(hipMemcpyAsync(p->hx, x_host, in_b, hipMemcpyHostToDevice, p->stream));
  }
  if (W_host && p->auto_target > 0 && p->logN <= p->loglmax) {
    // single-workgroup transforms compute every bin of every row anyway: round-off costs nothing, no need to look
    const double floor_tol = p->prec == 64 ? kDefaultTolerance64 : kDefaultTolerance32;
    if (floor_tol != p->tolerance) { for (auto& t : p->slots) t.key.clear(); p->tolerance = floor_tol; }
  } else if (W_host && p->auto_target > 0) {
    // accuracy target of THIS call = auto_target / (dynamic range of its spectrum relative to white noise), a power of
    // ten (so that calls with like spectra share one cached row table), never looser than the target itself
    rc = cwt_forward_fft(p, p->hx, n0, p->hxhat);
    double tol = 0;
    if (!rc) rc = cwt_plan_auto_tolerance(p, p->hxhat, p->auto_target, &tol);
    if (rc) return rc;
    if (tol != p->tolerance) { for (auto& t : p->slots) t.key.clear(); p->tolerance = tol; }
  }
  if (W_host) rc = cwt_transform(p, p->hx, n0, mother, param, dt, scales, nrows, p->hxhat, p->hW, n0, n0);
  else rc = cwt_forward_fft(p, p->hx, n0, p->hxhat);
  if (rc) return rc;
  if (staged) {
    if (xh_b) HIPCHECK(hipMemcpyAsync(stage + in_b, p->hxhat, xh_b, hipMemcpyDeviceToHost, p->stream));
    if (w_b) HIPCHECK(hipMemcpyAsync(stage + in_b + xh_b, p->hW, w_b, hipMemcpyDeviceToHost, p->stream));
    HIPCHECK(hipStreamSynchronize(p->stream));
    if (xh_b) std::memcpy(xhat_host, stage + in_b, xh_b);
    if (w_b) std::memcpy(W_host, stage + in_b + xh_b, w_b);
    return CWT_OK;
  }
  if (xhat_host)
    HIPCHECK(hipMemcpyAsync(xhat_host, p->hxhat, size_t(p->N) * 2 * es, hipMemcpyDeviceToHost, p->stream));
  if (W_host) return copy_d2h(p, W_host, p->hW, size_t(nrows) * size_t(n0) * 2 * es);
  HIPCHECK(hipStreamSynchronize(p->stream));
  return CWT_OK;
}

int cwt_plan_timings(cwt_plan* p, int cap, const char** names, double* total_ms, int* launches, int* n) {
  if (!p || !n) return fail(CWT_EINVAL, "NULL argument");
  HIPCHECK(hipStreamSynchronize(p->stream));
  double tot[KC_COUNT] = {0};
  int cnt[KC_COUNT] = {0};
  for (auto& t : p->timed) {
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, t.a, t.b));
    tot[t.cls] += ms;
    cnt[t.cls]++;
    p->free_events.push_back(t.a);
    p->free_events.push_back(t.b);
  }
  p->timed.clear();
  int k = 0;
  for (int c = 0; c < KC_COUNT; ++c) {
    if (!cnt[c]) continue;
    if (k < cap) {
      if (names) names[k] = kClassNames[c];
      if (total_ms) total_ms[k] = tot[c];
      if (launches) launches[k] = cnt[c];
    }
    ++k;
  }
  *n = k;
  return CWT_OK;
}

int cwt_plan_row_classes(cwt_plan* p, int* codes, int cap, int* n) {
  if (!p || !n) return fail(CWT_EINVAL, "NULL argument");
  const int total = int(p->rt->table.size());
  const int n_aux = p->rt->aux_first >= 0 ? p->rt->aols_nbatch : 0;
  *n = total - n_aux;
  if (!codes) return CWT_OK;
  for (int i = 0; i < total; ++i) {
    const RowDesc& rd = p->rt->table[i];
    // 0 single-workgroup, 1 band-limited, 2 band-limited K = 2048, 3 two-pass, 4 overlap-save, 5 overlap-save on half-size tiles
    const int small_end = p->rt->ols_first + (p->rt->ols_grp[0].logp != p->rt->ols_grp[1].logp ? p->rt->ols_grp[0].nrows : 0);
    if (n_aux && i >= p->rt->aux_first && i < p->rt->aux_first + n_aux) continue;      // the mask pseudo-rows of the k_aols rows
    // ... 6 overlap-save on the band-passed complex signal (rows clipped at Nyquist)
    // 7 band-limited row in polynomial form (logK = log2 of its interval count, nterms = its degree)
    const int kind = i < p->rt->n_small ? 0 : i < p->rt->wide_first ? (rd.logK == 11 ? 2 : 1) : i < p->rt->ols_first ? 3 :
                     i >= p->rt->poly_first ? 7 : i >= p->rt->aols_first ? 6 : i < small_end ? 5 : 4;
    if (rd.out_row >= 0 && rd.out_row < cap) codes[rd.out_row] = kind * 10000 + rd.logK * 100 + rd.nterms;
  }
  return CWT_OK;
}

int cwt_plan_classify(cwt_plan* p, int mother, double param, double dt, const double* scales, int nrows, int64_t ncols,
                      int with_signal, int* codes) {
  if (!p || !scales || !codes) return fail(CWT_EINVAL, "NULL argument");
  HIPCHECK(hipSetDevice(p->device));
  int rc = prepare_rows_table(p, with_signal != 0, mother, param, dt, scales, nrows, ncols, ncols);
  if (rc) return rc;
  int n = 0;
  return cwt_plan_row_classes(p, codes, nrows, &n);
}

extern "C++" {
namespace {
// Cost model of one rank's step, microseconds at N = 2^20: per kernel class a fixed part (launch ramp and tail; for the
// overlap-save classes the block spectra of that tile size) + a per-row part.  Fitted to the per-class launch durations of
// bench.py on BASELINE configs 2 / 3 (profiles/r03_per_class.txt) and the per-rank runs of profiles/r03_shards.txt; the
// overlap-save, band-passed and polynomial terms of fp64 refitted (least squares) to the 15 per-rank runs of
// profiles/r04_shards.txt.
struct ShardCost { double fwd, tp_fixed, tp_row, k2048_fixed, k2048_row, ols_fixed, ols_row, olsh_fixed, olsh_row, nar_fixed, nar_row, nar_term,
                   aols_fixed, aols_row, poly_fixed, poly_row, poly_coef; };
constexpr ShardCost kShardCost64 = {30.0, 18.0, 9.8, 30.0, 6.1, 22.0, 3.25, 44.0, 3.2, 8.0, 2.85, 0.9, 32.0, 2.9, 12.5, 2.67, 2.7};
constexpr ShardCost kShardCost32 = {27.0, 14.0, 5.3, 8.0, 5.4, 28.0, 2.3, 20.0, 1.9, 4.0, 1.75, 0.55, 40.0, 2.3, 22.0, 1.4, 1.1};

// Estimated step time of a rank that owns rows [lo, hi) (codes as cwt_plan_row_classes reports them).  nscale = transform
// length / 2^20: per-row parts scale with it, per-launch parts do not; chunk = rows per two-pass launch pair.
double shard_cost(const int* codes, int lo, int hi, const ShardCost& c, double nscale, int chunk) {
  double total = 0;
  bool seen_tp = false, seen_big = false, seen_ols = false, seen_olsh = false, seen_nar = false, seen_aols = false, seen_poly = false;
  int n_tp = 0;
  for (int i = lo; i < hi; ++i) {
    const int kind = codes[i] / 10000, logk = (codes[i] / 100) % 100, terms = codes[i] % 100;
    if (kind == 3) { ++n_tp; if (!seen_tp) { seen_tp = true; total += c.tp_fixed; } total += c.tp_row * nscale; }
    else if (kind == 2) { if (!seen_big) { seen_big = true; total += c.k2048_fixed; } total += c.k2048_row * nscale; }
    else if (kind == 4) { if (!seen_ols) { seen_ols = true; total += c.ols_fixed; } total += c.ols_row * nscale; }
    else if (kind == 5) { if (!seen_olsh) { seen_olsh = true; total += c.olsh_fixed; } total += c.olsh_row * nscale; }
    else if (kind == 7) {
      // stage 2 per row (a little more per degree) + the row's share of stage 1: (D + 1) K' coefficients, priced at the
      // measured 2.2 us (fp64) of a K' = 16384, D = 8 row (profiles/r04_shards.txt)
      if (!seen_poly) { seen_poly = true; total += c.poly_fixed; }
      total += c.poly_row * nscale * (1.0 + 0.015 * std::max(0, terms - 8));
      total += c.poly_coef * double((terms + 1) << logk) / double(9 << 14);
    }
    else if (kind == 6) { if (!seen_aols) { seen_aols = true; total += c.aols_fixed * std::max(nscale, 0.5); } total += c.aols_row * nscale; }
    else {
      if (!seen_nar) { seen_nar = true; total += c.nar_fixed; }
      const double per = c.nar_row * nscale;
      total += per;
      if (kind == 1) {                      // longer transforms per residue, shorter store segments
        const int K = 1 << logk;
        total += per * (K >= 1024 ? 0.25 : K >= 512 ? 0.13 : K >= 32 ? 0.05 : -0.05);
        if (terms > 1) total += nscale * c.nar_term * (terms - 1);
      }
    }
  }
  if (n_tp > chunk) total += c.tp_fixed * ((n_tp - 1) / chunk);
  if (seen_tp || seen_big || seen_nar || seen_aols || seen_poly) total += c.fwd * std::max(nscale, 0.5);   // some row needs the spectrum
  return total;
}
}  // namespace
}  // extern "C++"

int cwt_shard_codes(const int* codes, int nrows, int precision, double nscale, int chunk_rows, int world, int* first,
                    int* count) {
  if (!codes || !first || !count) return fail(CWT_EINVAL, "NULL argument");
  if (nrows < 0 || world < 1 || (precision != 32 && precision != 64) || !(nscale > 0) || chunk_rows < 1)
    return fail(CWT_EINVAL, "bad shard arguments");
  const ShardCost& c = precision == 64 ? kShardCost64 : kShardCost32;
  const int n = nrows;
  std::vector<int> bounds(size_t(world) + 1, n);
  bounds[0] = 0;
  if (world > 1 && n > 0) {
    // largest shard minimised by bisection on the limit: greedy fill of contiguous shards (cost is monotone in hi)
    auto cuts_for = [&](double limit, std::vector<int>* out) {
      int lo = 0;
      for (int r = 0; r < world; ++r) {
        int a = lo, b = n;
        while (a < b) {
          const int m = (a + b + 1) / 2;
          if (shard_cost(codes, lo, m, c, nscale, chunk_rows) <= limit) a = m; else b = m - 1;
        }
        const int hi = lo < n ? std::max(a, lo + 1) : lo;
        if (out) (*out)[size_t(r) + 1] = std::min(hi, n);
        lo = std::min(hi, n);
      }
      return lo >= n;
    };
    double lo_t = 0, hi_t = shard_cost(codes, 0, n, c, nscale, chunk_rows);
    for (int it = 0; it < 40; ++it) {
      const double mid = 0.5 * (lo_t + hi_t);
      if (cuts_for(mid, nullptr)) hi_t = mid; else lo_t = mid;
    }
    cuts_for(hi_t, &bounds);
    bounds[size_t(world)] = n;
    // the greedy fill leaves the slack in the last shard and may strand one or two rows of a kernel class in a shard:
    // move every boundary by up to 4 rows where that lowers the larger of the two neighbouring shards
    for (int pass = 0; pass < 3; ++pass)
      for (int i = 1; i < world; ++i) {
        const int lo = bounds[size_t(i) - 1], hi = bounds[size_t(i) + 1];
        int best = bounds[size_t(i)];
        double best_cost = -1;
        for (int b = std::max(lo, bounds[size_t(i)] - 4); b <= std::min(hi, bounds[size_t(i)] + 4); ++b) {
          const double cost = std::max(shard_cost(codes, lo, b, c, nscale, chunk_rows), shard_cost(codes, b, hi, c, nscale, chunk_rows));
          if (best_cost < 0 || cost < best_cost - 1e-9) { best = b; best_cost = cost; }
        }
        bounds[size_t(i)] = best;
      }
  }
  for (int r = 0; r < world; ++r) { first[r] = bounds[size_t(r)]; count[r] = bounds[size_t(r) + 1] - bounds[size_t(r)]; }
  return CWT_OK;
}

int cwt_shard_cost(const int* codes, int nrows, int precision, double nscale, int chunk_rows, double* cost_us) {
  if (!codes || !cost_us) return fail(CWT_EINVAL, "NULL argument");
  if (nrows < 0 || (precision != 32 && precision != 64) || !(nscale > 0) || chunk_rows < 1)
    return fail(CWT_EINVAL, "bad shard arguments");
  *cost_us = shard_cost(codes, 0, nrows, precision == 64 ? kShardCost64 : kShardCost32, nscale, chunk_rows);
  return CWT_OK;
}

int cwt_plan_balanced_shards(cwt_plan* p, int mother, double param, double dt, const double* scales, int nrows,
                             int64_t ncols, int world, int* first, int* count) {
  if (!p || !scales || !first || !count) return fail(CWT_EINVAL, "NULL argument");
  std::vector<int> codes(size_t(std::max(nrows, 1)));
  int rc = cwt_plan_classify(p, mother, param, dt, scales, nrows, ncols, 1, codes.data());
  if (rc) return rc;
  return cwt_shard_codes(codes.data(), nrows, p->prec, double(p->N) / double(1 << 20), chunk_rows_of(p), world, first, count);
}

int cwt_plan_last_split(cwt_plan* p, int counts[6]) {
  if (!p || !counts) return fail(CWT_EINVAL, "NULL argument");
  for (int i = 0; i < 6; ++i) counts[i] = p->split[i];
  return CWT_OK;
}

int cwt_plan_last_split8(cwt_plan* p, int counts[8]) {
  if (!p || !counts) return fail(CWT_EINVAL, "NULL argument");
  for (int i = 0; i < 8; ++i) counts[i] = p->split[i];
  return CWT_OK;
}

}  // extern "C"
