// abi.hip -- the exported C functions of libcwt_hip.so (include/cwt_hip.h has the contract and the reference lines each entry
// point stands in for).  Argument checks, the row-table cache lookups, dispatch on the plan's precision into the launches of
// launch_impl.hpp (compiled in launch_f64.hip / launch_f32.hip); the few small kernels launched directly from here
// (spectrum range, cross spectrum, time mean, coherence histogram) are instantiated in this file.
#include "launch_impl.hpp"

using namespace cwtd;
using namespace cwt;

namespace {
int prepare_rows_table(cwt_plan* p, bool have_signal, int mother, double param, double dt, const double* scales,
                       int nrows, int64_t ldw, int64_t ncols) {
  if (nrows < 1 || nrows > p->max_rows) return fail(CWT_EINVAL, "nrows must be in [1, max_rows]");
  if (ncols < 1 || ncols > p->N || ldw < ncols) return fail(CWT_EINVAL, "need 1 <= ncols <= nfft and ldw >= ncols");
  if (!(dt > 0) || !std::isfinite(dt)) return fail(CWT_EINVAL, "dt must be positive");
  const std::vector<double> key = call_key(0, {p->tolerance, double(mother), param, dt, double(nrows), have_signal ? 1.0 : 0.0, double(ncols)},
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
    if (!rc && p->rt->poly_rtab_elems)
      rc = p->prec == 64 ? fill_poly_tables<double>(p) : fill_poly_tables<float>(p);
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
  if (!rc && p->logN >= 18 && !p->profile) rc = ensure_distinct_queues(p);
  if (rc) return rc;
  const Mother mo = mother_of(mother, param);
  return p->prec == 64 ? rows_impl<double>(p, xhat_dev, mo, nrows, W_dev, ldw, ncols, x_dev, n0)
                       : rows_impl<float>(p, xhat_dev, mo, nrows, W_dev, ldw, ncols, x_dev, n0);
}

// The overlap-save rows need the signal only: cwt_transform queues them on side stream 1 BEFORE the forward FFT, so
// that they run beside it and beside the two-pass chain; rows_impl then skips them and joins the stream at its end.

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

// Cost model of one rank's step, microseconds at N = 2^20: per kernel class a fixed part (launch ramp and tail; for the
// overlap-save classes the block spectra of that tile size) + a per-row part.  Fitted to the per-class launch durations of
// bench.py on BASELINE configs 2 / 3 (profiles/r03_per_class.txt) and the per-rank runs of profiles/r03_shards.txt; the
// overlap-save, band-passed and polynomial terms of fp64 refitted (least squares) to the 15 per-rank runs of
// profiles/r04_shards.txt.
struct ShardCost { double fwd, tp_fixed, tp_row, k2048_fixed, k2048_row, ols_fixed, ols_row, olsh_fixed, olsh_row, nar_fixed, nar_row, nar_term,
                   aols_fixed, aols_row, poly_fixed, poly_row, poly_coef; };
constexpr ShardCost kShardCost64 = {30.0, 18.0, 9.8, 30.0, 6.1, 22.0, 3.25, 44.0, 3.2, 8.0, 2.85, 0.9, 38.0, 2.9, 12.5, 2.67, 2.7};
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
    else if (kind == 5) {
      // (round 6: the rows of the narrow block supports are the ones with the long halos -- less of every block kept, and those on
      // 8192-point blocks come in a second launch behind their own block spectra: +10 % [measured per rank, profiles/r06_shards.txt])
      if (!seen_olsh) { seen_olsh = true; total += c.olsh_fixed; }
      total += c.olsh_row * nscale * (logk <= 8 ? 1.10 : 1.0);
    }
    else if (kind == 7) {
      // stage 2 per row (a little more per degree) + the row's share of stage 1: (D + 1) K' coefficients, priced at the
      // measured 2.2 us (fp64) of a K' = 16384, D = 8 row (profiles/r04_shards.txt)
      if (!seen_poly) { seen_poly = true; total += c.poly_fixed; }
      // (round 6, per (K', degree) class, profiles/r06_sessions.txt session r: the rows slow down with K' -- more coefficient sets per
      // workgroup -- beyond what their share of stage 1 says: +5 % at K' = 1024, +12 % at 2048 / 4096, +10 % above)
      const double kprime = logk >= 13 ? 0.10 : logk >= 11 ? 0.12 : logk == 10 ? 0.05 : 0.0;
      total += c.poly_row * nscale * (1.0 + 0.015 * std::max(0, terms - 8) + kprime);
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

// =============================================================================================
// ---- hardware queues ---------------------------------------------------------------------------------------------------
// A long transform runs on four streams (the caller's + three side streams).  The runtime multiplexes ALL streams of the
// process onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in the order they were created; two of the plan's
// streams that land on ONE queue execute in submission order and the overlap the schedule was built for is gone [measured:
// fp32 Paul 0.600 -> 0.618 / 0.637 ms with one / two idle streams created before the plan, fp64 Morlet 0.935 -> 0.952;
// the config-3 blocks of a default bench run, whose plans come after a dozen others, 6-10 % slower than alone].  HIP has no
// query for a stream's queue, so the plan MEASURES it: a one-thread kernel on stream a waits (bounded: 200 us) for a flag that
// a one-thread kernel on stream b sets; submitted in that order, b can only get through if it sits on another queue.  A side
// stream that shares a queue with the caller's stream or with an earlier side stream is parked (kept alive, idle, so that the
// runtime's least-used-queue choice moves on) and replaced.  Once per plan and caller's stream, ~0.3 ms; only for transforms
// long enough to use the side streams.  Option "queue_probe" = 0 turns it off.
#if defined(CWT_HIP_EMULATED)
int cwtd::ensure_distinct_queues(cwt_plan*) { return CWT_OK; }     // (the CPU stand-in of the tests runs kernels in launch order)
#else
__global__ void k_queue_probe_wait(int* flag, int* seen, long long ticks) {
  const long long t0 = wall_clock64();                              // 100 MHz
  int s = 0, turns = 0;
  do {
    s = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if (!s) __builtin_amdgcn_s_sleep(16);                           // ~0.5 us
  } while (!s && wall_clock64() - t0 < ticks && ++turns < 20000);   // (the turn count bounds the wait whatever the clock does)
  *seen = s;
}
__global__ void k_queue_probe_set(int* flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

// true: kernels of a and b run side by side (different hardware queues)
static int queues_differ(cwt_plan* p, hipStream_t a, hipStream_t b, bool* differ) {
  HIPCHECK(hipMemsetAsync(p->probe_dev, 0, 2 * sizeof(int), a));
  HIPCHECK(hipEventRecord(p->ev_probe, a));
  HIPCHECK(hipStreamWaitEvent(b, p->ev_probe, 0));
  hipLaunchKernelGGL(k_queue_probe_wait, dim3(1), dim3(1), 0, a, p->probe_dev, p->probe_dev + 1, 20000LL);
  hipLaunchKernelGGL(k_queue_probe_set, dim3(1), dim3(1), 0, b, p->probe_dev);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipStreamSynchronize(a));
  HIPCHECK(hipStreamSynchronize(b));
  int h[2] = {0, 0};
  HIPCHECK(hipMemcpy(h, p->probe_dev, sizeof(h), hipMemcpyDeviceToHost));
  *differ = h[1] != 0;
  return CWT_OK;
}

int cwtd::ensure_distinct_queues(cwt_plan* p) {
  if (!p->queue_probe) return CWT_OK;
  // one probe per caller's stream (a caller that alternates between two streams does not pay 0.3 ms per call), remembered for
  // the last few; never while the caller's stream is being captured (the probe synchronises)
  for (hipStream_t s : p->probed_streams) if (s == p->stream) return CWT_OK;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(p->stream, &cap) != hipSuccess) (void)hipGetLastError();
  else if (cap != hipStreamCaptureStatusNone) return CWT_OK;
  if (!p->probe_dev && hipMalloc(reinterpret_cast<void**>(&p->probe_dev), 2 * sizeof(int)) != hipSuccess)
    return fail(CWT_ENOMEM, "device allocation failed");
  if (!p->ev_probe) HIPCHECK(hipEventCreateWithFlags(&p->ev_probe, hipEventDisableTiming));
  static const bool verbose = std::getenv("CWT_QUEUE_PROBE_VERBOSE") != nullptr;
  hipStream_t* mine[3] = {&p->side[1], &p->side[0], &p->side2};     // by the work they carry: overlap-save chain first
  hipStream_t fixed[4] = {p->stream, nullptr, nullptr, nullptr};
  bool replaced = false;
  for (int i = 0; i < 3; ++i) {
    for (int attempt = 0;; ++attempt) {
      bool ok = true;
      for (int j = 0; j <= i && ok; ++j) {
        const int rc = queues_differ(p, fixed[j], *mine[i], &ok);
        if (rc) return rc;
      }
      if (ok || attempt == 8 || p->spacers.size() >= 16) {          // (a busy GPU can make a probe time out: the parked streams are capped)
        if (verbose) std::fprintf(stderr, "[cwt] side stream %d: %s after %d replacement(s)\n", i, ok ? "own hardware queue" : "still shares a queue", attempt);
        break;
      }
      p->spacers.push_back(*mine[i]);                               // idle from here on; destroyed with the plan
      HIPCHECK(create_side_stream(mine[i]));
      ++p->queue_collisions;
      replaced = true;
    }
    fixed[i + 1] = *mine[i];
  }
  if (replaced) p->probed_streams.clear();                          // what was measured against the old side streams is stale
  if (p->probed_streams.size() >= 8) p->probed_streams.erase(p->probed_streams.begin());
  p->probed_streams.push_back(p->stream);
  p->queues_probed = true;
  p->probed_main = p->stream;
  return CWT_OK;
}
#endif


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

// Events that only order streams (fork / join of the side streams): no timestamps.  CWT_EVENT_TIMING=1 creates them with
// timestamps as before round 6 (A/B of the hand-over latency).
static hipError_t order_event(hipEvent_t* e) {
  static const bool timing = [] { const char* v = std::getenv("CWT_EVENT_TIMING"); return v && std::atoi(v) != 0; }();
  return timing ? hipEventCreate(e) : hipEventCreateWithFlags(e, hipEventDisableTiming);
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
  p->serial_rows = precision == 64 ? 2 : 0;       // [measured, round 6: c2 -1 %, fp64 Paul -5.7 %; fp32 DOG +-0, fp32 Paul +1.4 %]
  if (const char* e = std::getenv("CWT_TOLERANCE")) {   // default accuracy target of plans created from here on
    const double t = std::atof(e);
    if (t > 0 && t <= 1e-2) p->tolerance = t;
  }
  if (const char* e = std::getenv("CWT_QUEUE_PROBE")) p->queue_probe = std::atoi(e) != 0;   // (diagnostic: plans the caller does not create itself)
  int rc = precision == 64 ? build_tables<double>(p) : build_tables<float>(p);
  if (!rc) rc = precision == 64 ? set_func_attrs<double>() : set_func_attrs<float>();
  for (int i = 0; i < 2 && !rc; ++i) {
    if (create_side_stream(&p->side[i]) != hipSuccess ||
        order_event(&p->ev_a[i]) != hipSuccess || order_event(&p->ev_b[i]) != hipSuccess)
      rc = fail(CWT_EHIP, "cannot create side streams/events");
  }
  if (!rc && order_event(&p->ev_fork) != hipSuccess) rc = fail(CWT_EHIP, "cannot create event");
  if (!rc && order_event(&p->ev_ols) != hipSuccess) rc = fail(CWT_EHIP, "cannot create event");
  if (!rc && (create_side_stream(&p->side2) != hipSuccess || order_event(&p->ev_big) != hipSuccess))
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
  for (hipStream_t sp : p->spacers) (void)hipStreamDestroy(sp);
  if (p->ev_probe) (void)hipEventDestroy(p->ev_probe);
  if (p->probe_dev) (void)hipFree(p->probe_dev);
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
    if (t.prt_dev) (void)hipFree(t.prt_dev);
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
  else if (k == "aols_zc") p->aols_zc = value != 0;
  else if (k == "aols_long") p->aols_long = value != 0;
  else if (k == "poly") p->poly = value != 0;
  else if (k == "coef_small") p->coef_small = value != 0;
  else if (k == "poly_carrier") p->poly_carrier = value != 0;
  else if (k == "poly_cheb") p->poly_cheb = value != 0;
  else if (k == "poly_degree") { if (value < 2 || value > POLY_MAX_DEGREE) return fail(CWT_EINVAL, "poly_degree in [2, 24]"); p->poly_degree = int(value); }
  else if (k == "queue_probe") { p->queue_probe = value != 0; }
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
  else if (k == "ols_small_big") p->ols_small_big = value != 0;
  else if (k == "serial_s1_once") p->serial_s1_once = value != 0;
  else if (k == "ols_big_min_halo") { if (value < 64 || value > 8192) return fail(CWT_EINVAL, "ols_big_min_halo in [64, 8192]"); p->ols_big_min_halo = int(value); }
  else if (k == "ols_early") p->ols_early = value != 0;
  else if (k == "aols_small_b") p->aols_small_b = value != 0;
  else if (k == "fft_aside_small") p->fft_aside_small = value != 0;
  else if (k == "serial_rows") { if (value < 0 || value > 3) return fail(CWT_EINVAL, "serial_rows: 0 ... 3"); p->serial_rows = int(value); }
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
  p->tolerance = rel_tol;     // (part of the row tables' cache key: a loop whose measured target flips between two values -- Monte-Carlo
  return CWT_OK;              // surrogates near a threshold of the automatic mode -- keeps both tables; a rebuild at N = 2^23 is 1.6 s)
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
  hipLaunchKernelGGL((k_spectrum_fold<0>), dim3(1), dim3(192), 0, p->stream, part, groups, p->range_dev);
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

int cwt_device_memory(int device, size_t* free_bytes, size_t* total_bytes) {
  if (!free_bytes || !total_bytes) return fail(CWT_EINVAL, "NULL argument");
  HIPCHECK(hipSetDevice(device));
  HIPCHECK(hipMemGetInfo(free_bytes, total_bytes));
  return CWT_OK;
}
int cwt_device_synchronize(int device) {
  HIPCHECK(hipSetDevice(device));
  HIPCHECK(hipDeviceSynchronize());
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
  if (!rc && p->logN >= 18 && !p->profile) rc = ensure_distinct_queues(p);     // (transforms that use the side streams)
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
      p->ols_first_on_main = p->serial_rows >= 2 && serial_schedule(p, true) && p->rt->ols_grp[0].nrows > 0;
      r = p->prec == 64 ? launch_ols_early<double>(p, x_dev, n0, W_dev, ldw, ncols)
                        : launch_ols_early<float>(p, x_dev, n0, W_dev, ldw, ncols);
      if (r) return r;
    }
    // serial_rows = 2: the forward FFT on side stream 0 (the bands + coefficients of the polynomial rows follow it there), so that
    // the first overlap-save rows start on the caller's stream as soon as their block spectra exist
    const bool fft_aside = p->serial_rows >= 2 && serial_schedule(p, p->ols_launched != 0);
    hipStream_t caller = p->stream;
    if (fft_aside) {
      if (!p->ols_launched) HIPCHECK(hipEventRecord(p->ev_fork, caller));
      HIPCHECK(hipStreamWaitEvent(p->side[0], p->ev_fork, 0));
      p->stream = p->side[0];
      p->fft_small = p->fft_aside_small;
    }
    r = p->prec == 64 ? fft_rows_impl<double, IN_REAL>(p, x_dev, 0, 1, n0, xhat_dev)
                      : fft_rows_impl<float, IN_REAL>(p, x_dev, 0, 1, n0, xhat_dev);
    p->stream = caller;
    p->fft_small = 0;
    if (fft_aside && !r) {
      HIPCHECK(hipEventRecord(p->ev_a[1], p->side[0]));
      p->spectrum_ready = p->ev_a[1];
    }
    if (!r) r = p->prec == 64 ? rows_impl<double>(p, xhat_dev, mo, nrows, W_dev, ldw, ncols, x_dev, n0)
                              : rows_impl<float>(p, xhat_dev, mo, nrows, W_dev, ldw, ncols, x_dev, n0);
    p->ols_launched = 0;
    p->ols_first_on_main = 0;
    p->spectrum_ready = nullptr;
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
  const std::vector<double> key = call_key(2, {p->tolerance, double(mother), param, dt, double(nbatch), double(xhat_ld), double(nrows)},
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
  const std::vector<double> key = call_key(3, {p->tolerance, double(mother), param, dt, double(nbatch), double(nrows), double(ncols)},
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
  const std::vector<double> key = call_key(1, {p->tolerance, double(mother), param, double(spec_ld), double(nrows)},
                                           {{a, nrows}, {amp_re, nrows}, {amp_im, nrows}});
  if (!select_table(p, key)) {
    double cre, cim;
    int rc = mother_constant(mother, param, &cre, &cim);   // validates mother / order only
    if (!rc) rc = build_row_table(p, mother, param, a, amp_re, amp_im, spec_ld, nrows);
    if (!rc) rc = upload_row_table(p, key);
    if (!rc && p->rt->poly_rtab_elems)                    // polynomial rows: the tables of their economised weights
      rc = p->prec == 64 ? fill_poly_tables<double>(p) : fill_poly_tables<float>(p);
    if (rc) { p->rt->key.clear(); return rc; }
  }
  set_split(p);
  Mother mo;
  mo.kind = mother; mo.m = int(std::lround(param)); mo.p = param; mo.table = nullptr;
  return p->prec == 64 ? rows_impl<double>(p, spec_dev, mo, nrows, W_dev, ldw, ncols)
                       : rows_impl<float>(p, spec_dev, mo, nrows, W_dev, ldw, ncols);
}


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


int cwt_random_normal(cwt_plan* p, uint64_t seed, uint64_t offset, int64_t n, double scale, void* out_dev) {
  if (!p || !out_dev) return fail(CWT_EINVAL, "NULL argument");
  if (n < 1) return fail(CWT_EINVAL, "n must be >= 1");
  HIPCHECK(hipSetDevice(p->device));
  return p->prec == 64 ? random_normal_impl<double>(p, seed, offset, n, scale, out_dev)
                       : random_normal_impl<float>(p, seed, offset, n, scale, out_dev);
}

int cwt_ar1_filter(cwt_plan* p, const void* e_dev, int64_t tau, int64_t n, double g, void* out_dev) {
  if (!p || !e_dev || !out_dev) return fail(CWT_EINVAL, "NULL argument");
  if (n < 1 || tau < 0) return fail(CWT_EINVAL, "need n >= 1 and tau >= 0");
  if (!(std::fabs(g) < 1.0)) return fail(CWT_EINVAL, "the AR(1) coefficient must be inside (-1, 1)");
  if (e_dev == out_dev) return fail(CWT_EINVAL, "the filter cannot run in place");
  HIPCHECK(hipSetDevice(p->device));
  return p->prec == 64 ? ar1_filter_impl<double>(p, e_dev, tau, n, g, out_dev) : ar1_filter_impl<float>(p, e_dev, tau, n, g, out_dev);
}

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
        p->tolerance = floor_tol;
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
    p->tolerance = floor_tol;
  } else if (W_host && p->auto_target > 0) {
    // accuracy target of THIS call = auto_target / (dynamic range of its spectrum relative to white noise), a power of
    // ten (so that calls with like spectra share one cached row table), never looser than the target itself
    rc = cwt_forward_fft(p, p->hx, n0, p->hxhat);
    double tol = 0;
    if (!rc) rc = cwt_plan_auto_tolerance(p, p->hxhat, p->auto_target, &tol);
    if (rc) return rc;
    p->tolerance = tol;
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
