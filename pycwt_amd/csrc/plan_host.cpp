// plan_host.cpp -- host side without kernels: error state, row classification (which row takes which form: the searches for
// supports, halos, degrees; build_row_table), the row-table cache, scratch buffers, device -> host copies.  Plain C++ against
// the HIP runtime API (streams, events, allocations); compiles in seconds and runs on the CPU stand-in of tests/emu unchanged.
// Reference lines: the filter bank of pycwt/wavelet.py:102-104 is never built; what is decided here is how each row of
// wavelet.py:105-106 is computed instead.
#include "plan.hpp"

namespace cwtd {

thread_local std::string g_err;
uint64_t g_scratch_gen = 0;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

const char* const kClassNames[KC_COUNT] = {"fwd_small", "fwd_pass_a",  "fwd_pass_b", "small",  "direct", "narrow",
                                           "narrow_many", "narrow_big", "pass_a",     "pass_b", "icwt",   "elementwise",
                                           "ols_fwd", "ols", "ols_small", "aols_pre", "aols", "poly_coef", "poly"};

int ilog2(int64_t v) {
  int l = 0;
  while ((int64_t(1) << l) < v) ++l;
  return l;
}

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


Tolerances tolerances(const cwt_plan* p) {
  const double t = p->tolerance > 0 ? p->tolerance : (p->prec == 64 ? kDefaultTolerance64 : kDefaultTolerance32);
  Tolerances r;
  r.support = std::max(t * 0.1, p->prec == 64 ? 1e-18 : 1e-9);
  r.halo = std::max(t * 0.1, p->prec == 64 ? 1e-17 : 5e-7);
  r.clip = std::max(t, p->prec == 64 ? 1e-16 : 1e-8);
  return r;
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
double time_halo_factor_compute(int mother, double param, double eps);
double time_halo_factor(int mother, double param, double eps) {
  // depends on the mother and the accuracy target only, never on the scales: the last few answers are kept (the quadrature
  // below is 0.1 - 0.4 ms of every row-table build otherwise)
  struct Entry { int mother; double param, eps, value; };
  static std::mutex mu;
  static std::vector<Entry> memo;
  {
    std::lock_guard<std::mutex> lk(mu);
    for (const auto& e : memo) if (e.mother == mother && e.param == param && e.eps == eps) return e.value;
  }
  const double v = time_halo_factor_compute(mother, param, eps);
  std::lock_guard<std::mutex> lk(mu);
  if (memo.size() >= 32) memo.erase(memo.begin());
  memo.push_back({mother, param, eps, v});
  return v;
}
double time_halo_factor_compute(int mother, double param, double eps) {
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



// ---- rows clipped at Nyquist: overlap-save on the band-passed complex signal (k_aols_*) ------------------------
// in-place radix-2 inverse DFT (e^{+2 pi i k n / n}, unnormalised) of a power-of-two length; host helper of aols_halo
void host_ifft(std::vector<std::complex<double>>& v) {
  const size_t n = v.size();
  // twiddles e^{2 pi i k / n}, k < n/2, of the last length used (the halo search calls this with one length per table);
  // plain real arithmetic: std::complex's operator* checks for NaN on every product
  static thread_local std::vector<double> twr, twi;
  if (twr.size() != n / 2) {
    twr.resize(n / 2); twi.resize(n / 2);
    for (size_t k = 0; k < n / 2; ++k) {
      const double ang = 6.283185307179586476925 * double(k) / double(n);
      twr[k] = std::cos(ang); twi[k] = std::sin(ang);
    }
  }
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(v[i], v[j]);
  }
  double* d = reinterpret_cast<double*>(v.data());          // (re, im) pairs
  for (size_t len = 2; len <= n; len <<= 1) {
    const size_t half = len / 2, stride = n / len;
    for (size_t i = 0; i < n; i += len) {
      for (size_t k = 0; k < half; ++k) {
        const double wr = twr[k * stride], wi = twi[k * stride];
        double* a = d + 2 * (i + k);
        double* b = d + 2 * (i + k + half);
        const double br = b[0] * wr - b[1] * wi, bi = b[0] * wi + b[1] * wr;
        b[0] = a[0] - br; b[1] = a[1] - bi;
        a[0] += br; a[1] += bi;
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
// The window u on that grid (the same for every row of a table: computed once, two erfc per point otherwise)
std::vector<double> aols_window_grid(const AolsGeom& g, int hmax) {
  const int n = 4 * hmax, k0 = int(std::ceil(g.f_s * n));
  std::vector<double> w(static_cast<size_t>(n));
  for (int q = 0; q < n; ++q) {
    const int kappa = k0 + (((q - k0) % n) + n) % n;
    w[size_t(q)] = host_aols_window(g, double(kappa) / double(n));
  }
  return w;
}
int aols_halo(int mother, double param, double aN, const AolsGeom& g, double eps, int hmax, const std::vector<double>& window) {
  const int n = 4 * hmax;
  std::vector<std::complex<double>> e(size_t(n), std::complex<double>(0.0, 0.0));
  const int k0 = int(std::ceil(g.f_s * n));
  double in_band = 0, beyond = 0;
  for (int q = 0; q < n; ++q) {
    const int kappa = k0 + (((q - k0) % n) + n) % n;
    const double f = double(kappa) / double(n);
    const double u = window[size_t(q)];
    const double v = u != 0.0 ? host_profile(mother, param, aN * f) * u : 0.0;
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

// The same search for a Paul row continued through f = 0 (AolsGeom::zc_c / zc_w, k_aols_gtab): e = IFFT(E),
// E(kappa) = f^m e^-f erfc((-f - c) / w) / 2 at f = aN kappa / n over the signed bins kappa in [-n/2, n/2).  The tail is
// measured against the L1 mass of the TRUE wavelet (positive bins only): the continuation's lobe below 0 is up to `amp`
// times larger than the filter itself and dominates the kernel's own mass, but the band-passed signal has nothing there.
// Returns 0 if no halo <= hmax qualifies; *amp = max|E| / max of the true filter (what rounding noise is multiplied by).
int aols_halo_zc(int m, double aN, double c, double w, double eps, int hmax, double* amp, int* exact = nullptr, double z = 0) {
  const int n = 4 * hmax;
  if (z > 0 && !(aN > 4.0 * (c + 6.0 * w))) return 0;      // the two continuations would overlap above Nyquist
  std::vector<std::complex<double>> e(static_cast<size_t>(n), std::complex<double>(0.0, 0.0));
  std::vector<std::complex<double>> t(static_cast<size_t>(n), std::complex<double>(0.0, 0.0));
  double emax = 0, tmax = 0;
  for (int q = 0; q < n; ++q) {
    const int kappa = q > n / 2 ? q - n : q;
    const double f = aN * double(kappa) / double(n);
    const double arg = (-f - c) / w;
    const double g = std::pow(f, m) * std::exp(-std::max(f, -700.0));
    double v = arg > 9.0 ? 0.0 : g * 0.5 * std::erfc(arg);
    if (z > 0 && q > n / 2) {                               // continued past Nyquist too (k_aols_gtab)
      const double fc = double(q) / double(n), fu = aN * fc;
      if (fc < 0.76) v += std::pow(fu, m) * std::exp(-fu) * 0.5 * std::erfc(z * (fc - 0.625) / 0.125);
    }
    e[size_t(q)] = v;
    t[size_t(q)] = kappa > 0 && q != n / 2 ? g : 0.0;
    emax = std::max(emax, std::fabs(v));
    tmax = std::max(tmax, std::fabs(t[size_t(q)].real()));
  }
  if (!(tmax > 0)) return 0;
  *amp = emax / tmax;
  host_ifft(e);
  host_ifft(t);
  double total = 0;
  for (int i = 0; i < n; ++i) total += std::abs(t[size_t(i)]);
  std::vector<double> ring(size_t(n / 2) + 1, 0.0);
  for (int i = 0; i < n; ++i) ring[size_t(std::min(i, n - i))] += std::abs(e[size_t(i)]);
  double tail = 0;
  int best = 0;
  for (int d = n / 2; d > 0; --d) {
    tail += ring[size_t(d)];
    if (tail > eps * total) break;
    if (exact) *exact = d - 1;
    if ((d - 1) % 64 == 0 && d - 1 >= 64 && d - 1 <= hmax) best = d - 1;
  }
  return best;
}

// The halo of such a row is its scale times a constant as long as the filter has died out long before Nyquist (the kernel is
// the same function of t / s): the constant, from one search at a scale of 100 samples, kept per (m, c, w, eps).
double zc_halo_factor(int m, double c, double w, double eps) {
  struct Entry { int m; double c, w, eps, value; };
  static std::mutex mu;
  static std::vector<Entry> memo;
  {
    std::lock_guard<std::mutex> lk(mu);
    for (const auto& e : memo) if (e.m == m && e.c == c && e.w == w && e.eps == eps) return e.value;
  }
  double amp = 0;
  int exact = 0;
  const double s_ref = 100.0;
  (void)aols_halo_zc(m, 2.0 * 3.14159265358979323846 * s_ref, c, w, eps, 4096, &amp, &exact);
  const double v = exact > 0 ? double(exact) / s_ref : 0.0;
  std::lock_guard<std::mutex> lk(mu);
  if (memo.size() >= 32) memo.erase(memo.begin());
  memo.push_back({m, c, w, eps, v});
  return v;
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
constexpr int POLY_SAMPLES = 257;
struct PolyBandSamples {     // the row's filter relative to its largest value on the bins, and |kappa| (+ 1), at <= 257 bins of the band
  int npts = 0;
  double g[POLY_SAMPLES], kap[POLY_SAMPLES];
};
void poly_band_samples(int mother, double param, double a, int kc, int k_lo, int nband, double log_best, PolyBandSamples* out) {
  const int npts = std::min(nband, POLY_SAMPLES);
  out->npts = npts;
  for (int i = 0; i < npts; ++i) {
    const int k = k_lo + (npts > 1 ? int((long(nband - 1) * i) / (npts - 1)) : 0);
    const double lg = profile_log_rel(mother, param, a * double(k)) - log_best;
    out->g[i] = std::isfinite(lg) ? std::exp(std::min(lg, 0.0)) : 0.0;
    out->kap[i] = double(k - kc);                                           // signed distance from the sampling carrier kc
  }
}
// shift: the carrier sits `shift` bins above the one the samples were taken around
// cheb: the expansion is the Chebyshev series of e^{i theta u} (Jacobi-Anger: J_0 + 2 sum_k i^k J_k(theta) T_k(u)) cut at degree D and
// re-expanded in monomials (k_poly_rtab); what is cut is 2 sum_{k > D} |J_k(theta)| <= 2 (theta/2)^(D+1) / (D+1)! / (1 - theta / (2 D + 4))
// -- 2^D below the Taylor remainder of the same degree.
// dmax: degrees above it are of no interest to the caller (the search of the carrier stops a candidate where it cannot win any more)
// stride: every stride-th sample only (the search; the degree of the winner is then taken from all samples)
int poly_degree_for(const PolyBandSamples& b, int logk, double eps, double shift = 0.0, bool cheb = false, int dmax = POLY_MAX_DEGREE + 1,
                    int stride = 1) {
  const double tscale = 3.14159265358979323846 / double(1 << logk);
  double term[POLY_SAMPLES], th[POLY_SAMPLES];
  int n = 0;
  for (int i = 0; i < b.npts; i += stride, ++n) {
    term[n] = cheb ? 2.0 * b.g[i] : b.g[i];
    th[n] = (std::fabs(b.kap[i] - shift) + 1.0) * tscale * (cheb ? 0.5 : 1.0);   // + 1: the sampling skips neighbours
  }
  for (int d = 0; d <= std::min(dmax, POLY_MAX_DEGREE + 1); ++d) {
    double worst = 0;
    const double inv = 1.0 / double(d + 1);
    for (int i = 0; i < n; ++i) {
      term[i] *= th[i] * inv;
      const double tail = cheb ? 1.0 / (1.0 - std::min(th[i] / double(d + 2), 0.9)) : 1.0;
      worst = std::max(worst, term[i] * tail);
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
                    const double* amp_im, int64_t spec_ld, int nrows, const int* tab_klo,
                    const int* tab_nband, int rows_per_signal, int64_t tab_ld,
                    int64_t ols_ncols, int64_t out_ncols) {
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
  std::vector<char> wide_clipped, wide_unclipped;
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
    rd.kc_off = 0;
    rd.rtab_off = -1;
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
        bool small_big = false;
        if (ols_logp_s && p->ols_small_big && halo > p->ols_small_max_halo && halo <= (1 << (ols_logp_s + 1)) / 8 &&
            p->logN >= ols_logp_s + 3) {                  // (halo <= 1/8 block: three quarters of every block transform are kept)
          // blocks of two half-size tiles: the same 8192-point blocks as the default tile's, on 256-thread workgroups (four per CU)
          RowDesc two = rd;
          describe(ols_logp_s + 1, ols_logp_s, two);
          if (two.logK <= ols_logp_s - 3) { od = two; lb = ols_logp_s + 1; grp = 0; small_big = true; }
        }
        if (small_big) {
        } else if (ols_logp_s && halo <= p->ols_small_max_halo) {
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
      int poly_logk = 0, poly_deg = 0, poly_shift = 0;
      if (poly_ok && rd.nband > 0) {
        const int lk_max = std::min(p->poly_max_logk, p->logN - POLY_MIN_LOGR);
        const int kc = rd.k_lo + (rd.nband >> 1);
        PolyBandSamples band;
        if (std::max(8, ilog2(rd.nband)) <= lk_max) poly_band_samples(mother, param, rd.a, kc, rd.k_lo, rd.nband, row_best, &band);
        // The carrier k_c: the band's centre minimises the largest |theta|, but the bound weighs theta by the filter -- for a
        // lopsided filter (Paul: peak at 9 % of its band; DOG) a carrier nearer the peak needs a lower degree at the same K',
        // i.e. fewer or shorter coefficient planes.  Candidates: centre + c nband / 16, c = -7 ... 7 (the centre wins ties).
        // (a filter whose peak sits within 1/16 band of the centre -- Morlet -- keeps the centre: the search is 15 evaluations of the
        // degree rule per K', the whole cost of classifying a new scale grid)
        int cmax = 0;
        if (p->poly_carrier && band.npts > 16) {
          int ipk = 0;
          for (int i = 1; i < band.npts; ++i) if (band.g[i] > band.g[ipk]) ipk = i;
          if (std::abs(2 * ipk - (band.npts - 1)) * 8 > band.npts) cmax = 7;
        }
        bool searched = false;
        int c_prev = 0;
        for (int lk = std::max(8, ilog2(rd.nband)); lk <= lk_max; ++lk) {
          int best_deg = POLY_MAX_DEGREE + 2, best_c = 0;
          auto try_c = [&](int cc) {
            const double shift = double(cc) * double(rd.nband) / 16.0;
            const int deg = poly_degree_for(band, lk, tol.support, std::round(shift), p->poly_cheb != 0, best_deg, cmax ? 4 : 1);
            if (deg < best_deg || (deg == best_deg && std::abs(cc) < std::abs(best_c))) { best_deg = deg; best_c = cc; }
          };
          if (!searched) {                                   // all candidates at the first K', the neighbours of the winner after that
            for (int c = 0; c <= cmax; ++c)
              for (int sgn = (c ? -1 : 1); sgn <= 1; sgn += 2) try_c(sgn * c);
            searched = true;
          } else {
            for (int cc = std::max(-cmax, c_prev - 1); cc <= std::min(cmax, c_prev + 1); ++cc) try_c(cc);
          }
          c_prev = best_c;
          if (cmax && best_deg <= POLY_MAX_DEGREE)            // the winner's degree from every sample
            best_deg = poly_degree_for(band, lk, tol.support, std::round(double(best_c) * double(rd.nband) / 16.0), p->poly_cheb != 0);
          if (best_deg > POLY_MAX_DEGREE) continue;
          poly_logk = lk; poly_deg = best_deg;
          poly_shift = int(std::round(double(best_c) * double(rd.nband) / 16.0));
          if (best_deg <= p->poly_degree) break;
        }
      }
      if (poly_logk) {
        rd.logK = poly_logk;
        rd.nterms = poly_deg;
        rd.kc_off = (rd.nband >> 1) + poly_shift;
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
        wide_unclipped.push_back(aols_ok && vanishes && rd.nband > 0 && amp_im[j] == 0.0);
      }
    }
  }
  // Rows clipped at Nyquist (so far two-pass rows) that can run as overlap-save rows on the band-passed complex signal:
  // one mask and one window for all of them (from the smallest scale), the halo of each from its kernel, one halo class.
  std::vector<RowDesc> aols_rows, aols2_rows;
  AolsGeom ag{}, ag2{};
  int aols_logp = 12, aols_ks = 1;
  // Paul rows whose filter has died out at Nyquist (two-pass rows so far: the kink of f^m H(f) at f = 0 gives the wavelet its
  // 1/t^(m+1) tail, a halo of ~250 scales at 1e-10): the profile continued THROUGH f = 0 and cut below it by a taper of the
  // row's own width (AolsGeom::zc_*, k_aols_gtab).  Taper centre c, width c / 6 (u(0) = 1 - 1e-17).  The lobe below 0 is
  // ~c^m e^c / (2 m^m e^-m) times the filter; the band-passed signal has nothing there but rounding noise, so that factor times
  // the arithmetic's epsilon must stay a tenth of the accuracy target: c = 4 in fp64 at 1e-9 (x 1.5e3), not available in fp32.
  const int paul_m = int(std::lround(param));
  double zc_c = 0;
  if (aols_ok && p->aols_zc && mother == MOTHER_PAUL && paul_m >= 1 && rows_per_signal == 0) {
    const double mach = p->prec == 64 ? 1.2e-16 : 6e-8;
    const double budget = 0.1 * std::max(tol.clip, 10 * mach) / mach;
    const double peak = std::pow(double(paul_m), paul_m) * std::exp(-double(paul_m));
    for (double c : {4.0, 3.5, 3.0, 2.5, 2.0})
      if (std::pow(c, paul_m) * std::exp(c) * 0.5 / peak <= budget) { zc_c = c; break; }
  }
  if (aols_ok) {
    double a_min = 0;
    bool any_zc = false;
    for (size_t i = 0; i < wide_rows.size(); ++i) {
      if (wide_clipped[i] && (a_min == 0 || wide_rows[i].a < a_min)) a_min = wide_rows[i].a;
      any_zc = any_zc || (zc_c > 0 && wide_unclipped[i]);
    }
    bool geom_ok = a_min > 0 || any_zc;
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
    std::vector<int> halos(wide_rows.size(), 0), halos2(wide_rows.size(), 0);
    std::vector<char> zc_row(wide_rows.size(), 0);
    int hmax_seen = 0, hmax2_seen = 0, cnt = 0, cnt2 = 0;
    const int rps = rows_per_signal > 0 ? rows_per_signal : nrows;
    if (geom_ok) {
      const double eps = std::max(tol.halo, p->prec == 64 ? 2e-14 : 5e-7);
      std::vector<int> halo_of_scale(size_t(rps), -1);     // (a numeric tail search each: once per scale, not per signal)
      const std::vector<double> window = aols_window_grid(ag, 512);
      std::vector<double> window2;                          // ... on the grid of the long-halo search (made when a row asks for it)
      ag.zc_c = ag2.zc_c = zc_c;
      ag.zc_w = ag2.zc_w = zc_c / 6.0;
      double zc_f_safe = 0, zc_factor = 0, zc_dummy = 0;
      if (zc_c > 0) {
        profile_support(MOTHER_PAUL, param, eps * 1e-3, &zc_dummy, &zc_f_safe);
        zc_factor = zc_halo_factor(paul_m, ag.zc_c, ag.zc_w, eps);
      }
      for (size_t i = 0; i < wide_rows.size(); ++i) {
        const bool zc_clipped = zc_c > 0 && wide_clipped[i] && wide_rows[i].a * double(N) > 4.0 * (ag.zc_c + 6.0 * ag.zc_w);
        if (zc_clipped) {                                  // filter alive at Nyquist AND room for both continuations
          double amp = 0;
          const int h = aols_halo_zc(paul_m, wide_rows[i].a * double(N), ag.zc_c, ag.zc_w, eps, 2048, &amp, nullptr, ag.z);
          if (h > 0 && h <= 512) { halos[i] = h; zc_row[i] = 1; ++cnt; hmax_seen = std::max(hmax_seen, h); continue; }
          if (h > 512) { halos2[i] = h; zc_row[i] = 1; ++cnt2; hmax2_seen = std::max(hmax2_seen, h); continue; }
        }
        if (zc_c > 0 && wide_unclipped[i]) {               // continued through f = 0: 4096-point tiles up to a halo of 512,
          double amp = 0;                                  // 8192-point tiles up to 2048 (a second class, below)
          // scale-invariant halo where the filter has died out far below eps long before Nyquist, the numeric search otherwise
          int h = 0;
          const double aN = wide_rows[i].a * double(N);
          if (0.5 * aN > zc_f_safe && zc_factor > 0) {
            const double want = zc_factor * aN / (2.0 * 3.14159265358979323846) * 1.01 + 2.0;
            h = want <= 2048.0 ? std::max(64, int((int64_t(std::ceil(want)) + 63) / 64 * 64)) : 0;
          } else {
            h = aols_halo_zc(paul_m, aN, ag.zc_c, ag.zc_w, eps, 2048, &amp);
          }
          if (h > 0 && h <= 512) { halos[i] = h; zc_row[i] = 1; ++cnt; hmax_seen = std::max(hmax_seen, h); }
          else if (h > 512) { halos2[i] = h; zc_row[i] = 1; ++cnt2; hmax2_seen = std::max(hmax2_seen, h); }
          continue;
        }
        if (!wide_clipped[i]) continue;
        int& h = halo_of_scale[size_t(wide_rows[i].out_row % rps)];
        if (h < 0) h = aols_halo(mother, param, wide_rows[i].a * double(N), ag, eps, 512, window);
        halos[i] = h;
        if (halos[i]) { ++cnt; hmax_seen = std::max(hmax_seen, halos[i]); }
        else if (p->aols_long && p->prec == 64 && rows_per_signal == 0 && mother != MOTHER_DOG && p->logN >= 16) {
          // clipped rows whose kernel is longer than the 4096-point tile allows (fp64 Paul, s = 2.2 ... 11: the kink of f^m H(f) at
          // f = 0 with no room for the continuation through it): the second class, 8192-point tiles, halos up to 2048
          if (window2.empty()) window2 = aols_window_grid(ag, 2048);
          const int h2 = aols_halo(mother, param, wide_rows[i].a * double(N), ag, eps, 2048, window2);
          if (h2 > 512) { halos2[i] = h2; ++cnt2; hmax2_seen = std::max(hmax2_seen, h2); }
        }
      }
      if (cnt2 && !cnt) {                                  // (the second class rides on the first one's band-passed signal: keep
        cnt2 = 0;                                          // the layout simple -- no first class, no second)
        std::fill(halos2.begin(), halos2.end(), 0);
      }
    }
    if (geom_ok && cnt % aols_nbatch == 0 && (cnt + cnt2) / aols_nbatch >= std::max(1, p->aols_min_rows) && cnt > 0) {
      aols_logp = 12;                                      // 4096-point tiles: four block transforms in flight per CU
      const int P = 1 << aols_logp, L = P - 2 * hmax_seen;
      ag.halo = hmax_seen;
      ag.nrows = cnt / aols_nbatch;                         // per signal
      ag.nblocks = int((out_ncols + L - 1) / L);
      ag.ksp = int(std::ceil(ag.f_s * double(P)));
      std::vector<RowDesc> keep;
      long toff = 0;
      std::vector<long> tab_of_scale(size_t(rps), -1);     // one filter table per scale, shared by the signals
      // the second class: 8192-point tiles, its own halo / block grid, its tables behind the first class's
      const int P2 = 1 << 13, L2 = P2 - 2 * hmax2_seen;
      ag2.f_s = ag.f_s; ag2.f1_lo = ag.f1_lo; ag2.z = ag.z;
      ag2.halo = hmax2_seen;
      ag2.nrows = cnt2;
      ag2.nblocks = cnt2 ? int((out_ncols + L2 - 1) / L2) : 0;
      ag2.ksp = int(std::ceil(ag2.f_s * double(P2)));
      long toff2 = long(ag.nrows) << aols_logp;
      for (size_t i = 0; i < wide_rows.size(); ++i) {
        if (halos2[i]) {
          RowDesc o = wide_rows[i];
          o.a = wide_rows[i].a * double(N >> 13);
          o.amp_re = amp_re[o.out_row] / double(P2);
          o.amp_im = 0.0;
          o.k_lo = ag2.ksp; o.nband = P2;
          o.logK = 13; o.nterms = 1;
          o.nyq_re = o.nyq_im = 0.0;
          o.aux_off = zc_row[i] ? 1 : 0;                    // continued through f = 0, or the plain window of the first class
          o.tab_off = toff2;
          toff2 += P2;
          aols2_rows.push_back(o);
          continue;
        }
        if (!halos[i]) { keep.push_back(wide_rows[i]); continue; }
        RowDesc o = wide_rows[i];
        o.a = wide_rows[i].a * double(N >> aols_logp);     // profile argument per block bin
        o.amp_re = amp_re[o.out_row] / double(P);          // 1/P of the block's inverse transform (x_M carries its own 1/N)
        o.amp_im = 0.0;
        o.k_lo = ag.ksp; o.nband = P;
        o.logK = aols_logp; o.nterms = 1;
        o.nyq_re = o.nyq_im = 0.0;
        o.aux_off = zc_row[i] ? 1 : 0;
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
    G.cls.n = 0; G.wgs = 0; G.wgs_base = 0; G.fwd_blocks[0] = G.fwd_blocks[1] = G.fwd_blocks[2] = 0; G.row_first = G.nrows = 0;
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
          // (complex64: a workgroup takes two blocks of a row, CWT_PAIR_F32)
          const long nunits = ols_pairs(p->prec / 8, grp.logp) ? (k.nblocks + 1) / 2 : k.nblocks;
          wg += ((nunits * ols_nbatch + 7) / 8) * 8 * (k.nrows / ols_nbatch) * G;
          blk += k.nblocks;
          xs += long(k.nblocks) * stride;
          lo_d = hi_d;
        }
        grp.fwd_blocks[lb - grp.logp] = blk;
        row0 += nr;
        if (lb == grp.logp) grp.wgs_base = wg;
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
  p->rt->n_aols2 = 0;
  if (!aols_rows.empty()) {
    p->rt->table.insert(p->rt->table.end(), aols_rows.begin(), aols_rows.end());
    p->rt->aols_logp = aols_logp;
    p->rt->aols_geom = ag;
    const bool pair32 = aols_pairs(p->prec / 8);                // a workgroup of k_aols_rows takes two blocks of a row
    p->rt->aols_wgs = long(((pair32 ? (ag.nblocks + 1) / 2 : ag.nblocks) + 7) / 8) * 8 * ag.nrows;
    p->rt->aols_gt_elems = long(ag.nrows) << aols_logp;
    p->rt->aols_nbatch = aols_nbatch;
    // second class (Paul rows continued through f = 0 on 8192-point tiles): right behind the first in the table
    p->rt->aols2_first = int(p->rt->table.size());
    p->rt->n_aols2 = int(aols2_rows.size());
    p->rt->n_aols += p->rt->n_aols2;
    if (!aols2_rows.empty()) {
      p->rt->table.insert(p->rt->table.end(), aols2_rows.begin(), aols2_rows.end());
      p->rt->aols2_geom = ag2;
      p->rt->aols2_wgs = long(((pair32 ? (ag2.nblocks + 1) / 2 : ag2.nblocks) + 7) / 8) * 8 * ag2.nrows;
      p->rt->aols_gt_elems += long(ag2.nrows) << 13;
    }
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
        pc.c[pc.n++] = PolyClass{r.logK, int(i) - ch.row_first, 0, 0, 0, 0};
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
        // the single launch on 4096-point tiles: K' > 4096 takes K' / 4096 workgroups per job, in groups of 8 jobs
        const long jobs = long(c.nrows) * c.ndeg, s = c.logK > 12 ? 1L << (c.logK - 12) : 1;
        c.wg_first1 = int(ch.wgs_all);
        ch.wgs_all += c.logK > 12 ? ((jobs + 7) / 8) * 8 * s : (((jobs << c.logK) + 4095) / 4096 + 7) / 8 * 8;
      }
    p->rt->poly_coef_elems = off;
    p->rt->poly_band_elems = boff;
    // tables of the economised weights: one per (K', D) pair, (D + 1) x (K' + 1) reals (|kappa| = 0 ... K'; the sign of an odd
    // degree at negative kappa is applied by the kernel)
    p->rt->poly_rtabs.clear();
    p->rt->poly_rtab_elems = 0;
    if (p->poly_cheb)
      for (RowDesc& r : poly_rows) {
        long at = -1;
        for (const auto& t : p->rt->poly_rtabs) if (t.logK == r.logK && t.deg == r.nterms) at = t.off;
        if (at < 0) {
          at = p->rt->poly_rtab_elems;
          p->rt->poly_rtabs.push_back({r.logK, r.nterms, at});
          p->rt->poly_rtab_elems += (long(r.nterms) + 1) * ((1L << r.logK) + 1);
        }
        r.rtab_off = at;
      }
    p->rt->table.insert(p->rt->table.end(), poly_rows.begin(), poly_rows.end());
  }
  return CWT_OK;
}

// Does the current row table take the serial schedule of rows_launch_serial (launch_impl.hpp)?  Long transforms whose rows are
// polynomial rows plus any of overlap-save / band-passed / two-pass rows, one signal, not while profiling (every timed kernel
// runs alone on the plan's stream then).
bool serial_schedule(const cwt_plan* p, bool ols_early) {
  const auto* rt = p->rt;
  return p->serial_rows && p->overlap_narrow && !p->profile && p->logN >= 18 && rt->n_poly && !rt->n_narrow && !rt->n_small &&
         (rt->n_wide || rt->n_ols || rt->n_aols) && (!rt->n_ols || ols_early) && rt->aols_nbatch == 1 && rt->ols_nbatch == 1;
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


// Makes the slot built from `key` current and returns true, or picks the least recently used slot for a rebuild
// (returns false; the caller builds p->rt->table and calls upload_row_table).  An empty key never matches.
bool select_table(cwt_plan* p, const std::vector<double>& key) {
  ++p->tick;
  if (!key.empty())
    for (auto& t : p->slots)
      if (t.key == key) { p->rt = &t; t.used = p->tick; return true; }
  static const bool verbose = std::getenv("CWT_TABLES_VERBOSE") != nullptr;      // (diagnostic: which calls rebuild a row table)
  if (verbose && key.size() > 4)
    std::fprintf(stderr, "[cwt] row table miss: kind %g tolerance %g mother %g rows/param %g %g (key of %zu)\n", key[0], key[1], key[2], key[3], key[4], key.size());
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

}  // namespace cwtd
