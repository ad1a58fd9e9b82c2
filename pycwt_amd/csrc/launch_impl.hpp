// launch_impl.hpp -- every kernel launch of the library, as templates over the arithmetic type T (float | double): twiddle
// tables, forward FFT, the row forms (single workgroup, band-limited, two-pass, overlap-save, band-passed, polynomial) and
// their scheduling over the plan's streams (rows_launch), filter tables, the callers' kernels (coherence helpers, reductions,
// Bluestein).  Included by launch_f64.hip / launch_f32.hip, which instantiate it for ONE precision each (the two halves of the
// device code compile side by side), and by abi.hip, which only sees the declarations (extern templates at the end).
#pragma once
#include <hip/hip_runtime.h>

#include "plan.hpp"
#include "cwt_kernels.hpp"

namespace cwtd {

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
inline void narrow_class_counts(const cwt_plan* p, int* n_small_k, int* n_big, int* n_many = nullptr) {
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
  if constexpr (MODE == IN_REAL && LOGR <= LOGP - 1 && LOGR >= 8) {
    if (p->fft_small) {      // the forward FFT beside the overlap-save rows (serial_rows = 2): half-size tiles get their turn on the CUs
      constexpr int LP = LOGP - 1;
      hipLaunchKernelGGL((k_pass_a_ct<T, LOGR, LP, MODE>), dim3(1u << (p->logN - LP), cnt), dim3(1 << (LP - 4)),
                         (size_t(1) << LP) * sizeof(T), st, in, rows, mo, tw_table<T>(p, LOGR), twn_of<T>(p), p->logN, n0, in_ld, Z);
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
  if constexpr (CONJ && LOGK <= LOGP - 1) {
    if (p->fft_small) return launch_pass_b_ct_lp<T, LOGK, LOGP - 1, CONJ>(p, rows, cnt, W, ldw, ncols, Z, st);
  }
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
// (g_only / d_only >= 0: only that tile group / only its blocks of 2^d tiles; d_only = -2: every block length but one tile)
template <typename T>
int launch_ols_fwd(cwt_plan* p, const void* x_dev, int64_t n0, hipStream_t st, hipEvent_t after_first = nullptr, int g_only = -1,
                   int d_only = -1) {
  int rc = CWT_OK;
  {
    for (int g = 0; g < 2 && !rc; ++g) {
      if (g == 1 && after_first) HIPCHECK(hipEventRecord(after_first, st));     // the half-size tiles' spectra exist
      const auto& G = p->rt->ols_grp[g];
      if (!G.nrows || (g_only >= 0 && g != g_only)) continue;
      for (int d = 0; d < 3 && !rc; ++d) {
        if (!G.fwd_blocks[d] || (d_only >= 0 && d != d_only) || (d_only == -2 && d == 0)) continue;
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
// part: -1 = every class of the group in one launch, 0 = the classes on blocks of one tile, 1 = the classes on longer blocks
template <typename T, int LOGP>
int launch_ols_rows_p(cwt_plan* p, int g, cplx<T>* W, int64_t ldw, int64_t ncols, hipStream_t st, int part = -1) {
  const cwt_plan::RowTable* rt = p->rt;
  const auto& G = rt->ols_grp[g];
  const long first = part == 1 ? G.wgs_base : 0, count = part == 0 ? G.wgs_base : G.wgs - first;
  if (count <= 0) return CWT_OK;
  static const bool once = (allow_big_lds(&k_ols_ct<T, LOGP>), true);
  (void)once;
  // (complex64: two blocks per workgroup, 8-byte exchange elements)
  const size_t lds = ((size_t(1) << LOGP) + (size_t(1) << (LOGP - 4))) * (ols_pairs(sizeof(T), LOGP) ? sizeof(pairf) : sizeof(T));
  return timed_launch(p, g == 0 ? KC_OLS_SMALL : KC_OLS, [&] {
    hipLaunchKernelGGL((k_ols_ct<T, LOGP>), dim3(unsigned(count)), dim3(1 << (LOGP - 4)), lds, st,
                       static_cast<const cplx<T>*>(p->xs), rt->rows_dev + rt->ols_first + G.row_first,
                       static_cast<const cplx<T>*>(rt->gt_dev), static_cast<const cplx<T>*>(p->tw_all), twn_of<T>(p),
                       p->logN, G.cls, W, long(ldw), long(ncols), unsigned(first));
  }, st);
}
template <typename T>
int launch_ols_rows(cwt_plan* p, cplx<T>* W, int64_t ldw, int64_t ncols, hipStream_t st, int g_only = -1, int part = -1) {
  int rc = CWT_OK;
  for (int g = 0; g < 2 && !rc; ++g) {        // the half-size tiles first (by far the longer launch since the rows with long
                                              // halos went to the polynomial form), then the default tile's rows
    const auto& G = p->rt->ols_grp[g];
    if (!G.nrows || (g_only >= 0 && g != g_only)) continue;
    switch (G.logp) {
      case 12: rc = launch_ols_rows_p<T, 12>(p, g, W, ldw, ncols, st, part); break;
      case 13: rc = launch_ols_rows_p<T, 13>(p, g, W, ldw, ncols, st, part); break;
      default: return fail(CWT_EINVAL, "overlap-save tile size");
    }
  }
  return rc;
}

// Rows clipped at Nyquist (k_aols_*): band-passed complex signal x_M = IFFT_N(xhat mask) through the two-pass kernels
// (the mask is the pseudo-row at aux_first: profile 1), its block spectra, then every (block, row) pair.
// ready != nullptr (one signal only): the band-passed signal and its block spectra on st, `ready` recorded behind them,
// the rows on st_rows behind that event.
// phase (one signal, ready != nullptr): 0 = everything, 1 = the band-passed signal and its block spectra on st + `ready` recorded,
// 2 = the rows on st_rows, which the CALLER has made wait for `ready`
template <typename T, int LOGP>
int launch_aols_p(cwt_plan* p, const void* xhat_dev, cplx<T>* W, int64_t ldw, int64_t ncols, hipStream_t st,
                  hipStream_t st_rows = nullptr, hipEvent_t ready = nullptr, int phase = 0) {      // (ready == nullptr: everything on st; the caller's stream may be the null stream)
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
  const size_t lds_rows = aols_pairs(sizeof(T)) ? 2 * lds : lds;      // complex64 rows: two blocks per workgroup
  for (int b0 = 0; b0 < nb; b0 += chunk) {
    const int cnt = std::min(chunk, nb - b0);
    bool ok = true;
    if (phase != 2) rc = timed_launch(p, KC_AOLS_PRE, [&] {
      ok = try_pass_a_ct<T, IN_SPECTRUM>(p, logR, xhat_dev, rt->rows_dev + rt->aux_first + b0, cnt, one, 0L, 0L, Z, st);
    }, st);
    if (!rc && !ok) rc = fail(CWT_EINVAL, "k_aols rows need the default geometry");
    if (!rc && phase != 2) rc = timed_launch(p, KC_AOLS_PRE, [&] {
      // (beside the overlap-save rows the default tile's 512-thread workgroups do not find a CU before those drain: 170-250 us
      // for 16; 256-thread workgroups get their turn)
      if (p->aols_small_b && ready && logK == 10 && default_logp<T>() == 13 && p->use_ct) { launch_pass_b_ct_lp<T, 10, 12, false>(p, nullptr, cnt, xm, p->N, p->N, Z, st); ok = true; }
      else ok = try_pass_b_ct<T, false>(p, logK, nullptr, cnt, xm, p->N, p->N, Z, st);
    }, st);
    if (!rc && !ok) rc = fail(CWT_EINVAL, "k_aols rows need the default geometry");
    if (!rc && phase != 2) rc = timed_launch(p, KC_AOLS_PRE, [&] {
      hipLaunchKernelGGL((k_aols_fwd<T, LOGP>), dim3(unsigned(g.nblocks), unsigned(cnt)), dim3(1 << (LOGP - 4)), lds, st, xm,
                         p->logN, g.halo, static_cast<const cplx<T>*>(p->tw_all), static_cast<cplx<T>*>(p->xsa));
    }, st);
    hipStream_t sr = st;
    if (!rc && ready && nb == 1) {
      if (phase != 2) HIPCHECK(hipEventRecord(ready, st));
      if (phase == 0) HIPCHECK(hipStreamWaitEvent(st_rows, ready, 0));
      sr = st_rows;
    }
    if (phase == 1) return rc;
    if (!rc) rc = timed_launch(p, KC_AOLS, [&] {
      hipLaunchKernelGGL((k_aols_rows<T, LOGP>), dim3(unsigned(rt->aols_wgs), unsigned(cnt)), dim3(1 << (LOGP - 4)), lds_rows, sr,
                         static_cast<const cplx<T>*>(p->xsa), rt->rows_dev + rt->aols_first + long(b0) * g.nrows,
                         static_cast<const T*>(rt->agt_dev), static_cast<const cplx<T>*>(p->tw_all), g,
                         static_cast<const cplx<T>*>(xhat_dev), long(p->N >> 1), W, long(ldw), long(ncols));
    }, sr);
    if (rc) return rc;
  }
  return CWT_OK;
}
// The second class of such rows (Paul continued through f = 0, 8192-point tiles; one signal): block spectra of the SAME
// band-passed signal on its own block grid, then its rows -- on the stream of the first class, behind it (both use p->xsa).
template <typename T>
int launch_aols_second(cwt_plan* p, const void* xhat_dev, cplx<T>* W, int64_t ldw, int64_t ncols, hipStream_t st) {
  const cwt_plan::RowTable* rt = p->rt;
  const AolsGeom& g = rt->aols2_geom;
  constexpr int LOGP = 13, P = 1 << LOGP;
  int rc = grow(&p->xsa, &p->xsa_bytes, size_t(g.nblocks) * size_t(P + 8) * sizeof(cplx<T>), st);
  if (rc) return rc;
  static const bool once = (allow_big_lds(&k_aols_fwd<T, LOGP>), allow_big_lds(&k_aols_rows<T, LOGP>), true);
  (void)once;
  const size_t lds = ((size_t(1) << LOGP) + (size_t(1) << (LOGP - 4))) * sizeof(T);
  rc = timed_launch(p, KC_AOLS_PRE, [&] {
    hipLaunchKernelGGL((k_aols_fwd<T, LOGP>), dim3(unsigned(g.nblocks), 1u), dim3(1 << (LOGP - 4)), lds, st,
                       static_cast<const cplx<T>*>(p->xm), p->logN, g.halo, static_cast<const cplx<T>*>(p->tw_all),
                       static_cast<cplx<T>*>(p->xsa));
  }, st);
  if (!rc) rc = timed_launch(p, KC_AOLS, [&] {
    hipLaunchKernelGGL((k_aols_rows<T, LOGP>), dim3(unsigned(rt->aols2_wgs), 1u), dim3(1 << (LOGP - 4)),
                       aols_pairs(sizeof(T)) ? 2 * lds : lds, st,
                       static_cast<const cplx<T>*>(p->xsa), rt->rows_dev + rt->aols2_first, static_cast<const T*>(rt->agt_dev),
                       static_cast<const cplx<T>*>(p->tw_all), g, static_cast<const cplx<T>*>(xhat_dev), long(p->N >> 1), W,
                       long(ldw), long(ncols));
  }, st);
  return rc;
}
template <typename T>
int launch_aols(cwt_plan* p, const void* xhat_dev, cplx<T>* W, int64_t ldw, int64_t ncols, hipStream_t st,
                hipStream_t st_rows = nullptr, hipEvent_t ready = nullptr, int phase = 0) {
  int rc = CWT_OK;
  switch (p->rt->aols_logp) {
    case 12: rc = launch_aols_p<T, 12>(p, xhat_dev, W, ldw, ncols, st, st_rows, ready, phase); break;
    default: return fail(CWT_EINVAL, "k_aols tile size");
  }
  if (!rc && p->rt->n_aols2 && phase != 1) rc = launch_aols_second<T>(p, xhat_dev, W, ldw, ncols, ready ? st_rows : st);
  return rc;
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
  const cplx<T>* tw = static_cast<const cplx<T>*>(p->tw_all);
  const T* rtab = static_cast<const T*>(rt->prt_dev);       // economised weights of the (K', D) pairs (rows with rtab_off >= 0)
  if (rt->poly_rtab_elems > 0 && (!rtab || rt->prt_bytes < size_t(rt->poly_rtab_elems) * sizeof(T)))
    return fail(CWT_EINVAL, "polynomial rows without their weight tables (fill_poly_tables was not run for this row table)");
  if (p->coef_small)         // every class on 256-thread workgroups, one launch (k_poly_coef_all)
    return timed_launch(p, KC_POLY_COEF, [&] {
      hipLaunchKernelGGL((k_poly_coef_all<T>), dim3(unsigned(ch.wgs_all)), dim3(256), ((size_t(1) << 12) + (size_t(1) << 8)) * sizeof(T), st,
                         static_cast<const cplx<T>*>(band), rows, tw, ch.cls, coef, rtab); }, st);
  // largest tiles first: the 16384-point workgroups take a whole CU each and should find the chip as empty as it gets
  auto lds_of = [](int lp) { return ((size_t(1) << lp) + (size_t(1) << (lp - 4))) * sizeof(T); };
  const bool split = st2 && ch.wgs[2] && (ch.wgs[1] || ch.wgs[0]);
  hipStream_t s2 = split ? st2 : st;
  if (split) {
    HIPCHECK(hipEventRecord(p->ev_big, st));             // the bands are ready
    HIPCHECK(hipStreamWaitEvent(st2, p->ev_big, 0));
  }
  if (!rc && ch.wgs[2]) rc = timed_launch(p, KC_POLY_COEF, [&] {
    hipLaunchKernelGGL((k_poly_coef<T, 14>), dim3(unsigned(ch.wgs[2])), dim3(1024), lds_of(14), st,
                       static_cast<const cplx<T>*>(band), rows, tw, ch.cls, coef, rtab); }, st);
  if (!rc && ch.wgs[1]) rc = timed_launch(p, KC_POLY_COEF, [&] {
    hipLaunchKernelGGL((k_poly_coef<T, 13>), dim3(unsigned(ch.wgs[1])), dim3(512), lds_of(13), s2,
                       static_cast<const cplx<T>*>(band), rows, tw, ch.cls, coef, rtab); }, s2);
  if (!rc && ch.wgs[0]) rc = timed_launch(p, KC_POLY_COEF, [&] {
    hipLaunchKernelGGL((k_poly_coef<T, 12>), dim3(unsigned(ch.wgs[0])), dim3(256), lds_of(12), s2,
                       static_cast<const cplx<T>*>(band), rows, tw, ch.cls, coef, rtab); }, s2);
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

// Two-pass rows (forms T), chunk by chunk on the plan's stream through the one intermediate buffer.
template <typename T>
int launch_wide_rows(cwt_plan* p, const void* xhat_dev, const Mother& mo, cplx<T>* W, int64_t ldw, int64_t ncols) {
  const int logN = p->logN, logP = std::min(p->log_wg_points, logN), threads = 1 << (logP - 4);
  const size_t lds = (size_t(1) << logP) * sizeof(T);
  const int logK = two_pass_logk(p), logR = logN - logK;
  const int chunk = balanced_chunk(p, p->rt->n_wide);
  const int nchunks = (p->rt->n_wide + chunk - 1) / chunk;
  int rc = ensure_z(p, chunk);
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
  return CWT_OK;
}

// The schedule of a long transform with polynomial rows (option "serial_rows", default): every kernel that WRITES W runs
// on the caller's stream, one after the other -- overlap-save rows (half-size tiles, then the default tile), rows on the
// band-passed signal, two-pass rows, polynomial rows -- and everything they need is prepared on the side streams beside the
// first of them: block spectra on side stream 1 (queued before the forward FFT by cwt_transform), behind them the band-passed
// signal and its block spectra; bands + interval coefficients on side stream 0 (+ side2).  Why: heavy kernels side by side cost
// more than one after the other (EXPERIMENTS R5.2), a stream-to-stream hand-over costs 15-20 us where the waiting stream is idle
// -- so the hand-overs sit where the event completed long before the wait is reached, and the step ends on the caller's stream
// (the next call, or whatever the caller queues, follows at a kernel boundary instead of a join).
template <typename T>
int rows_launch_serial(cwt_plan* p, const void* xhat_dev, const Mother& mo, void* W_dev, int64_t ldw, int64_t ncols,
                       hipEvent_t spectrum_ready) {
  const cwt_plan::RowTable* rt = p->rt;
  const cplx<T>* xhat = static_cast<const cplx<T>*>(xhat_dev);
  cplx<T>* W = static_cast<cplx<T>*>(W_dev);
  hipStream_t M = p->stream, S0 = p->side[0], S1 = p->side[1];
  int rc = CWT_OK;
  // the one intermediate buffer serves the band-passed signal (side stream 1) and the two-pass rows (caller's stream): sized
  // for both before either is queued
  if (rt->n_aols || rt->n_wide) rc = ensure_z(p, rt->n_wide ? balanced_chunk(p, rt->n_wide) : 1);
  if (rc) return rc;
  if (!spectrum_ready) {                                  // the forward FFT ran on the caller's stream
    spectrum_ready = p->ev_a[1];
    HIPCHECK(hipEventRecord(spectrum_ready, M));
  }
  if (rt->n_poly) {
    HIPCHECK(hipStreamWaitEvent(S0, spectrum_ready, 0));
    rc = launch_poly_coef<T>(p, xhat, mo, 0, S0, p->side2);
    if (rc) return rc;
    HIPCHECK(hipEventRecord(p->ev_a[0], S0));
  }
  // side stream 1 is ONE in-order chain -- block spectra of the longer blocks / of the default tile (queued by cwt_transform), then
  // the band-passed signal and its block spectra -- so the caller's stream waits for its END once, behind the first overlap-save
  // launch (the chain is long done then: 130 of 250 us), instead of once per consumer: a wait costs the stream 7-8 us, a kernel
  // boundary 2 [measured]
  const bool s1_once = rt->n_aols && p->serial_rows != 3 && p->serial_s1_once;
  if (rt->n_aols) HIPCHECK(hipStreamWaitEvent(S1, spectrum_ready, 0));   // (the rows wait for the band-passed signal, made from the spectrum)
  if (s1_once) {
    rc = launch_aols<T>(p, xhat_dev, W, ldw, ncols, S1, M, p->ev_b[1], 1);
    if (rc) return rc;
  }
  const bool g0_split = p->ols_first_on_main && rt->ols_grp[0].wgs > rt->ols_grp[0].wgs_base;   // longer blocks on the half-size tiles:
  if (rt->n_ols) {                                        // block spectra queued by cwt_transform on side stream 1
    if (!p->ols_first_on_main) HIPCHECK(hipStreamWaitEvent(M, p->ev_b[0], 0));
    rc = launch_ols_rows<T>(p, W, ldw, ncols, M, 0, g0_split ? 0 : -1);   // their spectra come from side stream 1, behind ev_ols
    if (rc) return rc;
  }
  // serial_rows = 3: ONE wait on the caller's stream for everything the side streams prepare (each wait is a barrier packet that
  // costs the stream 5-8 us even when its event completed long ago): side stream 1 = block spectra of the default tile, then
  // (behind the coefficients' event) the band-passed signal, then the event the rows of that signal wait for -- which therefore
  // come before the default tile's rows.
  const bool one_wait = p->serial_rows == 3 && rt->n_aols && rt->n_poly;
  if (one_wait) {
    HIPCHECK(hipStreamWaitEvent(S1, p->ev_a[0], 0));
    rc = launch_aols<T>(p, xhat_dev, W, ldw, ncols, S1, M, p->ev_b[1]);
    if (rc) return rc;
  }
  if (s1_once) HIPCHECK(hipStreamWaitEvent(M, p->ev_b[1], 0));
  if (rt->n_ols && (rt->ols_grp[1].nrows || g0_split)) {
    if (!one_wait && !s1_once) HIPCHECK(hipStreamWaitEvent(M, p->ev_ols, 0));
    if (g0_split) rc = launch_ols_rows<T>(p, W, ldw, ncols, M, 0, 1);
    if (!rc && rt->ols_grp[1].nrows) rc = launch_ols_rows<T>(p, W, ldw, ncols, M, 1);
    if (rc) return rc;
  }
  if (rt->n_aols && !one_wait) {   // band-passed signal + block spectra on side stream 1 (behind the block spectra of the signal: they
                                   // have the two overlap-save launches to get done), the rows on the caller's stream
    rc = launch_aols<T>(p, xhat_dev, W, ldw, ncols, S1, M, p->ev_b[1], s1_once ? 2 : 0);
    if (rc) return rc;
  }
  // The polynomial rows BEFORE the two-pass rows: k_poly_rows starts every workgroup with a fetch of its coefficient sets and runs
  // at the store rate only while the planes sit in the Infinity Cache; a two-pass chunk in between moves ~200 MB through it
  // (fp64 Paul, 73 MB of planes, [measured]: 4.65 us per row behind the two-pass rows, 2.9 in front of them).
  if (rt->n_poly) {
    if (!one_wait) HIPCHECK(hipStreamWaitEvent(M, p->ev_a[0], 0));
    const int nchunks = int(rt->poly_chunks.size());
    for (int c = 0; c < nchunks && !rc; ++c) {            // chunk c's rows, then chunk c + 1's coefficients
      rc = launch_poly_rows<T>(p, c, W, ldw, ncols, M);
      if (!rc && c + 1 < nchunks) rc = launch_poly_coef<T>(p, xhat, mo, c + 1, M, nullptr);
    }
    if (rc) return rc;
  }
  if (rt->n_wide) {
    HIPCHECK(hipStreamWaitEvent(M, spectrum_ready, 0));
    rc = launch_wide_rows<T>(p, xhat_dev, mo, W, ldw, ncols);
  }
  return rc;
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
  if (serial_schedule(p, ols_early))
    return rows_launch_serial<T>(p, xhat_dev, mo, W_dev, ldw, ncols, p->spectrum_ready);
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
  // polynomial rows on the plan's own stream: before the two-pass rows, while their planes sit in the Infinity Cache (see
  // rows_launch_serial)
  const bool poly_first = p->rt->n_poly && !poly_on_side && p->rt->n_wide;
  auto poly_rows_on = [&](hipStream_t ps) {
    const int nchunks = int(p->rt->poly_chunks.size());
    int r = CWT_OK;
    for (int c = 0; c < nchunks && !r; ++c) {              // chunk c's rows, then chunk c + 1's coefficients, on one stream
      r = launch_poly_rows<T>(p, c, W, ldw, ncols, ps);
      if (!r && c + 1 < nchunks) r = launch_poly_coef<T>(p, xhat, mo, c + 1, ps, poly_on_side ? p->side2 : nullptr);
    }
    return r;
  };
  if (poly_first) {
    rc = poly_rows_on(p->stream);
    if (rc) return rc;
  }
  if (p->rt->n_wide) {                 // two-pass rows, chunk by chunk on the plan's stream (one intermediate buffer)
    rc = launch_wide_rows<T>(p, xhat_dev, mo, W, ldw, ncols);
    if (rc) return rc;
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
  if (p->rt->n_poly && !poly_first) {  // second half, on the same stream as the first (joined below when that is a side stream)
    narrow_on_side = poly_on_side;
    rc = poly_rows_on(poly_on_side ? p->side[0] : p->stream);
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

inline Mother mother_of(int mother, double param) {
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
  if (t->n_aols2) {    // the second class (8192-point tiles): its rows carry the offsets of their tables behind the first's
    const dim3 grid2((1 << 13) / 256, t->aols2_geom.nrows);
    if (mo.kind == MOTHER_MORLET) hipLaunchKernelGGL((k_aols_gtab<T, MOTHER_MORLET>), grid2, block, 0, p->stream, t->rows_dev + t->aols2_first, mo, 13, t->aols2_geom, gt);
    else hipLaunchKernelGGL((k_aols_gtab<T, MOTHER_PAUL>), grid2, block, 0, p->stream, t->rows_dev + t->aols2_first, mo, 13, t->aols2_geom, gt);
  }
  HIPCHECK(hipGetLastError());
  return CWT_OK;
}

// Tables of the economised monomial weights of the polynomial rows (k_poly_rtab), one per (K', D) pair, on the plan's stream.
template <typename T>
int fill_poly_tables(cwt_plan* p) {
  cwt_plan::RowTable* t = p->rt;
  int rc = grow(&t->prt_dev, &t->prt_bytes, size_t(t->poly_rtab_elems) * sizeof(T), p->stream);
  if (rc) return rc;
  T* out = static_cast<T*>(t->prt_dev);
  for (const auto& e : t->poly_rtabs)
    hipLaunchKernelGGL((k_poly_rtab<T>), dim3(((1u << e.logK) + 256u) / 256u, unsigned(e.deg + 1)), dim3(256), 0, p->stream, e.logK, e.deg,
                       out + e.off);
  HIPCHECK(hipGetLastError());
  return CWT_OK;
}

template <typename T>
int launch_ols_early(cwt_plan* p, const void* x_dev, int64_t n0, void* W_dev, int64_t ldw, int64_t ncols) {
  int rc = grow(&p->xs, &p->xs_bytes, size_t(p->rt->ols_xs_elems) * sizeof(cplx<T>), p->stream);
  if (rc) return rc;
  HIPCHECK(hipEventRecord(p->ev_fork, p->stream));        // after the previous call's work and the row-table upload
  HIPCHECK(hipStreamWaitEvent(p->side[1], p->ev_fork, 0));
  if (p->ols_first_on_main) {                           // serial_rows = 2: the first rows' spectra where the rows will follow
    rc = launch_ols_fwd<T>(p, x_dev, n0, p->stream, nullptr, 0, 0);
    if (!rc) rc = launch_ols_fwd<T>(p, x_dev, n0, p->side[1], nullptr, 0, -2);    // (the half-size tiles' longer blocks)
    if (!rc) rc = launch_ols_fwd<T>(p, x_dev, n0, p->side[1], nullptr, 1);
  } else {
    rc = launch_ols_fwd<T>(p, x_dev, n0, p->side[1], p->ev_b[0]);     // (the rows follow in rows_launch)
  }
  if (rc) return rc;
  HIPCHECK(hipEventRecord(p->ev_ols, p->side[1]));      // serial schedule: all block spectra exist (the other one records it again behind the rows)
  (void)W_dev; (void)ldw; (void)ncols;
  p->ols_launched = 1;
  return CWT_OK;
}

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

template <typename T, bool POWER>
int reduce_scales_impl(cwt_plan* p, const void* W_dev, int64_t ldw, int64_t ncols, int nrows,
                       const double* weights, double coeff, void* out_dev) {
  int rc = upload_reals<T>(p, weights, nrows);
  if (rc) return rc;
  const unsigned blocks = unsigned((ncols + ICWT_THREADS - 1) / ICWT_THREADS);
  return timed_launch(p, KC_ICWT, [&] {
    hipLaunchKernelGGL((k_icwt<T, POWER>), dim3(blocks), dim3(ICWT_THREADS), 0, p->stream,
                       static_cast<const cplx<T>*>(W_dev), long(ldw), long(ncols), nrows,
                       static_cast<const T*>(p->weights_dev), T(coeff), static_cast<T*>(out_dev));
  });
}

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

// Surrogate series on the device (cwt_random_normal, cwt_ar1_filter)
template <typename T>
int random_normal_impl(cwt_plan* p, uint64_t seed, uint64_t offset, int64_t n, double scale, void* out) {
  const unsigned blocks = unsigned(((n + 1) / 2 + 255) / 256);
  return timed_launch(p, KC_ELEMENTWISE, [&] {
    hipLaunchKernelGGL((k_normal_fill<T>), dim3(blocks), dim3(256), 0, p->stream, (unsigned long long)seed,
                       (unsigned long long)offset, long(n), scale, static_cast<T*>(out));
  });
}
template <typename T>
int ar1_filter_impl(cwt_plan* p, const void* e, int64_t tau, int64_t n, double g, void* out) {
  // history a segment may forget: g^warm <= 1e-17
  const double lg = std::log(std::fabs(g));
  const int64_t warm = lg < 0 ? std::min<int64_t>(int64_t(std::ceil(39.2 / -lg)) + 1, tau + n) : tau + n;
  const int seg = int(std::max<int64_t>(64, std::min<int64_t>(4096, warm / 4)));    // warm-up at most ~4x the useful work
  const unsigned blocks = unsigned(((n + seg - 1) / seg + 255) / 256);
  return timed_launch(p, KC_ELEMENTWISE, [&] {
    hipLaunchKernelGGL((k_ar1_filter<T>), dim3(blocks), dim3(256), 0, p->stream, static_cast<const T*>(e), long(tau), long(n), g,
                       long(warm), seg, static_cast<T*>(out));
  });
}

// ---- one precision per translation unit ----------------------------------------------------------------------------------------
#define CWT_LAUNCH_TEMPLATES(X, T)                                                                                                  \
  X int build_tables<T>(cwt_plan*);                                                                                                 \
  X int set_func_attrs<T>();                                                                                                        \
  X int fft_rows_impl<T, IN_REAL>(cwt_plan*, const void*, int64_t, int, int64_t, void*);                                            \
  X int fft_rows_impl<T, IN_CPLX>(cwt_plan*, const void*, int64_t, int, int64_t, void*);                                            \
  X int rows_impl<T>(cwt_plan*, const void*, const Mother&, int, void*, int64_t, int64_t, const void*, int64_t);                    \
  X int fill_ols_tables<T>(cwt_plan*, const Mother&);                                                                               \
  X int fill_aols_tables<T>(cwt_plan*, const Mother&);                                                                              \
  X int fill_poly_tables<T>(cwt_plan*);                                                                                             \
  X int launch_ols_early<T>(cwt_plan*, const void*, int64_t, void*, int64_t, int64_t);                                              \
  X int wct_products_impl<T>(cwt_plan*, const void*, const void*, const double*, int, int64_t, int64_t, void*, void*, void*);       \
  X int boxcar_impl<T>(cwt_plan*, const void*, int, int64_t, int64_t, const double*, int, void*);                                   \
  X int coherence_impl<T>(cwt_plan*, const void*, const void*, int, int64_t, int64_t, void*);                                       \
  X int reduce_scales_impl<T, true>(cwt_plan*, const void*, int64_t, int64_t, int, const double*, double, void*);                   \
  X int reduce_scales_impl<T, false>(cwt_plan*, const void*, int64_t, int64_t, int, const double*, double, void*);                  \
  X int random_normal_impl<T>(cwt_plan*, uint64_t, uint64_t, int64_t, double, void*);                                             \
  X int ar1_filter_impl<T>(cwt_plan*, const void*, int64_t, int64_t, double, void*);                                               \
  X int forward_fft_n_impl<T>(cwt_plan*, const void*, int64_t, void*);                                                              \
  X int transform_rows_n_impl<T>(cwt_plan*, const void*, int64_t, int, double, double, const double*, int, void*, int64_t);

#ifndef CWT_LAUNCH_TU
CWT_LAUNCH_TEMPLATES(extern template, double)
CWT_LAUNCH_TEMPLATES(extern template, float)
#endif

}  // namespace cwtd
