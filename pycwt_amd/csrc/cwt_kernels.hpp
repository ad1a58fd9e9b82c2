// cwt_kernels.hpp -- the HIP kernels of the CWT hot path (gfx950).
//
// Math (pycwt/wavelet.py:91-106): W[j, n] = (1/N) sum_k xhat[k] F_j[k] e^{+2 pi i k n / N},
// F_j[k] = sqrt(2 pi s_j / dt) conj(psi_ft(s_j w_k)).  F_j is never stored: every kernel that
// consumes the spectrum evaluates profile(s_j w_k) on the fly and multiplies by a per-row
// complex amplitude that carries the norm, the mother's constant and the 1/N of the inverse FFT.
//
// Kernels (T = float | double):
//   k_small   N <= lmax            one workgroup FFT per row (also the forward FFT of the signal)
//   k_direct  N <= 8               plain DFT (sizes below the radix-16 engine)
//   k_narrow  band-limited rows    single pass: aliased K_j-point FFTs, N = K_j * R_j
//   k_pass_a  wide rows, pass 1    column FFTs over k1 (k = q + K k1)
//   k_pass_b  wide rows, pass 2    twiddle e^{2 pi i q r / N} on load, row FFTs over q, LDS transpose, store W[R m + r]
// (the inter-pass twiddle sits in pass B, which is memory bound, rather than in pass A, which is VALU / latency
// bound: measured pass A -5 % fp64 / -16 % fp32, pass B +2 %)
//   k_icwt    TC98 eq. 11 column reduction (wavelet.py:169-170)
#pragma once
#include <hip/hip_runtime.h>

#include "fft_engine.hpp"

#ifndef CWT_MAX_THREADS
#define CWT_MAX_THREADS 1024
#endif
// Minimum resident waves per SIMD the compiler must allow for (second __launch_bounds__ argument, i.e. the VGPR
// budget: 4 -> 128, 5 -> 96, 6 -> 80, 8 -> 64 registers) of the compile-time kernels, per precision.  Measured
// defaults; override with -D for tuning runs.
// Overlap-save block transforms with TB <= 2^this residues per tile use the padded exchange layout.  Measured with 3 and 4:
// the bank conflicts of the TB = 8 / 16 stage-0 writes (16-36 % of those kernels' LDS cycles) go away, the step does not move.
#ifndef CWT_OLS_PAD_LOGTB
#define CWT_OLS_PAD_LOGTB 2
#endif
#ifndef CWT_LB_NARROW_F64
#define CWT_LB_NARROW_F64 4
#endif
#ifndef CWT_LB_NARROW_F32
#define CWT_LB_NARROW_F32 4
#endif
#ifndef CWT_LB_NARROW_F32_BIG
#define CWT_LB_NARROW_F32_BIG 8   // the 16384-point (1024-thread) tiles of the fp32 K = 1024 rows: 64 VGPRs -> TWO workgroups per CU
#endif                            // instead of one (2.50 -> 1.79 us per row); the K <= 512 rows on 8192-point tiles lose at 6 and 8
#ifndef CWT_LB_OLS_F64
#define CWT_LB_OLS_F64 4
#endif
#ifndef CWT_LB_OLS_F32
#define CWT_LB_OLS_F32 6      // 80 VGPRs -> three 512-thread workgroups per CU: overlap-save kernel -7 % (fp32 DOG / Paul)
#endif
#ifndef CWT_LB_OLS_F32_HALF
#define CWT_LB_OLS_F32_HALF 6 // the same kernel on half-size tiles (256 threads)
#endif
#ifndef CWT_LB_OLS_F64_HALF
#define CWT_LB_OLS_F64_HALF 4
#endif
#ifndef CWT_LB_PASS_A_F64
#define CWT_LB_PASS_A_F64 4
#endif
#ifndef CWT_LB_PASS_A_F32
#define CWT_LB_PASS_A_F32 4
#endif
#ifndef CWT_LB_PASS_B_F64
#define CWT_LB_PASS_B_F64 4
#endif
#ifndef CWT_LB_PASS_B_F32
#define CWT_LB_PASS_B_F32 8
#endif
namespace cwt {

enum : int { MOTHER_MORLET = 0, MOTHER_PAUL = 1, MOTHER_DOG = 2, MOTHER_TABLE = 3 };
enum : int { IN_SPECTRUM = 0, IN_REAL = 1, IN_CPLX = 2 };   // IN_CPLX: complex rows, conjugated on load

// One row (scale) of the transform, prepared on the host in double precision.
struct RowDesc {
  double a;        // s_j * 2 pi / (N dt): profile argument = a * signed bin index
  double amp_re;   // complex amplitude: sqrt(s w_1 N) * mother constant / N  (conj applied)
  double amp_im;
  int k_lo;        // first signed bin index of the filter's support, >= -N/2
  int nband;       // number of bins in the support; k_lo + nband - 1 <= N/2 - 1
  int out_row;     // destination row of W
  int logK;        // k_narrow: log2 of this row's FFT length
  int nterms;      // k_narrow_ct: ceil(nband / K) aliased bins per FFT input (1 unless K = 1024)
  long spec_off;   // element offset of this row's spectrum (0: all rows share one spectrum)
  long tab_off;    // MOTHER_TABLE: element offset of this row's explicit filter F_j[0..N); rows with tables or coefficient
                   // planes of their own (overlap-save, polynomial): element offset of those
  long aux_off;    // polynomial rows: element offset of the row's filtered band (k_poly_band)
  double nyq_re;   // k_aols rows of a two-sided real filter (DOG): F_j at the Nyquist bin / N, the one bin outside the mask
  double nyq_im;   //   and its mirror image
};

struct Mother {
  int kind;           // MOTHER_*
  int m;              // integer order for Paul / DOG
  double p;           // f0 (Morlet) or m
  const void* table;  // MOTHER_TABLE: rows x N complex filter bank on the device (custom mothers)
};

// Tables for e^{2 pi i t / N}, t < N, as a product of a coarse and a fine root of unity.
template <typename T>
struct TwN {
  const cplx<T>* hi;  // hi[i] = e^{2 pi i (i << shift) / N}
  const cplx<T>* lo;  // lo[i] = e^{2 pi i i / N}, i < (1 << shift)
  int shift;
  __device__ __forceinline__ cplx<T> operator()(unsigned t) const {
    return cmul<T>(hi[t >> shift], lo[t & ((1u << shift) - 1u)]);
  }
};

// exp(x) for x <= 0 (every profile argument is non-positive inside the filter's support).
// fp64: n = rint(x*log2 e), Cody-Waite reduction to |f| <= ln2/2, degree-13 Taylor polynomial
// (truncation 4e-18), ldexp; ~19 instructions and 1-2 ulp, against ~40 for the library call.
__device__ __forceinline__ double exp_(double x) {
  const double n = rint(x * 1.4426950408889634074);
  double f = fma(n, -6.93147180369123816490e-01, x);
  f = fma(n, -1.90821492927058770002e-10, f);
  double p = 1.6059043836821614599e-10;              // 1/13!
  p = fma(p, f, 2.0876756987868098979e-09);          // 1/12!
  p = fma(p, f, 2.5052108385441718775e-08);          // 1/11!
  p = fma(p, f, 2.7557319223985890653e-07);          // 1/10!
  p = fma(p, f, 2.7557319223985890653e-06);          // 1/9!
  p = fma(p, f, 2.4801587301587301587e-05);          // 1/8!
  p = fma(p, f, 1.9841269841269841270e-04);          // 1/7!
  p = fma(p, f, 1.3888888888888888889e-03);          // 1/6!
  p = fma(p, f, 8.3333333333333333333e-03);          // 1/5!
  p = fma(p, f, 4.1666666666666666667e-02);          // 1/4!
  p = fma(p, f, 1.6666666666666666667e-01);          // 1/3!
  p = fma(p, f, 0.5);
  p = fma(p, f, 1.0);
  p = fma(p, f, 1.0);
  const double nc = n < -1100.0 ? -1100.0 : n;       // deep underflow -> 0 without int overflow
  return ldexp(p, int(nc));
}
__device__ __forceinline__ float exp_(float x) { return __expf(x); }

template <typename T>
__device__ __forceinline__ T ipow(T b, int e) {
  T r = T(1);
  while (e > 0) {
    if (e & 1) r *= b;
    b *= b;
    e >>= 1;
  }
  return r;
}

// Real profile of psi_ft at f = s*w (mothers.py:26-28, 118-122, 170-173) without the mother's
// constant factor; Paul is 0 for f <= 0 (the mathematically intended value, see cwt_hip.h).
template <typename T>
__device__ __forceinline__ T profile(const Mother& mo, T f) {
  if (mo.kind == MOTHER_MORLET) {
    const T d = f - T(mo.p);
    return exp_(T(-0.5) * d * d);
  }
  const T pw = ipow<T>(f, mo.m);
  if (mo.kind == MOTHER_PAUL) return f > T(0) ? pw * exp_(-f) : T(0);
  return pw * exp_(T(-0.5) * f * f);
}

// x * F_row[ks] for a built-in mother, ks inside the row's band.
template <typename T>
__device__ __forceinline__ cplx<T> filter_value(const cplx<T> x, const RowDesc& rd, const Mother& mo, int ks) {
  const T g = profile<T>(mo, T(rd.a) * T(ks));
  const T gr = g * T(rd.amp_re), gi = g * T(rd.amp_im);
  return mk<T>(x.x * gr - x.y * gi, x.x * gi + x.y * gr);
}

// xhat[k] * F_row[k] for signed bin ks (0 outside the row's band).
template <typename T>
__device__ __forceinline__ cplx<T> filtered_bin(const cplx<T>* __restrict__ xhat, const RowDesc& rd,
                                                const Mother& mo, int ks, int nmask) {
  const unsigned d = unsigned(ks - rd.k_lo);
  if (d >= unsigned(rd.nband)) return mk<T>(T(0), T(0));
  const cplx<T> x = (xhat + rd.spec_off)[ks & nmask];
  if (mo.kind == MOTHER_TABLE) {   // explicit filter bank: F_j[k] was evaluated by the host from psi_ft
    const cplx<T> f = (static_cast<const cplx<T>*>(mo.table) + rd.tab_off)[ks & nmask];
    const T s = T(rd.amp_re);
    return mk<T>((x.x * f.x - x.y * f.y) * s, (x.x * f.y + x.y * f.x) * s);
  }
  return filter_value<T>(x, rd, mo, ks);
}

__device__ __forceinline__ int signed_bin(int k, int N) { return k < (N >> 1) ? k : k - N; }

// ---------------------------------------------------------------------------------------------
// k_small: whole transform of length N = 2^logN (16..lmax) inside one workgroup, TB rows per WG.
// MODE IN_SPECTRUM: rows of W.  MODE IN_REAL: forward FFT of the zero-padded real signal
// (nrows = 1, out = conj(inverse(x))).
template <typename T, int MODE>
__global__ void __launch_bounds__(CWT_MAX_THREADS)
k_small(const void* __restrict__ in, const RowDesc* __restrict__ rows, int nrows, Mother mo,
        const cplx<T>* __restrict__ tw, int logN, int logTB, long n0, long in_ld,
        cplx<T>* __restrict__ out, long ldw, long ncols) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  T* lds = reinterpret_cast<T*>(lds_raw);
  const int N = 1 << logN, logNT = logN - 4, NT = 1 << logNT;
  Geo<T, false> g;
  g.logL = logN; g.logTB = logTB;
  g.j = threadIdx.x & (NT - 1);
  g.t = threadIdx.x >> logNT;
  const int row = blockIdx.x * (1 << logTB) + g.t;
  const bool live = row < nrows;
  T re[16], im[16];
  if constexpr (MODE == IN_REAL) {
    const T* x = static_cast<const T*>(in) + long(row) * in_ld;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = g.j + (e << logNT);
      re[e] = (live && k < n0) ? x[k] : T(0);
      im[e] = T(0);
    }
  } else if constexpr (MODE == IN_CPLX) {
    const cplx<T>* x = static_cast<const cplx<T>*>(in) + long(row) * in_ld;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = g.j + (e << logNT);
      const cplx<T> v = (live && k < n0) ? x[k] : mk<T>(T(0), T(0));
      re[e] = v.x; im[e] = -v.y;
    }
  } else {
    const cplx<T>* xhat = static_cast<const cplx<T>*>(in);
    RowDesc rd;
    if (live) rd = rows[row]; else { rd.nband = 0; rd.k_lo = 0; rd.a = 0; rd.amp_re = 0; rd.amp_im = 0; rd.out_row = 0; rd.spec_off = 0; rd.tab_off = 0; rd.aux_off = 0; rd.nyq_re = 0; rd.nyq_im = 0; }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = g.j + (e << logNT);
      const cplx<T> v = filtered_bin<T>(xhat, rd, mo, signed_bin(k, N), N - 1);
      re[e] = v.x; im[e] = v.y;
    }
  }
  wg_ifft<T, false>(re, im, lds, g, tw);
  if (!live) return;
  const long orow = (MODE != IN_SPECTRUM) ? long(row) : long(rows[row].out_row);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const long m = g.j + (e << logNT);
    if (m < ncols) out[orow * ldw + m] = mk<T>(re[e], MODE != IN_SPECTRUM ? -im[e] : im[e]);
  }
}

// k_direct: N <= 8.  One thread per output element.
template <typename T, int MODE>
__global__ void k_direct(const void* __restrict__ in, const RowDesc* __restrict__ rows, int nrows,
                         Mother mo, int logN, long n0, long in_ld, cplx<T>* __restrict__ out, long ldw,
                         long ncols) {
  const int N = 1 << logN;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = idx >> logN, m = idx & (N - 1);
  if (row >= nrows || m >= ncols) return;
  double sr = 0, si = 0;
  for (int k = 0; k < N; ++k) {
    double yr, yi;
    if constexpr (MODE == IN_REAL) {
      yr = (k < n0) ? double((static_cast<const T*>(in) + long(row) * in_ld)[k]) : 0.0;
      yi = 0;
    } else if constexpr (MODE == IN_CPLX) {
      const cplx<T> v = (k < n0) ? (static_cast<const cplx<T>*>(in) + long(row) * in_ld)[k] : mk<T>(T(0), T(0));
      yr = v.x; yi = -v.y;
    } else {
      const cplx<T> v = filtered_bin<T>(static_cast<const cplx<T>*>(in), rows[row], mo,
                                        signed_bin(k, N), N - 1);
      yr = v.x; yi = v.y;
    }
    const double ang = 6.283185307179586476925 * double((k * m) & (N - 1)) / double(N);
    const double c = cos(ang), s = sin(ang);
    sr += yr * c - yi * s;
    si += yr * s + yi * c;
  }
  const long orow = (MODE != IN_SPECTRUM) ? long(row) : long(rows[row].out_row);
  out[orow * ldw + m] = mk<T>(T(sr), T(MODE != IN_SPECTRUM ? -si : si));
}

// slot e *= first * step^e  (the inter-pass twiddle e^{2 pi i q r / N} walked along one index)
template <typename T>
__device__ __forceinline__ void twiddle_slots(T (&re)[16], T (&im)[16], cplx<T> cur, const cplx<T> step) {
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const T x = re[e], y = im[e];
    re[e] = x * cur.x - y * cur.y;
    im[e] = x * cur.y + y * cur.x;
    if (e < 15) cur = cmul<T>(cur, step);
  }
}

// ---------------------------------------------------------------------------------------------
// k_narrow: rows whose filter support is <= K = 2^logK bins.  N = K*R, n = R*m + r:
//   W[R m + r] = sum_{q<K} ( Y[k(q)] e^{2 pi i k(q) r / N} ) e^{2 pi i q m / K},
// k(q) = the only in-band bin congruent to q mod K.  grid = (R/TB, rows of this class);
// PLANES layout, lanes run along r so that the stores of W are TB*sizeof(complex) contiguous.
template <typename T>
__global__ void __launch_bounds__(CWT_MAX_THREADS)
k_narrow(const cplx<T>* __restrict__ xhat, const RowDesc* __restrict__ rows, Mother mo,
         const cplx<T>* __restrict__ tw, TwN<T> twn, int logN, int logK, int logTB,
         cplx<T>* __restrict__ W, long ldw, long ncols) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  T* lds = reinterpret_cast<T*>(lds_raw);
  const int N = 1 << logN, K = 1 << logK, logNT = logK - 4, NT = 1 << logNT;
  const int logR = logN - logK;
  const RowDesc rd = rows[blockIdx.y];
  Geo<T, true> g;
  g.logL = logK; g.logTB = logTB;
  g.t = threadIdx.x & ((1 << logTB) - 1);
  g.j = threadIdx.x >> logTB;
  const unsigned r = (blockIdx.x << logTB) + g.t;

  // phase 0: Y[q] = xhat[k(q)] * F[k(q)], q < K, shared by the TB FFTs of this workgroup
  cplx<T>* ytile = reinterpret_cast<cplx<T>*>(lds);
  for (int q = threadIdx.x; q < K; q += blockDim.x) {
    const int d = (q - rd.k_lo) & (K - 1);
    ytile[q] = filtered_bin<T>(xhat, rd, mo, rd.k_lo + d, N - 1);
  }
  __syncthreads();

  // phase 1: slot e <- Y[q_e] * e^{2 pi i k(q_e) r / N}, q_e = j + e*NT; the twiddle advances by
  // e^{2 pi i NT r / N} per slot and by an extra e^{-2 pi i K r / N} when k(q) wraps around the band
  T re[16], im[16];
  {
    const unsigned nm = unsigned(N - 1);
    int d = (g.j - rd.k_lo) & (K - 1);
    cplx<T> cur = twn((unsigned(rd.k_lo + d) * r) & nm);
    const cplx<T> step = twn((unsigned(NT) * r) & nm);
    const cplx<T> stepw = cmul<T>(step, twn((0u - (r << logK)) & nm));
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const cplx<T> y = ytile[g.j + (e << logNT)];
      re[e] = y.x * cur.x - y.y * cur.y;
      im[e] = y.x * cur.y + y.y * cur.x;
      const int dn = (d + NT) & (K - 1);
      cur = cmul<T>(cur, dn < d ? stepw : step);
      d = dn;
    }
  }
  __syncthreads();  // ytile aliases the exchange buffer

  wg_ifft<T, true>(re, im, lds, g, tw);

  cplx<T>* wrow = W + long(rd.out_row) * ldw;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const long n = (long(g.j + (e << logNT)) << logR) + r;
    if (n < ncols) store_w<T>(wrow + n, re[e], im[e]);
  }
}

// ---------------------------------------------------------------------------------------------
// k_pass_a: first pass of the two-pass transform, N = R*K, input bin k = q + K*k1:
//   Z[r][q] = sum_{k1<R} Y[q + K k1] e^{2 pi i k1 r / R}      (the twiddle e^{2 pi i q r / N} is applied by pass B)
// grid = (K/TQ, rows in chunk).  PLANES layout: lanes run along q (coalesced reads of xhat and
// coalesced stores of Z rows).  MODE IN_REAL reads the zero-padded real signal instead.
template <typename T, int MODE>
__global__ void __launch_bounds__(CWT_MAX_THREADS)
k_pass_a(const void* __restrict__ in, const RowDesc* __restrict__ rows, Mother mo,
         const cplx<T>* __restrict__ tw, TwN<T> twn, int logN, int logK, int logTQ, long n0, long in_ld,
         cplx<T>* __restrict__ Z) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  T* lds = reinterpret_cast<T*>(lds_raw);
  const int N = 1 << logN, logR = logN - logK, logNT = logR - 4, NT = 1 << logNT;
  Geo<T, true> g;
  g.logL = logR; g.logTB = logTQ;
  g.t = threadIdx.x & ((1 << logTQ) - 1);
  g.j = threadIdx.x >> logTQ;
  const int q = (blockIdx.x << logTQ) + g.t;
  T re[16], im[16];
  if constexpr (MODE == IN_REAL) {
    const T* x = static_cast<const T*>(in) + long(blockIdx.y) * in_ld;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const long k = q + (long(g.j + (e << logNT)) << logK);
      re[e] = k < n0 ? x[k] : T(0);
      im[e] = T(0);
    }
  } else if constexpr (MODE == IN_CPLX) {
    const cplx<T>* x = static_cast<const cplx<T>*>(in) + long(blockIdx.y) * in_ld;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const long k = q + (long(g.j + (e << logNT)) << logK);
      const cplx<T> v = k < n0 ? x[k] : mk<T>(T(0), T(0));
      re[e] = v.x; im[e] = -v.y;
    }
  } else {
    const cplx<T>* xhat = static_cast<const cplx<T>*>(in);
    const RowDesc rd = rows[blockIdx.y];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = q + ((g.j + (e << logNT)) << logK);
      const cplx<T> v = filtered_bin<T>(xhat, rd, mo, signed_bin(k, N), N - 1);
      re[e] = v.x; im[e] = v.y;
    }
  }
  wg_ifft<T, true>(re, im, lds, g, tw);
  cplx<T>* z = Z + (long(blockIdx.y) << logN) + q;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const long r = g.j + (e << logNT);
    z[r << logK] = mk<T>(re[e], im[e]);
  }
}

// k_pass_b: second pass: W[R m + r] = sum_{q<K} (Z[r][q] e^{2 pi i q r / N}) e^{2 pi i q m / K}.
// grid = (R/TB, rows in chunk).  ROWS layout for the FFT (coalesced reads of Z rows), then an LDS
// transpose so that lanes run along r for the stores.  CONJ: store conj (forward transform).
template <typename T, bool CONJ>
__global__ void __launch_bounds__(CWT_MAX_THREADS)
k_pass_b(const cplx<T>* __restrict__ Z, const RowDesc* __restrict__ rows,
         const cplx<T>* __restrict__ tw, TwN<T> twn, int logN, int logK, int logTB, cplx<T>* __restrict__ W,
         long ldw, long ncols) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  T* lds = reinterpret_cast<T*>(lds_raw);
  const int logR = logN - logK, logNT = logK - 4, NT = 1 << logNT;
  Geo<T, false> g;
  g.logL = logK; g.logTB = logTB;
  g.j = threadIdx.x & (NT - 1);
  g.t = threadIdx.x >> logNT;
  const long r0 = long(blockIdx.x) << logTB;
  const cplx<T>* z = Z + (long(blockIdx.y) << logN) + ((r0 + g.t) << logK) + g.j;
  T re[16], im[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const cplx<T> v = z[e << logNT];
    re[e] = v.x; im[e] = v.y;
  }
  twiddle_slots<T>(re, im, twn(unsigned(r0 + g.t) * unsigned(g.j)), twn(unsigned(r0 + g.t) << logNT));
  wg_ifft<T, false>(re, im, lds, g, tw);

  // transpose: element (t, m) -> linear index m*TB + t; thread reads back linear tid + c*blockDim
  // (every LDS read of wg_ifft is already fenced by the barrier that ends its last exchange)
#pragma unroll
  for (int e = 0; e < 16; ++e)
    lds[lds_swizzle<T>(((g.j + (e << logNT)) << logTB) | g.t)] = re[e];
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 16; ++c) re[c] = lds[lds_swizzle<T>(threadIdx.x + c * blockDim.x)];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16; ++e)
    lds[lds_swizzle<T>(((g.j + (e << logNT)) << logTB) | g.t)] = im[e];
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 16; ++c) im[c] = lds[lds_swizzle<T>(threadIdx.x + c * blockDim.x)];

  const long orow = rows ? long(rows[blockIdx.y].out_row) : long(blockIdx.y);
  cplx<T>* wrow = W + orow * ldw;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int idx = threadIdx.x + c * blockDim.x;
    const long m = idx >> logTB, t = idx & ((1 << logTB) - 1);
    const long n = (m << logR) + r0 + t;
    if (n < ncols) store_w<T>(wrow + n, re[c], CONJ ? -im[c] : im[c]);
  }
}

// =============================================================================================
// Compile-time specialised versions of k_narrow / k_pass_a / k_pass_b for the default geometry
// (LOGP = log2 of the points per workgroup: 13 for fp64, 14 for fp32).  Same math and same launch
// grids as the generic kernels above; the host picks them when the geometry matches.
// Global accesses are written as (uniform pointer)[32-bit lane offset] so that they compile to
// SGPR-base + VGPR-offset instructions instead of 64-bit per-lane address arithmetic.
// Epilogue shared by the ROWS-layout kernels: thread (t, j) holds outputs m = j + e*NT of FFT t (residue
// r0 + t).  LDS transpose (t, m) -> m*TB + t with one pad element per 16, read back linearly in tid, so
// that consecutive lanes store consecutive r: W[R m + r0 + t] in TB*sizeof(complex)-byte segments.
template <typename T, int LOGK, int LOGP, bool CONJ>
__device__ __forceinline__ void transpose_store(T (&re)[16], T (&im)[16], T* lds, int t, int j,
                                                cplx<T>* __restrict__ wrow, int logR, unsigned r0,
                                                long ncols) {
  constexpr int LOGTB = LOGP - LOGK, NT = 1 << (LOGK - 4), BD = 1 << (LOGP - 4);
  constexpr int TS = (BD) + (BD >> 4);                    // physical stride of BD elements
  const int wa = (j << LOGTB) + t, wbase = wa + (wa >> 4);
  const int rbase = int(threadIdx.x) + (int(threadIdx.x) >> 4);
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16; ++e) lds[wbase + e * TS] = re[e];
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 16; ++c) re[c] = lds[rbase + c * TS];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16; ++e) lds[wbase + e * TS] = im[e];
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 16; ++c) im[c] = lds[rbase + c * TS];
  const unsigned m0 = threadIdx.x >> LOGTB, tt = threadIdx.x & ((1 << LOGTB) - 1);
  const unsigned off = (m0 << logR) + r0 + tt;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const unsigned step_c = unsigned(c * (BD >> LOGTB)) << logR;
    if (long(off) + step_c < ncols) store_w<T>(wrow + step_c + off, re[c], CONJ ? -im[c] : im[c]);
  }
  (void)NT;
}

// Workgroup ids are dealt round-robin to the 8 XCDs (id & 7).  The kernels below store 128-B segments at
// a 16-KiB stride, neighbouring segments coming from neighbouring tiles; with the natural map every XCD's
// L2 owns every 8th line of each run.  This map gives XCD c the contiguous tile range
// [c n/8, (c+1) n/8) instead, so one L2 writes back whole multi-KiB runs (measured on the store
// pattern alone: 5.46 -> 5.76 TB/s, tools/microbench/mem_patterns.hip).
// In the transform itself it pays for complex128 only (A/B on one box: pass B -3 %, band-limited -2 %;
// complex64 pass B +6 %, pass A +3 % in either precision), so only those kernels use it.
__device__ __forceinline__ unsigned xcd_tile_any() {
  const unsigned x = blockIdx.x, n = gridDim.x;
  return (n & 7u) ? x : (x & 7u) * (n >> 3) + (x >> 3);
}
template <typename T>
__device__ __forceinline__ unsigned xcd_tile() {
  return sizeof(T) == 8 ? xcd_tile_any() : blockIdx.x;
}
// Pass A on half-size tiles stores 64-B segments with plain stores; those only merge into whole lines when
// both halves meet in one L2, i.e. under the XCD-aware map (store pattern alone: 3.9 -> 4.8 TB/s).
template <typename T, int LOGTQ>
__device__ __forceinline__ unsigned pass_a_tile() {
  return ((2 * sizeof(T)) << LOGTQ) == 64 ? xcd_tile_any() : blockIdx.x;
}

// Phase PH of NQ of the multi-term input stage of narrow_ct_body (see there): the Y tiles of all NTERMS terms for the
// FFT inputs q in [PH K/NQ, (PH+1) K/NQ) go to LDS, the thread's slots e in [PH 16/NQ, (PH+1) 16/NQ) are finished.
// Compile-time recursion over PH keeps every register-array index a constant.
template <typename T, int LOGK, int LOGP, int NTERMS, int NQ, int PH>
__device__ __forceinline__ void narrow_phases(const cplx<T>* __restrict__ xhat, const RowDesc& rd, const Mother& mo,
                                              int N, cplx<T>* ytile, int j, const cplx<T>& rho, const cplx<T>& step,
                                              const cplx<T>& stepw, cplx<T>& cur, int& d, T (&re)[16], T (&im)[16]) {
  constexpr int K = 1 << LOGK, NT = K >> 4, KQ = K / NQ, EQ = 16 / NQ;
  if constexpr (PH > 0) __syncthreads();                   // every thread is done reading the previous phase
  for (int idx = threadIdx.x; idx < NTERMS * KQ; idx += (1 << (LOGP - 4))) {
    const int i = idx / KQ, q = PH * KQ + (idx - i * KQ);
    const int dq = (q - rd.k_lo) & (K - 1);
    ytile[idx] = filtered_bin<T>(xhat, rd, mo, rd.k_lo + dq + (i << LOGK), N - 1);
  }
  __syncthreads();
#pragma unroll
  for (int el = 0; el < EQ; ++el) {
    constexpr int E0 = PH * EQ;
    cplx<T> h = ytile[(NTERMS - 1) * KQ + j + el * NT];
#pragma unroll
    for (int i = NTERMS - 2; i >= 0; --i) {
      const cplx<T> y = ytile[i * KQ + j + el * NT];
      h = mk<T>(h.x * rho.x - h.y * rho.y + y.x, h.x * rho.y + h.y * rho.x + y.y);
    }
    re[E0 + el] = h.x * cur.x - h.y * cur.y;
    im[E0 + el] = h.x * cur.y + h.y * cur.x;
    if constexpr (PH + 1 < NQ) {       // finish the slot before the next phase overwrites the tiles: left alone the
      keep_here(re[E0 + el]);         // compiler sinks this arithmetic below the barrier and spills the NTERMS raw
      keep_here(im[E0 + el]);         // tile values per slot instead of keeping the one result
    }
    const int dn = (d + NT) & (K - 1);
    cur = cmul<T>(cur, dn < d ? stepw : step);
    d = dn;
  }
  if constexpr (PH + 1 < NQ)
    narrow_phases<T, LOGK, LOGP, NTERMS, NQ, PH + 1>(xhat, rd, mo, N, ytile, j, rho, step, stepw, cur, d, re, im);
}

template <typename T, int LOGK, int LOGP, int NTERMS>
__device__ __forceinline__ void narrow_ct_body(const cplx<T>* __restrict__ xhat, const RowDesc& rd,
                                               const Mother& mo, const cplx<T>* __restrict__ tw_all,
                                               const TwN<T>& twn, int logN, cplx<T>* __restrict__ W, long ldw,
                                               long ncols, T* lds) {
  constexpr int LOGTB = LOGP - LOGK, K = 1 << LOGK, LOGNT = LOGK - 4, NT = 1 << LOGNT;
  // PLANES layout: lanes run along the residue r, stores go straight from registers in 128-B segments.
  // (Measured alternative, rejected: ROWS layout with barrier-free wave-local FFTs + LDS transpose of
  // the outputs -- 4 % slower in fp64, 29 % slower in fp32.)
  using F = ct::Fft<T, LOGK, LOGTB, true>;
  const int N = 1 << logN, logR = logN - LOGK;
  const cplx<T>* tw = tw_all + (K - 2);                 // table of e^{2 pi i p / K}
  F f;
  f.t = threadIdx.x & ((1 << LOGTB) - 1);
  f.j = threadIdx.x >> LOGTB;
  const unsigned r = (xcd_tile<T>() << LOGTB) + f.t;

  // Input of the K-point FFT for output residue r (n = R m + r):
  //   Z_r[q] = sum_{i < nterms} Y[k_i(q)] e^{2 pi i k_i(q) r / N},  k_i(q) = k_lo + ((q - k_lo) mod K) + i K
  // NTERMS = 1 for rows whose support fits K bins; several terms otherwise (K >= 1024 only), which
  // still beats a two-pass transform while the row needs no intermediate in memory.
  // Per term: the Y tile (K complex) is built cooperatively in LDS, then every thread walks its 16
  // inputs with a running twiddle that advances by e^{2 pi i NT r / N} per slot and by an extra
  // e^{-2 pi i K r / N} where k_0(q) wraps around the band start.
  // The exchange buffer holds P reals = P/2 complex = BT tiles of K bins.  More terms than that go through it in NQ
  // phases over the FFT input index: phase ph holds all NTERMS terms of the K/NQ inputs q in [ph K/NQ, (ph+1) K/NQ),
  // i.e. of the thread's slots e in [ph 16/NQ, (ph+1) 16/NQ), which are finished (Horner, twiddle) before the next
  // phase overwrites the tiles -- nothing but finished FFT inputs stays in registers across the phases.
  cplx<T>* ytile = reinterpret_cast<cplx<T>*>(lds);
  const unsigned nm = unsigned(N - 1);
  constexpr int BT = (1 << LOGP) / (2 * K);
  static_assert(BT >= 1, "tile too small for one term");
  constexpr int NQ = NTERMS <= BT ? 1 : NTERMS <= 2 * BT ? 2 : NTERMS <= 4 * BT ? 4 : 8;
  static_assert(NQ <= 8 && NTERMS * (K / NQ) <= BT * K, "too many terms for the exchange buffer");
  // Z_r[q] = e^{2 pi i k_0(q) r / N} * sum_i Y_i[q] rho^i (Horner), rho = e^{2 pi i K r / N}; the common factor is a
  // running product over the 16 slots that picks up rho^-1 where k_0(q) wraps
  const cplx<T> step = twn((unsigned(NT) * r) & nm);
  const cplx<T> rho = twn((r << LOGK) & nm);
  const cplx<T> stepw = cmul<T>(step, mk<T>(rho.x, -rho.y));
  int d = (f.j - rd.k_lo) & (K - 1);
  cplx<T> cur = twn((unsigned(rd.k_lo + d) * r) & nm);
  T re[16], im[16];
  narrow_phases<T, LOGK, LOGP, NTERMS, NQ, 0>(xhat, rd, mo, N, ytile, f.j, rho, step, stepw, cur, d, re, im);
  __syncthreads();  // the tiles alias the exchange buffer
  f.run(re, im, lds, tw);
  cplx<T>* wrow = W + long(rd.out_row) * ldw;
  const unsigned off = (unsigned(f.j) << logR) + r;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const unsigned step_e = unsigned(e * NT) << logR;
    if (long(off) + step_e < ncols) store_w<T>(wrow + step_e + off, re[e], im[e]);
  }
}

// waves per SIMD the band-limited kernels are compiled for (see the CWT_LB_* defaults at the top)
template <typename T, int LOGP>
constexpr int narrow_waves_per_simd() {
  return sizeof(T) == 8 ? CWT_LB_NARROW_F64 : LOGP >= 14 ? CWT_LB_NARROW_F32_BIG : CWT_LB_NARROW_F32;
}

// All band-limited rows of a transform in ONE launch: blockIdx.y walks the row table (sorted by
// class), every workgroup branches once to the body specialised for its row's (K, terms).  One
// launch instead of one per class removes ~10 kernel boundaries and partial last waves per transform.
template <typename T, int LOGP>
__global__ void __launch_bounds__(1 << (LOGP - 4), (narrow_waves_per_simd<T, LOGP>()))
k_narrow_ct_all(const cplx<T>* __restrict__ xhat, const RowDesc* __restrict__ rows, Mother mo,
                const cplx<T>* __restrict__ tw_all, TwN<T> twn, int logN, cplx<T>* __restrict__ W, long ldw,
                long ncols) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  T* lds = reinterpret_cast<T*>(lds_raw);
  const RowDesc rd = rows[blockIdx.y];
#define CWT_NARROW_CASE(LK, NT_)                                                                    \
  case (LK) + 100 * (NT_):                                                                           \
    narrow_ct_body<T, LK, LOGP, NT_>(xhat, rd, mo, tw_all, twn, logN, W, ldw, ncols, lds);          \
    break;
  switch (rd.logK + 100 * rd.nterms) {
    CWT_NARROW_CASE(4, 1) CWT_NARROW_CASE(5, 1) CWT_NARROW_CASE(6, 1) CWT_NARROW_CASE(7, 1)
    CWT_NARROW_CASE(8, 1) CWT_NARROW_CASE(9, 1) CWT_NARROW_CASE(10, 1) CWT_NARROW_CASE(10, 2)
    CWT_NARROW_CASE(10, 3) CWT_NARROW_CASE(10, 4)
    default: break;
  }
#undef CWT_NARROW_CASE
}


// Band-limited rows with 5..16 aliased terms of K = 1024 bins (support up to 16384 bins): a kernel of their own so
// that the common cases above keep their register allocation.
template <typename T, int LOGP>
__global__ void __launch_bounds__(1 << (LOGP - 4), (narrow_waves_per_simd<T, LOGP>()))
k_narrow_ct_many(const cplx<T>* __restrict__ xhat, const RowDesc* __restrict__ rows, Mother mo,
                 const cplx<T>* __restrict__ tw_all, TwN<T> twn, int logN, cplx<T>* __restrict__ W, long ldw,
                 long ncols) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  T* lds = reinterpret_cast<T*>(lds_raw);
  const RowDesc rd = rows[blockIdx.y];
#define CWT_MANY_CASE(NT_)                                                                           \
  case NT_: narrow_ct_body<T, 10, LOGP, NT_>(xhat, rd, mo, tw_all, twn, logN, W, ldw, ncols, lds); break;
  switch (rd.nterms) {
    CWT_MANY_CASE(5) CWT_MANY_CASE(6) CWT_MANY_CASE(7) CWT_MANY_CASE(8) CWT_MANY_CASE(9) CWT_MANY_CASE(10)
    CWT_MANY_CASE(11) CWT_MANY_CASE(12) CWT_MANY_CASE(13) CWT_MANY_CASE(14) CWT_MANY_CASE(15) CWT_MANY_CASE(16)
    default: break;
  }
#undef CWT_MANY_CASE
}

// fp64 only: rows whose support needs K = 2048 (or 2..4 aliased terms of 2048 bins, support <= 8192) run with
// 16384 points per workgroup (1024 threads, 128 KiB of LDS, one workgroup per CU) so that the stores stay
// 128-byte segments (TB = 8).  Converts rows of support 4096..8192 from the two-pass transform (48 B per
// sample*scale of traffic) to the single-pass form (16 B).
template <typename T>
__global__ void __launch_bounds__(1024, 4)
k_narrow_ct_big(const cplx<T>* __restrict__ xhat, const RowDesc* __restrict__ rows, Mother mo,
                const cplx<T>* __restrict__ tw_all, TwN<T> twn, int logN, cplx<T>* __restrict__ W, long ldw,
                long ncols) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  T* lds = reinterpret_cast<T*>(lds_raw);
  const RowDesc rd = rows[blockIdx.y];
  switch (rd.nterms) {
    case 1: narrow_ct_body<T, 11, 14, 1>(xhat, rd, mo, tw_all, twn, logN, W, ldw, ncols, lds); break;
    case 2: narrow_ct_body<T, 11, 14, 2>(xhat, rd, mo, tw_all, twn, logN, W, ldw, ncols, lds); break;
    case 3: narrow_ct_body<T, 11, 14, 3>(xhat, rd, mo, tw_all, twn, logN, W, ldw, ncols, lds); break;
    case 4: narrow_ct_body<T, 11, 14, 4>(xhat, rd, mo, tw_all, twn, logN, W, ldw, ncols, lds); break;
    case 5: narrow_ct_body<T, 11, 14, 5>(xhat, rd, mo, tw_all, twn, logN, W, ldw, ncols, lds); break;
    case 6: narrow_ct_body<T, 11, 14, 6>(xhat, rd, mo, tw_all, twn, logN, W, ldw, ncols, lds); break;
    case 7: narrow_ct_body<T, 11, 14, 7>(xhat, rd, mo, tw_all, twn, logN, W, ldw, ncols, lds); break;
    case 8: narrow_ct_body<T, 11, 14, 8>(xhat, rd, mo, tw_all, twn, logN, W, ldw, ncols, lds); break;
    default: break;
  }
}

template <typename T, int LOGR, int LOGP, int MODE>
__global__ void __launch_bounds__(1 << (LOGP - 4), (sizeof(T) == 8 ? CWT_LB_PASS_A_F64 : CWT_LB_PASS_A_F32))
k_pass_a_ct(const void* __restrict__ in, const RowDesc* __restrict__ rows, Mother mo,
            const cplx<T>* __restrict__ tw, TwN<T> twn, int logN, long n0, long in_ld,
            cplx<T>* __restrict__ Z) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  T* lds = reinterpret_cast<T*>(lds_raw);
  constexpr int LOGTQ = LOGP - LOGR, LOGNT = LOGR - 4, NT = 1 << LOGNT;
  using F = ct::Fft<T, LOGR, LOGTQ, true>;
  const int N = 1 << logN, logK = logN - LOGR;
  F f;
  f.t = threadIdx.x & ((1 << LOGTQ) - 1);
  f.j = threadIdx.x >> LOGTQ;
  const unsigned q = (pass_a_tile<T, LOGP - LOGR>() << LOGTQ) + f.t;
  const unsigned k0 = q + (unsigned(f.j) << logK);        // bin of slot 0; slot e adds (e*NT) << logK
  T re[16], im[16];
  if constexpr (MODE == IN_REAL) {
    const T* x = static_cast<const T*>(in) + long(blockIdx.y) * in_ld;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const unsigned step_e = unsigned(e * NT) << logK;
      re[e] = (long(k0) + step_e < n0) ? (x + step_e)[k0] : T(0);
      im[e] = T(0);
    }
  } else if constexpr (MODE == IN_CPLX) {
    const cplx<T>* x = static_cast<const cplx<T>*>(in) + long(blockIdx.y) * in_ld;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const unsigned step_e = unsigned(e * NT) << logK;
      const cplx<T> v = (long(k0) + step_e < n0) ? (x + step_e)[k0] : mk<T>(T(0), T(0));
      re[e] = v.x; im[e] = -v.y;
    }
  } else {
    const cplx<T>* xhat = static_cast<const cplx<T>*>(in);
    const RowDesc rd = rows[blockIdx.y];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = int(k0 + (unsigned(e * NT) << logK));
      const cplx<T> v = filtered_bin<T>(xhat, rd, mo, signed_bin(k, N), N - 1);
      re[e] = v.x; im[e] = v.y;
    }
  }
  f.run(re, im, lds, tw);
  cplx<T>* z = Z + (long(blockIdx.y) << logN);
  const unsigned off = (unsigned(f.j) << logK) + q;
#pragma unroll
  for (int e = 0; e < 16; ++e) (z + (long(e * NT) << logK))[off] = mk<T>(re[e], im[e]);
}

// Pass A for a row whose support spans only c <= C = 2^LOGC of the R = 2^LOGR bins k1 of every column q
// (k = q + K k1): the column transform has c consecutive non-zero inputs u_q[a] = Y[k_lo + d0(q) + K a],
// so with r = R2 m' + r' (R2 = R / C) it is, for each r', a C-point FFT of the aliased twiddled inputs
//   V[q'] = u_q[a(q')] e^{2 pi i (k1_base + a(q')) r' / R},  a(q') = (q' - k1_base) mod C,
// exactly the band-limited form of k_narrow one level down.  8 columns x R2 residues per workgroup (the
// same 8R points as the full pass A), log2 C instead of log2 R butterfly levels, and only 8 C filter
// evaluations per workgroup.  Z layout is unchanged.
template <typename T, int LOGR, int LOGP, int LOGC>
__device__ __forceinline__ void pass_a_band_body(const cplx<T>* __restrict__ xhat, const RowDesc& rd,
                                                 const Mother& mo, const cplx<T>* __restrict__ tw_all,
                                                 const TwN<T>& twn, int logN, cplx<T>* __restrict__ z, T* lds) {
  constexpr int LOGTQ = LOGP - LOGR, LOGR2 = LOGR - LOGC, LOGTB = LOGTQ + LOGR2;
  constexpr int C = 1 << LOGC, LOGNT = LOGC - 4, NT = 1 << LOGNT, TQ = 1 << LOGTQ;
  using F = ct::Fft<T, LOGC, LOGTB, true>;
  const int N = 1 << logN, logK = logN - LOGR, K = 1 << logK;
  const unsigned nm = unsigned(N - 1);
  F f;
  f.t = threadIdx.x & ((1 << LOGTB) - 1);
  f.j = threadIdx.x >> LOGTB;
  const int tq = f.t & (TQ - 1);
  const unsigned rp = unsigned(f.t) >> LOGTQ;                       // residue r' < R2
  const int q0 = pass_a_tile<T, LOGP - LOGR>() << LOGTQ;

  cplx<T>* ytile = reinterpret_cast<cplx<T>*>(lds);                 // [a][tq], a < C
  for (int idx = threadIdx.x; idx < (C << LOGTQ); idx += (1 << (LOGP - 4))) {
    const int d0 = (q0 + (idx & (TQ - 1)) - rd.k_lo) & (K - 1);
    ytile[idx] = filtered_bin<T>(xhat, rd, mo, rd.k_lo + d0 + ((idx >> LOGTQ) << logK), N - 1);
  }
  __syncthreads();
  const int q = q0 + tq;
  const int kb = rd.k_lo + ((q - rd.k_lo) & (K - 1));               // signed bin of this column's first term
  const int k1b = (kb - q) >> logK;                                 // its k1 (may be negative): exact division
  T re[16], im[16];
  {
    int a = (f.j - k1b) & (C - 1);
    cplx<T> cur = twn(((unsigned(k1b + a) * rp) << logK) & nm);     // e^{2 pi i (k1b + a) r' / R}
    const cplx<T> step = twn(((unsigned(NT) * rp) << logK) & nm);
    const cplx<T> stepw = cmul<T>(step, twn((0u - ((rp << LOGC) << logK)) & nm));
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const cplx<T> y = ytile[(a << LOGTQ) + tq];
      re[e] = y.x * cur.x - y.y * cur.y;
      im[e] = y.x * cur.y + y.y * cur.x;
      const int an = (a + NT) & (C - 1);
      cur = cmul<T>(cur, an < a ? stepw : step);
      a = an;
    }
  }
  __syncthreads();                                                   // the tile aliases the exchange buffer
  f.run(re, im, lds, tw_all + (C - 2));
  // slot e holds m' = j + e*NT  ->  r = R2 m' + r'
  const unsigned r0 = (unsigned(f.j) << LOGR2) + rp;
  const unsigned off = (r0 << logK) + unsigned(q);
#pragma unroll
  for (int e = 0; e < 16; ++e) (z + ((long(e * NT) << LOGR2) << logK))[off] = mk<T>(re[e], im[e]);
}

// Pass A of a chunk of wide rows: every workgroup branches once on its row's class (rd.logK: 0 = all
// R inputs may be non-zero -> full column FFT; 4/6/8 -> support spans <= 16/64/256 bins k1).
template <typename T, int LOGR, int LOGP>
__global__ void __launch_bounds__(1 << (LOGP - 4), (sizeof(T) == 8 ? CWT_LB_PASS_A_F64 : CWT_LB_PASS_A_F32))
k_pass_a_ct_rows(const cplx<T>* __restrict__ xhat, const RowDesc* __restrict__ rows, Mother mo,
                 const cplx<T>* __restrict__ tw_all, TwN<T> twn, int logN, cplx<T>* __restrict__ Z) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  T* lds = reinterpret_cast<T*>(lds_raw);
  const RowDesc rd = rows[blockIdx.y];
  cplx<T>* z = Z + (long(blockIdx.y) << logN);
  if constexpr (LOGR > 4)
    if (rd.logK == 4) {
      pass_a_band_body<T, LOGR, LOGP, 4>(xhat, rd, mo, tw_all, twn, logN, z, lds);
      return;
    }
  if constexpr (LOGR > 6)
    if (rd.logK == 6) {
      pass_a_band_body<T, LOGR, LOGP, 6>(xhat, rd, mo, tw_all, twn, logN, z, lds);
      return;
    }
  if constexpr (LOGR > 8)
    if (rd.logK == 8) {
      pass_a_band_body<T, LOGR, LOGP, 8>(xhat, rd, mo, tw_all, twn, logN, z, lds);
      return;
    }
  // full column FFT (same code as k_pass_a_ct<..., IN_SPECTRUM>)
  constexpr int LOGTQ = LOGP - LOGR, LOGNT = LOGR - 4, NT = 1 << LOGNT;
  using F = ct::Fft<T, LOGR, LOGTQ, true>;
  const int N = 1 << logN, logK = logN - LOGR;
  F f;
  f.t = threadIdx.x & ((1 << LOGTQ) - 1);
  f.j = threadIdx.x >> LOGTQ;
  const unsigned q = (pass_a_tile<T, LOGP - LOGR>() << LOGTQ) + f.t;
  const unsigned k0 = q + (unsigned(f.j) << logK);
  T re[16], im[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int k = int(k0 + (unsigned(e * NT) << logK));
    const cplx<T> v = filtered_bin<T>(xhat, rd, mo, signed_bin(k, N), N - 1);
    re[e] = v.x; im[e] = v.y;
  }
  f.run(re, im, lds, tw_all + ((1 << LOGR) - 2));
  const unsigned off = (unsigned(f.j) << logK) + q;
#pragma unroll
  for (int e = 0; e < 16; ++e) (z + (long(e * NT) << logK))[off] = mk<T>(re[e], im[e]);
}

template <typename T, int LOGK, int LOGP, bool CONJ>
__global__ void __launch_bounds__(1 << (LOGP - 4), (sizeof(T) == 8 ? CWT_LB_PASS_B_F64 : CWT_LB_PASS_B_F32))
k_pass_b_ct(const cplx<T>* __restrict__ Z, const RowDesc* __restrict__ rows,
            const cplx<T>* __restrict__ tw, TwN<T> twn, int logN, cplx<T>* __restrict__ W, long ldw, long ncols) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  T* lds = reinterpret_cast<T*>(lds_raw);
  constexpr int LOGTB = LOGP - LOGK, LOGNT = LOGK - 4, NT = 1 << LOGNT, BD = 1 << (LOGP - 4);
  using F = ct::Fft<T, LOGK, LOGTB, false>;
  const int logR = logN - LOGK;
  F f;
  f.j = threadIdx.x & (NT - 1);
  f.t = threadIdx.x >> LOGNT;
  const unsigned r0 = xcd_tile<T>() << LOGTB;
  const cplx<T>* z = Z + (long(blockIdx.y) << logN) + (long(r0) << LOGK);
  const unsigned zoff = (unsigned(f.t) << LOGK) + f.j;
  T re[16], im[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const cplx<T> v = (z + e * NT)[zoff];
    re[e] = v.x; im[e] = v.y;
  }
  twiddle_slots<T>(re, im, twn((r0 + unsigned(f.t)) * unsigned(f.j)), twn((r0 + unsigned(f.t)) << LOGNT));
#pragma unroll
  for (int e = 0; e < 16; ++e) { keep_here(re[e]); keep_here(im[e]); }   // twiddle done before the FFT's registers fill up
  f.run(re, im, lds, tw);

  const long orow = rows ? long(rows[blockIdx.y].out_row) : long(blockIdx.y);
  transpose_store<T, LOGK, LOGP, CONJ>(re, im, lds, f.t, f.j, W + orow * ldw, logR, r0, ncols);
}


// =============================================================================================
// Overlap-save rows (k_ols_fwd, k_ols_ct): wide-band rows whose wavelet is COMPACT IN TIME.
//
// W[j, n] = sum_m x[m] h_j[(n - m) mod N], h_j = IFFT_N(F_j).  Where F_j is not clipped at the Nyquist bins, h_j is the
// sampled wavelet psi((t)/s)/s up to the filter-support threshold, negligible beyond |t| > H = c_H * s/dt samples
// (c_H from the mother's tail mass, `time_halo_factor`).  For an output block [n0, n0 + L), L = P - 2H, take the P
// input samples x[n0 - H .. n0 + L + H) (indices mod N, zero beyond the signal as the padded reference has them):
//   y = IFFT_P( FFT_P(x_block) * G_j ),  G_j[k'] = F_j[k' N / P]   (decimating the spectrum = wrapping h_j to period P)
// and y[H .. H + L) = W[j, n0 .. n0 + L) up to the neglected tail: no intermediate in memory, no N-point transform,
// fully contiguous stores.  FFT_P(x_block) is shared by every row of a halo class: k_ols_fwd writes the half
// spectra X_b[0 .. P/2] (x is real) of all blocks of all classes once per transform, k_ols_ct reads them through L2
// (all rows of one block run on the same XCD).  The block transform is itself band limited (support B P / N bins),
// so it runs as P/K aliased K-point FFTs exactly like k_narrow one level down, on ONE workgroup tile: TB = P / K
// residues x K points = the whole block, n_local = thread + e * P/16.  Rows with long halos and narrow block
// supports use blocks of P_b = 2P points instead (the kept fraction (P_b - 2H) / P_b rises): the P_b / K residues of
// a block are split over P_b / P workgroups, each storing TB-element segments (n_local = (P_b / K) m + r).
struct OlsClass {
  int wg_first;    // first workgroup of this class in the k_ols_ct launch (multiple of 8)
  int blk_first;   // first workgroup (= block) of this class in the k_ols_fwd<T, logb> launch
  int nblocks;     // output blocks of L = 2^logb - 2*halo columns
  int nrows;       // rows of this class
  int row_first;   // their first entry in the row table passed to k_ols_ct
  int halo;        // H (multiple of 64)
  int logb;        // log2 of the block length P_b >= P (workgroup tile): P_b / P workgroups share one block transform
  int nsig;        // signals of a batched call (1 otherwise): nrows = nsig x rows per signal, scale by scale
  long xs_off;     // element offset of this class's block spectra (nblocks x (P_b/2 + 8) complex)
};
constexpr int OLS_MAX_CLASSES = 16;
struct OlsClasses {
  OlsClass c[OLS_MAX_CLASSES];
  int wg_first[OLS_MAX_CLASSES];   // copy of c[i].wg_first (INT_MAX beyond n): one scalar load finds a workgroup's class
  int n;
};
template <int LOGP> constexpr int ols_stride() { return (1 << (LOGP - 1)) + 8; }   // complex elements per block spectrum

// profile() with the mother known at compile time (straight-line code: the loads of neighbouring bins can be
// scheduled together)
template <typename T, int MK>
__device__ __forceinline__ T profile_k(const Mother& mo, T f) {
  if constexpr (MK == MOTHER_MORLET) {
    const T d = f - T(mo.p);
    return exp_(T(-0.5) * d * d);
  } else if constexpr (MK == MOTHER_PAUL) {
    return f > T(0) ? ipow<T>(f, mo.m) * exp_(-f) : T(0);
  } else {
    return ipow<T>(f, mo.m) * exp_(T(-0.5) * f * f);
  }
}

// X_b[k] for signed block bin ks from the stored half spectrum (X_b[-k] = conj X_b[k]); bins outside the row's band
// read entry 0 (any valid address); their filter table entry is 0
template <typename T>
__device__ __forceinline__ cplx<T> ols_load(const cplx<T>* __restrict__ xb, const RowDesc& rd, int ks) {
  const bool in = unsigned(ks - rd.k_lo) < unsigned(rd.nband);
  return xb[in ? (ks < 0 ? -ks : ks) : 0];
}
// x * G_row[ks], G from the row's filter table entry g (0 outside the band)
template <typename T>
__device__ __forceinline__ cplx<T> ols_apply(cplx<T> x, cplx<T> g, int ks) {
  if (ks < 0) x.y = -x.y;
  return mk<T>(x.x * g.x - x.y * g.y, x.x * g.y + x.y * g.x);
}

// Filter tables of the overlap-save rows: gt[tab_off + q] = amp * profile(a * k(q)), q < K (K = 2^logK: the row's block
// FFT length; k(q) = the band bin aliased to q, or the signed bin q itself when K = P), 0 outside the band.  Written once
// per row table (the table is cached with it), so that the row kernel multiplies instead of evaluating one exp per band
// bin per workgroup (K = 8192: 16 per thread, a third of that kernel's instructions).
template <typename T, int MK>
__global__ void k_ols_gtab(const RowDesc* __restrict__ rows, Mother mo, int logP, cplx<T>* __restrict__ gt) {
  const RowDesc rd = rows[blockIdx.y];
  const int K = 1 << rd.logK, q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= K) return;
  const int ks = rd.logK == logP ? signed_bin(q, K) : rd.k_lo + ((q - rd.k_lo) & (K - 1));
  cplx<T> g = mk<T>(T(0), T(0));
  if (unsigned(ks - rd.k_lo) < unsigned(rd.nband)) {
    const T v = profile_k<T, MK>(mo, T(rd.a) * T(ks));
    g = mk<T>(v * T(rd.amp_re), v * T(rd.amp_im));
  }
  gt[rd.tab_off + q] = g;
}


// Half spectra X_b[0 .. P_b/2] of the input blocks of the overlap-save classes (x is real), from a complex transform of
// HALF the block length (the classic real-input packing): with
// z[n] = x[2n] + i x[2n+1], n < M = P_b / 2, and Z = FFT_M(z),
//   X_b[k] = E[k] + e^{-2 pi i k / P_b} O[k],  E[k] = (Z[k] + conj Z[M-k]) / 2,  O[k] = (Z[k] - conj Z[M-k]) / (2i),  k <= M
// (Z[M] = Z[0]).  One workgroup of M/16 threads per block: half the butterflies and half the registers / LDS of the
// complex transform of the zero-imaginary block, twice the workgroups in flight per CU; the mirrored operand Z[M-k] comes
// through one extra pass of the exchange buffer.  Blocks of two workgroup tiles (P_b = 2P) are ONE M = P transform.
template <typename T, int LOGM>
__global__ void __launch_bounds__(1 << (LOGM - 4), 4)
k_ols_fwd_r(const T* __restrict__ x, long n0, int logN, OlsClasses cls, const cplx<T>* __restrict__ tw_all, TwN<T> twn,
            cplx<T>* __restrict__ xs, long x_ld, long xs_sig) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  T* lds = reinterpret_cast<T*>(lds_raw);
  x += long(blockIdx.y) * x_ld;                             // batched call: blockIdx.y = signal
  xs += long(blockIdx.y) * xs_sig;
  constexpr int M = 1 << LOGM, NT = M >> 4, LOGB = LOGM + 1, PB = 1 << LOGB;
  using F = ct::Fft<T, LOGM, 0, false>;
  const int wg = int(blockIdx.x);
  int c = 0;
  for (int i = 0; i < cls.n; ++i)
    if (cls.c[i].logb == LOGB && wg >= cls.c[i].blk_first) c = i;
  const int blk = wg - cls.c[c].blk_first, H = cls.c[c].halo, L = PB - 2 * H;
  const long nmask = (1L << logN) - 1;
  const long first = long(blk) * L - H;                    // even: L and H are multiples of 64
  F f;
  f.t = 0;
  f.j = threadIdx.x;
  T re[16], im[16];
  // forward = conj(inverse(conj z)): feed (x[2n], -x[2n+1])
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const long n = (first + 2 * (f.j + e * NT)) & nmask;   // even, so n + 1 does not wrap
    re[e] = n < n0 ? x[n] : T(0);
    im[e] = n + 1 < n0 ? -x[n + 1] : T(0);
  }
  f.run(re, im, lds, tw_all + (M - 2));
  // slot e holds conj(Z[k]), k = j + e NT.  Mirror pass: slot e <- the same plane at position (M - k) mod M
  T mr[16], mi[16];
  const int self = f.phys(f.j);
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16; ++e) lds[self + e * F::pstride(NT)] = re[e];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16; ++e) mr[e] = lds[f.phys((M - f.j - e * NT) & (M - 1))];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16; ++e) lds[self + e * F::pstride(NT)] = im[e];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16; ++e) mi[e] = lds[f.phys((M - f.j - e * NT) & (M - 1))];
  cplx<T>* out = xs + cls.c[c].xs_off + long(blk) * ((PB >> 1) + 8);
  // e^{2 pi i p / P_b}: from the table of length P_b where it exists (P_b <= 16384), else from the N-point tables
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int k = f.j + e * NT;
    // Z[k] = (re, -im), conj Z[M-k] = (mr, +mi)
    const T er = T(0.5) * (re[e] + mr[e]), ei = T(0.5) * (mi[e] - im[e]);          // E = (Z + conj Zm) / 2
    const T dr = T(0.5) * (re[e] - mr[e]), di = T(0.5) * (-im[e] - mi[e]);         // D = (Z - conj Zm) / 2,  O = D / i = (di, -dr)
    cplx<T> w;                                                                     // e^{+2 pi i k / P_b}; we need its conjugate
    if constexpr (LOGB <= 14) w = (tw_all + (PB - 2))[k];
    else w = twn(unsigned(k) << (logN - LOGB));
    const T orr = di, oi = -dr;
    out[k] = mk<T>(er + orr * w.x + oi * w.y, ei + oi * w.x - orr * w.y);          // E + conj(w) O
    if (k == 0) out[M] = mk<T>(er - orr, T(0));                                    // X[M] = Re Z[0] - Im Z[0]
  }
}

// Block transform with K = P: every thread filters its own 16 bins (rows whose block support exceeds P/2 bins).
template <typename T, int LOGP>
__device__ __forceinline__ void ols_full_body(const cplx<T>* __restrict__ xb, const RowDesc& rd,
                                              const cplx<T>* __restrict__ gt, const cplx<T>* __restrict__ tw_all,
                                              cplx<T>* __restrict__ wout, int H, int nlim, T* lds) {
  constexpr int P = 1 << LOGP, NT = P >> 4;
  using F = ct::Fft<T, LOGP, 0, false>;
  F f;
  f.t = 0;
  f.j = threadIdx.x;
  T re[16], im[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {                          // all loads first, then the arithmetic
    const cplx<T> v = ols_load<T>(xb, rd, signed_bin(f.j + e * NT, P));
    re[e] = v.x; im[e] = v.y;
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const cplx<T> v = ols_apply<T>(mk<T>(re[e], im[e]), gt[f.j + e * NT], signed_bin(f.j + e * NT, P));
    re[e] = v.x; im[e] = v.y;
  }
  f.run(re, im, lds, tw_all + (P - 2));
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int nl = f.j + e * NT - H;
    if (nl >= 0 && nl < nlim) store_w<T>(wout + nl, re[e], im[e]);
  }
}

// Block transform for a block support <= K = 2^LOGK < P bins: TB = P / K aliased K-point FFTs (residue r = lane index
// t, n_local = TB m + t), inputs Z_r[q] = Y[k(q)] e^{2 pi i k(q) r / P} with the filtered band Y built once in LDS.
// The host aligns the band start k_lo to a multiple of K/16, so that k(q) = k_lo + ((q - k_lo) mod K) wraps between
// the same two slots for every thread: slots e >= ew = 16 - ((-k_lo mod K) / NT) carry an extra e^{-2 pi i K r / P}.
template <typename T, int LOGK, int LOGP>
__device__ __forceinline__ void ols_band_body(const cplx<T>* __restrict__ xb, const RowDesc& rd,
                                              const cplx<T>* __restrict__ gt,
                                              const cplx<T>* __restrict__ tw_all, const TwN<T>& twn, int logN,
                                              cplx<T>* __restrict__ wout, int H, int nlim, T* lds, int logx, unsigned g) {
  // logx = log2(P_b / P), g < P_b / P: this workgroup's residues are r = g TB + t of the P_b / K of the block
  constexpr int LOGTB = LOGP - LOGK, K = 1 << LOGK, NT = K >> 4, BD = 1 << (LOGP - 4);
  using F = ct::Fft<T, LOGK, LOGTB, true, (LOGTB <= CWT_OLS_PAD_LOGTB)>;
  F f;
  f.t = threadIdx.x & ((1 << LOGTB) - 1);
  f.j = threadIdx.x >> LOGTB;
  const unsigned r = (g << LOGTB) + unsigned(f.t);
  cplx<T>* ytile = reinterpret_cast<cplx<T>*>(lds);
  constexpr int NQ = K > BD ? K / BD : 1;                 // band bins per thread
  cplx<T> yv[NQ], gv[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int q = int(threadIdx.x) + i * BD;
    if (q < K) {
      yv[i] = ols_load<T>(xb, rd, rd.k_lo + ((q - rd.k_lo) & (K - 1)));
      gv[i] = gt[q];
    }
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int q = int(threadIdx.x) + i * BD;
    if (q < K) ytile[q] = ols_apply<T>(yv[i], gv[i], rd.k_lo + ((q - rd.k_lo) & (K - 1)));
  }
  __syncthreads();
  const int sh = logN - LOGP - logx;
  const unsigned pm = (1u << (LOGP + logx)) - 1u;
  const cplx<T> step = twn(((unsigned(NT) * r) & pm) << sh);
  const cplx<T> rhoc = twn(((0u - (r << LOGK)) & pm) << sh);           // e^{-2 pi i K r / P}
  const int c0 = ((0 - rd.k_lo) & (K - 1)) >> (LOGK - 4);               // (-k_lo mod K) / NT, k_lo = 0 mod NT
  const int ew = 16 - c0;                                               // uniform: first slot after the wrap
  cplx<T> cur = twn(((unsigned(rd.k_lo + f.j + c0 * NT) * r) & pm) << sh);
  T re[16], im[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const cplx<T> y = ytile[f.j + e * NT];
    re[e] = y.x * cur.x - y.y * cur.y;
    im[e] = y.x * cur.y + y.y * cur.x;
    if (e < 15) cur = cmul<T>(cur, step);
    if (e + 1 == ew) cur = cmul<T>(cur, rhoc);                          // uniform branch
  }
  __syncthreads();                                       // the band tile aliases the exchange buffer
  f.run(re, im, lds, tw_all + (K - 2));
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    // n_local = (P_b / K) (j + e NT) + r; with P_b = P this is thread + e P/16
    const int nl = (((f.j + e * NT) << (LOGTB + logx)) | int(r)) - H;
    if (nl >= 0 && nl < nlim) store_w<T>(wout + nl, re[e], im[e]);
  }
}

// All overlap-save rows of a transform in one launch: 1-D grid, class c owns workgroups [wg_first, next wg_first);
// inside a class the 8 XCDs (workgroup id & 7) take every 8th block and walk all rows of a block back to back, so that
// a block spectrum is fetched into one L2 once and read there by every row.
template <typename T, int LOGP>
__global__ void __launch_bounds__(1 << (LOGP - 4), (sizeof(T) == 8 ? (LOGP == 12 ? CWT_LB_OLS_F64_HALF : CWT_LB_OLS_F64)
                                                                : (LOGP == 12 ? CWT_LB_OLS_F32_HALF : CWT_LB_OLS_F32)))
k_ols_ct(const cplx<T>* __restrict__ xs, const RowDesc* __restrict__ rows, const cplx<T>* __restrict__ gtab,
         const cplx<T>* __restrict__ tw_all, TwN<T> twn, int logN, OlsClasses cls, cplx<T>* __restrict__ W, long ldw,
         long ncols) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  T* lds = reinterpret_cast<T*>(lds_raw);
  constexpr int P = 1 << LOGP;
  // this workgroup's class: every class record is read from the kernel arguments at a fixed address (one round of scalar
  // loads for all 16) and selected with uniform compares -- no load whose address depends on an earlier load
  OlsClass oc = cls.c[0];
#pragma unroll
  for (int i = 1; i < OLS_MAX_CLASSES; ++i)
    if (int(blockIdx.x) >= cls.wg_first[i]) oc = cls.c[i];
  const unsigned local = blockIdx.x - unsigned(oc.wg_first);
  const int logx = oc.logb - LOGP;
  const unsigned g = (local >> 3) & ((1u << logx) - 1u);      // which part of the block's residues
  // (signal, block) pairs vb = signal * nblocks + block: XCD (workgroup id & 7) takes every 8th pair and walks the
  // nrs rows of that signal's class back to back; the class's rows are stored scale by scale, nsig signals each
  const unsigned seq = local >> (3 + logx), nsig = unsigned(oc.nsig), nrs = unsigned(oc.nrows) / nsig;
  const unsigned vb = (seq / nrs) * 8u + (local & 7u);
  if (vb >= unsigned(oc.nblocks) * nsig) return;
  const unsigned sig = vb / unsigned(oc.nblocks), blk = vb - sig * unsigned(oc.nblocks);
  const RowDesc rd = rows[oc.row_first + int((seq % nrs) * nsig + sig)];
  const int H = oc.halo, L = (P << logx) - 2 * H;
  const cplx<T>* xb = xs + rd.spec_off + oc.xs_off + long(blk) * ((P << logx) / 2 + 8);   // spec_off: the row's signal (batch)
  const long col0 = long(blk) * L;
  const long left = ncols - col0;
  const int nlim = left < L ? int(left) : L;
  cplx<T>* wout = W + long(rd.out_row) * ldw + col0;
  const cplx<T>* gt = gtab + rd.tab_off;
#define CWT_OLS_CASE(LK)                                                                            \
  case LK:                                                                                           \
    if constexpr (LK < LOGP) ols_band_body<T, LK, LOGP>(xb, rd, gt, tw_all, twn, logN, wout, H, nlim, lds, logx, g); \
    else ols_full_body<T, LOGP>(xb, rd, gt, tw_all, wout, H, nlim, lds);                            \
    break;
  switch (rd.logK) {
    CWT_OLS_CASE(4) CWT_OLS_CASE(5) CWT_OLS_CASE(6) CWT_OLS_CASE(7) CWT_OLS_CASE(8) CWT_OLS_CASE(9)
    CWT_OLS_CASE(10) CWT_OLS_CASE(11) CWT_OLS_CASE(12) CWT_OLS_CASE(13)
    default: ols_full_body<T, LOGP>(xb, rd, gt, tw_all, wout, H, nlim, lds); break;
  }
#undef CWT_OLS_CASE
}

// =============================================================================================
// Overlap-save rows on the BAND-PASSED COMPLEX signal (k_aols_*): rows whose filter is CLIPPED at the Nyquist bins.
//
// F_j[k] = amp G(a k) on the signed bins k in [-N/2, N/2) (wavelet.py:94, 102-104).  Where G has not died out at the
// Nyquist bins the cyclic filter jumps there, h_j = IFFT_N(F_j) has a 1/t tail and no overlap-save on the real signal is
// possible (these rows were the two-pass rows: 48 B per sample*scale).  But with a mask over the bins [k_s, N/2),
//     xhat F_j  =  (xhat mask) E_j,     E_j(f) = amp G(a N f) u(f),  f = k/N in [f_s, f_s + 1),
// where G is continued PAST Nyquist (f > 1/2: no wrap) and u is a smooth window: 1 on the part of the mask that carries
// the filter, erfc tapers over the rest of the circle.  E_j is cyclically smooth, so its kernel e_j is short (the wavelet
// itself convolved with the taper's kernel), and W_j = x_M (*) e_j is an overlap-save convolution of the complex
// band-passed signal x_M = IFFT_N(xhat mask), which is computed ONCE per transform (one two-pass row) and shared by all
// such rows.  Valid because xhat mask vanishes wherever E_j differs from F_j: Morlet's negative-frequency part (below the
// support threshold from bin k_s down), Paul's Heaviside (k_s = 1).
// DOG (two-sided real profile P(-f) = (-1)^m P(f), real signal): with the mask over the positive bins 1 .. N/2 - 1 and
// y = x_M (*) e_j (table = sign |amp| P u), the negative bins contribute (-1)^m conj: W = 2 Re y (m even) or -2 Im y (m odd,
// amp = i sign |amp|), plus the Nyquist bin, which the reference counts once, at -pi / dt: + F_j[N/2] xhat[N/2] (-1)^n / N
// (RowDesc::nyq_*).  RowDesc::nterms of such a row: 1 = y itself, 2 = 2 Re y, 3 = -2 Im y.
struct AolsGeom {
  int nrows;       // rows of the class
  int nblocks;     // output blocks of L = P - 2 halo columns
  int halo;        // H (multiple of 64)
  int ksp;         // first unwrapped bin of the block grid: a block bin q stands for kappa = ksp + ((q - ksp) mod P)
  double f_s;      // low edge of the mask in cycles per sample (<= 1/N)
  double f1_lo;    // the window is 1 on [f1_lo, 1/2]
  double z;        // erfc argument at the ends of a taper: u = erfc(z)/2 there
};

// window u(f), f in [f_s, f_s + 1)
__device__ __forceinline__ double aols_window(const AolsGeom& g, double f) {
  if (f > 0.5) {
    const double hw = 0.5 * (g.f_s + 0.5), c = 0.5 + hw;
    return 0.5 * erfc(g.z * (f - c) / hw);
  }
  if (f < g.f1_lo) {
    const double hw = 0.5 * (g.f1_lo - g.f_s), c = g.f_s + hw;
    return hw > 0 ? 0.5 * erfc(g.z * (c - f) / hw) : 0.0;
  }
  return 1.0;
}

// Filter tables of those rows: gt[tab_off + q] = amp_re * G(a_b kappa(q)) u(kappa(q) / P), q < P, real (the mother's
// constant is real for Morlet and Paul).  Evaluated in double for either precision; once per scale grid.
template <typename T, int MK>
__global__ void k_aols_gtab(const RowDesc* __restrict__ rows, Mother mo, int logP, AolsGeom g, T* __restrict__ gt) {
  const RowDesc rd = rows[blockIdx.y];
  const int P = 1 << logP, q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= P) return;
  const int kappa = g.ksp + ((q - g.ksp) & (P - 1));
  const double v = profile_k<double, MK>(mo, rd.a * double(kappa)) * aols_window(g, double(kappa) / double(P));
  gt[rd.tab_off + q] = T(v * rd.amp_re);
}

// Spectra of the input blocks of x_M (complex, N-periodic): block b covers x_M[b L - H .. b L - H + P).  One workgroup
// per block, forward transform as conj(inverse(conj)); all P bins are kept (x_M is not real).  blockIdx.y = signal of a
// batch: its x_M at xm + y N, its block spectra at xs + y nblocks (P + 8).
template <typename T, int LOGP>
__global__ void __launch_bounds__(1 << (LOGP - 4), 4)
k_aols_fwd(const cplx<T>* __restrict__ xm, int logN, int halo, const cplx<T>* __restrict__ tw_all,
           cplx<T>* __restrict__ xs) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  T* lds = reinterpret_cast<T*>(lds_raw);
  constexpr int P = 1 << LOGP, NT = P >> 4;
  using F = ct::Fft<T, LOGP, 0, false>;
  const long nmask = (1L << logN) - 1;
  const long first = long(blockIdx.x) * (P - 2 * halo) - halo;
  xm += long(blockIdx.y) << logN;
  F f;
  f.t = 0;
  f.j = threadIdx.x;
  T re[16], im[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const cplx<T> v = xm[(first + f.j + e * NT) & nmask];
    re[e] = v.x; im[e] = -v.y;
  }
  f.run(re, im, lds, tw_all + (P - 2));
  cplx<T>* out = xs + (long(blockIdx.y) * gridDim.x + long(blockIdx.x)) * (P + 8);
#pragma unroll
  for (int e = 0; e < 16; ++e) out[f.j + e * NT] = mk<T>(re[e], -im[e]);
}

// The rows: workgroup = (block, row).  The 8 XCDs (workgroup id & 7) take every 8th block and walk all rows of it back to
// back, so that a block spectrum is fetched into one L2 once.  y = IFFT_P(X_b * table), columns [H, H + L) are stored.
// blockIdx.y = signal of a batch: its rows at rows + y g.nrows, its block spectra at xs + y nblocks (P + 8); xhat = the
// spectra of the batch (the Nyquist bin of a row's signal at xhat[rd.spec_off + N / 2], two-sided filters only).
template <typename T, int LOGP>
__global__ void __launch_bounds__(1 << (LOGP - 4), (sizeof(T) == 8 ? (LOGP == 12 ? CWT_LB_OLS_F64_HALF : CWT_LB_OLS_F64)
                                                                : (LOGP == 12 ? CWT_LB_OLS_F32_HALF : CWT_LB_OLS_F32)))
k_aols_rows(const cplx<T>* __restrict__ xs, const RowDesc* __restrict__ rows, const T* __restrict__ gtab,
            const cplx<T>* __restrict__ tw_all, AolsGeom g, const cplx<T>* __restrict__ xhat, long nyq, cplx<T>* __restrict__ W,
            long ldw, long ncols) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  T* lds = reinterpret_cast<T*>(lds_raw);
  constexpr int P = 1 << LOGP, NT = P >> 4;
  using F = ct::Fft<T, LOGP, 0, false>;
  const unsigned seq = blockIdx.x >> 3;
  const unsigned blk = (seq / unsigned(g.nrows)) * 8u + (blockIdx.x & 7u);
  if (blk >= unsigned(g.nblocks)) return;
  const RowDesc rd = rows[blockIdx.y * unsigned(g.nrows) + seq % unsigned(g.nrows)];
  const int H = g.halo, L = P - 2 * H;
  const cplx<T>* xb = xs + (long(blockIdx.y) * g.nblocks + long(blk)) * (P + 8);
  const T* gt = gtab + rd.tab_off;
  const long col0 = long(blk) * L, left = ncols - col0;
  const int nlim = left < L ? int(left) : L;
  cplx<T>* wout = W + long(rd.out_row) * ldw + col0;
  F f;
  f.t = 0;
  f.j = threadIdx.x;
  T re[16], im[16], gv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {                          // all loads first, then the arithmetic
    const cplx<T> v = xb[f.j + e * NT];
    re[e] = v.x; im[e] = v.y;
    gv[e] = gt[f.j + e * NT];
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) { re[e] *= gv[e]; im[e] *= gv[e]; }
  f.run(re, im, lds, tw_all + (P - 2));
  if (rd.nterms == 1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int nl = f.j + e * NT - H;
      if (nl >= 0 && nl < nlim) store_w<T>(wout + nl, re[e], im[e]);
    }
  } else {                                                // two-sided real filter of a real signal (see above)
    const cplx<T> xn = xhat[rd.spec_off + nyq];
    const T nr = T(rd.nyq_re) * xn.x - T(rd.nyq_im) * xn.y, ni = T(rd.nyq_re) * xn.y + T(rd.nyq_im) * xn.x;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int nl = f.j + e * NT - H;
      const T v = rd.nterms == 2 ? T(2) * re[e] : T(-2) * im[e];
      const T sg = ((col0 + nl) & 1) ? T(-1) : T(1);      // (-1)^n
      if (nl >= 0 && nl < nlim) store_w<T>(wout + nl, v + sg * nr, sg * ni);
    }
  }
}

// =============================================================================================
// Band-limited rows in POLYNOMIAL form (k_poly_coef, k_poly_rows): no tile structure in the kernel that writes W.
//
// A row whose filter lives on the bins k_c + kappa, kappa in [-B/2, B/2), is a carrier times a slowly varying envelope:
//     W[n] = e^{2 pi i k_c n / N} v(n),     v(n) = sum_kappa Y[kappa] e^{2 pi i kappa n / N},  Y = xhat F_j / N.
// Cut the row into K' >= B intervals of R = N / K' samples: n = R m + r, u = (r - R/2) / (R/2) in [-1, 1).  Then
//     e^{2 pi i kappa n / N} = e^{2 pi i kappa m / K'} e^{i pi kappa / K'} e^{i theta u},   theta = pi kappa / K'  (|theta| <= pi B / (2 K'))
// and with e^{i theta u} = sum_d (i theta)^d u^d / d! cut at degree D
//     v(R m + r) = sum_{d <= D} a_d[m] u^d,    a_d = IFFT_K'( Y[kappa] e^{i pi kappa / K'} (i theta)^d / d! ).
// Stage 1 (k_poly_coef): D + 1 short inverse FFTs per row -> coefficient planes a_d[0 .. K'), a few per cent of the row's
// bytes.  Stage 2 (k_poly_rows): every output is one Horner evaluation, one modulation and one contiguous non-temporal
// store: a streaming kernel with many waves per CU and no FFT, which runs at the contiguous-store rate of the chip instead
// of the 128-byte-segment rate of a K-point transform per residue.  What it must not do is start every wave with a fetch
// of its own coefficients (latency bound, tools/microbench/stream_poly2.hip): a workgroup fetches the sets of all the
// intervals it touches once, into LDS, and covers POLY_PASSES x 256 lanes x 16 bytes with them (stream_poly3.hip: 6.1 - 6.7
// TB/s at two passes for R = 64 ... 4096, degree 8; one pass 3.4, four 5.8).
// The host picks K' and D per row: D is the smallest even degree with  F(kappa)/F_max * |theta|^(D+1)/(D+1)! <= the support
// threshold on every bin, i.e. the truncation is treated like the band limit itself.
constexpr int POLY_MAX_CLASSES = 8;       // K' = 2^8 ... 2^14 + one spare
constexpr int POLY_LOGP = 14;             // largest K' = points per workgroup of the largest k_poly_coef tile (1024 threads)
constexpr int POLY_MAX_DEGREE = 24;
#ifndef CWT_POLY_PASSES
#define CWT_POLY_PASSES 2
#endif
constexpr int POLY_PASSES = CWT_POLY_PASSES;   // passes of 256 lanes x 16 bytes per workgroup of k_poly_rows (measured: 1, 3, 4 slower)
constexpr int POLY_MIN_LOGR = 6;          // shortest interval: 64 samples
struct PolyClass {
  int logK;        // log2 K'
  int row_first;   // first row of the class in the row table handed to the kernels
  int nrows;
  int ndeg;        // degrees computed per row of this class = 1 + the largest degree in it
  int wg_first;    // first workgroup of the class in ITS k_poly_coef launch (one launch per tile size: 4096-point tiles for
                   // K' <= 4096, 8192 for K' = 8192, 16384 for K' = 16384)
};
struct PolyClasses {
  PolyClass c[POLY_MAX_CLASSES];
  int n;
};

// 1 / d!, d <= POLY_MAX_DEGREE
__device__ __forceinline__ double inv_factorial(int d) {
  constexpr double t[POLY_MAX_DEGREE + 1] = {
      1.0, 1.0, 0.5, 1.6666666666666666e-01, 4.1666666666666664e-02, 8.3333333333333332e-03, 1.3888888888888889e-03,
      1.9841269841269841e-04, 2.4801587301587302e-05, 2.7557319223985893e-06, 2.7557319223985888e-07,
      2.5052108385441720e-08, 2.0876756987868100e-09, 1.6059043836821613e-10, 1.1470745597729725e-11,
      7.6471637318198164e-13, 4.7794773323873853e-14, 2.8114572543455206e-15, 1.5619206968586226e-16,
      8.2206352466243295e-18, 4.1103176233121648e-19, 1.9572941063391263e-20, 8.8967913924505741e-22,
      3.8681701706306835e-23, 1.6117375710961184e-24};
  return t[d];
}

// The filtered, phase-shifted band of every polynomial row in the input order of its K'-point transforms:
//   yb[band_off + q] = Y[kappa(q)] e^{i pi kappa(q) / K'},  kappa(q) = the band bin congruent to q mod K' (0 if there is none)
// -- computed once per row (one filter evaluation per bin), read by the D + 1 transforms of the row.  grid = (K'_max / 256, rows).
template <typename T>
__global__ void __launch_bounds__(256)
k_poly_band(const cplx<T>* __restrict__ xhat, const RowDesc* __restrict__ rows, Mother mo, TwN<T> twn, int logN,
            cplx<T>* __restrict__ yb) {
  const RowDesc rd = rows[blockIdx.y];
  const int K = 1 << rd.logK, q = blockIdx.x * 256 + threadIdx.x;
  if (q >= K) return;
  const int N = 1 << logN;
  const int kc = rd.k_lo + (rd.nband >> 1), klo = rd.k_lo - kc;
  const int kap = klo + ((q - klo) & (K - 1));
  const cplx<T> y = filtered_bin<T>(xhat, rd, mo, kc + kap, N - 1);        // 0 outside the band
  const cplx<T> ph = twn((unsigned(kap) << (logN - rd.logK - 1)) & unsigned(N - 1));   // e^{2 pi i kappa (R/2) / N}
  yb[rd.aux_off + q] = cmul<T>(y, ph);
}

// One workgroup = 2^(LOGP - LOGK) transforms of K' = 2^LOGK points, ROWS layout (lanes run along the interval index m, so the
// planes are written in whole lines; K' <= 1024: a transform lives in one wavefront and needs no workgroup barrier).
// Transform `job` of the class = (row, degree): job = row * ndeg + d; its input is the row's band times (i theta)^d / d!.
// Tiles of 4096 points (256 threads, four workgroups per CU) wherever K' allows: these launches sit on the critical path of the
// step (k_poly_rows waits for them) and are latency bound -- one 16384-point workgroup per CU for everything measured 67 us.
template <typename T, int LOGK, int LOGP>
__device__ __forceinline__ void poly_coef_body(const cplx<T>* __restrict__ yb, const RowDesc* __restrict__ rows,
                                               const cplx<T>* __restrict__ tw_all, const PolyClass& pc,
                                               unsigned local_wg, cplx<T>* __restrict__ coef, T* lds) {
  constexpr int LOGTB = LOGP - LOGK, TB = 1 << LOGTB, K = 1 << LOGK, LOGNT = LOGK - 4, NT = 1 << LOGNT;
  using F = ct::Fft<T, LOGK, LOGTB, false>;
  F f;
  f.j = threadIdx.x & (NT - 1);
  f.t = threadIdx.x >> LOGNT;
  const int job = int(local_wg) * TB + f.t;
  const int rowi = job / pc.ndeg, d = job - rowi * pc.ndeg;
  RowDesc rd;
  bool live = rowi < pc.nrows;
  if (live) rd = rows[pc.row_first + rowi];
  live = live && d <= rd.nterms;                              // nterms = the row's degree D
  T re[16], im[16];
  if (live) {
    const cplx<T>* y = yb + rd.aux_off + f.j;
#pragma unroll
    for (int e = 0; e < 16; ++e) {                            // all loads first
      const cplx<T> v = y[e * NT];
      re[e] = v.x; im[e] = v.y;
    }
    const int klo = -(rd.nband >> 1);                         // kappa of the first band bin
    const T tscale = T(3.14159265358979323846 / double(K));
    const T ifact = T(inv_factorial(d));
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int q = f.j + e * NT;
      const int kap = klo + ((q - klo) & (K - 1));
      const T pw = ipow<T>(T(kap) * tscale, d) * ifact;       // theta^d / d!
      T vr = re[e] * pw, vi = im[e] * pw;
      if (d & 1) { const T tmp = vr; vr = -vi; vi = tmp; }    // times i^d
      if (d & 2) { vr = -vr; vi = -vi; }
      re[e] = vr; im[e] = vi;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) { re[e] = T(0); im[e] = T(0); }
  }
  f.run(re, im, lds, tw_all + (K - 2));
  if (!live) return;
  cplx<T>* out = coef + rd.tab_off + (long(d) << LOGK) + f.j;  // plane d of the row
#pragma unroll
  for (int e = 0; e < 16; ++e) out[e * NT] = mk<T>(re[e], im[e]);
}

template <typename T, int LOGP>
__global__ void __launch_bounds__(1 << (LOGP - 4), 4)
k_poly_coef(const cplx<T>* __restrict__ yb, const RowDesc* __restrict__ rows, const cplx<T>* __restrict__ tw_all,
            PolyClasses cls, cplx<T>* __restrict__ coef) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  T* lds = reinterpret_cast<T*>(lds_raw);
  // the classes of this launch: log2 K' in (LOGP - 1, LOGP] for the two large tiles, <= 12 for the 4096-point tile
  constexpr int LK_LO = LOGP == 12 ? 8 : LOGP, LK_HI = LOGP;
  PolyClass pc = cls.c[0];
  bool found = false;
#pragma unroll
  for (int i = 0; i < POLY_MAX_CLASSES; ++i)
    if (i < cls.n && cls.c[i].logK >= LK_LO && cls.c[i].logK <= LK_HI && int(blockIdx.x) >= cls.c[i].wg_first) { pc = cls.c[i]; found = true; }
  if (!found) return;
  const unsigned local = blockIdx.x - unsigned(pc.wg_first);
#define CWT_POLY_CASE(LK) \
  case LK: if constexpr (LK >= LK_LO && LK <= LK_HI) poly_coef_body<T, LK, LOGP>(yb, rows, tw_all, pc, local, coef, lds); break;
  switch (pc.logK) {
    CWT_POLY_CASE(8) CWT_POLY_CASE(9) CWT_POLY_CASE(10) CWT_POLY_CASE(11) CWT_POLY_CASE(12) CWT_POLY_CASE(13)
    CWT_POLY_CASE(14)
    default: break;
  }
#undef CWT_POLY_CASE
}

// Stage 2.  One workgroup = 256 lanes x POLY_PASSES passes; a lane stores 16 bytes per pass (one complex128 or two adjacent
// complex64 outputs).  sc[i][d] = a_d[m0 + i] for the intervals m0 ... the workgroup touches.
template <typename T, int D>
__device__ __forceinline__ void poly_rows_body(const RowDesc& rd, const cplx<T>* __restrict__ coef, const TwN<T>& twn,
                                               int logN, cplx<T>* __restrict__ W, long ldw, long ncols, cplx<T>* sc) {
  constexpr int PT = sizeof(T) == 8 ? 1 : 2, SPAN = 256 * PT, I = POLY_PASSES;
  const int logR = logN - rd.logK;
  const unsigned nmask = unsigned((1 << logN) - 1);
  const unsigned n0 = blockIdx.x * unsigned(SPAN * I);
  const unsigned m0 = n0 >> logR;
  const unsigned last = (n0 + unsigned(SPAN * I) - 1u) & nmask;          // (the grid covers ncols <= N outputs)
  const unsigned nint = ((last >= n0 ? last : nmask) >> logR) - m0 + 1u;
  const cplx<T>* a = coef + rd.tab_off + m0;
  for (unsigned t = threadIdx.x; t < nint * unsigned(D + 1); t += 256u) {
    const unsigned i = t / unsigned(D + 1), d = t - i * unsigned(D + 1);
    sc[t] = a[(long(d) << rd.logK) + i];
  }
  const int kc = rd.k_lo + (rd.nband >> 1);
  const unsigned nl = n0 + threadIdx.x * PT;
  cplx<T> w = twn((unsigned(kc) * nl) & nmask);
  const cplx<T> step = twn((unsigned(kc) * unsigned(SPAN)) & nmask);    // uniform: one pass further
  cplx<T> adj = mk<T>(T(1), T(0));
  if constexpr (PT == 2) adj = twn(unsigned(kc) & nmask);                // e^{2 pi i k_c / N}: the lane's second output
  const T scale = T(2) / T(1u << logR);
  cplx<T>* wrow = W + long(rd.out_row) * ldw;
  __syncthreads();
#pragma unroll
  for (int p = 0; p < I; ++p) {
    const unsigned n = nl + unsigned(p * SPAN);
    cplx<T> o[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const unsigned ni = n + unsigned(i);
      const cplx<T>* c = sc + ((ni >> logR) - m0) * unsigned(D + 1);
      const T u = T(int(ni & ((1u << logR) - 1u))) * scale - T(1);
      T pr = c[D].x, pi = c[D].y;
#pragma unroll
      for (int d = D - 1; d >= 0; --d) { const cplx<T> cd = c[d]; pr = fma(pr, u, cd.x); pi = fma(pi, u, cd.y); }
      const cplx<T> wi = i == 0 ? w : cmul<T>(w, adj);
      o[i] = mk<T>(pr * wi.x - pi * wi.y, pr * wi.y + pi * wi.x);
    }
    if constexpr (PT == 1) {
      if (long(n) < ncols) store_w<T>(wrow + n, o[0].x, o[0].y);
    } else {
      if (long(n) + 1 < ncols && ((reinterpret_cast<size_t>(wrow + n) & 15u) == 0)) {
        typedef T vec4 __attribute__((vector_size(4 * sizeof(T))));
        vec4 v = {o[0].x, o[0].y, o[PT - 1].x, o[PT - 1].y};
        __builtin_nontemporal_store(v, reinterpret_cast<vec4*>(wrow + n));
      } else {
        if (long(n) < ncols) store_w<T>(wrow + n, o[0].x, o[0].y);
        if (long(n) + 1 < ncols) store_w<T>(wrow + n + 1, o[PT - 1].x, o[PT - 1].y);
      }
    }
    if (p + 1 < I) w = cmul<T>(w, step);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_poly_rows(const RowDesc* __restrict__ rows, const cplx<T>* __restrict__ coef, TwN<T> twn, int logN,
            cplx<T>* __restrict__ W, long ldw, long ncols) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  cplx<T>* sc = reinterpret_cast<cplx<T>*>(lds_raw);
  const RowDesc rd = rows[blockIdx.y];
#define CWT_POLYR_CASE(DD) case DD: poly_rows_body<T, DD>(rd, coef, twn, logN, W, ldw, ncols, sc); break;
  switch (rd.nterms) {
    CWT_POLYR_CASE(2) CWT_POLYR_CASE(4) CWT_POLYR_CASE(6) CWT_POLYR_CASE(8) CWT_POLYR_CASE(10) CWT_POLYR_CASE(12)
    CWT_POLYR_CASE(14) CWT_POLYR_CASE(16) CWT_POLYR_CASE(18) CWT_POLYR_CASE(20) CWT_POLYR_CASE(22) CWT_POLYR_CASE(24)
    default: break;
  }
#undef CWT_POLYR_CASE
}

// ---------------------------------------------------------------------------------------------
// Element-wise helpers of the coherence path (pycwt/wavelet.py:499-514, mothers.py:97-102).
// All matrices are rows x ld, row-major, n < ncols valid.

// P[j,n] = (|W1|^2 + i |W2|^2) / s_j   (both auto-spectra ride through ONE complex smoothing pass:
//                                       the smoothing kernel is real, so Re/Im stay separate)
// C[j,n] = W1 conj(W2) / s_j ;  A[j,n] = angle(W1 conj(W2))
template <typename T>
__global__ void k_wct_products(const cplx<T>* __restrict__ W1, const cplx<T>* __restrict__ W2,
                               const T* __restrict__ inv_s, long ld, long ncols, cplx<T>* __restrict__ P,
                               cplx<T>* __restrict__ C, T* __restrict__ A) {
  const long n = long(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= ncols) return;
  const long i = long(blockIdx.y) * ld + n;
  const cplx<T> a = W1[i], b = W2[i];
  const T is = inv_s[blockIdx.y];
  P[i] = mk<T>((a.x * a.x + a.y * a.y) * is, (b.x * b.x + b.y * b.y) * is);
  const T cr = a.x * b.x + a.y * b.y, ci = a.y * b.x - a.x * b.y;
  C[i] = mk<T>(cr * is, ci * is);
  A[i] = atan2(ci, cr);
}

// Boxcar along the scale axis = scipy.signal.convolve2d(T, win[:, None], 'same') (zero boundary):
// out[j] = sum_i win[i] T[j + (L-1)/2 - i]
template <typename T>
__global__ void k_boxcar_scales(const cplx<T>* __restrict__ in, int nrows, long ld, long ncols,
                                const T* __restrict__ win, int L, cplx<T>* __restrict__ out) {
  const long n = long(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= ncols) return;
  const int j = blockIdx.y, c = (L - 1) / 2;
  T sr = 0, si = 0;
  for (int i = 0; i < L; ++i) {
    const int jj = j + c - i;
    if (jj >= 0 && jj < nrows) {
      const cplx<T> v = in[long(jj) * ld + n];
      sr += win[i] * v.x;
      si += win[i] * v.y;
    }
  }
  out[long(j) * ld + n] = mk<T>(sr, si);
}

// Same sums (same order), but every workgroup walks RB consecutive rows of its 256 columns and keeps the last L
// input rows in a per-thread ring in LDS: every input element is read from memory (RB + L - 1) / RB times
// instead of L times (L = 14 rows for the default dj = 1/12: 332 GB -> 34 GB per smoothing at BASELINE config 5).
template <typename T>
__global__ void k_boxcar_scales_ring(const cplx<T>* __restrict__ in, int nrows, long ld, long ncols,
                                     const T* __restrict__ win, int L, cplx<T>* __restrict__ out, int RB) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  cplx<T>* ring = reinterpret_cast<cplx<T>*>(lds_raw) + threadIdx.x;       // slot s at ring[s * blockDim.x]
  const long n = long(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool live = n < ncols;
  const int j0 = blockIdx.y * RB, c = (L - 1) / 2, jend = (j0 + RB < nrows) ? j0 + RB : nrows;
  const int bias = L * (nrows / L + 2);                                     // keeps (jj + bias) positive
  auto fetch = [&](int jj) {
    return (live && jj >= 0 && jj < nrows) ? in[long(jj) * ld + n] : mk<T>(T(0), T(0));
  };
  for (int jj = j0 + c - L + 1; jj < j0 + c; ++jj) ring[((jj + bias) % L) * blockDim.x] = fetch(jj);
  for (int j = j0; j < jend; ++j) {
    ring[((j + c + bias) % L) * blockDim.x] = fetch(j + c);
    T sr = 0, si = 0;
    for (int i = 0; i < L; ++i) {
      const int jj = j + c - i;
      if (jj >= 0 && jj < nrows) {
        const cplx<T> v = ring[((jj + bias) % L) * blockDim.x];
        sr += win[i] * v.x;
        si += win[i] * v.y;
      }
    }
    if (live) out[long(j) * ld + n] = mk<T>(sr, si);
  }
}

// WCT = |S12|^2 / (S1 S2) with S = S1 + i S2
template <typename T>
__global__ void k_wct_coherence(const cplx<T>* __restrict__ S, const cplx<T>* __restrict__ S12, long ld,
                                long ncols, T* __restrict__ out) {
  const long n = long(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= ncols) return;
  const long i = long(blockIdx.y) * ld + n;
  const cplx<T> s = S[i], c = S12[i];
  out[i] = (c.x * c.x + c.y * c.y) / (s.x * s.y);
}

// ---------------------------------------------------------------------------------------------
// Transform lengths that are not powers of two (the reference's pyfftw branch transforms at len(signal) without
// padding, helpers.py:15-19): Bluestein's identity 2kn = k^2 + n^2 - (k - n)^2 turns a length-n0 DFT into chirp
// multiplications and one circular convolution of power-of-two length M >= 2 n0 - 1, which runs on the FFT engine.
// chirp(m) = e^{sgn * pi i m^2 / n0}; m^2 is reduced mod 2 n0 in integers, so the angle is exact to the last bit.
__device__ __forceinline__ void chirp(long m, long n0, int sgn, double* c, double* s) {
  const unsigned long long r = (unsigned long long)(m * m) % (unsigned long long)(2 * n0);
  sincospi(double(sgn) * double(r) / double(n0), s, c);
}

// out[r, n] = in[r, n] * chirp(n) * scale, n < n0.  MODE IN_REAL: real input; IN_CPLX: complex input (out may be in).
template <typename T, int MODE>
__global__ void k_chirp_mul(const void* in, long in_ld, long n0, int sgn, double scale, cplx<T>* out, long out_ld) {
  const long n = long(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= n0) return;
  double c, s;
  chirp(n, n0, sgn, &c, &s);
  c *= scale; s *= scale;
  const long r = blockIdx.y;
  double xr, xi = 0;
  if constexpr (MODE == IN_REAL) xr = double((static_cast<const T*>(in) + r * in_ld)[n]);
  else { const cplx<T> v = (static_cast<const cplx<T>*>(in) + r * in_ld)[n]; xr = v.x; xi = v.y; }
  out[r * out_ld + n] = mk<T>(T(xr * c - xi * s), T(xr * s + xi * c));
}

// The convolution kernel of length M: b[m] = chirp(m) for |m| < n0 (indices mod M), 0 elsewhere.
template <typename T>
__global__ void k_chirp_kernel(long n0, long M, int sgn, cplx<T>* b) {
  const long m = long(blockIdx.x) * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const long dist = m < n0 ? m : (M - m < n0 ? M - m : -1);
  double c = 0, s = 0;
  if (dist >= 0) chirp(dist, n0, sgn, &c, &s);
  b[m] = mk<T>(T(c), T(s));
}

// A[j, k] = xhat[k] * amp_j * profile(a_j * signed_bin(k)) * chirp(k), k < n0: the filtered spectrum of row j
// (wavelet.py:102-105 at transform length n0) premultiplied for the inverse Bluestein convolution.
template <typename T>
__global__ void k_bluestein_band(const cplx<T>* __restrict__ xhat, const double* __restrict__ a,
                                 const double* __restrict__ amp_re, const double* __restrict__ amp_im, Mother mo,
                                 long n0, cplx<T>* __restrict__ A, long ld) {
  const long k = long(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= n0) return;
  const int j = blockIdx.y;
  const long sk = k < (n0 + 1) / 2 ? k : k - n0;          // numpy.fft.fftfreq order for even and odd n0
  const double g = profile<double>(mo, a[j] * double(sk));
  const double gr = g * amp_re[j], gi = g * amp_im[j];
  const cplx<T> x = xhat[k];
  const double yr = double(x.x) * gr - double(x.y) * gi, yi = double(x.x) * gi + double(x.y) * gr;
  double c, s;
  chirp(k, n0, +1, &c, &s);
  A[long(j) * ld + k] = mk<T>(T(yr * c - yi * s), T(yr * s + yi * c));
}

// ---------------------------------------------------------------------------------------------
// k_icwt: out[n] = coeff * sum_j g(W[j, n]) * w[j]; POWER = false: g = Re (TC98 eq. 11 with w = 1/sqrt(s_j),
// wavelet.py:169-170); POWER = true: g = |.|^2 (scale-averaged power with w = 1/s_j on the selected scales,
// TC98 eq. 24 as used in sample/simple_sample.py:87-91)
template <typename T, bool POWER>
__global__ void k_icwt(const cplx<T>* __restrict__ W, long ldw, long ncols, int nrows,
                       const T* __restrict__ w, T coeff, T* __restrict__ out) {
  const long n = long(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= ncols) return;
  T acc[4] = {0, 0, 0, 0};
  int j = 0;
  for (; j + 4 <= nrows; j += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const cplx<T> v = W[long(j + u) * ldw + n];
      acc[u] += (POWER ? (v.x * v.x + v.y * v.y) : v.x) * w[j + u];
    }
  }
  for (; j < nrows; ++j) {
    const cplx<T> v = W[long(j) * ldw + n];
    acc[0] += (POWER ? (v.x * v.x + v.y * v.y) : v.x) * w[j];
  }
  out[n] = coeff * ((acc[0] + acc[1]) + (acc[2] + acc[3]));
}

// Cross wavelet spectrum W12 = W1 conj(W2) (pycwt/wavelet.py:399).  `out` may be W1 (every thread reads its own
// element of both inputs before it writes).
template <typename T>
__global__ void k_cross_spectrum(const cplx<T>* W1, const cplx<T>* __restrict__ W2, long ld, long ncols,
                                 cplx<T>* out) {
  const long n = long(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= ncols) return;
  const long i = long(blockIdx.y) * ld + n;
  const cplx<T> a = W1[i], b = W2[i];
  out[i] = mk<T>(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}

// Monte-Carlo significance of the coherence (pycwt/wavelet.py:609-630): per-scale histogram of floor(R2 * nbins)
// over the columns [lo_j, hi_j) that lie outside the cone of influence; values outside [0, nbins) and NaNs are
// skipped.  One LDS histogram per workgroup, merged into the global one (accumulated over the draws).
template <typename T>
__global__ void k_coherence_hist(const T* __restrict__ R2, long ld, const long* __restrict__ lo,
                                 const long* __restrict__ hi, int nbins, unsigned long long* __restrict__ hist) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  unsigned* h = reinterpret_cast<unsigned*>(lds_raw);
  const int row = blockIdx.y;
  for (int b = threadIdx.x; b < nbins; b += blockDim.x) h[b] = 0u;
  __syncthreads();
  const T* r = R2 + long(row) * ld;
  const long stop = hi[row];
  for (long n = lo[row] + long(blockIdx.x) * blockDim.x + threadIdx.x; n < stop; n += long(gridDim.x) * blockDim.x) {
    const T v = floor(r[n] * T(nbins));
    if (v >= T(0) && v < T(nbins)) atomicAdd(&h[int(v)], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nbins; b += blockDim.x)
    if (h[b]) atomicAdd(&hist[long(row) * nbins + b], (unsigned long long)h[b]);
}

// k_spectrum_range: out[0] = max_k |xhat[k]|^2, out[1] = sum_k |xhat[k]|^2 over the n bins, out[2 + w] = the sum over the
// QUARTER-OCTAVE window w = 4 b + q of the positive half, 2^b (4 + q) / 4 <= k < 2^b (5 + q) / 4 (bounds rounded up; below
// bin 4 most windows are empty and the others hold one bin): the dynamic range of the spectrum at the resolution of a
// row's pass band (the narrowest built-in filter, Morlet(6), is ~3/4 octave wide at its 1-sigma points), by which a caller
// divides the accuracy it wants (cwt_spectrum_range).  Two launches: every workgroup reduces a contiguous slice (fp64
// accumulation; the windows a slice touches follow from the leading-zero counts of its ends) into
// part[workgroup][2 + WINDOWS]; one workgroup folds those.
constexpr int SPECTRUM_OCTAVES = 32;
constexpr int SPECTRUM_WINDOWS = 4 * SPECTRUM_OCTAVES;
constexpr int SPECTRUM_SLOTS = 2 + SPECTRUM_WINDOWS;
// first bin of window w (w = SPECTRUM_WINDOWS: one past the last)
__host__ __device__ inline long spectrum_window_lo(int w) {
  const int b = w >> 2, q = w & 3;
  const long num = (1L << b) * (4 + q);
  return (num + 3) >> 2;
}
__host__ __device__ inline int spectrum_window_of(long k) {   // k >= 1
  int b = 0;
  while ((2L << b) <= k) ++b;
  int w = 4 * b;
  while (w + 1 < 4 * b + 4 && spectrum_window_lo(w + 1) <= k) ++w;
  return w;
}
template <typename T>
__global__ void __launch_bounds__(256) k_spectrum_range(const cplx<T>* __restrict__ xhat, long n, double* __restrict__ part) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  double* red = reinterpret_cast<double*>(lds_raw);        // 256 doubles of reduction scratch + the workgroup's slots
  double* acc = red + 256;
  if (threadIdx.x < SPECTRUM_SLOTS) acc[threadIdx.x] = 0;
  const long per = (n + gridDim.x - 1) / gridDim.x;
  const long k0 = long(blockIdx.x) * per, k1 = k0 + per < n ? k0 + per : n;
  // windows this slice can touch: [w_lo, w_hi]
  const int w_lo = k0 < 1 ? 0 : spectrum_window_of(k0);
  const int w_hi = k1 < 2 ? 0 : spectrum_window_of(k1 - 1);
  double mx = 0, sm = 0;
  __syncthreads();
  for (int w = w_lo; w <= w_hi && w < SPECTRUM_WINDOWS; ++w) {
    long lo = spectrum_window_lo(w), hi = spectrum_window_lo(w + 1);
    if (hi > n / 2) hi = n / 2;
    if (lo < k0) lo = k0;
    if (hi > k1) hi = k1;
    if (hi <= lo) continue;                                   // (uniform: an empty window below bin 4, or outside the slice)
    double o = 0;
    for (long k = lo + threadIdx.x; k < hi; k += 256) {
      const cplx<T> v = xhat[k];
      o += double(v.x) * double(v.x) + double(v.y) * double(v.y);
    }
    red[threadIdx.x] = o;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
      if (int(threadIdx.x) < s2) red[threadIdx.x] += red[threadIdx.x + s2];
      __syncthreads();
    }
    if (threadIdx.x == 0) acc[2 + w] = red[0];
    __syncthreads();
  }
  for (long k = k0 + threadIdx.x; k < k1; k += 256) {          // (a second pass over the slice: it sits in the L2 now)
    const cplx<T> v = xhat[k];
    const double a = double(v.x) * double(v.x) + double(v.y) * double(v.y);
    mx = a > mx || a != a ? a : mx;                             // NaN propagates
    sm += a;
  }
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (int(threadIdx.x) < s2) { const double o = red[threadIdx.x + s2]; if (o > red[threadIdx.x] || o != o) red[threadIdx.x] = o; }
    __syncthreads();
  }
  if (threadIdx.x == 0) acc[0] = red[0];
  __syncthreads();
  red[threadIdx.x] = sm;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (int(threadIdx.x) < s2) red[threadIdx.x] += red[threadIdx.x + s2];
    __syncthreads();
  }
  if (threadIdx.x == 0) acc[1] = red[0];
  __syncthreads();
  if (threadIdx.x < SPECTRUM_SLOTS) part[long(blockIdx.x) * SPECTRUM_SLOTS + threadIdx.x] = acc[threadIdx.x];
}

// out[q] = fold of part[g][q] over the g workgroups of k_spectrum_range (max for q = 0, sums otherwise); one workgroup.
__global__ void __launch_bounds__(192) k_spectrum_fold(const double* __restrict__ part, int groups, double* __restrict__ out) {
  const int q = threadIdx.x;
  if (q >= SPECTRUM_SLOTS) return;
  double r = 0;
  for (int g = 0; g < groups; ++g) {
    const double o = part[long(g) * SPECTRUM_SLOTS + q];
    if (q == 0) { if (o > r || o != o) r = o; }
    else r += o;
  }
  out[q] = r;
}

// k_time_mean: out[j] = (1/ncols) sum_n |W[j, n]|^2  -- the global wavelet spectrum (power.mean(axis=1),
// sample/simple_sample.py:79).  One workgroup of 256 threads per row, fp64 accumulation.
template <typename T>
__global__ void k_time_mean(const cplx<T>* __restrict__ W, long ldw, long ncols, T* __restrict__ out) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  double* part = reinterpret_cast<double*>(lds_raw);
  const cplx<T>* row = W + long(blockIdx.x) * ldw;
  double acc = 0;
  for (long n = threadIdx.x; n < ncols; n += blockDim.x) {
    const cplx<T> v = row[n];
    acc += double(v.x) * double(v.x) + double(v.y) * double(v.y);
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (int(threadIdx.x) < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = T(part[0] / double(ncols));
}

}  // namespace cwt
