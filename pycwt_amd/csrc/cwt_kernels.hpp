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

#include "cwt_types.hpp"
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
// Overlap-save block transforms with K >= 256: the per-residue rotation from a table in LDS, the per-thread factor and the
// wrap of the aliased index folded into stage 1 (ols_band_body).  0 = the running product with the wrap (A/B).
#ifndef CWT_OLS_ROT_TABLE
#define CWT_OLS_ROT_TABLE 1
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
#ifndef CWT_LB_OLS_F32_HALF   // the same kernel on half-size tiles (256 threads).  With block pairs (CWT_PAIR_F32) the data alone
#define CWT_LB_OLS_F32_HALF (CWT_PAIR_F32 ? 3 : 6)   // are 64 registers: 168 (3 waves per SIMD, no scratch) 2.02 us per row, 128
#endif                                               // (45 spilled) 2.93, one block per workgroup at 80: 2.28
#ifndef CWT_LB_AOLS_F32
#define CWT_LB_AOLS_F32 (CWT_PAIR_F32 ? 4 : 6)       // k_aols_rows<float>: 106 registers as block pairs
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

}  // namespace cwt

#include "cwt_kernels_rows.hpp"      // overlap-save, band-passed, polynomial rows
#include "cwt_kernels_callers.hpp"   // coherence helpers, Bluestein, icwt, spectrum range
