// cwt_kernels_callers.hpp -- kernels of the callers of the path: coherence helpers, boxcar, Bluestein chirps,
// icwt / scale reductions, spectrum range, time mean.  Included by cwt_kernels.hpp.
#pragma once
#include "cwt_kernels.hpp"

namespace cwt {

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
// A pure read stream over a matrix that is read once: NON-TEMPORAL loads (the lines are not kept in L2 / the Infinity Cache) and
// 128-thread workgroups with 8 rows in flight per thread [measured, tools/microbench/icwt_read.hip, profiles/r06_icwt_read.txt:
// plain loads 5.4-5.6 TB/s whatever the issue pattern; non-temporal 6.1, with 128-thread workgroups 6.3 TB/s].
template <typename T>
__device__ __forceinline__ cplx<T> load_once(const cplx<T>* p) {
  typedef T vec2 __attribute__((vector_size(2 * sizeof(T))));
  const vec2 v = __builtin_nontemporal_load(reinterpret_cast<const vec2*>(p));
  return mk<T>(v[0], v[1]);
}
constexpr int ICWT_THREADS = 128, ICWT_INFLIGHT = 8;
template <typename T, bool POWER>
__global__ void __launch_bounds__(ICWT_THREADS)
k_icwt(const cplx<T>* __restrict__ W, long ldw, long ncols, int nrows,
       const T* __restrict__ w, T coeff, T* __restrict__ out) {
  const long n = long(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= ncols) return;
  constexpr int U = ICWT_INFLIGHT;
  T acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = T(0);
  int j = 0;
  for (; j + U <= nrows; j += U) {
    cplx<T> v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = load_once<T>(W + long(j + u) * ldw + n);
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] += (POWER ? (v[u].x * v[u].x + v[u].y * v[u].y) : v[u].x) * w[j + u];
  }
  for (; j < nrows; ++j) {
    const cplx<T> v = load_once<T>(W + long(j) * ldw + n);
    acc[0] += (POWER ? (v.x * v.x + v.y * v.y) : v.x) * w[j];
  }
  out[n] = coeff * (((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7])));
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
template <int UNUSED = 0>   // (a template only so that the definition may live in a header shared by several translation units)
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
  double a4[4] = {0, 0, 0, 0};                              // four non-temporal loads in flight per thread (a row is read once)
  long n = threadIdx.x;
  for (; n + 3 * long(blockDim.x) < ncols; n += 4 * long(blockDim.x)) {
    cplx<T> v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = load_once<T>(row + n + u * long(blockDim.x));
#pragma unroll
    for (int u = 0; u < 4; ++u) a4[u] += double(v[u].x) * double(v[u].x) + double(v[u].y) * double(v[u].y);
  }
  for (; n < ncols; n += blockDim.x) {
    const cplx<T> v = load_once<T>(row + n);
    a4[0] += double(v.x) * double(v.x) + double(v.y) * double(v.y);
  }
  const double acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (int(threadIdx.x) < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = T(part[0] / double(ncols));
}

// ---------------------------------------------------------------------------------------------
// Surrogate series of the Monte-Carlo significance (pycwt/wavelet.py:609-613, helpers.py:146-173) made on the device.
// Philox4x32-10 (Salmon et al. 2011), counter = (index of the pair of outputs, stream offset), key = seed; Box-Muller on two
// 53-bit uniforms gives two independent N(0, 1) deviates per counter.  Reproducible per (seed, offset, index) and independent
// of the launch geometry; NOT the sequence of NumPy's generator -- the host path stays the seed-for-seed one.
struct Philox {
  unsigned c[4], k[2];
  __host__ __device__ static inline void mulhilo(unsigned a, unsigned b, unsigned* hi, unsigned* lo) {
    const unsigned long long p = (unsigned long long)a * (unsigned long long)b;
    *hi = unsigned(p >> 32);
    *lo = unsigned(p);
  }
  __host__ __device__ inline void round() {
    unsigned hi0, lo0, hi1, lo1;
    mulhilo(0xD2511F53u, c[0], &hi0, &lo0);
    mulhilo(0xCD9E8D57u, c[2], &hi1, &lo1);
    const unsigned n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  }
  __host__ __device__ inline void run10() {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      round();
      if (r < 9) { k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u; }
    }
  }
};

// out[i] = scale * N(0, 1), i < n.  Thread t makes outputs 2t and 2t + 1.
template <typename T>
__global__ void __launch_bounds__(256) k_normal_fill(unsigned long long seed, unsigned long long offset, long n, double scale,
                                                     T* __restrict__ out) {
  const long t = long(blockIdx.x) * 256 + threadIdx.x;
  if (2 * t >= n) return;
  Philox g;
  g.c[0] = unsigned(t); g.c[1] = unsigned((unsigned long long)t >> 32);
  g.c[2] = unsigned(offset); g.c[3] = unsigned(offset >> 32);
  g.k[0] = unsigned(seed); g.k[1] = unsigned(seed >> 32);
  g.run10();
  // two uniforms in (0, 1]: 53 bits each (u1 never 0: the logarithm is finite)
  const double u1 = (double((((unsigned long long)g.c[0]) << 21) ^ (g.c[1] >> 11)) + 1.0) * (1.0 / 9007199254740992.0);
  const double u2 = double((((unsigned long long)g.c[2]) << 21) ^ (g.c[3] >> 11)) * (1.0 / 9007199254740992.0);
  const double r = sqrt(-2.0 * log(u1)) * scale, ang = 6.283185307179586476925 * u2;
  out[2 * t] = T(r * cos(ang));
  if (2 * t + 1 < n) out[2 * t + 1] = T(r * sin(ang));
}

// AR(1) filter y[i] = g y[i-1] + e[i] over e[0 .. tau + n) started from y[-1] = 0, the first tau outputs dropped
// (scipy.signal.lfilter([1, 0], [1, -g], e, axis=0)[tau:], what helpers.py:170 means): out[j] = y[tau + j], j < n.
// One thread per segment of SEG outputs; it runs the recursion from `warm` samples before its segment -- all the way from
// e[0] where that is nearer, else far enough that the forgotten history is below g^warm <= 1e-17 of the signal.
template <typename T>
__global__ void __launch_bounds__(256) k_ar1_filter(const T* __restrict__ e, long tau, long n, double g, long warm, int seg,
                                                    T* __restrict__ out) {
  const long s = (long(blockIdx.x) * 256 + threadIdx.x) * seg;      // first output of this thread
  if (s >= n) return;
  const long first = tau + s;                                        // its index in e
  long i = first - warm;
  if (i < 0) i = 0;
  double y = 0;
  for (; i < first; ++i) y = g * y + double(e[i]);
  const long end = s + seg < n ? s + seg : n;
  for (long j = s; j < end; ++j) {
    y = g * y + double(e[tau + j]);
    out[j] = T(y);
  }
}

}  // namespace cwt
