// fft_engine.hpp -- workgroup-level inverse Stockham FFT for gfx950 (CDNA4).
//
// One FFT of length L = 2^logL (16 <= L <= 4096) is spread over L/16 threads;
// every thread keeps 16 complex points in registers (slot e <-> position
// j + e*L/16), does radix-16 butterflies in registers (4x4 decomposition,
// constant twiddles), and exchanges data with the other threads of the same
// FFT through LDS between stages (Stockham autosort: stage with sub-length Ns
// reads position jj + i*L/r and writes (jj-k)*r + k + i*Ns, k = jj mod Ns).
// If log2 L is not a multiple of 4 the last stage is a radix-2/4/8 stage with
// 8/4/2 butterflies per thread, so the register layout on exit equals the
// layout on entry.
//
// A workgroup holds TB independent FFTs.  Two LDS layouts:
//   ROWS   : element `pos` of FFT `t` at t*L + pos      (lanes run along pos)
//   PLANES : element `pos` of FFT `t` at pos*TB + t     (lanes run along t)
// Both go through an XOR swizzle of the low address bits so that the strided
// Stockham writes and the contiguous reads are bank-conflict free on gfx950
// (ds_write_b64: 16-lane groups over 32 banks; ds_read_b64: 32-lane groups over
// 64 banks; MI355X_MICROARCH.md, LDS table).  Real and imaginary parts are
// exchanged one after the other through the same buffer (P*sizeof(T) bytes for
// P complex points per workgroup), which keeps two 8192-point fp64 workgroups
// resident per CU.
//
// Direction: e^{+2*pi*i*k*n/L} (inverse, unnormalised).  The forward transform
// of a real signal is obtained by the caller as conj(inverse(x)).
#pragma once
#include <hip/hip_runtime.h>

namespace cwt {

template <typename T> struct C2;
template <> struct C2<double> { using type = double2; };
template <> struct C2<float> { using type = float2; };
template <typename T> using cplx = typename C2<T>::type;

// Two independent transforms in the lanes of one thread: a pair of floats that the compiler keeps in an aligned register pair
// and drives with gfx950's packed fp32 instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two results per issue
// slot).  The engine below is generic over the DATA type T (float, double or pairf); twiddles and butterfly constants are
// scalars of sc_t<T> and broadcast to both halves (op_sel: no extra instruction).  An LDS element of pairf is 8 bytes, so
// the layouts behave as they do for double.
typedef float pairf __attribute__((vector_size(8)));
template <typename T> struct ScalarOf { using type = T; };
template <> struct ScalarOf<pairf> { using type = float; };
template <typename T> using sc_t = typename ScalarOf<T>::type;

template <typename T>
__host__ __device__ __forceinline__ cplx<T> mk(T x, T y) {
  cplx<T> c;
  c.x = x;
  c.y = y;
  return c;
}
// Streaming (non-temporal) store of one complex value of W: W is written once and never re-read by
// the transform, and keeping it out of the Infinity Cache leaves that cache to the two-pass
// intermediate (measured on MI355X: pass B 1.07 -> 0.77 ms at 16-row chunks, band-limited rows -11 %).
template <typename T>
__device__ __forceinline__ void store_w(cplx<T>* p, T re, T im) {
  typedef T vec2 __attribute__((vector_size(2 * sizeof(T))));
  vec2 v = {re, im};
  __builtin_nontemporal_store(v, reinterpret_cast<vec2*>(p));
}

// Pins a value to a register at this point of the program (an empty asm that "modifies" it): the compiler must
// finish computing it here and may not sink the computation past later barriers.  Used where a result is produced
// long before its use and sinking would keep many more inputs alive than outputs (narrow_phases).
template <typename T>
__device__ __forceinline__ void keep_here(T& v) {
#if defined(__AMDGCN__)
  asm volatile("" : "+v"(v));
#else
  (void)v;
#endif
}

// stage twiddle e^{2 pi i idx / L} from the table (timing-only lab ablation: arithmetic instead of the load)
template <typename T>
__device__ __forceinline__ cplx<T> stage_tw(const cplx<T>* __restrict__ tw, int idx) {
  return tw[idx];
}

template <typename T>
__device__ __forceinline__ cplx<T> cmul(cplx<T> a, cplx<T> b) {
  return mk<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// XOR swizzle of an LDS element index (elements of sizeof(T) bytes).
template <typename T>
__device__ __forceinline__ int lds_swizzle(int a) {
  if constexpr (sizeof(T) == 8) return a ^ ((a >> 4) & 15);
  else return a ^ ((a >> 5) & 31);
}

// Placement of the calling thread inside the workgroup's set of FFTs.
template <typename T, bool PLANES>
struct Geo {
  int logL;   // log2 of the FFT length
  int logTB;  // log2 of the number of FFTs in the workgroup
  int t;      // which FFT
  int j;      // thread index inside the FFT, 0 .. L/16-1
  __device__ __forceinline__ int addr(int pos) const {
    const int a = PLANES ? ((pos << logTB) | t) : ((t << logL) | pos);
    return lds_swizzle<T>(a);
  }
};

// ---- register butterflies (inverse direction) ----------------------------
template <typename T>
__device__ __forceinline__ void r4(T ar, T ai, T br, T bi, T cr, T ci, T dr, T di,  // inputs 0..3
                                   T& y0r, T& y0i, T& y1r, T& y1i, T& y2r, T& y2i, T& y3r, T& y3i) {
  const T s0r = ar + cr, s0i = ai + ci;
  const T s1r = ar - cr, s1i = ai - ci;
  const T s2r = br + dr, s2i = bi + di;
  const T s3r = br - dr, s3i = bi - di;
  y0r = s0r + s2r; y0i = s0i + s2i;
  y2r = s0r - s2r; y2i = s0i - s2i;
  y1r = s1r - s3i; y1i = s1i + s3r;   // s1 + i*s3
  y3r = s1r + s3i; y3i = s1i - s3r;   // s1 - i*s3
}

template <typename T> struct K16 {
  static constexpr T C = T(0.92387953251128675613L);  // cos(pi/8)
  static constexpr T S = T(0.38268343236508977173L);  // sin(pi/8)
  static constexpr T H = T(0.70710678118654752440L);  // sqrt(1/2)
};

// multiply (r,i) by e^{+2*pi*i*Q/16} for the Q that occur in the 4x4 split
template <typename T, int Q>
__device__ __forceinline__ void rot16(T& r, T& i) {
  using R = sc_t<T>;
  constexpr R C = K16<R>::C, S = K16<R>::S, H = K16<R>::H;
  T x = r, y = i;
  if constexpr (Q == 0) { return; }
  else if constexpr (Q == 1) { r = x * C - y * S; i = x * S + y * C; }
  else if constexpr (Q == 2) { r = (x - y) * H; i = (x + y) * H; }
  else if constexpr (Q == 3) { r = x * S - y * C; i = x * C + y * S; }
  else if constexpr (Q == 4) { r = -y; i = x; }
  else if constexpr (Q == 6) { r = (-x - y) * H; i = (x - y) * H; }
  else if constexpr (Q == 9) { r = y * S - x * C; i = -x * S - y * C; }
  else { static_assert(Q < 0, "unsupported rotation"); }
}

// 16-point inverse DFT, natural order in and out.
template <typename T>
__device__ __forceinline__ void bfly16(T (&re)[16], T (&im)[16]) {
  // step 1: DFT4 over a for each b (input index 4a+b); result c kept at slot 4c+b
#pragma unroll
  for (int b = 0; b < 4; ++b)
    r4<T>(re[b], im[b], re[4 + b], im[4 + b], re[8 + b], im[8 + b], re[12 + b], im[12 + b],
          re[b], im[b], re[4 + b], im[4 + b], re[8 + b], im[8 + b], re[12 + b], im[12 + b]);
  // step 2: slot 4c+b *= W16^(b*c)
  rot16<T, 1>(re[5], im[5]);  rot16<T, 2>(re[6], im[6]);   rot16<T, 3>(re[7], im[7]);
  rot16<T, 2>(re[9], im[9]);  rot16<T, 4>(re[10], im[10]); rot16<T, 6>(re[11], im[11]);
  rot16<T, 3>(re[13], im[13]); rot16<T, 6>(re[14], im[14]); rot16<T, 9>(re[15], im[15]);
  // step 3: DFT4 over b for each c; output d of group c is natural index c + 4d
  T xr[16], xi[16];
#pragma unroll
  for (int c = 0; c < 4; ++c)
    r4<T>(re[4 * c], im[4 * c], re[4 * c + 1], im[4 * c + 1], re[4 * c + 2], im[4 * c + 2],
          re[4 * c + 3], im[4 * c + 3],
          xr[c], xi[c], xr[c + 4], xi[c + 4], xr[c + 8], xi[c + 8], xr[c + 12], xi[c + 12]);
#pragma unroll
  for (int n = 0; n < 16; ++n) { re[n] = xr[n]; im[n] = xi[n]; }
}

// R-point inverse DFT (R = 2, 4, 8) on local arrays, natural order in and out.
template <typename T, int R>
__device__ __forceinline__ void bfly_small(T (&re)[R], T (&im)[R]) {
  if constexpr (R == 2) {
    const T ar = re[0], ai = im[0], br = re[1], bi = im[1];
    re[0] = ar + br; im[0] = ai + bi; re[1] = ar - br; im[1] = ai - bi;
  } else if constexpr (R == 4) {
    r4<T>(re[0], im[0], re[1], im[1], re[2], im[2], re[3], im[3],
          re[0], im[0], re[1], im[1], re[2], im[2], re[3], im[3]);
  } else {
    static_assert(R == 8, "radix");
    // input index 2a+b: DFT4 over a for b = 0,1 -> slot 2c+b; twiddle W8^(b*c); DFT2 over b
#pragma unroll
    for (int b = 0; b < 2; ++b)
      r4<T>(re[b], im[b], re[2 + b], im[2 + b], re[4 + b], im[4 + b], re[6 + b], im[6 + b],
            re[b], im[b], re[2 + b], im[2 + b], re[4 + b], im[4 + b], re[6 + b], im[6 + b]);
    rot16<T, 2>(re[3], im[3]);  // W8^1
    rot16<T, 4>(re[5], im[5]);  // W8^2
    rot16<T, 6>(re[7], im[7]);  // W8^3
    T xr[8], xi[8];
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // natural index c + 4d
      xr[c] = re[2 * c] + re[2 * c + 1];     xi[c] = im[2 * c] + im[2 * c + 1];
      xr[c + 4] = re[2 * c] - re[2 * c + 1]; xi[c + 4] = im[2 * c] - im[2 * c + 1];
    }
#pragma unroll
    for (int n = 0; n < 8; ++n) { re[n] = xr[n]; im[n] = xi[n]; }
  }
}

// element m *= w^m, m = 1..R-1 (running product: two live twiddle registers).  A product tree (w^2, w^4, w^8 by squaring,
// depth 4 instead of 15 at the same 14 complex multiplications) measured +1 % in fp64 and +17 % in fp32 (its 15 live
// powers cost registers: spills at the 80-VGPR budget of the fp32 kernels); removing the chain altogether (timing only)
// is worth -2.5 % of the fp64 step.
template <typename T, int R, typename S>
__device__ __forceinline__ void twiddle_chain(T (&re)[R], T (&im)[R], S wr, S wi) {
  S pr = wr, pi = wi;
#pragma unroll
  for (int m = 1; m < R; ++m) {
    const T x = re[m], y = im[m];
    re[m] = x * pr - y * pi;
    im[m] = x * pi + y * pr;
    if (m + 1 < R) {
      const S nr = pr * wr - pi * wi;
      pi = pr * wi + pi * wr;
      pr = nr;
    }
  }
}

// element m *= start * ratio^m, m = 0..R-1 (a stage's twiddles with a per-thread factor and a per-slot ratio folded in)
template <typename T, int R, typename S>
__device__ __forceinline__ void twiddle_chain_from(T (&re)[R], T (&im)[R], S pr, S pi, S wr, S wi) {
#pragma unroll
  for (int m = 0; m < R; ++m) {
    const T x = re[m], y = im[m];
    re[m] = x * pr - y * pi;
    im[m] = x * pi + y * pr;
    if (m + 1 < R) {
      const S nr = pr * wr - pi * wi;
      pi = pr * wi + pi * wr;
      pr = nr;
    }
  }
}

// Stockham exchange of one real plane: slot n goes to position base + n*Ns,
// slot e comes back from position j + e*NT.
template <typename T, bool PLANES>
__device__ __forceinline__ void exchange_plane(T (&v)[16], T* lds, const Geo<T, PLANES>& g, int base,
                                               int logNs, int logNT) {
#pragma unroll
  for (int n = 0; n < 16; ++n) lds[g.addr(base + (n << logNs))] = v[n];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16; ++e) v[e] = lds[g.addr(g.j + (e << logNT))];
  __syncthreads();
}

// last stage when log2 L is not a multiple of 4: radix R = 2^rem, 16/R butterflies per thread
template <typename T, int R>
__device__ __forceinline__ void partial_stage(T (&re)[16], T (&im)[16], int j, int logL, int logNs,
                                              const cplx<sc_t<T>>* __restrict__ tw) {
  constexpr int NB = 16 / R;
  constexpr int LOGR = (R == 2) ? 1 : (R == 4) ? 2 : 3;
  const int logNT = logL - 4;
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int jj = j + (u << logNT);
    const int k = jj & ((1 << logNs) - 1);
    const cplx<sc_t<T>> w = stage_tw<sc_t<T>>(tw, k << (logL - logNs - LOGR));
    T lr[R], li[R];
#pragma unroll
    for (int i = 0; i < R; ++i) { lr[i] = re[u + i * NB]; li[i] = im[u + i * NB]; }
    twiddle_chain<T, R>(lr, li, w.x, w.y);
    bfly_small<T, R>(lr, li);
#pragma unroll
    for (int i = 0; i < R; ++i) { re[u + i * NB] = lr[i]; im[u + i * NB] = li[i]; }
  }
}

// Inverse FFT of length 2^g.logL across the threads of one FFT.
//   in : slot e holds x[g.j + e*L/16]       out: slot e holds X[g.j + e*L/16]
//   lds: P reals of workgroup scratch, tw[p] = e^{2*pi*i*p/L}, p < L.
// Every thread of the workgroup must call this (it contains barriers).
template <typename T, bool PLANES>
__device__ __forceinline__ void wg_ifft(T (&re)[16], T (&im)[16], T* lds, const Geo<T, PLANES>& g,
                                        const cplx<T>* __restrict__ tw) {
  const int logL = g.logL;
  const int logNT = logL - 4;
  const int nfull = logL >> 2;
  const int rem = logL & 3;
  int logNs = 0;
  for (int s = 0; s < nfull; ++s) {
    const int k = g.j & ((1 << logNs) - 1);
    if (s > 0) {
      const cplx<T> w = stage_tw<T>(tw, k << (logL - logNs - 4));
      twiddle_chain<T, 16>(re, im, w.x, w.y);
    }
    bfly16<T>(re, im);
    if (s == nfull - 1 && rem == 0) break;  // results already sit at j + n*L/16
    const int base = ((g.j - k) << 4) + k;
    exchange_plane<T, PLANES>(re, lds, g, base, logNs, logNT);
    exchange_plane<T, PLANES>(im, lds, g, base, logNs, logNT);
    logNs += 4;
  }
  if (rem == 1) partial_stage<T, 2>(re, im, g.j, logL, logNs, tw);
  else if (rem == 2) partial_stage<T, 4>(re, im, g.j, logL, logNs, tw);
  else if (rem == 3) partial_stage<T, 8>(re, im, g.j, logL, logNs, tw);
}


// =============================================================================================
// Compile-time specialised engine for the hot geometries (same algorithm as wg_ifft above).
// All strides are constants, so every LDS access is `base register + immediate offset`, the
// stage loop disappears, and the register allocator sees straight-line code.
//
// LDS layouts (element = one real of type T; re and im planes are exchanged one after the other):
//   PLANES: element pos of FFT t at pos*TB + t, no padding.  Reads (consecutive lanes ->
//           consecutive elements) are conflict free; stage-0 writes with TB = 8 (fp64) are a
//           harmless 2-way conflict, everything else is conflict free.
//   ROWS  : a = t*L + pos, physical index a + (a >> 4)  (one pad element per 16): the stride-16
//           Stockham writes become stride 17, and all offsets stay additive because every
//           stride used is a multiple of 16 elements.  Requires L >= 256.
// Synchronisation: PLANES -> workgroup barriers.  ROWS with every FFT inside one wavefront (L <= 1024) ->
// no s_barrier at all: LDS operations of one wave execute in order, so a compiler-level fence is
// enough between the write and the read phase.
namespace ct {

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// PADDED (PLANES only, TB <= 8, TB*L >= 256): the planes layout with one pad element per 16, a + (a >> 4), a = pos*TB + t.
// Without it the stage-0 writes of a workgroup with few FFTs (lanes 16*TB elements apart) all fall on the same banks
// (32-way conflict at TB = 2, 16-way at TB = 4); with it lane pairs are 17*TB apart.  Offsets stay immediates: every
// stride is a multiple of 16 elements except the stage-0 slot stride TB, whose pad (n*TB) >> 4 does not depend on t.
// WAVE (PLANES only): the TB FFTs of this instance live inside ONE wavefront (TB * L / 16 <= 64 threads, lds = that
// wave's own slice): wave-level synchronisation instead of workgroup barriers, so the waves of a workgroup run on their own.
template <typename T, int LOGL, int LOGTB, bool PLANES, bool PADDED = false, bool WAVE = false>
struct Fft {
  static constexpr int L = 1 << LOGL, TB = 1 << LOGTB, LOGNT = LOGL - 4, NT = 1 << LOGNT;
  static constexpr int NFULL = LOGL / 4, REM = LOGL % 4;
  static constexpr bool WAVE_LOCAL = (!PLANES && NT <= 64) || WAVE;   // every FFT lives inside one wavefront
  static_assert(!WAVE || (PLANES && (TB * NT <= 64)), "WAVE: all FFTs of the instance inside one wavefront");
  static_assert(PLANES || LOGL >= 8, "ROWS layout needs L >= 256");
  static_assert(!PADDED || (PLANES && LOGTB <= 4 && LOGL + LOGTB >= 8), "padded planes: TB <= 16, tile >= 256");
  static constexpr int LDS_ELEMS = (PLANES && !PADDED) ? (TB * L) : (TB * L + ((TB * L) >> 4));

  int t, j;   // FFT index in the workgroup, thread index in the FFT

  __device__ __forceinline__ static void sync() {
    if constexpr (WAVE_LOCAL) wave_sync(); else __syncthreads();
  }
  // physical element index of position pos (pos may carry any multiple-of-16 part)
  __device__ __forceinline__ int phys(int pos) const {
    if constexpr (PADDED) { const int a = (pos << LOGTB) + t; return a + (a >> 4); }
    else if constexpr (PLANES) return (pos << LOGTB) + t;
    else { const int a = (t << LOGL) + pos; return a + (a >> 4); }
  }
  // physical stride that corresponds to a logical stride (multiple of 16 for ROWS)
  static constexpr int pstride(int s) {
    return PADDED ? ((s << LOGTB) + ((s << LOGTB) >> 4)) : PLANES ? (s << LOGTB) : (s + (s >> 4));
  }

  // one real plane: slot n -> position wbase + n*Ns ; slot e <- position j + e*NT
  template <int LOGNS>
  __device__ __forceinline__ void exchange(T (&v)[16], T* lds, int wphys, int rphys) const {
    constexpr int WS = (LOGNS == 0) ? (PLANES ? TB : 1) : pstride(1 << LOGNS);
    constexpr int RS = (NT >= 16 || PLANES) ? pstride(NT) : 0;
    if constexpr (PADDED && LOGNS == 0) {      // slot stride TB < 16: wphys = phys(16 j), slot n adds n*TB + its pad
#pragma unroll
      for (int n = 0; n < 16; ++n) lds[wphys + n * TB + ((n * TB) >> 4)] = v[n];
    } else {
#pragma unroll
      for (int n = 0; n < 16; ++n) lds[wphys + n * WS] = v[n];
    }
    sync();
    if constexpr (NT >= 16 || PLANES) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = lds[rphys + e * RS];
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = lds[phys(j + e * NT)];
    }
    sync();
  }

  template <int S>
  __device__ __forceinline__ void full_stage(T (&re)[16], T (&im)[16], T* lds,
                                             const cplx<sc_t<T>>* __restrict__ tw, int rphys) const {
    constexpr int LOGNS = 4 * S;
    const int k = j & ((1 << LOGNS) - 1);
    if constexpr (S > 0) {
      const cplx<sc_t<T>> w = stage_tw<sc_t<T>>(tw, k << (LOGL - LOGNS - 4));
      twiddle_chain<T, 16>(re, im, w.x, w.y);
    }
    bfly16<T>(re, im);
    if constexpr (S == NFULL - 1 && REM == 0) return;
    const int wphys = phys(((j - k) << 4) + k);
    exchange<LOGNS>(re, lds, wphys, rphys);
    exchange<LOGNS>(im, lds, wphys, rphys);
  }

  // As run(), for callers that (i) hand the 16 inputs of stage 0 over in the cyclic slot order (e + ew) mod 16 and (ii) left a
  // factor A(source thread) = A0 sigma^(16 j_src) ... out of them -- see ols_band_body: stage 1 then multiplies slot m by
  // start (w sigma)^m instead of w^m, with start = W16^(c ew) A0 of THIS thread (c = j mod 16).  Needs two full stages.
  __device__ __forceinline__ void run_pre(T (&re)[16], T (&im)[16], T* lds, const cplx<sc_t<T>>* __restrict__ tw,
                                          cplx<sc_t<T>> start, cplx<sc_t<T>> sigma) const {
    static_assert(NFULL >= 2, "run_pre: L >= 256");
    const int rphys = phys(j);
    full_stage<0>(re, im, lds, tw, rphys);
    {
      const int k = j & 15;
      const cplx<sc_t<T>> w = stage_tw<sc_t<T>>(tw, k << (LOGL - 8));
      const cplx<sc_t<T>> ratio = cmul<sc_t<T>>(w, sigma);
      twiddle_chain_from<T, 16>(re, im, start.x, start.y, ratio.x, ratio.y);
      bfly16<T>(re, im);
      if constexpr (!(NFULL == 2 && REM == 0)) {
        const int wphys = phys(((j - k) << 4) + k);
        exchange<4>(re, lds, wphys, rphys);
        exchange<4>(im, lds, wphys, rphys);
      }
    }
    if constexpr (NFULL >= 3) full_stage<2>(re, im, lds, tw, rphys);
    if constexpr (REM == 1) partial_stage<T, 2>(re, im, j, LOGL, 4 * NFULL, tw);
    if constexpr (REM == 2) partial_stage<T, 4>(re, im, j, LOGL, 4 * NFULL, tw);
    if constexpr (REM == 3) partial_stage<T, 8>(re, im, j, LOGL, 4 * NFULL, tw);
  }

  // in: slot e = x[j + e*NT]; out: slot e = X[j + e*NT].  lds: LDS_ELEMS reals.
  __device__ __forceinline__ void run(T (&re)[16], T (&im)[16], T* lds,
                                      const cplx<sc_t<T>>* __restrict__ tw) const {
    const int rphys = phys(j);
    full_stage<0>(re, im, lds, tw, rphys);
    if constexpr (NFULL >= 2) full_stage<1>(re, im, lds, tw, rphys);
    if constexpr (NFULL >= 3) full_stage<2>(re, im, lds, tw, rphys);
    if constexpr (REM == 1) partial_stage<T, 2>(re, im, j, LOGL, 4 * NFULL, tw);
    if constexpr (REM == 2) partial_stage<T, 4>(re, im, j, LOGL, 4 * NFULL, tw);
    if constexpr (REM == 3) partial_stage<T, 8>(re, im, j, LOGL, 4 * NFULL, tw);
  }
};

}  // namespace ct
}  // namespace cwt
