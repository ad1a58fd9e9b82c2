// cwt_kernels_rows.hpp -- the row kernels of the fast forms: overlap-save on the signal (k_ols_*), on the band-passed complex
// signal (k_aols_*), polynomial rows (k_poly_*).  Included by cwt_kernels.hpp.
#pragma once
#include "cwt_kernels.hpp"

namespace cwt {

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
  const int sh = logN - LOGP - logx;
  const unsigned pm = (1u << (LOGP + logx)) - 1u;
  const int c0 = ((0 - rd.k_lo) & (K - 1)) >> (LOGK - 4);               // (-k_lo mod K) / NT, k_lo = 0 mod NT
  T re[16], im[16];
  if constexpr (LOGK >= 8 && CWT_OLS_ROT_TABLE) {
    // Slot e' <- bin k_lo + j + e' NT, which sits at FFT position j + ((e' - c0) mod 16) NT: the 16 inputs of a thread in
    // the cyclic order that makes their bins consecutive -- no wrap inside the thread.  Its rotation is
    //   e^{2 pi i (k_lo + j + e' NT) r / P_b} = A(j, r) S(r)^e',   S^e' = e^{2 pi i r e' / (16 M)}  (M = P_b / K residues)
    // S^e' comes from a table of 16 x TB entries in LDS (one look-up per thread to build it), A is left out here: stage 0 is
    // linear in a thread's inputs, its outputs go to 16 other threads, and what arrives in thread j', slot m, came from
    // thread (j' >> 4) + m NT/16 -- so stage 1 applies A together with its own twiddles as start (w sigma)^m, and the cyclic
    // order as the constant W16^(c ew) of its output index c = j' mod 16 (Fft::run_pre).  16 complex multiplications per
    // thread here instead of 32 + the 15 of the wrap.
    constexpr int TB = 1 << LOGTB;
    cplx<T>* ttab = ytile + K;
    if (int(threadIdx.x) < 16 * TB) {
      const unsigned e = threadIdx.x >> LOGTB, rr = (g << LOGTB) + (threadIdx.x & (TB - 1));
      ttab[threadIdx.x] = (tw_all + ((16u << (LOGTB + logx)) - 2u))[rr * e];
    }
    __syncthreads();
    const cplx<T> a0 = twn(((unsigned(rd.k_lo + (f.j >> 4)) * r) & pm) << sh);
    const cplx<T> sigma = twn(((unsigned(NT >> 4) * r) & pm) << sh);
    const int ew = (16 - c0) & 15;
    const cplx<T> phi = (tw_all + 14)[((f.j & 15) * ew) & 15];           // W16^(c ew)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const cplx<T> y = ytile[f.j + (((e + ew) & 15) << (LOGK - 4))];
      const cplx<T> tt = ttab[(e << LOGTB) + f.t];
      re[e] = y.x * tt.x - y.y * tt.y;
      im[e] = y.x * tt.y + y.y * tt.x;
    }
    __syncthreads();                                       // the band tile aliases the exchange buffer
    f.run_pre(re, im, lds, tw_all + (K - 2), cmul<T>(phi, a0), sigma);
  } else {
    __syncthreads();
    const cplx<T> step = twn(((unsigned(NT) * r) & pm) << sh);
    const cplx<T> rhoc = twn(((0u - (r << LOGK)) & pm) << sh);           // e^{-2 pi i K r / P}
    const int ew = 16 - c0;                                               // uniform: first slot after the wrap
    cplx<T> cur = twn(((unsigned(rd.k_lo + f.j + c0 * NT) * r) & pm) << sh);
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
  }
  const unsigned nlim_u = nlim > 0 ? unsigned(nlim) : 0u;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    // n_local = (P_b / K) (j + e NT) + r; with P_b = P this is thread + e P/16
    const int nl = (((f.j + e * NT) << (LOGTB + logx)) | int(r)) - H;
    if (unsigned(nl) < nlim_u) store_w<T>(wout + nl, re[e], im[e]);
  }
}

// ---- complex64: TWO blocks of a row per workgroup -------------------------------------------------------------------
// A block transform is ~1000 vector instructions per thread of a 4096-point tile (2.3 us per row of 2^20 outputs against 1.0 for
// its bytes), of which the compiler packs a third on its own.  gfx950 issues packed fp32 instructions -- two results per slot --
// so in complex64 a workgroup transforms blocks 2u and 2u + 1 of its row TOGETHER: every data register is a pairf (block 2u in
// the low half, 2u + 1 in the high half), filter table, twiddles and addresses are shared, the exchange buffer holds 8-byte
// elements.  The odd block out at the end of a row is transformed twice and stored once (nlim of the second half <= 0).
// Which kernels: ols_pairs() / aols_pairs() in cwt_types.hpp (the host sizes the grids from the same functions);
// -DCWT_PAIR_F32=0 keeps one block per workgroup everywhere (A/B).  [measured, fp32 DOG / Paul at N = 2^20: the instruction
// count per block falls 1.8 x, the 4096-point rows gain 11 % (2.28 -> 2.02 us), the rows on the band-passed signal 10-20 %
// (2.35-2.47 -> 1.94-2.14): these kernels are NOT bound by the issue rate alone]
__device__ __forceinline__ pairf pair_of(float a, float b) { pairf v = {a, b}; return v; }

template <int LOGP>
__device__ __forceinline__ void ols_full_body2(const float2* __restrict__ xb0, const float2* __restrict__ xb1, const RowDesc& rd,
                                               const float2* __restrict__ gt, const float2* __restrict__ tw_all,
                                               float2* __restrict__ w0, float2* __restrict__ w1, int H, int nlim0, int nlim1,
                                               pairf* lds) {
  constexpr int P = 1 << LOGP, NT = P >> 4;
  using F = ct::Fft<pairf, LOGP, 0, false>;
  F f;
  f.t = 0;
  f.j = threadIdx.x;
  pairf re[16], im[16];
  float2 gv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {                          // all loads first, then the arithmetic
    const int ks = signed_bin(f.j + e * NT, P);
    const float2 a = ols_load<float>(xb0, rd, ks), b = ols_load<float>(xb1, rd, ks);
    re[e] = pair_of(a.x, b.x);
    im[e] = ks < 0 ? pair_of(-a.y, -b.y) : pair_of(a.y, b.y);
    gv[e] = gt[f.j + e * NT];
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const pairf x = re[e], y = im[e];
    re[e] = x * gv[e].x - y * gv[e].y;
    im[e] = x * gv[e].y + y * gv[e].x;
  }
  f.run(re, im, lds, tw_all + (P - 2));
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int nl = f.j + e * NT - H;
    if (nl >= 0 && nl < nlim0) store_w<float>(w0 + nl, re[e][0], im[e][0]);
    if (nl >= 0 && nl < nlim1) store_w<float>(w1 + nl, re[e][1], im[e][1]);
  }
}

template <int LOGK, int LOGP>
__device__ __forceinline__ void ols_band_body2(const float2* __restrict__ xb0, const float2* __restrict__ xb1, const RowDesc& rd,
                                               const float2* __restrict__ gt, const float2* __restrict__ tw_all,
                                               const TwN<float>& twn, int logN, float2* __restrict__ w0, float2* __restrict__ w1,
                                               int H, int nlim0, int nlim1, pairf* lds, int logx, unsigned g) {
  constexpr int LOGTB = LOGP - LOGK, K = 1 << LOGK, NT = K >> 4, BD = 1 << (LOGP - 4);
  using F = ct::Fft<pairf, LOGK, LOGTB, true, (LOGTB <= CWT_OLS_PAD_LOGTB)>;
  F f;
  f.t = threadIdx.x & ((1 << LOGTB) - 1);
  f.j = threadIdx.x >> LOGTB;
  const unsigned r = (g << LOGTB) + unsigned(f.t);
  pairf* ytile = lds;                                     // the filtered band of both blocks: (re, im) pairs, 16 bytes per bin
  constexpr int NQ = K > BD ? K / BD : 1;
  float2 ya[NQ], yb[NQ], gv[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int q = int(threadIdx.x) + i * BD;
    if (q < K) {
      const int ks = rd.k_lo + ((q - rd.k_lo) & (K - 1));
      ya[i] = ols_load<float>(xb0, rd, ks);
      yb[i] = ols_load<float>(xb1, rd, ks);
      gv[i] = gt[q];
    }
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int q = int(threadIdx.x) + i * BD;
    if (q < K) {
      const int ks = rd.k_lo + ((q - rd.k_lo) & (K - 1));
      const pairf x = pair_of(ya[i].x, yb[i].x), y = ks < 0 ? pair_of(-ya[i].y, -yb[i].y) : pair_of(ya[i].y, yb[i].y);
      ytile[2 * q] = x * gv[i].x - y * gv[i].y;
      ytile[2 * q + 1] = x * gv[i].y + y * gv[i].x;
    }
  }
  const int sh = logN - LOGP - logx;
  const unsigned pm = (1u << (LOGP + logx)) - 1u;
  const int c0 = ((0 - rd.k_lo) & (K - 1)) >> (LOGK - 4);
  pairf re[16], im[16];
  if constexpr (LOGK >= 8 && CWT_OLS_ROT_TABLE) {        // as ols_band_body: rotation from an LDS table, the rest folded into stage 1
    constexpr int TB = 1 << LOGTB;
    float2* ttab = reinterpret_cast<float2*>(ytile + 2 * K);
    if (int(threadIdx.x) < 16 * TB) {
      const unsigned e = threadIdx.x >> LOGTB, rr = (g << LOGTB) + (threadIdx.x & (TB - 1));
      ttab[threadIdx.x] = (tw_all + ((16u << (LOGTB + logx)) - 2u))[rr * e];
    }
    __syncthreads();
    const float2 a0 = twn(((unsigned(rd.k_lo + (f.j >> 4)) * r) & pm) << sh);
    const float2 sigma = twn(((unsigned(NT >> 4) * r) & pm) << sh);
    const int ew = (16 - c0) & 15;
    const float2 phi = (tw_all + 14)[((f.j & 15) * ew) & 15];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int q = f.j + (((e + ew) & 15) << (LOGK - 4));
      const pairf yx = ytile[2 * q], yy = ytile[2 * q + 1];
      const float2 tt = ttab[(e << LOGTB) + f.t];
      re[e] = yx * tt.x - yy * tt.y;
      im[e] = yx * tt.y + yy * tt.x;
    }
    __syncthreads();                                     // the band tile aliases the exchange buffer
    f.run_pre(re, im, lds, tw_all + (K - 2), cmul<float>(phi, a0), sigma);
  } else {
    __syncthreads();
    const float2 step = twn(((unsigned(NT) * r) & pm) << sh);
    const float2 rhoc = twn(((0u - (r << LOGK)) & pm) << sh);           // e^{-2 pi i K r / P}
    const int ew = 16 - c0;
    float2 cur = twn(((unsigned(rd.k_lo + f.j + c0 * NT) * r) & pm) << sh);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const pairf yx = ytile[2 * (f.j + e * NT)], yy = ytile[2 * (f.j + e * NT) + 1];
      re[e] = yx * cur.x - yy * cur.y;
      im[e] = yx * cur.y + yy * cur.x;
      if (e < 15) cur = cmul<float>(cur, step);
      if (e + 1 == ew) cur = cmul<float>(cur, rhoc);                     // uniform branch
    }
    __syncthreads();                                       // the band tile aliases the exchange buffer
    f.run(re, im, lds, tw_all + (K - 2));
  }
  const unsigned lim0 = nlim0 > 0 ? unsigned(nlim0) : 0u, lim1 = nlim1 > 0 ? unsigned(nlim1) : 0u;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int nl = (((f.j + e * NT) << (LOGTB + logx)) | int(r)) - H;
    if (unsigned(nl) < lim0) store_w<float>(w0 + nl, re[e][0], im[e][0]);
    if (unsigned(nl) < lim1) store_w<float>(w1 + nl, re[e][1], im[e][1]);
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
         long ncols, unsigned wg0) {
  // wg0 (a multiple of 8): this launch covers the workgroups wg0 ... of the class list -- the classes on blocks of one tile and the
  // classes on longer blocks go in two launches where their block spectra come from two kernels (serial schedule)
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  T* lds = reinterpret_cast<T*>(lds_raw);
  constexpr int P = 1 << LOGP;
  const unsigned wg = blockIdx.x + wg0;
  // this workgroup's class: every class record is read from the kernel arguments at a fixed address (one round of scalar
  // loads for all 16) and selected with uniform compares -- no load whose address depends on an earlier load
  OlsClass oc = cls.c[0];
#pragma unroll
  for (int i = 1; i < OLS_MAX_CLASSES; ++i)
    if (int(wg) >= cls.wg_first[i]) oc = cls.c[i];
  const unsigned local = wg - unsigned(oc.wg_first);
  const int logx = oc.logb - LOGP;
  const unsigned g = (local >> 3) & ((1u << logx) - 1u);      // which part of the block's residues
  // (signal, block) pairs vb = signal * nblocks + block: XCD (workgroup id & 7) takes every 8th pair and walks the
  // nrs rows of that signal's class back to back; the class's rows are stored scale by scale, nsig signals each
  const unsigned seq = local >> (3 + logx), nsig = unsigned(oc.nsig), nrs = unsigned(oc.nrows) / nsig;
  // unit = one block, or (complex64) the blocks 2u, 2u + 1
  constexpr bool PAIR = ols_pairs(sizeof(T), LOGP);
  const unsigned nunits = PAIR ? (unsigned(oc.nblocks) + 1u) >> 1 : unsigned(oc.nblocks);
  const unsigned vb = (seq / nrs) * 8u + (local & 7u);
  if (vb >= nunits * nsig) return;
  const unsigned sig = vb / nunits, unit = vb - sig * nunits, blk = PAIR ? 2u * unit : unit;
  const RowDesc rd = rows[oc.row_first + int((seq % nrs) * nsig + sig)];
  const int H = oc.halo, L = (P << logx) - 2 * H;
  const cplx<T>* xb = xs + rd.spec_off + oc.xs_off + long(blk) * ((P << logx) / 2 + 8);   // spec_off: the row's signal (batch)
  const long col0 = long(blk) * L;
  const long left = ncols - col0;
  const int nlim = left < L ? int(left) : L;
  cplx<T>* wout = W + long(rd.out_row) * ldw + col0;
  const cplx<T>* gt = gtab + rd.tab_off;
  if constexpr (PAIR) {
    const bool two = blk + 1u < unsigned(oc.nblocks);
    const float2* xb1 = two ? xb + ((P << logx) / 2 + 8) : xb;
    const long left1 = left - L;
    const int nlim1 = !two ? 0 : left1 < L ? int(left1) : L;
    pairf* lds2 = reinterpret_cast<pairf*>(lds_raw);
#define CWT_OLS_CASE2(LK)                                                                                                    \
  case LK:                                                                                                                    \
    if constexpr (LK < LOGP) ols_band_body2<LK, LOGP>(xb, xb1, rd, gt, tw_all, twn, logN, wout, wout + L, H, nlim, nlim1, lds2, logx, g); \
    else ols_full_body2<LOGP>(xb, xb1, rd, gt, tw_all, wout, wout + L, H, nlim, nlim1, lds2);                                  \
    break;
    switch (rd.logK) {
      CWT_OLS_CASE2(4) CWT_OLS_CASE2(5) CWT_OLS_CASE2(6) CWT_OLS_CASE2(7) CWT_OLS_CASE2(8) CWT_OLS_CASE2(9)
      CWT_OLS_CASE2(10) CWT_OLS_CASE2(11) CWT_OLS_CASE2(12) CWT_OLS_CASE2(13)
      default: ols_full_body2<LOGP>(xb, xb1, rd, gt, tw_all, wout, wout + L, H, nlim, nlim1, lds2); break;
    }
#undef CWT_OLS_CASE2
    return;
  }
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
  double v;
  if (MK == MOTHER_PAUL && rd.aux_off == 1) {
    // continuation through f = 0 (AolsGeom::zc_*): block bins above P/2 are the negative bins
    const double f = rd.a * double(kappa > (P >> 1) ? kappa - P : kappa);
    const double arg = (-f - g.zc_c) / g.zc_w;
    v = arg > 9.0 ? 0.0 : ipow<double>(f, mo.m) * exp(-f) * 0.5 * erfc(arg);      // (erfc(9) / 2 = 2e-37: beyond, exp(-f) may overflow)
    if (kappa > (P >> 1)) {
      // a row whose filter has not died out at Nyquist: ALSO continued past Nyquist and tapered to nothing by 3/4 cycle per
      // sample (the host only accepts the row if the taper below 0 starts above that: a s w_N > 4 (c + 6 w))
      const double fc = double(kappa) / double(P);
      const double up = profile_k<double, MK>(mo, rd.a * double(kappa)) * 0.5 * erfc(g.z * (fc - 0.625) / 0.125);
      v += fc < 0.76 ? up : 0.0;
    }
  } else {
    v = profile_k<double, MK>(mo, rd.a * double(kappa)) * aols_window(g, double(kappa) / double(P));
  }
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
__global__ void __launch_bounds__(1 << (LOGP - 4), (sizeof(T) == 8 ? (LOGP == 12 ? CWT_LB_OLS_F64_HALF : CWT_LB_OLS_F64) : CWT_LB_AOLS_F32))
k_aols_rows(const cplx<T>* __restrict__ xs, const RowDesc* __restrict__ rows, const T* __restrict__ gtab,
            const cplx<T>* __restrict__ tw_all, AolsGeom g, const cplx<T>* __restrict__ xhat, long nyq, cplx<T>* __restrict__ W,
            long ldw, long ncols) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  T* lds = reinterpret_cast<T*>(lds_raw);
  constexpr int P = 1 << LOGP, NT = P >> 4;
  constexpr bool PAIR = aols_pairs(sizeof(T));                 // complex64: blocks 2u, 2u + 1 in one workgroup (see ols_band_body2)
  using F = ct::Fft<T, LOGP, 0, false>;
  const unsigned seq = blockIdx.x >> 3;
  const unsigned unit = (seq / unsigned(g.nrows)) * 8u + (blockIdx.x & 7u);
  const unsigned blk = PAIR ? 2u * unit : unit;
  if (blk >= unsigned(g.nblocks)) return;
  const RowDesc rd = rows[blockIdx.y * unsigned(g.nrows) + seq % unsigned(g.nrows)];
  const int H = g.halo, L = P - 2 * H;
  const cplx<T>* xb = xs + (long(blockIdx.y) * g.nblocks + long(blk)) * (P + 8);
  const T* gt = gtab + rd.tab_off;
  const long col0 = long(blk) * L, left = ncols - col0;
  const int nlim = left < L ? int(left) : L;
  cplx<T>* wout = W + long(rd.out_row) * ldw + col0;
  if constexpr (PAIR) {
    using F2 = ct::Fft<pairf, LOGP, 0, false>;
    const bool two = blk + 1u < unsigned(g.nblocks);
    const float2* xb1 = two ? xb + (P + 8) : xb;
    const long left1 = left - L;
    const int nlim1 = !two ? 0 : left1 < L ? int(left1) : L;
    float2* wout1 = wout + L;
    F2 f;
    f.t = 0;
    f.j = threadIdx.x;
    pairf re[16], im[16];
    float gv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {                        // all loads first, then the arithmetic
      const float2 a = xb[f.j + e * NT], b = xb1[f.j + e * NT];
      re[e] = pair_of(a.x, b.x); im[e] = pair_of(a.y, b.y);
      gv[e] = gt[f.j + e * NT];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) { re[e] *= gv[e]; im[e] *= gv[e]; }
    f.run(re, im, reinterpret_cast<pairf*>(lds_raw), tw_all + (P - 2));
    if (rd.nterms == 1) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int nl = f.j + e * NT - H;
        if (nl >= 0 && nl < nlim) store_w<float>(wout + nl, re[e][0], im[e][0]);
        if (nl >= 0 && nl < nlim1) store_w<float>(wout1 + nl, re[e][1], im[e][1]);
      }
    } else {                                              // two-sided real filter of a real signal (see above)
      const float2 xn = xhat[rd.spec_off + nyq];
      const float nr = float(rd.nyq_re) * xn.x - float(rd.nyq_im) * xn.y, ni = float(rd.nyq_re) * xn.y + float(rd.nyq_im) * xn.x;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int nl = f.j + e * NT - H;
        const pairf v = rd.nterms == 2 ? 2.0f * re[e] : -2.0f * im[e];
        const float sg = ((col0 + nl) & 1) ? -1.0f : 1.0f;            // (-1)^n
        const float sg1 = (L & 1) ? -sg : sg;
        if (nl >= 0 && nl < nlim) store_w<float>(wout + nl, v[0] + sg * nr, sg * ni);
        if (nl >= 0 && nl < nlim1) store_w<float>(wout1 + nl, v[1] + sg1 * nr, sg1 * ni);
      }
    }
    return;
  }
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

// Chebyshev-economised weights (option poly_cheb, the default).  e^{i theta u} = J_0(theta) + 2 sum_{k >= 1} i^k J_k(theta) T_k(u) on
// u in [-1, 1] (Jacobi-Anger); cut at k = D the error is 2 sum_{k > D} |J_k(theta)| <= ~2 (theta/2)^(D+1) / (D+1)! -- 2^D below the
// Taylor remainder theta^(D+1) / (D+1)! of the same degree, so the same degree serves twice the theta, i.e. HALF the intervals K'
// (half the coefficient planes and half the short transforms behind them) for every row whose K' the degree decided.  Re-expanded
// in monomials, T_k(u) = sum_d t_{k,d} u^d, the coefficient of u^d is i^d r_d(theta) with
//     r_d(theta) = sum_{k = d, d + 2, ... <= D}  eps_k |t_{k,d}| J_k(theta),   |t_{k,d}| = (k/2) (k - j - 1)! / (j! d!) 2^d,  j = (k - d)/2
// (eps_0 = 1, eps_k = 2; the signs of t_{k,d} and of i^k cancel: every term is positive for theta > 0, r_d -> theta^d / d! for small
// theta), so k_poly_coef multiplies a band bin by r_d(theta_kappa) where it multiplied by theta^d / d!, and k_poly_rows evaluates the
// same Horner form.  r_d depends on (K', D, d, |kappa|) only: one table of (D + 1) x (K' + 1) reals per (K', D) pair of the scale grid,
// written once per row table; r_d(-theta) = (-1)^d r_d(theta).  grid = ((K' + 256) / 256, D + 1), evaluated in double.
template <typename T>
__global__ void __launch_bounds__(256)
k_poly_rtab(int logK, int D, T* __restrict__ out) {
  const int K = 1 << logK, q = blockIdx.x * 256 + threadIdx.x, d = blockIdx.y;
  if (q > K) return;
  const double h = 0.5 * 3.14159265358979323846 * double(q) / double(K);          // theta / 2
  const double h2 = h * h;
  double r = 0.0;
  for (int k = d; k <= D; k += 2) {
    // J_k(theta) = (theta/2)^k / k! * sum_m (-1)^m (theta/2)^(2m) / (m! (k+1)...(k+m))      (theta <= pi: 30 terms are plenty)
    double term = 1.0, sum = 1.0;
    for (int m = 1; m <= 30; ++m) {
      term *= -h2 / (double(m) * double(k + m));
      sum += term;
    }
    double pref = 1.0;                                       // (theta/2)^k / k!
    for (int i = 1; i <= k; ++i) pref *= h / double(i);
    double w = 1.0;                                          // eps_k |t_{k,d}| = k (k - j - 1)! / (j! d!) 2^d, j = (k - d) / 2  (k >= 1)
    if (k > 0) {
      const int j = (k - d) >> 1;
      w = double(k);
      if (j == 0) w /= double(d);                            // (d - 1)! / d!
      for (int i = d + 1; i <= k - j - 1; ++i) w *= double(i);   // (k - j - 1)! / d!  (empty for j <= 1)
      for (int i = 2; i <= j; ++i) w /= double(i);
      for (int i = 0; i < d; ++i) w *= 2.0;
    }
    r += w * pref * sum;
  }
  out[long(d) * (K + 1) + q] = T(r);
}

// theta^d / d! (Taylor) or r_d(theta_kappa) from the row's table
template <typename T>
__device__ __forceinline__ T poly_weight(const T* __restrict__ rtab, const RowDesc& rd, int K, int kap, int d, T tscale, T ifact) {
  if (rd.rtab_off < 0) return ipow<T>(T(kap) * tscale, d) * ifact;
  const T r = rtab[rd.rtab_off + long(d) * (K + 1) + (kap < 0 ? -kap : kap)];
  return (kap < 0 && (d & 1)) ? -r : r;
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
  const int kc = rd.k_lo + rd.kc_off, klo = rd.k_lo - kc;
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
                                               unsigned local_wg, cplx<T>* __restrict__ coef, T* lds, const T* __restrict__ rtab) {
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
    const int klo = -rd.kc_off;                               // kappa of the first band bin
    const T tscale = T(3.14159265358979323846 / double(K));
    const T ifact = T(inv_factorial(d));
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int q = f.j + e * NT;
      const int kap = klo + ((q - klo) & (K - 1));
      const T pw = poly_weight<T>(rtab, rd, K, kap, d, tscale, ifact);     // theta^d / d!, or its economised counterpart
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
            PolyClasses cls, cplx<T>* __restrict__ coef, const T* __restrict__ rtab) {
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
  case LK: if constexpr (LK >= LK_LO && LK <= LK_HI) poly_coef_body<T, LK, LOGP>(yb, rows, tw_all, pc, local, coef, lds, rtab); break;
  switch (pc.logK) {
    CWT_POLY_CASE(8) CWT_POLY_CASE(9) CWT_POLY_CASE(10) CWT_POLY_CASE(11) CWT_POLY_CASE(12) CWT_POLY_CASE(13)
    CWT_POLY_CASE(14)
    default: break;
  }
#undef CWT_POLY_CASE
}

// K' = 8192 / 16384 on 256-thread workgroups (option "coef_small"): the 512- / 1024-thread tiles above take half a CU / a whole
// CU each and do not get one while the overlap-save rows keep refilling the chip with 256-thread workgroups (k_poly_coef<14>:
// 346 us for 30 us of work, and k_poly_rows waits for it).  One decimation-in-frequency step splits a K'-point transform
// into S = K' / 4096 independent 4096-point transforms, one workgroup each:
//     X[S m + h] = IFFT_4096( u_h )[m],   u_h[q] = ( sum_{a < S} x[q + 4096 a] e^{2 pi i a h / S} ) e^{2 pi i q h / K'},   q < 4096
// -- every workgroup reads the whole band (S loads per point, from the cache) and stores every S-th coefficient of its plane; the
// S workgroups of a job are eight workgroup ids apart (same XCD, same L2: the partial lines merge there).
template <typename T, int LOGS>
__device__ __forceinline__ void poly_coef_split_body(const cplx<T>* __restrict__ yb, const RowDesc* __restrict__ rows,
                                                     const cplx<T>* __restrict__ tw_all, const PolyClass& pc,
                                                     unsigned local_wg, cplx<T>* __restrict__ coef, T* lds, const T* __restrict__ rtab) {
  constexpr int S = 1 << LOGS, LOGK = 12 + LOGS, K = 1 << LOGK, NT = 256;
  using F = ct::Fft<T, 12, 0, false>;
  F f;
  f.t = 0;
  f.j = threadIdx.x;
  const unsigned idx = local_wg % unsigned(8 * S), job = (local_wg / unsigned(8 * S)) * 8u + (idx & 7u), h = idx >> 3;
  const int rowi = int(job) / pc.ndeg, d = int(job) - rowi * pc.ndeg;
  if (rowi >= pc.nrows) return;                               // (uniform over the workgroup)
  const RowDesc rd = rows[pc.row_first + rowi];
  if (d > rd.nterms) return;
  const int klo = -rd.kc_off;
  const T tscale = T(3.14159265358979323846 / double(K));
  const T ifact = T(inv_factorial(d));
  const cplx<T>* y = yb + rd.aux_off + f.j;
  T re[16], im[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) { re[e] = T(0); im[e] = T(0); }
#pragma unroll 1
  for (int a = 0; a < S; ++a) {
    const unsigned turn = (unsigned(a) * h * unsigned(4 / S) + unsigned(d)) & 3u;     // e^{2 pi i a h / S} i^d as quarter turns
#pragma unroll
    for (int g = 0; g < 16; g += 8) {                         // eight loads in flight (the accumulators hold 64 registers already)
      cplx<T> v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = y[(g + e) * NT + (a << 12)];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int q = f.j + (g + e) * NT + (a << 12);
        const int kap = klo + ((q - klo) & (K - 1));
        const T pw = poly_weight<T>(rtab, rd, K, kap, d, tscale, ifact);     // theta^d / d!, or its economised counterpart
        T vr = v[e].x * pw, vi = v[e].y * pw;
        if (turn & 1u) { const T tmp = vr; vr = -vi; vi = tmp; }
        if (turn & 2u) { vr = -vr; vi = -vi; }
        re[g + e] += vr; im[g + e] += vi;
      }
      if (g == 0) { keep_here(re[0]); keep_here(im[0]); }      // (keeps the second group's loads behind the first group's sums)
    }
  }
  if (h) {
    const cplx<T>* tk = tw_all + (K - 2);
#pragma unroll
    for (int g = 0; g < 16; g += 4) {
      cplx<T> w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = tk[unsigned(f.j + (g + e) * NT) * h];     // q h < 3 * 4096 <= K' - 1: no wrap
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const T x = re[g + e], yv = im[g + e];
        re[g + e] = x * w[e].x - yv * w[e].y;
        im[g + e] = x * w[e].y + yv * w[e].x;
      }
    }
  }
  f.run(re, im, lds, tw_all + (4096 - 2));
  cplx<T>* out = coef + rd.tab_off + (long(d) << LOGK) + (long(f.j) << LOGS) + h;   // plane d of the row, element S m + h
#pragma unroll
  for (int e = 0; e < 16; ++e) out[long(e * NT) << LOGS] = mk<T>(re[e], im[e]);
}

// every class of a chunk in ONE launch of 256-thread workgroups: PolyClass::wg_first1 = the class's first workgroup in it
template <typename T>
__global__ void __launch_bounds__(256, 4)
k_poly_coef_all(const cplx<T>* __restrict__ yb, const RowDesc* __restrict__ rows, const cplx<T>* __restrict__ tw_all,
                PolyClasses cls, cplx<T>* __restrict__ coef, const T* __restrict__ rtab) {
  HIP_DYNAMIC_SHARED(double2, lds_raw)
  T* lds = reinterpret_cast<T*>(lds_raw);
  PolyClass pc = cls.c[0];
#pragma unroll
  for (int i = 1; i < POLY_MAX_CLASSES; ++i)
    if (i < cls.n && int(blockIdx.x) >= cls.c[i].wg_first1) pc = cls.c[i];
  const unsigned local = blockIdx.x - unsigned(pc.wg_first1);
#define CWT_POLY_CASE(LK) case LK: poly_coef_body<T, LK, 12>(yb, rows, tw_all, pc, local, coef, lds, rtab); break;
  switch (pc.logK) {
    CWT_POLY_CASE(8) CWT_POLY_CASE(9) CWT_POLY_CASE(10) CWT_POLY_CASE(11) CWT_POLY_CASE(12)
    case 13: poly_coef_split_body<T, 1>(yb, rows, tw_all, pc, local, coef, lds, rtab); break;
    case 14: poly_coef_split_body<T, 2>(yb, rows, tw_all, pc, local, coef, lds, rtab); break;
    default: break;
  }
#undef CWT_POLY_CASE
}

// Stage 2.  One workgroup = 256 lanes x POLY_PASSES passes; a lane stores 16 bytes per pass (one complex128 or two adjacent
// complex64 outputs).  sc[i][d] = a_d[m0 + i] for the intervals m0 ... the workgroup touches.
template <typename T, int D>
__device__ __forceinline__ void poly_rows_body(const RowDesc& rd, const cplx<T>* __restrict__ coef, const TwN<T>& twn,
                                               int logN, cplx<T>* __restrict__ W, long ldw, long ncols, cplx<T>* sc) {
  constexpr int PT = sizeof(T) == 8 ? 1 : 2, SPAN = 256 * PT, I = POLY_PASSES, WSPAN = 64 * PT;
  static_assert((I & (I - 1)) == 0, "POLY_PASSES: a power of two (a wavefront's span must divide the interval length)");
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
  // A wavefront covers I x WSPAN CONSECUTIVE outputs (its I passes side by side, not SPAN apart): for R >= I x WSPAN they lie in
  // one interval, and a coefficient read from LDS serves all I x PT outputs of a lane.  The reads -- (D + 1) x 16 bytes per lane,
  // all lanes at one address -- are what the kernel is bound by beside its stores: per class [measured, round 6 session q]
  // D = 4 rows store 6.9 TB/s, D = 8 rows of the same K' 5.6.
  const int kc = rd.k_lo + rd.kc_off;
  const unsigned nl = n0 + (threadIdx.x >> 6) * unsigned(WSPAN * I) + (threadIdx.x & 63u) * unsigned(PT);
  cplx<T> w = twn((unsigned(kc) * nl) & nmask);
  const cplx<T> step = twn((unsigned(kc) * unsigned(WSPAN)) & nmask);   // uniform: one pass further
  cplx<T> adj = mk<T>(T(1), T(0));
  if constexpr (PT == 2) adj = twn(unsigned(kc) & nmask);                // e^{2 pi i k_c / N}: the lane's second output
  const T scale = T(2) / T(1u << logR);
  cplx<T>* wrow = W + long(rd.out_row) * ldw;
  __syncthreads();
  T pr[I][PT], pi[I][PT], u[I][PT];
#pragma unroll
  for (int p = 0; p < I; ++p)
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const unsigned ni = nl + unsigned(p * WSPAN + i);
      u[p][i] = T(int(ni & ((1u << logR) - 1u))) * scale - T(1);
    }
  if ((1 << logR) >= WSPAN * I) {                                        // (uniform) one interval for the whole lane
    const cplx<T>* c = sc + ((nl >> logR) - m0) * unsigned(D + 1);
    const cplx<T> top = c[D];
#pragma unroll
    for (int p = 0; p < I; ++p)
#pragma unroll
      for (int i = 0; i < PT; ++i) { pr[p][i] = top.x; pi[p][i] = top.y; }
#pragma unroll
    for (int d = D - 1; d >= 0; --d) {
      const cplx<T> cd = c[d];
#pragma unroll
      for (int p = 0; p < I; ++p)
#pragma unroll
        for (int i = 0; i < PT; ++i) { pr[p][i] = fma(pr[p][i], u[p][i], cd.x); pi[p][i] = fma(pi[p][i], u[p][i], cd.y); }
    }
  } else {                                                               // an interval per pass (R >= WSPAN: the PT outputs of a pass share it)
#pragma unroll
    for (int p = 0; p < I; ++p) {
      const cplx<T>* c = sc + (((nl + unsigned(p * WSPAN)) >> logR) - m0) * unsigned(D + 1);
      const cplx<T> top = c[D];
#pragma unroll
      for (int i = 0; i < PT; ++i) { pr[p][i] = top.x; pi[p][i] = top.y; }
#pragma unroll
      for (int d = D - 1; d >= 0; --d) {
        const cplx<T> cd = c[d];
#pragma unroll
        for (int i = 0; i < PT; ++i) { pr[p][i] = fma(pr[p][i], u[p][i], cd.x); pi[p][i] = fma(pi[p][i], u[p][i], cd.y); }
      }
    }
  }
#pragma unroll
  for (int p = 0; p < I; ++p) {
    const unsigned n = nl + unsigned(p * WSPAN);
    cplx<T> o[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const cplx<T> wi = i == 0 ? w : cmul<T>(w, adj);
      o[i] = mk<T>(pr[p][i] * wi.x - pi[p][i] * wi.y, pr[p][i] * wi.y + pi[p][i] * wi.x);
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

// Rows of degree <= CWT_POLY_SCALAR_D64 / _D32 (complex128 / complex64) take their coefficient sets through the scalar data path
// (poly_rows_body_s: no LDS, no barrier, 32 vector registers), the others stage them in LDS (poly_rows_body).  [measured, round 6
// sessions y - ab, per (K', degree) class in complex128: D = 4 rows 2.59 -> 2.39 us per row (7.0 TB/s), D = 6 2.54 -> 2.71, D = 8 2.88 ->
// 3.31 -- nine scalar round trips per wavefront cost more than one LDS staging per workgroup; config 2 with D <= 4: -0.7 ... -0.9 %.
// complex64: not used -- a pass of a wavefront is 128 outputs there, two intervals at R = 64 (two sets and a select per lane: 71
// registers), and the step is +0.9 % (five interleaved repeats, Paul and DOG) although a first, all-scalar build looked better]
#ifndef CWT_POLY_SCALAR_D64
#define CWT_POLY_SCALAR_D64 4
#endif
#ifndef CWT_POLY_SCALAR_D32
#define CWT_POLY_SCALAR_D32 0
#endif
__device__ __forceinline__ unsigned wave_uniform(unsigned v) {
#if defined(__AMDGCN__)
  return __builtin_amdgcn_readfirstlane(v);
#else
  return v;
#endif
}
// As poly_rows_body, but a wavefront reads the coefficient set(s) of its own interval(s) itself: the addresses are uniform over the
// wavefront (R >= 64 outputs of one pass lie in one interval), so the loads are scalar loads and the Horner FMAs take the coefficient from
// scalar registers -- no staging in LDS, no workgroup barrier, no LDS reads.
template <typename T, int D>
__device__ __forceinline__ void poly_rows_body_s(const RowDesc& rd, const cplx<T>* __restrict__ coef, const TwN<T>& twn,
                                                 int logN, cplx<T>* __restrict__ W, long ldw, long ncols) {
  constexpr int PT = sizeof(T) == 8 ? 1 : 2, SPAN = 256 * PT, I = POLY_PASSES, WSPAN = 64 * PT;
  const int logR = logN - rd.logK;
  const unsigned nmask = unsigned((1 << logN) - 1);
  const unsigned n0 = blockIdx.x * unsigned(SPAN * I);
  const unsigned nw = n0 + wave_uniform(threadIdx.x >> 6) * unsigned(WSPAN * I);      // first output of this wavefront
  const unsigned nl = nw + (threadIdx.x & 63u) * unsigned(PT);
  const int kc = rd.k_lo + rd.kc_off;
  cplx<T> w = twn((unsigned(kc) * nl) & nmask);
  const cplx<T> step = twn((unsigned(kc) * unsigned(WSPAN)) & nmask);
  cplx<T> adj = mk<T>(T(1), T(0));
  if constexpr (PT == 2) adj = twn(unsigned(kc) & nmask);
  const T scale = T(2) / T(1u << logR);
  cplx<T>* wrow = W + long(rd.out_row) * ldw;
  const cplx<T>* a = coef + rd.tab_off;
  T pr[I][PT], pi[I][PT], u[I][PT];
#pragma unroll
  for (int p = 0; p < I; ++p)
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const unsigned ni = nl + unsigned(p * WSPAN + i);
      u[p][i] = T(int(ni & ((1u << logR) - 1u))) * scale - T(1);
    }
  if ((1 << logR) >= WSPAN * I) {
    const cplx<T>* c = a + ((nw & nmask) >> logR);
    const cplx<T> top = c[long(D) << rd.logK];
#pragma unroll
    for (int p = 0; p < I; ++p)
#pragma unroll
      for (int i = 0; i < PT; ++i) { pr[p][i] = top.x; pi[p][i] = top.y; }
#pragma unroll
    for (int d = D - 1; d >= 0; --d) {
      const cplx<T> cd = c[long(d) << rd.logK];
#pragma unroll
      for (int p = 0; p < I; ++p)
#pragma unroll
        for (int i = 0; i < PT; ++i) { pr[p][i] = fma(pr[p][i], u[p][i], cd.x); pi[p][i] = fma(pi[p][i], u[p][i], cd.y); }
    }
  } else {                                                               // an interval per pass (R >= WSPAN: see k_poly_rows)
#pragma unroll
    for (int p = 0; p < I; ++p) {
      const cplx<T>* c = a + (((nw + unsigned(p * WSPAN)) & nmask) >> logR);
      const cplx<T> top = c[long(D) << rd.logK];
#pragma unroll
      for (int i = 0; i < PT; ++i) { pr[p][i] = top.x; pi[p][i] = top.y; }
#pragma unroll
      for (int d = D - 1; d >= 0; --d) {
        const cplx<T> cd = c[long(d) << rd.logK];
#pragma unroll
        for (int i = 0; i < PT; ++i) { pr[p][i] = fma(pr[p][i], u[p][i], cd.x); pi[p][i] = fma(pi[p][i], u[p][i], cd.y); }
      }
    }
  }
#pragma unroll
  for (int p = 0; p < I; ++p) {
    const unsigned n = nl + unsigned(p * WSPAN);
    cplx<T> o[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const cplx<T> wi = i == 0 ? w : cmul<T>(w, adj);
      o[i] = mk<T>(pr[p][i] * wi.x - pi[p][i] * wi.y, pr[p][i] * wi.y + pi[p][i] * wi.x);
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
  constexpr int SMAX = sizeof(T) == 8 ? CWT_POLY_SCALAR_D64 : CWT_POLY_SCALAR_D32;
  static_assert(sizeof(T) == 8 || SMAX == 0, "poly_rows_body_s: one interval per pass of a wavefront (R >= 128 in complex64 is not guaranteed)");
#define CWT_POLYR_CASE(DD)                                                                      \
  case DD:                                                                                      \
    if constexpr (DD <= SMAX) poly_rows_body_s<T, DD>(rd, coef, twn, logN, W, ldw, ncols);      \
    else poly_rows_body<T, DD>(rd, coef, twn, logN, W, ldw, ncols, sc);                         \
    break;
  switch (rd.nterms) {
    CWT_POLYR_CASE(2) CWT_POLYR_CASE(4) CWT_POLYR_CASE(6) CWT_POLYR_CASE(8) CWT_POLYR_CASE(10) CWT_POLYR_CASE(12)
    CWT_POLYR_CASE(14) CWT_POLYR_CASE(16) CWT_POLYR_CASE(18) CWT_POLYR_CASE(20) CWT_POLYR_CASE(22) CWT_POLYR_CASE(24)
    default: break;
  }
#undef CWT_POLYR_CASE
}

}  // namespace cwt
