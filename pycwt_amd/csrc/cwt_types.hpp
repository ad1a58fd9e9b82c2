// cwt_types.hpp -- the plain-C++ records shared by the host side (row classification, launches) and the kernels: one row
// of the transform, the mother, the class tables of the overlap-save / polynomial rows.  No HIP in here: plan_host.cpp
// compiles with a plain C++ compiler.
#pragma once

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
  int kc_off;      // polynomial rows: carrier bin k_c = k_lo + kc_off (the band's centre for a symmetric filter; nearer the filter's
                   // peak for a lopsided one -- Paul, DOG -- where that lowers the degree: build_row_table)
  long rtab_off;   // polynomial rows: element offset of the row's (K', D) table of monomial weights r_d(theta_kappa) of the Chebyshev-
                   // economised expansion (k_poly_rtab); -1 = Taylor weights theta^d / d!
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

// ---- overlap-save rows (kernels: cwt_kernels_rows.hpp) ----
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

// ---- rows clipped at Nyquist on the band-passed complex signal ----
struct AolsGeom {
  int nrows;       // rows of the class
  int nblocks;     // output blocks of L = P - 2 halo columns
  int halo;        // H (multiple of 64)
  int ksp;         // first unwrapped bin of the block grid: a block bin q stands for kappa = ksp + ((q - ksp) mod P)
  double f_s;      // low edge of the mask in cycles per sample (<= 1/N)
  double f1_lo;    // the window is 1 on [f1_lo, 1/2]
  double z;        // erfc argument at the ends of a taper: u = erfc(z)/2 there
  // Rows with RowDesc::aux_off == 1 (Paul, filter NOT clipped at Nyquist): the profile f^m e^-f is continued analytically
  // THROUGH f = 0 to negative arguments (block bins above P/2 stand for the negative bins) and cut there by the row's own
  // taper u(f) = erfc((-f - zc_c) / zc_w) / 2 in the profile's argument f = s w -- the kink at f = 0 that gives the Paul
  // wavelet its 1/t^(m+1) tail is gone, the kernel is as compact as a Gaussian's (halo ~16 s instead of ~250 s at 1e-10).
  double zc_c, zc_w;
};

// ---- band-limited rows in polynomial form ----
// complex64: the overlap-save kernels on 4096-point tiles (k_ols_ct<float, 12>, k_aols_rows<float, .>) transform two blocks
// of a row per workgroup in packed fp32 arithmetic (cwt_kernels_rows.hpp, ols_band_body2); the host sizes their grids in block
// PAIRS.  0 = one block per workgroup everywhere.  Not the 8192-point tiles of k_ols_ct: 512 threads x 128+ registers leave one
// workgroup per CU (measured 3.3 -> 4.0-4.2 us per row).
#ifndef CWT_PAIR_F32
#define CWT_PAIR_F32 1
#endif
constexpr bool ols_pairs(int real_bytes, int logp) { return CWT_PAIR_F32 != 0 && real_bytes == 4 && logp == 12; }
constexpr bool aols_pairs(int real_bytes) { return CWT_PAIR_F32 != 0 && real_bytes == 4; }
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
  int wg_first1;   // ... and in the single launch of k_poly_coef_all (256-thread workgroups for every K'; a multiple of 8)
};
struct PolyClasses {
  PolyClass c[POLY_MAX_CLASSES];
  int n;
};

// ---- cwt_spectrum_range ----
constexpr int SPECTRUM_OCTAVES = 32;
constexpr int SPECTRUM_WINDOWS = 4 * SPECTRUM_OCTAVES;
constexpr int SPECTRUM_SLOTS = 2 + SPECTRUM_WINDOWS;

}  // namespace cwt
