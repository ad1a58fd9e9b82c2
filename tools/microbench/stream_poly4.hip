// stream_poly4.hip -- round 6: where is the ceiling of k_poly_rows, and which structure gets closest?
//
// stream_poly3.hip priced the LDS-staged form with ZERO twiddle tables (its outputs were all zero: constant data clocks
// higher and costs the memory less power, MI355X_MICROARCH.md "DVFS give-back").  Here every table is real and the outputs
// are random-looking, as in the product.  Variants (same bytes, 64 rows x 2^20 complex128):
//   fill     store only, value = a hash of the index (the ceiling for random data), 1 output per lane, 256-thread workgroups
//   base     the product's structure: coefficient sets of the workgroup's intervals -> LDS, I passes of 256 lanes x 16 B
//   wide     a lane stores two ADJACENT outputs per pass (32 B per lane, 8 KB per pass)
//   run R    base with the workgroup -> chunk map permuted so that one XCD (workgroup id mod 8) writes R consecutive
//            chunks (R = 1 is the identity)
//   plain    base with ordinary stores instead of non-temporal ones
//   persist  G workgroups per CU loop over chunks; the coefficient sets of chunk c + 1 are fetched into the other half of the
//            LDS before chunk c is evaluated and stored
//   wg512 / wg128  base with 512- / 128-thread workgroups (same bytes per workgroup: I passes of 4 KB)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double v2 __attribute__((vector_size(16)));

__device__ __forceinline__ double2 twn(const double2* hi, const double2* lo, unsigned t) {
  const double2 a = hi[t >> 10], b = lo[t & 1023u];
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

template <bool NT>
__device__ __forceinline__ void st16(double2* p, double re, double im) {
  v2 o = {re, im};
  if constexpr (NT) __builtin_nontemporal_store(o, reinterpret_cast<v2*>(p));
  else *reinterpret_cast<v2*>(p) = o;
}

// workgroup id -> chunk id: XCD x = b & 7 writes runs of RUN consecutive chunks
template <int RUN>
__device__ __forceinline__ unsigned chunk_of(unsigned b) {
  if constexpr (RUN == 1) return b;
  const unsigned x = b & 7u, i = b >> 3, t = i % RUN, g = i / RUN;
  return (g * 8u + x) * RUN + t;
}

__global__ void __launch_bounds__(256) k_fill(double2* __restrict__ W) {
  const size_t n = size_t(blockIdx.y) * (size_t(1) << 20) + size_t(blockIdx.x) * 256 + threadIdx.x;
  unsigned h = unsigned(n) * 2654435761u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  const double a = double(int(h)) * 4.656612873077393e-10, b = double(int(h * 3266489917u)) * 4.656612873077393e-10;
  st16<true>(W + n, a, b);
}

template <int TH, bool NT, bool ZERO>
__global__ void __launch_bounds__(TH) k_fill2(double2* __restrict__ W) {
  const size_t n = size_t(blockIdx.y) * (size_t(1) << 20) + size_t(blockIdx.x) * TH + threadIdx.x;
  unsigned h = unsigned(n) * 2654435761u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  const double a = ZERO ? 0.0 : double(int(h)) * 4.656612873077393e-10, b = ZERO ? 0.0 : double(int(h * 3266489917u)) * 4.656612873077393e-10;
  st16<NT>(W + n, a, b);
}
// re-writes the coefficients (same values) so that they sit in the caches as k_poly_coef leaves the product's
__global__ void k_touch(double2* p, size_t n) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) { double2 v = p[i]; v.x += 0.0; p[i] = v; }
}

// D = degree, I = passes, TH = threads, WIDE = two adjacent outputs per lane, RUN = chunk permutation, NT = non-temporal
template <int D, int I, int TH, bool WIDE, int RUN, bool NT>
__global__ void __launch_bounds__(TH) k_rows(double2* __restrict__ W, const double2* __restrict__ coef, int logK,
                                             const double2* __restrict__ hi, const double2* __restrict__ lo, const int* kcs) {
  extern __shared__ double2 sc[];                       // [interval][d]
  constexpr unsigned PT = WIDE ? 2 : 1, SPAN = TH * PT;
  const int logN = 20, logR = logN - logK;
  const unsigned row = blockIdx.y;
  const unsigned n0 = chunk_of<RUN>(blockIdx.x) * (SPAN * I);
  const unsigned m0 = n0 >> logR;
  const unsigned nint = (((n0 + SPAN * I - 1u) >> logR) - m0 + 1u) * (D + 1);
  const double2* a = coef + (((size_t(row) << logK) + m0) * (D + 1));
  for (unsigned t = threadIdx.x; t < nint; t += TH) sc[t] = a[t];
  const int kc = kcs[row];
  const unsigned nmask = (1u << logN) - 1u;
  double2 w = twn(hi, lo, (unsigned(kc) * (n0 + threadIdx.x * PT)) & nmask);
  const double2 st = twn(hi, lo, (unsigned(kc) * SPAN) & nmask);
  double2 adj = make_double2(1.0, 0.0);
  if constexpr (WIDE) adj = twn(hi, lo, unsigned(kc) & nmask);
  const double scale = 2.0 / double(1u << logR);
  double2* out = W + (size_t(row) << logN) + n0 + threadIdx.x * PT;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < I; ++i) {
#pragma unroll
    for (unsigned k = 0; k < PT; ++k) {
      const unsigned n = n0 + i * SPAN + threadIdx.x * PT + k;
      const double2* c = sc + ((n >> logR) - m0) * (D + 1);
      const double u = double(int(n & ((1u << logR) - 1u))) * scale - 1.0;
      double pr = c[D].x, pi = c[D].y;
#pragma unroll
      for (int d = D - 1; d >= 0; --d) { const double2 cd = c[d]; pr = fma(pr, u, cd.x); pi = fma(pi, u, cd.y); }
      const double2 wk = k == 0 ? w : cmul(w, adj);
      st16<NT>(out + i * SPAN + k, pr * wk.x - pi * wk.y, pr * wk.y + pi * wk.x);
    }
    if (i + 1 < I) w = cmul(w, st);
  }
}

// persistent: gridDim.x workgroups walk the chunks of ALL rows (chunk = 256 * I outputs) in steps of gridDim.x
template <int D, int I>
__global__ void __launch_bounds__(256) k_persist(double2* __restrict__ W, const double2* __restrict__ coef, int logK,
                                                 const double2* __restrict__ hi, const double2* __restrict__ lo, const int* kcs,
                                                 unsigned nchunks_row, unsigned rows) {
  extern __shared__ double2 sc[];                       // two halves of `half` elements
  const int logN = 20, logR = logN - logK;
  const unsigned half = ((256u * I) >> logR) * (D + 1) + 2 * (D + 1);
  const unsigned total = nchunks_row * rows, nmask = (1u << logN) - 1u;
  const double scale = 2.0 / double(1u << logR);
  auto fetch = [&](unsigned c, double2 (&r)[2], unsigned& cnt) {      // <= 2 values per lane at these sizes
    const unsigned row = c / nchunks_row, n0 = (c - row * nchunks_row) * (256u * I), m0 = n0 >> logR;
    cnt = (((n0 + 256u * I - 1u) >> logR) - m0 + 1u) * (D + 1);
    const double2* a = coef + (((size_t(row) << logK) + m0) * (D + 1));
    r[0] = threadIdx.x < cnt ? a[threadIdx.x] : make_double2(0, 0);
    r[1] = threadIdx.x + 256u < cnt ? a[threadIdx.x + 256u] : make_double2(0, 0);
  };
  unsigned c = blockIdx.x, buf = 0, cnt;
  double2 r[2];
  if (c < total) { fetch(c, r, cnt); sc[threadIdx.x] = r[0]; if (threadIdx.x + 256u < half) sc[threadIdx.x + 256u] = r[1]; }
  __syncthreads();
  for (; c < total; c += gridDim.x) {
    const unsigned nxt = c + gridDim.x;
    if (nxt < total) fetch(nxt, r, cnt);                // in flight while this chunk is evaluated
    const unsigned row = c / nchunks_row, n0 = (c - row * nchunks_row) * (256u * I), m0 = n0 >> logR;
    const int kc = kcs[row];
    double2 w = twn(hi, lo, (unsigned(kc) * (n0 + threadIdx.x)) & nmask);
    const double2 st = twn(hi, lo, (unsigned(kc) * 256u) & nmask);
    double2* out = W + (size_t(row) << logN) + n0 + threadIdx.x;
    const double2* base = sc + buf * half;
#pragma unroll
    for (int i = 0; i < I; ++i) {
      const unsigned n = n0 + i * 256u + threadIdx.x;
      const double2* cc = base + ((n >> logR) - m0) * (D + 1);
      const double u = double(int(n & ((1u << logR) - 1u))) * scale - 1.0;
      double pr = cc[D].x, pi = cc[D].y;
#pragma unroll
      for (int d = D - 1; d >= 0; --d) { const double2 cd = cc[d]; pr = fma(pr, u, cd.x); pi = fma(pi, u, cd.y); }
      st16<true>(out + i * 256, pr * w.x - pi * w.y, pr * w.y + pi * w.x);
      if (i + 1 < I) w = cmul(w, st);
    }
    if (nxt < total) {
      double2* dst = sc + (buf ^ 1u) * half;
      dst[threadIdx.x] = r[0];
      if (threadIdx.x + 256u < half) dst[threadIdx.x + 256u] = r[1];
    }
    buf ^= 1u;
    __syncthreads();
  }
}

static double2 *W, *coef, *hi, *lo; static int* kcs;
static hipEvent_t e0, e1;
static const int rows = 64;
static const size_t N = size_t(1) << 20;

static int g_burst = 1;      // launches between the two events (1: a host round trip after every launch, as stream_poly3.hip times)
template <class F> float timeit(F&& launch, size_t ncoef) {
  float tot = 0; const int reps = 8;
  for (int i = 0; i < reps + 2; ++i) {
    CK(hipEventRecord(e0));
    for (int b = 0; b < g_burst; ++b) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (i >= 2) tot += ms / g_burst;
  }
  (void)ncoef;
  return tot / reps;
}
static void report(const char* name, int D, int logK, float ms) {
  printf("%-34s degree %2d K' = 2^%-2d: %7.3f ms  %7.1f GB/s  %5.2f us/row\n", name, D, logK, ms, rows * N * 16.0 / ms / 1e6, ms * 1e3 / rows);
  fflush(stdout);
}

template <int D, int I, int TH, bool WIDE, int RUN, bool NT>
void run(const char* name, int logK) {
  const unsigned span = TH * (WIDE ? 2 : 1) * I;
  const dim3 grid(unsigned(N / span), rows);
  const size_t lds = ((size_t(span) >> (20 - logK)) + 2) * (D + 1) * 16;
  report(name, D, logK, timeit([&] { hipLaunchKernelGGL((k_rows<D, I, TH, WIDE, RUN, NT>), grid, dim3(TH), lds, 0, W, coef, logK, hi, lo, kcs); }, 0));
}
template <int D, int I>
void run_persist(const char* name, int logK, int per_cu) {
  const unsigned nchunks_row = unsigned(N / (256 * I));
  const size_t half = ((size_t(256 * I) >> (20 - logK)) * (D + 1) + 2 * (D + 1));
  if (half > 512) { printf("%-34s skipped (more than two values per lane)\n", name); return; }
  char buf[96]; snprintf(buf, sizeof buf, "%s, %d WG/CU", name, per_cu);
  report(buf, D, logK, timeit([&] { hipLaunchKernelGGL((k_persist<D, I>), dim3(256 * per_cu), dim3(256), 2 * half * 16, 0, W, coef, logK, hi, lo, kcs, nchunks_row, unsigned(rows)); }, 0));
}

static void fill_tables(bool zero_tw, bool zero_coef, bool smooth_coef) {
  std::vector<double2> h(1024), l(1024);
  for (int i = 0; i < 1024; ++i) {
    const double ah = 6.283185307179586 * double(i) / 1024.0, al = 6.283185307179586 * double(i) / 1048576.0;
    h[i] = zero_tw ? make_double2(0, 0) : make_double2(cos(ah), sin(ah));
    l[i] = zero_tw ? make_double2(0, 0) : make_double2(cos(al), sin(al));
  }
  CK(hipMemcpy(hi, h.data(), 1024 * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(lo, l.data(), 1024 * 16, hipMemcpyHostToDevice));
  std::vector<double2> c(size_t(rows) * 13 * 16384);
  unsigned s = 12345u;
  size_t i = 0;
  for (auto& v : c) {
    s = s * 1664525u + 1013904223u; v.x = double(int(s)) * 4.656612873077393e-10; s = s * 1664525u + 1013904223u; v.y = double(int(s)) * 4.656612873077393e-10;
    if (zero_coef) v = make_double2(0, 0);
    if (smooth_coef) v = make_double2(1e-3 * double(i & 1023), 1.0);
    ++i;
  }
  CK(hipMemcpy(coef, c.data(), c.size() * 16, hipMemcpyHostToDevice));
}

int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]) == "warm") {      // the fair comparison: coefficients freshly written before every launch
    CK(hipMalloc(&W, rows * N * 16)); CK(hipMalloc(&coef, size_t(rows) * 13 * 16384 * 16)); CK(hipMalloc(&hi, 1024 * 16)); CK(hipMalloc(&lo, 1024 * 16));
    CK(hipMalloc(&kcs, rows * 4));
    int k[64]; for (int i = 0; i < 64; ++i) k[i] = 1000 + 37 * i;
    CK(hipMemcpy(kcs, k, rows * 4, hipMemcpyHostToDevice));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    fill_tables(false, false, false);
    auto warm = [&](int logK, int D) { const size_t n = (size_t(rows) * (D + 1)) << logK; hipLaunchKernelGGL(k_touch, dim3(unsigned((n + 255) / 256)), dim3(256), 0, 0, coef, n); };
    auto timed = [&](const char* name, int D, int logK, auto launch) {
      float tot = 0; const int reps = 8;
      for (int i = 0; i < reps + 2; ++i) {
        warm(logK, D);
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 2) tot += ms;
      }
      report(name, D, logK, tot / reps);
    };
    for (int rep = 0; rep < 2; ++rep) {
      timed("fill: hash values, nt, 256 threads", 0, 0, [&] { hipLaunchKernelGGL((k_fill2<256, true, false>), dim3(unsigned(N / 256), rows), dim3(256), 0, 0, W); });
      timed("fill: ZERO values, nt, 256 threads", 0, 0, [&] { hipLaunchKernelGGL((k_fill2<256, true, true>), dim3(unsigned(N / 256), rows), dim3(256), 0, 0, W); });
      timed("fill: hash values, plain stores", 0, 0, [&] { hipLaunchKernelGGL((k_fill2<256, false, false>), dim3(unsigned(N / 256), rows), dim3(256), 0, 0, W); });
      timed("fill: hash values, nt, 512 threads", 0, 0, [&] { hipLaunchKernelGGL((k_fill2<512, true, false>), dim3(unsigned(N / 512), rows), dim3(512), 0, 0, W); });
      timed("fill: hash values, nt, 128 threads", 0, 0, [&] { hipLaunchKernelGGL((k_fill2<128, true, false>), dim3(unsigned(N / 128), rows), dim3(128), 0, 0, W); });
      timed("fill: hash values, nt, 64 threads", 0, 0, [&] { hipLaunchKernelGGL((k_fill2<64, true, false>), dim3(unsigned(N / 64), rows), dim3(64), 0, 0, W); });
      for (int logK : {8, 11, 14}) {
        const dim3 g2(unsigned(N / 512), rows), g3(unsigned(N / 768) + 1, rows);
        auto lds = [&](unsigned span, int D) { return ((size_t(span) >> (20 - logK)) + 2) * (D + 1) * 16; };
        timed("warm base: 2 passes", 8, logK, [&] { hipLaunchKernelGGL((k_rows<8, 2, 256, false, 1, true>), g2, dim3(256), lds(512, 8), 0, W, coef, logK, hi, lo, kcs); });
        timed("warm run 8", 8, logK, [&] { hipLaunchKernelGGL((k_rows<8, 2, 256, false, 8, true>), g2, dim3(256), lds(512, 8), 0, W, coef, logK, hi, lo, kcs); });
        timed("warm plain stores", 8, logK, [&] { hipLaunchKernelGGL((k_rows<8, 2, 256, false, 1, false>), g2, dim3(256), lds(512, 8), 0, W, coef, logK, hi, lo, kcs); });
        timed("warm wg128: 4 passes", 8, logK, [&] { hipLaunchKernelGGL((k_rows<8, 4, 128, false, 1, true>), g2, dim3(128), lds(512, 8), 0, W, coef, logK, hi, lo, kcs); });
        timed("warm wg512: 1 pass", 8, logK, [&] { hipLaunchKernelGGL((k_rows<8, 1, 512, false, 1, true>), g2, dim3(512), lds(512, 8), 0, W, coef, logK, hi, lo, kcs); });
        const unsigned nchunks_row = unsigned(N / 512);
        const size_t half = ((size_t(512) >> (20 - logK)) * 9 + 18);
        if (half <= 512) {
          timed("warm persist 2 passes, 7 WG/CU", 8, logK, [&] { hipLaunchKernelGGL((k_persist<8, 2>), dim3(256 * 7), dim3(256), 2 * half * 16, 0, W, coef, logK, hi, lo, kcs, nchunks_row, unsigned(rows)); });
          timed("warm persist 2 passes, 4 WG/CU", 8, logK, [&] { hipLaunchKernelGGL((k_persist<8, 2>), dim3(256 * 4), dim3(256), 2 * half * 16, 0, W, coef, logK, hi, lo, kcs, nchunks_row, unsigned(rows)); });
        }
      }
    }
    return 0;
  }
  if (argc > 1 && std::string(argv[1]) == "data") {      // what does the DATA cost?  base kernel, K' = 2^11, degree 8
    CK(hipMalloc(&W, rows * N * 16)); CK(hipMalloc(&coef, size_t(rows) * 13 * 16384 * 16)); CK(hipMalloc(&hi, 1024 * 16)); CK(hipMalloc(&lo, 1024 * 16));
    CK(hipMalloc(&kcs, rows * 4));
    int k[64]; for (int i = 0; i < 64; ++i) k[i] = 1000 + 37 * i;
    CK(hipMemcpy(kcs, k, rows * 4, hipMemcpyHostToDevice));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int burst : {1, 10}) {
      g_burst = burst;
      printf("-- %d launch(es) per timed region\n", burst);
      report("fill (hash values, store only)", 0, 0, timeit([&] { hipLaunchKernelGGL(k_fill, dim3(unsigned(N / 256), rows), dim3(256), 0, 0, W); }, 0));
      for (int mode = 0; mode < 5; ++mode) {
        fill_tables(mode == 1 || mode == 3, mode == 2 || mode == 3, mode == 4);
        const char* names[5] = {"random coefficients, real carrier", "random coefficients, ZERO carrier", "ZERO coefficients, real carrier", "all zero", "smooth coefficients, real carrier"};
        for (int logK : {8, 11, 14}) {
          run<8, 2, 256, false, 1, true>(names[mode], logK);
        }
        run_persist<8, 2>(names[mode], 11, 7);
      }
    }
    return 0;
  }
  CK(hipMalloc(&W, rows * N * 16)); CK(hipMalloc(&coef, size_t(rows) * 13 * 16384 * 16)); CK(hipMalloc(&hi, 1024 * 16)); CK(hipMalloc(&lo, 1024 * 16));
  CK(hipMalloc(&kcs, rows * 4));
  {
    std::vector<double2> h(1024), l(1024);
    for (int i = 0; i < 1024; ++i) {
      const double ah = 6.283185307179586 * double(i) / 1024.0, al = 6.283185307179586 * double(i) / 1048576.0;
      h[i] = make_double2(cos(ah), sin(ah)); l[i] = make_double2(cos(al), sin(al));
    }
    CK(hipMemcpy(hi, h.data(), 1024 * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(lo, l.data(), 1024 * 16, hipMemcpyHostToDevice));
    std::vector<double2> c(size_t(rows) * 13 * 16384);
    unsigned s = 12345u;
    for (auto& v : c) { s = s * 1664525u + 1013904223u; v.x = double(int(s)) * 4.656612873077393e-10; s = s * 1664525u + 1013904223u; v.y = double(int(s)) * 4.656612873077393e-10; }
    CK(hipMemcpy(coef, c.data(), c.size() * 16, hipMemcpyHostToDevice));
    int k[64]; for (int i = 0; i < 64; ++i) k[i] = 1000 + 37 * i;
    CK(hipMemcpy(kcs, k, rows * 4, hipMemcpyHostToDevice));
  }
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    report("fill (hash values, store only)", 0, 0, timeit([&] { hipLaunchKernelGGL(k_fill, dim3(unsigned(N / 256), rows), dim3(256), 0, 0, W); }, 0));
    for (int logK : {8, 11, 14}) {
      run<8, 2, 256, false, 1, true>("base: 2 passes", logK);
      run<8, 1, 256, true, 1, true>("wide: 1 pass x 32 B", logK);
      run<8, 2, 256, true, 1, true>("wide: 2 passes x 32 B", logK);
      run<8, 3, 256, false, 1, true>("base: 3 passes", logK);
      run<8, 2, 256, false, 2, true>("run 2", logK);
      run<8, 2, 256, false, 4, true>("run 4", logK);
      run<8, 2, 256, false, 8, true>("run 8", logK);
      run<8, 2, 256, false, 32, true>("run 32", logK);
      run<8, 2, 256, false, 1, false>("plain stores", logK);
      run<8, 2, 512, false, 1, true>("wg512: 2 passes", logK);
      run<8, 1, 512, false, 1, true>("wg512: 1 pass", logK);
      run<8, 4, 128, false, 1, true>("wg128: 4 passes", logK);
      run<8, 2, 128, false, 1, true>("wg128: 2 passes", logK);
      run_persist<8, 2>("persist 2 passes", logK, 4);
      run_persist<8, 2>("persist 2 passes", logK, 7);
      run_persist<8, 1>("persist 1 pass", logK, 7);
      run_persist<8, 4>("persist 4 passes", logK, 7);
    }
    run<4, 2, 256, false, 1, true>("base: 2 passes", 8);
    run<4, 2, 256, false, 8, true>("run 8", 8);
    run<12, 2, 256, false, 1, true>("base: 2 passes", 14);
    run<12, 2, 256, false, 8, true>("run 8", 14);
  }
  return 0;
}
