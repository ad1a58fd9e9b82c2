// stream_poly2.hip showed what the first form of k_poly_rows ran into: a wave that starts with a scalar load of ITS interval's
// coefficients (a different address for every 16th workgroup: a scalar-cache miss) lives ~1.5 us for 1 KB of output -- latency
// bound at 4.7 TB/s; the same kernel with an always-hit address runs at 6.5 TB/s.  This form amortises the fetch:
//   a workgroup of 256 threads covers 256 * I consecutive outputs in I passes of 4 KB (the store window of the I = 1 kernel),
//   the coefficient sets of the intervals it touches are fetched ONCE, cooperatively, into LDS (one barrier), every pass
//   reads its set from LDS (per-lane address: lanes may straddle intervals, so R < 64 is allowed),
//   the modulation is one table look-up per lane and a running product over the passes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double v2 __attribute__((vector_size(16)));

__device__ __forceinline__ double2 twn(const double2* hi, const double2* lo, unsigned t) {
  const double2 a = hi[t >> 10], b = lo[t & 1023u];
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

template <int D, int I>
__global__ void __launch_bounds__(256) k_rows(double2* __restrict__ W, const double2* __restrict__ coef, int logK,
                                              const double2* __restrict__ hi, const double2* __restrict__ lo, const int* kcs) {
  extern __shared__ double2 sc[];                       // [interval][d]
  const int logN = 20, logR = logN - logK;
  const unsigned row = blockIdx.y;
  const unsigned n0 = blockIdx.x * (256u * I);
  const unsigned m0 = n0 >> logR;
  const unsigned nint = (((n0 + 256u * I - 1u) >> logR) - m0 + 1u) * (D + 1);   // complex values to stage
  const double2* a = coef + (((size_t(row) << logK) + m0) * (D + 1));
  for (unsigned t = threadIdx.x; t < nint; t += 256u) sc[t] = a[t];
  const int kc = kcs[row];
  double2 w = twn(hi, lo, (unsigned(kc) * (n0 + threadIdx.x)) & ((1u << logN) - 1u));
  const double2 st = twn(hi, lo, (unsigned(kc) * 256u) & ((1u << logN) - 1u));
  const double scale = 2.0 / double(1u << logR);
  double2* out = W + (size_t(row) << logN) + n0 + threadIdx.x;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < I; ++i) {
    const unsigned n = n0 + i * 256u + threadIdx.x;
    const double2* c = sc + ((n >> logR) - m0) * (D + 1);
    const double u = double(int(n & ((1u << logR) - 1u))) * scale - 1.0;
    double pr = c[D].x, pi = c[D].y;
#pragma unroll
    for (int d = D - 1; d >= 0; --d) { const double2 cd = c[d]; pr = fma(pr, u, cd.x); pi = fma(pi, u, cd.y); }
    v2 o = {pr * w.x - pi * w.y, pr * w.y + pi * w.x};
    __builtin_nontemporal_store(o, reinterpret_cast<v2*>(out) + i * 256);
    const double nx = w.x * st.x - w.y * st.y;
    w.y = w.x * st.y + w.y * st.x; w.x = nx;
  }
}

__global__ void k_fillc(double2* p, size_t n) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) p[i] = make_double2(1e-3 * (i & 1023), 1.0);
}

template <int D, int I>
void run(double2* W, double2* coef, const double2* hi, const double2* lo, const int* kcs, int rows, int logK, hipEvent_t e0, hipEvent_t e1) {
  const size_t N = size_t(1) << 20, ncoef = (size_t(rows) * (D + 1)) << logK;
  const dim3 grid(unsigned(N / (256 * I)), rows);
  const size_t lds = ((size_t(256 * I) >> (20 - logK)) + 2) * (D + 1) * 16;
  float tot = 0;
  const int reps = 6;
  for (int i = 0; i < reps + 2; ++i) {
    hipLaunchKernelGGL(k_fillc, dim3(unsigned((ncoef + 255) / 256)), dim3(256), 0, 0, coef, ncoef);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rows<D, I>), grid, dim3(256), lds, 0, W, coef, logK, hi, lo, kcs);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (i >= 2) tot += ms;
  }
  const float ms = tot / reps;
  printf("LDS-staged, %d passes, degree %2d K' = 2^%-2d (R = %5d, coefficients %6.1f MB): %7.3f ms  %7.1f GB/s  %5.2f us/row\n", I, D, logK,
         1 << (20 - logK), ncoef * 16.0 / 1e6, ms, rows * N * 16.0 / ms / 1e6, ms * 1e3 / rows);
}

int main() {
  const int rows = 64;
  const size_t N = size_t(1) << 20;
  double2 *W, *coef, *hi, *lo; int* kcs;
  CK(hipMalloc(&W, rows * N * 16)); CK(hipMalloc(&coef, size_t(rows) * 13 * 65536 * 16)); CK(hipMalloc(&hi, 1024 * 16)); CK(hipMalloc(&lo, 1024 * 16));
  CK(hipMalloc(&kcs, rows * 4));
  CK(hipMemset(hi, 0, 1024 * 16)); CK(hipMemset(lo, 0, 1024 * 16));
  int h[64]; for (int i = 0; i < 64; ++i) h[i] = 1000 + 37 * i;
  CK(hipMemcpy(kcs, h, rows * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int logK : {8, 11, 14}) {
    run<8, 1>(W, coef, hi, lo, kcs, rows, logK, e0, e1);
    run<8, 2>(W, coef, hi, lo, kcs, rows, logK, e0, e1);
    run<8, 4>(W, coef, hi, lo, kcs, rows, logK, e0, e1);
    run<8, 8>(W, coef, hi, lo, kcs, rows, logK, e0, e1);
  }
  run<4, 4>(W, coef, hi, lo, kcs, rows, 8, e0, e1);
  run<6, 4>(W, coef, hi, lo, kcs, rows, 10, e0, e1);
  run<12, 4>(W, coef, hi, lo, kcs, rows, 14, e0, e1);
  run<6, 4>(W, coef, hi, lo, kcs, rows, 16, e0, e1);
  run<4, 4>(W, coef, hi, lo, kcs, rows, 16, e0, e1);
  return 0;
}
