// stream_poly.hip, closer to the real kernel: 64 rows x 2^20 outputs, planar coefficients coef[row][d][m] (K' intervals per
// row, written by a first kernel, i.e. they come from L2 / Infinity Cache / HBM as in the transform), the modulation from the
// two-level root-of-unity table (two look-ups + one complex product per output).  Variants:
//   A  one output per lane (16 B store)            B  two ADJACENT outputs per lane (2 x 16 B stores, second twiddle = first * step)
//   C  as A with the twiddle look-up replaced by a running product inside the lane (lower bound: no table)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double v2 __attribute__((vector_size(16)));

__device__ __forceinline__ double2 twn(const double2* hi, const double2* lo, unsigned t) {
  const double2 a = hi[t >> 10], b = lo[t & 1023u];
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

template <int D, int VAR>
__global__ void __launch_bounds__(256) k_rows(double2* __restrict__ W, const double2* __restrict__ coef, int logK,
                                              const double2* __restrict__ hi, const double2* __restrict__ lo, const int* kcs,
                                              double scale0) {
  constexpr int PT = VAR == 1 ? 2 : 1;
  const int logN = 20, logR = logN - logK;
  const unsigned row = blockIdx.y;
  const unsigned n = (blockIdx.x * 256u + threadIdx.x) * PT;
  const unsigned m = (VAR == 3) ? (n >> logR) : __builtin_amdgcn_readfirstlane(n >> logR);
  double2 c[D + 1];
  if (VAR == 6) {
#pragma unroll
    for (int d = 0; d <= D; ++d) c[d] = make_double2(scale0 + d, scale0 - d);
  } else if (VAR == 4 || VAR == 5) {
    const double2* a = coef + (VAR == 5 ? size_t(0) : ((size_t(row) << logK) + m) * (D + 1));
#pragma unroll
    for (int d = 0; d <= D; ++d) c[d] = a[d];
  } else {
    const double2* a = coef + ((size_t(row) * (D + 1)) << logK) + m;
#pragma unroll
    for (int d = 0; d <= D; ++d) c[d] = a[size_t(d) << logK];
  }
  const int kc = kcs[row];
  const unsigned r = n & ((1u << logR) - 1u);
  const double scale = 2.0 / double(1u << logR);
  double2 w;
  if (VAR == 2) w = make_double2(1.0 - 1e-9 * r, 1e-9 * r);
  else w = twn(hi, lo, (unsigned(kc) * n) & ((1u << logN) - 1u));
  double2* out = W + (size_t(row) << logN) + n;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const double u = double(int(r) + i) * scale - 1.0;
    double pr = c[D].x, pi = c[D].y;
#pragma unroll
    for (int d = D - 1; d >= 0; --d) { pr = fma(pr, u, c[d].x); pi = fma(pi, u, c[d].y); }
    v2 o = {pr * w.x - pi * w.y, pr * w.y + pi * w.x};
    __builtin_nontemporal_store(o, reinterpret_cast<v2*>(out) + i);
    if (PT == 2 && i == 0) {
      const double2 st = twn(hi, lo, unsigned(kc) & ((1u << logN) - 1u));    // uniform: e^{2 pi i kc / N}
      const double nx = w.x * st.x - w.y * st.y;
      w.y = w.x * st.y + w.y * st.x; w.x = nx;
    }
  }
}

__global__ void k_fillc(double2* p, size_t n) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) p[i] = make_double2(1e-3 * (i & 1023), 1.0);
}

template <int D, int VAR>
void run(double2* W, double2* coef, const double2* hi, const double2* lo, const int* kcs, int rows, int logK, hipEvent_t e0, hipEvent_t e1) {
  const size_t N = size_t(1) << 20, ncoef = (size_t(rows) * (D + 1)) << logK;
  constexpr int PT = VAR == 1 ? 2 : 1;
  const dim3 grid(unsigned(N / (256 * PT)), rows);
  float tot = 0;
  const int reps = 6;
  for (int i = 0; i < reps + 2; ++i) {
    hipLaunchKernelGGL(k_fillc, dim3(unsigned((ncoef + 255) / 256)), dim3(256), 0, 0, coef, ncoef);   // coefficients freshly written
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rows<D, VAR>), grid, dim3(256), 0, 0, W, coef, logK, hi, lo, kcs, 0.25);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (i >= 2) tot += ms;
  }
  const float ms = tot / reps;
  printf("variant %c degree %2d K' = 2^%-2d (R = %5d, coefficients %6.1f MB): %7.3f ms  %7.1f GB/s  %5.2f us/row\n", "ABCDEFG"[VAR], D, logK,
         1 << (20 - logK), ncoef * 16.0 / 1e6, ms, rows * N * 16.0 / ms / 1e6, ms * 1e3 / rows);
}

int main() {
  const int rows = 64;
  const size_t N = size_t(1) << 20;
  double2 *W, *coef, *hi, *lo; int* kcs;
  CK(hipMalloc(&W, rows * N * 16)); CK(hipMalloc(&coef, size_t(rows) * 13 * 16384 * 16)); CK(hipMalloc(&hi, 1024 * 16)); CK(hipMalloc(&lo, 1024 * 16));
  CK(hipMalloc(&kcs, rows * 4));
  CK(hipMemset(hi, 0, 1024 * 16)); CK(hipMemset(lo, 0, 1024 * 16));
  int h[64]; for (int i = 0; i < 64; ++i) h[i] = 1000 + 37 * i;
  CK(hipMemcpy(kcs, h, rows * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int logK : {8, 11, 14}) {
    run<8, 0>(W, coef, hi, lo, kcs, rows, logK, e0, e1);
    run<8, 3>(W, coef, hi, lo, kcs, rows, logK, e0, e1);
    run<8, 4>(W, coef, hi, lo, kcs, rows, logK, e0, e1);
  }
  run<8, 5>(W, coef, hi, lo, kcs, rows, 8, e0, e1);
  run<8, 6>(W, coef, hi, lo, kcs, rows, 8, e0, e1);
  run<2, 5>(W, coef, hi, lo, kcs, rows, 8, e0, e1);
  run<2, 6>(W, coef, hi, lo, kcs, rows, 8, e0, e1);
  run<0, 4>(W, coef, hi, lo, kcs, rows, 8, e0, e1);
  run<2, 4>(W, coef, hi, lo, kcs, rows, 8, e0, e1);
  run<4, 4>(W, coef, hi, lo, kcs, rows, 8, e0, e1);
  run<4, 3>(W, coef, hi, lo, kcs, rows, 8, e0, e1);
  run<12, 3>(W, coef, hi, lo, kcs, rows, 14, e0, e1);
  run<12, 4>(W, coef, hi, lo, kcs, rows, 14, e0, e1);
  run<4, 0>(W, coef, hi, lo, kcs, rows, 8, e0, e1);
  run<4, 1>(W, coef, hi, lo, kcs, rows, 8, e0, e1);
  run<12, 0>(W, coef, hi, lo, kcs, rows, 14, e0, e1);
  run<12, 1>(W, coef, hi, lo, kcs, rows, 13, e0, e1);
  return 0;
}
