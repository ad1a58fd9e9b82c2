// Micro-benchmark for VERDICT r03 "Next #1c" (tile-less streaming kernel for the band-limited rows), second form.
// stream_fma.hip fetched every FMA operand from memory (791 GB/s: load bound, not a fair price).  Here the operands sit
// where a real kernel would have them: a band-limited row, demodulated, is a polynomial of degree D on each interval of
// R = N/K' samples (Taylor/Chebyshev coefficients from D+1 short inverse FFTs of the band, a negligible first stage), so
// an output costs one Horner evaluation with WAVE-UNIFORM complex coefficients (scalar loads -> SGPR operands), the
// modulation e^{2 pi i k_c n / N} as a running product, and one contiguous non-temporal 16-byte store:
//   W[n] = e^{2 pi i k_c n / N} * sum_d a_d[n div R] u^d,   u = ((n mod R) + 1/2) 2/R - 1
// 2 D + 8 fp64 FMA-class instructions per 16 B.  Kill criterion: < 6.3 TB/s at D = 10.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// one workgroup = 256 threads = 4 waves; wave w of workgroup b covers outputs [ (b*4 + w) * 64 * I, + 64 * I ), all inside
// one interval (R >= 64 I): lane l writes n0 + l + 64 i, i < I
template <int D, int I>
__global__ void __launch_bounds__(256) k_poly(double2* __restrict__ out, const double2* __restrict__ coef, int logR,
                                              double2 step, const double2* __restrict__ tw) {
  typedef double v2 __attribute__((vector_size(16)));
  const unsigned wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256u + threadIdx.x) >> 6);
  const unsigned lane = threadIdx.x & 63u;
  const size_t n0 = size_t(wave) * (64 * I);
  const unsigned m = unsigned(n0 >> logR);                       // uniform: the interval
  const double2* a = coef + size_t(m & 4095u) * (D + 1);         // uniform address -> scalar loads
  double2 c[D + 1];
#pragma unroll
  for (int d = 0; d <= D; ++d) c[d] = a[d];
  const unsigned r0 = unsigned(n0 & ((size_t(1) << logR) - 1)) + lane;
  const double scale = 2.0 / double(1u << logR);
  double2 w = tw[(wave * 64u + lane) & 1023u];                   // e^{2 pi i k_c (n0 + lane) / N}: one table look-up
#pragma unroll
  for (int i = 0; i < I; ++i) {
    const double u = (double(r0 + 64 * i) + 0.5) * scale - 1.0;
    double pr = c[D].x, pi = c[D].y;
#pragma unroll
    for (int d = D - 1; d >= 0; --d) { pr = fma(pr, u, c[d].x); pi = fma(pi, u, c[d].y); }
    v2 o = {pr * w.x - pi * w.y, pr * w.y + pi * w.x};
    __builtin_nontemporal_store(o, reinterpret_cast<v2*>(out) + n0 + lane + 64 * i);
    const double nx = w.x * step.x - w.y * step.y;
    w.y = w.x * step.y + w.y * step.x; w.x = nx;
  }
}

template <int D, int I>
void run(double2* out, const double2* coef, const double2* tw, size_t n, int logR, hipEvent_t e0, hipEvent_t e1) {
  const unsigned grid = unsigned(n / (size_t(256) * I));
  const double2 step = make_double2(0.999, 0.0447);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_poly<D, I>), dim3(grid), dim3(256), 0, 0, out, coef, logR, step, tw);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_poly<D, I>), dim3(grid), dim3(256), 0, 0, out, coef, logR, step, tw);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("degree %2d, %2d outputs/thread, R = 2^%d: %3d FMA-class / 16 B  %8.3f ms  %8.1f GB/s\n", D, I, logR, 2 * D + 8, ms,
         n * 16.0 / ms / 1e6);
}

int main() {
  const size_t n = size_t(1) << 28;                   // 4 GiB of complex128 = W of BASELINE config 2
  double2 *out, *coef, *tw;
  CK(hipMalloc(&out, n * 16)); CK(hipMalloc(&coef, 4096 * 17 * 16)); CK(hipMalloc(&tw, 1024 * 16));
  CK(hipMemset(coef, 0, 4096 * 17 * 16)); CK(hipMemset(tw, 0, 1024 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  run<0, 1>(out, coef, tw, n, 10, e0, e1);
  run<0, 4>(out, coef, tw, n, 10, e0, e1);
  run<6, 1>(out, coef, tw, n, 10, e0, e1);
  run<6, 4>(out, coef, tw, n, 10, e0, e1);
  run<8, 4>(out, coef, tw, n, 10, e0, e1);
  run<10, 1>(out, coef, tw, n, 10, e0, e1);
  run<10, 2>(out, coef, tw, n, 10, e0, e1);
  run<10, 4>(out, coef, tw, n, 10, e0, e1);
  run<10, 8>(out, coef, tw, n, 10, e0, e1);
  run<12, 4>(out, coef, tw, n, 10, e0, e1);
  run<16, 4>(out, coef, tw, n, 10, e0, e1);
  run<10, 4>(out, coef, tw, n, 8, e0, e1);
  run<10, 1>(out, coef, tw, n, 6, e0, e1);
  return 0;
}
