// Micro-benchmark for VERDICT r03 "Next #1c": can a tile-less streaming kernel (no LDS, no barriers, dozens of waves per
// CU) sustain the contiguous-store ceiling while it spends NF fp64 FMAs per 16-byte output?  A polyphase interpolator
// that upsamples a critically sampled band-limited row needs 2 x taps FMAs per complex output (taps = 16 ... 24).
//   out[i] = sum_t c[t] * v[(i >> 6) + t]   (NF/2 taps, complex value x real coefficient = 2 FMAs per tap)
// c and v come from small L1/L2-resident arrays, the store is one non-temporal 16-byte store per lane, contiguous.
// Kill criterion: < 6.3 TB/s at NF = 48 -> the idea is dropped (recorded in EXPERIMENTS.md).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int TAPS, int PER_THREAD>
__global__ void __launch_bounds__(256) k_stream(double2* __restrict__ out, const double2* __restrict__ v,
                                                const double* __restrict__ c, size_t n) {
  typedef double v2 __attribute__((vector_size(16)));
  const size_t base = (size_t(blockIdx.x) * blockDim.x) * PER_THREAD + threadIdx.x;
#pragma unroll
  for (int u = 0; u < PER_THREAD; ++u) {
    const size_t i = base + size_t(u) * blockDim.x;
    if (i >= n) return;
    const unsigned r = unsigned(i) & 63u;             // polyphase index: coefficients differ per output phase
    const unsigned m = unsigned(i >> 6) & 1023u;      // coarse sample index
    double sr = 0, si = 0;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const double2 x = v[m + t];
      const double h = c[r * TAPS + t];
      sr = fma(h, x.x, sr);
      si = fma(h, x.y, si);
    }
    v2 w = {sr, si};
    __builtin_nontemporal_store(w, reinterpret_cast<v2*>(out) + i);
  }
}

template <int TAPS, int PT>
void run(const char* name, double2* out, const double2* v, const double* c, size_t n, hipEvent_t e0, hipEvent_t e1) {
  const unsigned grid = unsigned((n + size_t(256) * PT - 1) / (size_t(256) * PT));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_stream<TAPS, PT>), dim3(grid), dim3(256), 0, 0, out, v, c, n);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_stream<TAPS, PT>), dim3(grid), dim3(256), 0, 0, out, v, c, n);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("%-34s taps %2d (%3d fp64 FMA / 16 B)  %8.3f ms  %8.1f GB/s  %6.1f TFLOP/s\n", name, TAPS, 2 * TAPS, ms,
         n * 16.0 / ms / 1e6, n * 4.0 * TAPS / ms / 1e9);
}

int main() {
  const size_t n = size_t(1) << 28;                   // 4 GiB of complex128 = W of BASELINE config 2
  double2 *out, *v; double* c;
  CK(hipMalloc(&out, n * 16)); CK(hipMalloc(&v, 2048 * 16)); CK(hipMalloc(&c, 64 * 64 * 8));
  CK(hipMemset(v, 0, 2048 * 16)); CK(hipMemset(c, 0, 64 * 64 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  run<0, 1>("store only, 1 output/thread", out, v, c, n, e0, e1);
  run<0, 4>("store only, 4 outputs/thread", out, v, c, n, e0, e1);
  run<8, 1>("1 output/thread", out, v, c, n, e0, e1);
  run<16, 1>("1 output/thread", out, v, c, n, e0, e1);
  run<24, 1>("1 output/thread", out, v, c, n, e0, e1);
  run<32, 1>("1 output/thread", out, v, c, n, e0, e1);
  run<16, 4>("4 outputs/thread", out, v, c, n, e0, e1);
  run<24, 4>("4 outputs/thread", out, v, c, n, e0, e1);
  run<32, 4>("4 outputs/thread", out, v, c, n, e0, e1);
  return 0;
}
