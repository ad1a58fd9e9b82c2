// Would a fused icwt epilogue pay?  ("fuse the eq.-11 reduction into the row kernels".)
// The fused form adds Re(W[j, n]) * w_j into out[n] from every row kernel, i.e. rows x N fp64 atomic adds with `rows`
// colliding addends per address.  This measures exactly that traffic -- hardware fp64 atomics (global_atomic_add_f64),
// the same 128-byte-segment pattern as the band-limited kernel (8 adjacent columns per lane group, 16 KiB stride) and
// the contiguous pattern of the overlap-save rows -- against the separate pass that exists (k_icwt: one read of W).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/icwt_atomics.hip -o tools/microbench/icwt_atomics && ./icwt_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_atomic_contig(double* acc, int logN) {
  const unsigned n = (blockIdx.x * blockDim.x + threadIdx.x) & ((1u << logN) - 1u);
  unsafeAtomicAdd(&acc[n], 1.0 + n * 1e-9);
}
// thread t of a workgroup of 512: lane group of 8 adjacent columns r0..r0+7, 16 slots m = j + 64 e at stride R = N / 1024
__global__ void k_atomic_strided(double* acc, int logN) {
  const unsigned R = 1u << (logN - 10), tile = blockIdx.x & ((R >> 3) - 1u);
  const unsigned t = threadIdx.x & 7u, j = threadIdx.x >> 3;
#pragma unroll
  for (int e = 0; e < 16; ++e) unsafeAtomicAdd(&acc[(j + 64u * e) * R + tile * 8u + t], 1.0 + e);
}
__global__ void k_read_rows(const double2* W, long ld, int rows, double* out, long N) {
  const long n = long(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double a = 0;
  for (int j = 0; j < rows; ++j) a += W[long(j) * ld + n].x;
  out[n] = a;
}
int main() {
  const int logN = 20, rows = 256;
  const long N = 1L << logN;
  double *acc, *out; double2* W;
  CK(hipMalloc(&acc, N * 8)); CK(hipMalloc(&out, N * 8)); CK(hipMalloc(&W, size_t(rows) * N * 16));
  CK(hipMemset(acc, 0, N * 8)); CK(hipMemset(W, 0, size_t(rows) * N * 16));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timeit = [&](const char* name, auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-58s %8.3f ms per %d rows x 2^%d\n", name, ms / 5, rows, logN);
  };
  timeit("fp64 atomic add, contiguous columns (overlap-save rows)", [&] {
    hipLaunchKernelGGL(k_atomic_contig, dim3(unsigned(rows * (N / 256))), dim3(256), 0, 0, acc, logN); });
  timeit("fp64 atomic add, 64-byte groups at 16 KiB stride (band-limited)", [&] {
    hipLaunchKernelGGL(k_atomic_strided, dim3(unsigned(rows * (N / 8192))), dim3(512), 0, 0, acc, logN); });
  timeit("separate pass: read of all of W, column sums (k_icwt's traffic)", [&] {
    hipLaunchKernelGGL(k_read_rows, dim3(unsigned(N / 256)), dim3(256), 0, 0, W, N, rows, out, N); });
  return 0;
}
