// Micro-benchmarks that decide tile shapes of the CWT kernels on MI355X:
//   fill      : 16 B/lane contiguous stores (write-only HBM ceiling)
//   copy      : 16 B/lane contiguous load+store
//   seg<T>    : stores of T contiguous complex128 (T*16 B) at stride R*16 B -- the store pattern of
//               k_narrow / k_pass_b with TB = T
//   wr_rd<S>  : kernel A writes S MiB, kernel B reads it back -- does the 256 MiB Infinity Cache keep
//               a just-written two-pass intermediate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k_fill(double2* p, size_t n, double v) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) p[i] = make_double2(v, -v);
}
__global__ void k_copy(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) b[i] = a[i];
}
__global__ void k_read(const double2* __restrict__ a, size_t n, double* sink) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  double s = 0;
  for (; i < n; i += stride) { double2 v = a[i]; s += v.x + v.y; }
  if (s == 1.2345e300) *sink = s;
}
// rows x N complex128; WG = 512 threads handles r-tile of T: element (m, t) -> row*N + m*R + r0 + t
template <int LOGT>
__global__ void k_seg(double2* W, int logN, int logK, double v) {
  const int T = 1 << LOGT;
  const int logR = logN - logK;
  const int t = threadIdx.x & (T - 1), j = threadIdx.x >> LOGT;
  const int NT = blockDim.x >> LOGT;                 // threads along m
  const size_t r0 = size_t(blockIdx.x) << LOGT;
  double2* row = W + (size_t(blockIdx.y) << logN);
  const int K = 1 << logK;
  for (int m = j; m < K; m += NT) row[(size_t(m) << logR) + r0 + t] = make_double2(v + m, v - t);
}

// k_seg with (a) optional non-temporal stores and (b) an XCD-aware tile map: workgroup ids go round-robin
// over the 8 XCDs, so tile x' = (x & 7) * (tiles/8) + (x >> 3) gives every XCD one contiguous eighth of
// each output row (its L2 then owns whole 2-KiB runs instead of every 8th 128-B line).
template <int LOGT, bool NT_STORE, bool XCD>
__global__ void k_seg2(double2* W, int logN, int logK, double v) {
  typedef double v2 __attribute__((vector_size(16)));
  const int T = 1 << LOGT;
  const int logR = logN - logK;
  const int t = threadIdx.x & (T - 1), j = threadIdx.x >> LOGT;
  const int NTH = blockDim.x >> LOGT;
  unsigned x = blockIdx.x;
  if (XCD) x = (x & 7) * (gridDim.x >> 3) + (x >> 3);
  const size_t r0 = size_t(x) << LOGT;
  double2* row = W + (size_t(blockIdx.y) << logN);
  const int K = 1 << logK;
  for (int m = j; m < K; m += NTH) {
    double2* q = row + (size_t(m) << logR) + r0 + t;
    if (NT_STORE) { v2 w = {v + m, v - t}; __builtin_nontemporal_store(w, reinterpret_cast<v2*>(q)); }
    else *q = make_double2(v + m, v - t);
  }
}

// "one FFT per wavefront" store pattern: wave w of the workgroup owns residue r0 + w, lane l slot e owns
// output m = l + 64 e: every store instruction writes 64 separate 16-B pieces; the neighbouring pieces
// of each 128-B line come from the other 7 waves of the SAME workgroup (same CU, same L2).
__global__ void k_seg_wave(double2* W, int logN, int logK, double v) {
  const int logR = logN - logK;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const size_t r = (size_t(blockIdx.x) << 3) + w;
  double2* row = W + (size_t(blockIdx.y) << logN);
  const int K = 1 << logK;
  for (int m = l; m < K; m += 64) row[(size_t(m) << logR) + r] = make_double2(v + m, v - w);
}
// same, but the 8 waves are made to run ~in lockstep by a barrier per store round
__global__ void k_seg_wave_sync(double2* W, int logN, int logK, double v) {
  const int logR = logN - logK;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const size_t r = (size_t(blockIdx.x) << 3) + w;
  double2* row = W + (size_t(blockIdx.y) << logN);
  const int K = 1 << logK;
  for (int m = l; m < K; m += 64) { row[(size_t(m) << logR) + r] = make_double2(v + m, v - w); __syncthreads(); }
}

// copy with non-temporal stores (the pass-B pattern: read the just-written intermediate, stream W out)
__global__ void k_copy_nt(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
  typedef double v2 __attribute__((vector_size(16)));
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) { double2 v = a[i]; v2 w = {v.x, v.y}; __builtin_nontemporal_store(w, reinterpret_cast<v2*>(b) + i); }
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
  const size_t rows = 64, logN = 20, N = size_t(1) << logN;
  const size_t n = rows * N;                        // 1 GiB of complex128
  double2 *A, *B; double* sink;
  CK(hipMalloc(&A, n * 16)); CK(hipMalloc(&B, n * 16)); CK(hipMalloc(&sink, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto bench = [&](const char* name, double bytes, auto&& f) {
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    const float ms = time_ms(e0, e1) / reps;
    printf("%-28s %8.3f ms  %8.1f GB/s\n", name, ms, bytes / ms / 1e6);
  };
  bench("fill 1GiB (16B/lane)", n * 16.0, [&] { hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, A, n, 1.5); });
  bench("fill 1GiB grid=n/256", n * 16.0, [&] { hipLaunchKernelGGL(k_fill, dim3(n / 256), dim3(256), 0, 0, A, n, 1.5); });
  bench("copy 1GiB->1GiB", 2 * n * 16.0, [&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, A, B, n); });
  bench("read 1GiB", n * 16.0, [&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, A, n, sink); });
  for (int logK : {10, 8}) {
    char nm[64];
    snprintf(nm, 64, "seg T=2  (32B)  K=2^%d", logK);
    bench(nm, n * 16.0, [&] { hipLaunchKernelGGL(k_seg<1>, dim3(N >> (logK + 1), rows), dim3(512), 0, 0, A, int(logN), logK, 1.0); });
    snprintf(nm, 64, "seg T=4  (64B)  K=2^%d", logK);
    bench(nm, n * 16.0, [&] { hipLaunchKernelGGL(k_seg<2>, dim3(N >> (logK + 2), rows), dim3(512), 0, 0, A, int(logN), logK, 1.0); });
    snprintf(nm, 64, "seg T=8  (128B) K=2^%d", logK);
    bench(nm, n * 16.0, [&] { hipLaunchKernelGGL(k_seg<3>, dim3(N >> (logK + 3), rows), dim3(512), 0, 0, A, int(logN), logK, 1.0); });
    snprintf(nm, 64, "seg T=16 (256B) K=2^%d", logK);
    bench(nm, n * 16.0, [&] { hipLaunchKernelGGL(k_seg<4>, dim3(N >> (logK + 4), rows), dim3(512), 0, 0, A, int(logN), logK, 1.0); });
    snprintf(nm, 64, "seg T=64 (1KiB) K=2^%d", logK);
    bench(nm, n * 16.0, [&] { hipLaunchKernelGGL(k_seg<6>, dim3(N >> (logK + 6), rows), dim3(512), 0, 0, A, int(logN), logK, 1.0); });
  }
  bench("seg2 128B plain", n * 16.0, [&] { hipLaunchKernelGGL((k_seg2<3, false, false>), dim3(N >> 13, rows), dim3(512), 0, 0, A, int(logN), 10, 1.0); });
  bench("seg2 128B plain xcd", n * 16.0, [&] { hipLaunchKernelGGL((k_seg2<3, false, true>), dim3(N >> 13, rows), dim3(512), 0, 0, A, int(logN), 10, 1.0); });
  bench("seg2 128B nt", n * 16.0, [&] { hipLaunchKernelGGL((k_seg2<3, true, false>), dim3(N >> 13, rows), dim3(512), 0, 0, A, int(logN), 10, 1.0); });
  bench("seg2 128B nt xcd", n * 16.0, [&] { hipLaunchKernelGGL((k_seg2<3, true, true>), dim3(N >> 13, rows), dim3(512), 0, 0, A, int(logN), 10, 1.0); });
  bench("seg2 64B nt", n * 16.0, [&] { hipLaunchKernelGGL((k_seg2<2, true, false>), dim3(N >> 12, rows), dim3(512), 0, 0, A, int(logN), 10, 1.0); });
  bench("seg2 64B nt xcd", n * 16.0, [&] { hipLaunchKernelGGL((k_seg2<2, true, true>), dim3(N >> 12, rows), dim3(512), 0, 0, A, int(logN), 10, 1.0); });
  bench("seg2 64B plain xcd", n * 16.0, [&] { hipLaunchKernelGGL((k_seg2<2, false, true>), dim3(N >> 12, rows), dim3(512), 0, 0, A, int(logN), 10, 1.0); });
  bench("seg2 256B nt xcd", n * 16.0, [&] { hipLaunchKernelGGL((k_seg2<4, true, true>), dim3(N >> 14, rows), dim3(512), 0, 0, A, int(logN), 10, 1.0); });
  bench("seg2 128B nt xcd K=2^11", n * 16.0, [&] { hipLaunchKernelGGL((k_seg2<3, true, true>), dim3(N >> 14, rows), dim3(1024), 0, 0, A, int(logN), 11, 1.0); });
  bench("seg2 128B nt     K=2^11", n * 16.0, [&] { hipLaunchKernelGGL((k_seg2<3, true, false>), dim3(N >> 14, rows), dim3(1024), 0, 0, A, int(logN), 11, 1.0); });
  bench("seg per-wave 16B pieces K=2^10", n * 16.0, [&] { hipLaunchKernelGGL(k_seg_wave, dim3(N >> (10 + 3), rows), dim3(512), 0, 0, A, int(logN), 10, 1.0); });
  bench("seg per-wave 16B + barrier", n * 16.0, [&] { hipLaunchKernelGGL(k_seg_wave_sync, dim3(N >> (10 + 3), rows), dim3(512), 0, 0, A, int(logN), 10, 1.0); });
  // write S MiB then read it back; time of the read only (events around the read), min of 5
  printf("write-then-read (time of the read kernel only):\n");
  for (size_t mib : {16, 32, 64, 128, 192, 256, 512, 1024}) {
    const size_t cnt = mib * 1024 * 1024 / 16;
    float best = 1e9f, bestw = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, A, cnt, double(rep));
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      bestw = fminf(bestw, time_ms(e0, e1));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, A, cnt, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      best = fminf(best, time_ms(e0, e1));
    }
    printf("  %5zu MiB  write %8.1f GB/s   read-after-write %8.1f GB/s\n", mib, mib * 1.048576 / bestw,
           mib * 1.048576 / best);
  }
  printf("fill S MiB (plain stores), then copy it with nt stores to a disjoint 1 GiB area (time of the copy):\n");
  for (size_t mib : {32, 64, 128, 192, 256, 512, 1024}) {
    const size_t cnt = mib * 1024 * 1024 / 16;
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipLaunchKernelGGL(k_fill, dim3(unsigned(cnt / 256)), dim3(256), 0, 0, A, cnt, double(rep));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_copy_nt, dim3(unsigned(cnt / 256)), dim3(256), 0, 0, A, B, cnt);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      best = fminf(best, time_ms(e0, e1));
    }
    printf("  %5zu MiB  read+write %8.1f GB/s combined\n", mib, 2 * mib * 1.048576 / best);
  }
  return 0;
}
