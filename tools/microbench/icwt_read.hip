// Column reduction over a rows x N complex128 matrix (the access pattern of k_icwt): how the loads are issued.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/icwt_read.hip -o tools/microbench/icwt_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int U, int C>   // U rows in flight per thread, C adjacent columns per thread
__global__ void __launch_bounds__(256) k_red(const double2* __restrict__ W, long ldw, long ncols, int nrows,
                                             const double* __restrict__ w, double* __restrict__ out) {
  const long n = (long(blockIdx.x) * blockDim.x + threadIdx.x) * C;
  if (n >= ncols) return;
  double acc[U][C];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int c = 0; c < C; ++c) acc[u][c] = 0;
  for (int j = 0; j + U <= nrows; j += U) {
    double2 v[U][C];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) v[u][c] = W[long(j + u) * ldw + n + c];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) acc[u][c] += v[u][c].x * w[j + u];
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    double s = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) s += acc[u][c];
    out[n + c] = s;
  }
}

// round 6: the same walk with non-temporal loads (the matrix is read once: keep it out of L2 / the Infinity Cache),
// and with 128- / 512- / 1024-thread workgroups
typedef double v2d __attribute__((vector_size(16)));
template <int U, int TH>
__global__ void __launch_bounds__(TH) k_red_nt(const double2* __restrict__ W, long ldw, long ncols, int nrows,
                                               const double* __restrict__ w, double* __restrict__ out) {
  const long n = long(blockIdx.x) * TH + threadIdx.x;
  if (n >= ncols) return;
  double acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = 0;
  for (int j = 0; j + U <= nrows; j += U) {
    v2d v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(W + long(j + u) * ldw + n));
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] += v[u][0] * w[j + u];
  }
  double s = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) s += acc[u];
  out[n] = s;
}
template <int U, int TH>
__global__ void __launch_bounds__(TH) k_red_th(const double2* __restrict__ W, long ldw, long ncols, int nrows,
                                               const double* __restrict__ w, double* __restrict__ out) {
  const long n = long(blockIdx.x) * TH + threadIdx.x;
  if (n >= ncols) return;
  double acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = 0;
  for (int j = 0; j + U <= nrows; j += U) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = W[long(j + u) * ldw + n];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] += v[u].x * w[j + u];
  }
  double s = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) s += acc[u];
  out[n] = s;
}

// rows split over blockIdx.y (partial sums, second tiny pass not timed): more workgroups, shorter chains
template <int U>
__global__ void __launch_bounds__(256) k_red_split(const double2* __restrict__ W, long ldw, long ncols, int nrows, int parts,
                                                   const double* __restrict__ w, double* __restrict__ part) {
  const long n = long(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= ncols) return;
  const int per = nrows / parts, j0 = blockIdx.y * per;
  double acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = 0;
  for (int j = j0; j + U <= j0 + per; j += U) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = W[long(j + u) * ldw + n];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] += v[u].x * w[j + u];
  }
  double s = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) s += acc[u];
  part[long(blockIdx.y) * ncols + n] = s;
}

#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("line %d: %s\n", __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const long N = 1 << 20; const int rows = 256;
  double2* W; double *w, *out;
  HK(hipMalloc(&W, size_t(rows) * N * 16)); HK(hipMalloc(&w, rows * 8)); HK(hipMalloc(&out, size_t(8) * N * 8));
  {   // random-looking data (zeros clock higher: MI355X_MICROARCH.md, DVFS give-back)
    std::vector<double> h(size_t(1) << 22);
    unsigned s = 99u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = double(int(s)) * 4.656612873077393e-10; }
    for (size_t off = 0; off < size_t(rows) * N * 2; off += h.size()) HK(hipMemcpy(reinterpret_cast<double*>(W) + off, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    std::vector<double> hw(rows, 0.5);
    HK(hipMemcpy(w, hw.data(), rows * 8, hipMemcpyHostToDevice));
  }
  hipEvent_t a, b; HK(hipEventCreate(&a)); HK(hipEventCreate(&b));
  auto time = [&](const char* what, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    (void)hipEventRecord(a);
    for (int i = 0; i < 10; ++i) launch();
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 10;
    std::printf("%-52s %7.3f ms  %7.1f GB/s\n", what, ms, double(rows) * N * 16 / ms / 1e6);
  };
  time("1 column / thread, 4 rows in flight (k_icwt)", [&] { hipLaunchKernelGGL((k_red<4, 1>), dim3(N / 256), dim3(256), 0, 0, W, N, N, rows, w, out); });
  time("1 column / thread, 8 rows in flight", [&] { hipLaunchKernelGGL((k_red<8, 1>), dim3(N / 256), dim3(256), 0, 0, W, N, N, rows, w, out); });
  time("1 column / thread, 16 rows in flight", [&] { hipLaunchKernelGGL((k_red<16, 1>), dim3(N / 256), dim3(256), 0, 0, W, N, N, rows, w, out); });
  time("2 columns / thread, 4 rows in flight", [&] { hipLaunchKernelGGL((k_red<4, 2>), dim3(N / 512), dim3(256), 0, 0, W, N, N, rows, w, out); });
  time("2 columns / thread, 8 rows in flight", [&] { hipLaunchKernelGGL((k_red<8, 2>), dim3(N / 512), dim3(256), 0, 0, W, N, N, rows, w, out); });
  time("rows in 2 parts, 8 rows in flight", [&] { hipLaunchKernelGGL((k_red_split<8>), dim3(N / 256, 2), dim3(256), 0, 0, W, N, N, rows, 2, w, out); });
  time("rows in 4 parts, 8 rows in flight", [&] { hipLaunchKernelGGL((k_red_split<8>), dim3(N / 256, 4), dim3(256), 0, 0, W, N, N, rows, 4, w, out); });
  time("rows in 8 parts, 4 rows in flight", [&] { hipLaunchKernelGGL((k_red_split<4>), dim3(N / 256, 8), dim3(256), 0, 0, W, N, N, rows, 8, w, out); });
  time("non-temporal loads, 4 rows in flight", [&] { hipLaunchKernelGGL((k_red_nt<4, 256>), dim3(N / 256), dim3(256), 0, 0, W, N, N, rows, w, out); });
  time("non-temporal loads, 8 rows in flight", [&] { hipLaunchKernelGGL((k_red_nt<8, 256>), dim3(N / 256), dim3(256), 0, 0, W, N, N, rows, w, out); });
  time("non-temporal loads, 16 rows in flight", [&] { hipLaunchKernelGGL((k_red_nt<16, 256>), dim3(N / 256), dim3(256), 0, 0, W, N, N, rows, w, out); });
  time("non-temporal, 8 in flight, 128-thread workgroups", [&] { hipLaunchKernelGGL((k_red_nt<8, 128>), dim3(N / 128), dim3(128), 0, 0, W, N, N, rows, w, out); });
  time("non-temporal, 8 in flight, 512-thread workgroups", [&] { hipLaunchKernelGGL((k_red_nt<8, 512>), dim3(N / 512), dim3(512), 0, 0, W, N, N, rows, w, out); });
  time("plain, 8 in flight, 64-thread workgroups", [&] { hipLaunchKernelGGL((k_red_th<8, 64>), dim3(N / 64), dim3(64), 0, 0, W, N, N, rows, w, out); });
  time("plain, 8 in flight, 128-thread workgroups", [&] { hipLaunchKernelGGL((k_red_th<8, 128>), dim3(N / 128), dim3(128), 0, 0, W, N, N, rows, w, out); });
  time("plain, 8 in flight, 1024-thread workgroups", [&] { hipLaunchKernelGGL((k_red_th<8, 1024>), dim3(N / 1024), dim3(1024), 0, 0, W, N, N, rows, w, out); });
  return 0;
}
