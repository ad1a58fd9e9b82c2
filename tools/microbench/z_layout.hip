// Which layout of the two-pass intermediate Z is cheaper on MI355X?  Memory traffic of the two passes
// only (no FFT arithmetic), N = 2^20 = R*K with R = K = 1024, complex128, chunks of 12 rows (192 MiB of Z):
//   layout RQ (shipped):  pass A stores Z[r][q] as 128-B segments at a 16-KiB stride,
//                         pass B loads contiguous 16-KiB runs;
//   layout QR:            pass A stores contiguous 16-KiB runs, pass B loads 128-B segments at a stride.
// Pass B always stores W[R m + r] as non-temporal 128-B segments (the output layout is fixed).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double v2 __attribute__((vector_size(16)));

// workgroup = 512 threads = 8 columns q (t) x 64 (j); slot e: r = j + 64 e
template <bool QR>
__global__ void __launch_bounds__(512) k_a(const double2* __restrict__ X, double2* __restrict__ Z) {
  const int t = threadIdx.x & 7, j = threadIdx.x >> 3;
  const unsigned q = blockIdx.x * 8 + t;
  double2* z = Z + (size_t(blockIdx.y) << 20);
  double2 acc[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = X[q + (unsigned(j + 64 * e) << 10)];     // column loads (L2-resident spectrum)
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const unsigned r = j + 64 * e;
    z[QR ? (q << 10) + r : (r << 10) + q] = acc[e];
  }
}

// workgroup = 512 threads = 8 residues r x 64; loads 16 values per thread, stores 16
template <bool QR, bool XCD>
__global__ void __launch_bounds__(512) k_b(const double2* __restrict__ Z, double2* __restrict__ W) {
  unsigned x = blockIdx.x;
  if (XCD) x = (x & 7) * (gridDim.x >> 3) + (x >> 3);
  const double2* z = Z + (size_t(blockIdx.y) << 20);
  double2* w = W + (size_t(blockIdx.y) << 20);
  double2 acc[16];
  const int t = threadIdx.x & 7, j = threadIdx.x >> 3;
  if (QR) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = z[(unsigned(j + 64 * e) << 10) + x * 8 + t];
  } else {
    const int jj = threadIdx.x & 63, tt = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = z[((x * 8 + tt) << 10) + jj + 64 * e];
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    v2 v = {acc[e].x, acc[e].y};
    __builtin_nontemporal_store(v, reinterpret_cast<v2*>(w + (unsigned(j + 64 * e) << 10) + x * 8 + t));
  }
}

// pass B of layout RQ plus NX LDS exchanges shaped like the FFT's (per plane: 16 b64 writes, wave barrier,
// 16 b64 reads at the transposed, padded index) -- no arithmetic: how much of the real kernel's time is LDS?
template <int NX, bool B128>
__global__ void __launch_bounds__(512) k_b_lds(const double2* __restrict__ Z, double2* __restrict__ W) {
  extern __shared__ double2 lds_raw[];
  double* lds = reinterpret_cast<double*>(lds_raw);
  unsigned x = blockIdx.x;
  x = (x & 7) * (gridDim.x >> 3) + (x >> 3);
  const double2* z = Z + (size_t(blockIdx.y) << 20);
  double2* w = W + (size_t(blockIdx.y) << 20);
  double2 acc[16];
  const int t = threadIdx.x & 7, j = threadIdx.x >> 3;
  const int jj = threadIdx.x & 63, tt = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = z[((x * 8 + tt) << 10) + jj + 64 * e];
  auto pad = [](int a) { return a + (a >> 4); };
#pragma unroll
  for (int xch = 0; xch < NX; ++xch) {
    if (B128) {
      double2* l2 = lds_raw + tt * 1088;
#pragma unroll
      for (int e = 0; e < 16; ++e) l2[pad(jj + 64 * e)] = acc[e];
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = l2[pad(jj * 16 + e)];
      __syncthreads();
    } else {
      double* l = lds + tt * 1088;
#pragma unroll
      for (int e = 0; e < 16; ++e) l[pad(jj + 64 * e)] = acc[e].x;
      __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e].x = l[pad(jj * 16 + e)];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int e = 0; e < 16; ++e) l[pad(jj + 64 * e)] = acc[e].y;
      __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e].y = l[pad(jj * 16 + e)];
      __builtin_amdgcn_wave_barrier();
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    v2 v = {acc[e].x, acc[e].y};
    __builtin_nontemporal_store(v, reinterpret_cast<v2*>(w + (unsigned(j + 64 * e) << 10) + x * 8 + t));
  }
}

int main() {
  const int rows = 12, chunks = 9;
  const size_t N = size_t(1) << 20;
  double2 *X, *Z, *W;
  CK(hipMalloc(&X, N * 16)); CK(hipMalloc(&Z, rows * N * 16)); CK(hipMalloc(&W, size_t(rows) * chunks * N * 16));
  CK(hipMemset(X, 0, N * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto&& a, auto&& b) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0));
      for (int c = 0; c < chunks; ++c) { a(); b(W + size_t(c) * rows * N); }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) best = fminf(best, ms);
    }
    printf("%-34s %7.3f ms for %d rows  (%5.2f us/row)\n", name, best, rows * chunks, 1e3 * best / (rows * chunks));
  };
  const dim3 g(128, rows), blk(512);
  auto a_rq = [&] { hipLaunchKernelGGL(k_a<false>, g, blk, 0, 0, X, Z); };
  auto a_qr = [&] { hipLaunchKernelGGL(k_a<true>, g, blk, 0, 0, X, Z); };
  auto nop = [&] {};
  auto nopb = [&](double2*) {};
  run("pass A only, Z[r][q] (strided st)", a_rq, nopb);
  run("pass A only, Z[q][r] (contig st)", a_qr, nopb);
  run("pass B only, Z[r][q] (contig ld)", nop, [&](double2* w) { hipLaunchKernelGGL((k_b<false, false>), g, blk, 0, 0, Z, w); });
  run("pass B only, Z[q][r] (strided ld)", nop, [&](double2* w) { hipLaunchKernelGGL((k_b<true, false>), g, blk, 0, 0, Z, w); });
  run("A+B Z[r][q]", a_rq, [&](double2* w) { hipLaunchKernelGGL((k_b<false, false>), g, blk, 0, 0, Z, w); });
  run("A+B Z[r][q] xcd", a_rq, [&](double2* w) { hipLaunchKernelGGL((k_b<false, true>), g, blk, 0, 0, Z, w); });
  run("A+B Z[q][r]", a_qr, [&](double2* w) { hipLaunchKernelGGL((k_b<true, false>), g, blk, 0, 0, Z, w); });
  run("A+B Z[q][r] xcd", a_qr, [&](double2* w) { hipLaunchKernelGGL((k_b<true, true>), g, blk, 0, 0, Z, w); });
  const size_t l64 = 8 * 1088 * 8, l128 = 8 * 1088 * 16;
  run("pass B + 1 LDS exchange (b64)", nop, [&](double2* w) { hipLaunchKernelGGL((k_b_lds<1, false>), g, blk, l64, 0, Z, w); });
  run("pass B + 2 LDS exchanges (b64)", nop, [&](double2* w) { hipLaunchKernelGGL((k_b_lds<2, false>), g, blk, l64, 0, Z, w); });
  run("pass B + 3 LDS exchanges (b64)", nop, [&](double2* w) { hipLaunchKernelGGL((k_b_lds<3, false>), g, blk, l64, 0, Z, w); });
  run("pass B + 3 LDS exchanges, 1 WG/CU", nop, [&](double2* w) { hipLaunchKernelGGL((k_b_lds<3, false>), g, blk, 100 * 1024, 0, Z, w); });
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_b_lds<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  run("pass B + 3 LDS exchanges (b128)", nop, [&](double2* w) { hipLaunchKernelGGL((k_b_lds<3, true>), g, blk, l128, 0, Z, w); });
  return 0;
}
