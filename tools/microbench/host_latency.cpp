// Where the time of a small pycwt_amd.cwt() call goes at the C boundary (504 samples x 97 scales, fp64), and whether the
// kernels should read the signal from / write W into page-locked host memory themselves instead of going through copies.
//   hipcc -O2 --offload-arch=gfx950 -Iinclude tools/microbench/host_latency.cpp -Lpycwt_amd -lcwt_hip -Wl,-rpath,$PWD/pycwt_amd -o /tmp/host_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "cwt_hip.h"

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { int rc_ = (x); if (rc_) { std::printf("line %d: rc %d %s\n", __LINE__, rc_, cwt_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("line %d: %s\n", __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

template <typename F> static double best(F f, int reps = 300) {
  for (int i = 0; i < 20; ++i) f();
  double b = 1e30;
  for (int i = 0; i < reps; ++i) { const double t = now(); f(); b = std::min(b, now() - t); }
  return b;
}
template <typename F> static double mean(F f, int reps = 300) {
  for (int i = 0; i < 20; ++i) f();
  const double t = now();
  for (int i = 0; i < reps; ++i) f();
  return (now() - t) / reps;
}

int main(int argc, char** argv) {
  const int64_t n0 = argc > 1 ? atol(argv[1]) : 504;
  const int rows = argc > 2 ? atoi(argv[2]) : 97;
  int64_t N = 1; while (N < n0) N <<= 1;
  cwt_plan* p = nullptr;
  CK(cwt_plan_create(&p, 0, N, 64, 1024));
  std::vector<double> x(n0), sj(rows);
  for (int64_t i = 0; i < n0; ++i) x[i] = std::sin(0.05 * i) + 0.3 * std::cos(1.7 * i);
  for (int j = 0; j < rows; ++j) sj[j] = 0.5 * std::pow(2.0, j / 12.0);
  const size_t wb = size_t(rows) * n0 * 16, xb = size_t(n0) * 8, hb = size_t(N) * 16;
  std::vector<char> W(wb), W2(wb), xh(hb);
  void *xd, *xhd, *Wd, *pin;
  CK(cwt_malloc(0, &xd, xb)); CK(cwt_malloc(0, &xhd, hb)); CK(cwt_malloc(0, &Wd, wb));
  HK(hipHostMalloc(&pin, (size_t(8) << 20)));
  char* pinc = static_cast<char*>(pin);
  hipStream_t st; HK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CK(cwt_plan_set_stream(p, st));
  auto sync = [&] { (void)hipStreamSynchronize(st); };

  std::printf("n0 = %ld, rows = %d, N = %ld: W %.0f KB\n", long(n0), rows, long(N), wb / 1e3);
  std::printf("%-64s %8s %8s\n", "", "best us", "mean us");
  auto line = [&](const char* what, auto f) { const double b = best(f), m = mean(f); std::printf("%-64s %8.1f %8.1f\n", what, b, m); };

  line("cwt_execute_host (W and spectrum)", [&] { cwt_execute_host(p, x.data(), n0, 0, 6.0, 0.25, sj.data(), rows, W.data(), xh.data()); });
  line("cwt_execute_host (W only)", [&] { cwt_execute_host(p, x.data(), n0, 0, 6.0, 0.25, sj.data(), rows, W.data(), nullptr); });
  {
    void* Wp = nullptr;
    CK(cwt_host_malloc(&Wp, wb));
    line("cwt_execute_host, W in a cwt_host_malloc buffer", [&] { cwt_execute_host(p, x.data(), n0, 0, 6.0, 0.25, sj.data(), rows, Wp, xh.data()); });
    std::printf("  same bytes as into pageable memory: %s\n", std::memcmp(W.data(), Wp, wb) ? "NO" : "yes");
    for (int wg : {512, 1024, 2048, 4096}) {               // rows per workgroup of the row kernel = wg / N
      if (wg < N) continue;
      CK(cwt_plan_set_option(p, "wg_points", wg));
      char what[96];
      std::snprintf(what, sizeof what, "  ... with %d-point workgroups (%d rows each)", wg, int(wg / N));
      line(what, [&] { cwt_execute_host(p, x.data(), n0, 0, 6.0, 0.25, sj.data(), rows, Wp, xh.data()); });
    }
    CK(cwt_host_free(Wp));
    CK(cwt_plan_set_option(p, "host_direct", 0));
    line("cwt_execute_host, option host_direct = 0 (copies: round 3)", [&] { cwt_execute_host(p, x.data(), n0, 0, 6.0, 0.25, sj.data(), rows, W.data(), xh.data()); });
    CK(cwt_plan_set_option(p, "host_direct", 1));
  }
  line("empty stream synchronize", [&] { sync(); });
  line("H2D 4 KB from pinned + sync", [&] { (void)hipMemcpyAsync(xd, pin, xb, hipMemcpyHostToDevice, st); sync(); });
  line("cwt_transform on device buffers + sync", [&] { cwt_transform(p, xd, n0, 0, 6.0, 0.25, sj.data(), rows, xhd, Wd, n0, n0); sync(); });
  line("cwt_transform, host API time only (no sync)", [&] { cwt_transform(p, xd, n0, 0, 6.0, 0.25, sj.data(), rows, xhd, Wd, n0, n0); });
  sync();
  line("cwt_forward_fft + sync", [&] { cwt_forward_fft(p, xd, n0, xhd); sync(); });
  line("D2H W to pinned + sync", [&] { (void)hipMemcpyAsync(pinc + (2 << 20), Wd, wb, hipMemcpyDeviceToHost, st); sync(); });
  line("D2H spectrum to pinned + sync", [&] { (void)hipMemcpyAsync(pinc + (1 << 20), xhd, hb, hipMemcpyDeviceToHost, st); sync(); });
  line("memcpy W pinned -> pageable", [&] { std::memcpy(W.data(), pinc + (2 << 20), wb); });
  line("H2D + transform + 2 x D2H + sync (what execute_host queues)", [&] {
    (void)hipMemcpyAsync(xd, pin, xb, hipMemcpyHostToDevice, st);
    cwt_transform(p, xd, n0, 0, 6.0, 0.25, sj.data(), rows, xhd, Wd, n0, n0);
    (void)hipMemcpyAsync(pinc + (1 << 20), xhd, hb, hipMemcpyDeviceToHost, st);
    (void)hipMemcpyAsync(pinc + (2 << 20), Wd, wb, hipMemcpyDeviceToHost, st);
    sync(); });
  // kernels on the page-locked buffer itself: signal read over PCIe, W (and the spectrum) written over PCIe
  std::memcpy(pin, x.data(), xb);
  line("cwt_transform: x, W in pinned host memory; spectrum on device", [&] { cwt_transform(p, pin, n0, 0, 6.0, 0.25, sj.data(), rows, xhd, pinc + (2 << 20), n0, n0); sync(); });
  line("cwt_transform: x, W, spectrum all in pinned host memory", [&] { cwt_transform(p, pin, n0, 0, 6.0, 0.25, sj.data(), rows, pinc + (1 << 20), pinc + (2 << 20), n0, n0); sync(); });
  line("cwt_transform: x on device, W pinned", [&] { cwt_transform(p, xd, n0, 0, 6.0, 0.25, sj.data(), rows, xhd, pinc + (2 << 20), n0, n0); sync(); });
  line("pinned x, W + D2H spectrum", [&] {
    cwt_transform(p, pin, n0, 0, 6.0, 0.25, sj.data(), rows, xhd, pinc + (2 << 20), n0, n0);
    (void)hipMemcpyAsync(pinc + (1 << 20), xhd, hb, hipMemcpyDeviceToHost, st); sync(); });
  // results agree?
  cwt_execute_host(p, x.data(), n0, 0, 6.0, 0.25, sj.data(), rows, W.data(), xh.data());
  cwt_transform(p, pin, n0, 0, 6.0, 0.25, sj.data(), rows, pinc + (1 << 20), pinc + (2 << 20), n0, n0); sync();
  std::printf("pinned-memory result identical to cwt_execute_host: W %s, spectrum %s\n",
              std::memcmp(W.data(), pinc + (2 << 20), wb) ? "NO" : "yes", std::memcmp(xh.data(), pinc + (1 << 20), hb) ? "NO" : "yes");
  return 0;
}
