/* A host in plain C (C99, no Python, no torch, no HIP headers) driving libcwt_hip.so through include/cwt_hip.h:
 * the drop-in boundary of SURVEY.md 8b as a compiled-language caller would use it.
 *
 *   gcc -std=c99 -O1 -I include tests/c_host/cwt_host.c -o cwt_host -L <dir of the library> -l<name> -lm
 *   ./cwt_host [log2 N] [precision 32|64]
 *
 * Workload and check: x[n] = cos(w_m n dt), w_m = 2 pi m / (N dt) -- for a power-of-two length the reference's
 * wavelet.py:91-106 then has the closed form
 *   W[j,n] = 1/2 sqrt(2 pi s_j / dt) [ conj psi_ft(s_j w_m) e^{+i w_m n dt} + conj psi_ft(-s_j w_m) e^{-i w_m n dt} ],
 * psi_ft(f) = pi^(-1/4) exp(-(f - f0)^2 / 2) for the Morlet (mothers.py:26-28, evaluated at negative f too).
 * Every row of W comes back and is compared with it.  Exit status 0 = within the tolerance. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cwt_hip.h"

#define CHECK(call)                                                                   \
  do {                                                                                \
    int rc_ = (call);                                                                 \
    if (rc_ != CWT_OK) {                                                              \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, cwt_last_error());          \
      return 2;                                                                       \
    }                                                                                 \
  } while (0)

int main(int argc, char** argv) {
  const int logn = argc > 1 ? atoi(argv[1]) : 14;
  const int prec = argc > 2 ? atoi(argv[2]) : 64;
  const int64_t N = (int64_t)1 << logn;
  const int rows = 24, m = 37;
  const double pi = 3.14159265358979323846, dt = 0.5, f0 = 6.0;
  const size_t es = prec == 64 ? sizeof(double) : sizeof(float);
  int ndev = 0;
  CHECK(cwt_device_count(&ndev));
  printf("backend %s, %d device(s), N = 2^%d, fp%d, %d scales\n", cwt_backend(), ndev, logn, prec, rows);
  if (ndev < 1) { fprintf(stderr, "no device\n"); return 2; }

  /* scales as wavelet.py:75-85 would choose them for this length: s0 = 2 dt / flambda, rows on a log grid up to N dt / 8 */
  const double flambda = 4.0 * pi / (f0 + sqrt(2.0 + f0 * f0));
  const double s0 = 2.0 * dt / flambda;
  double scales[24];
  for (int j = 0; j < rows; ++j) scales[j] = s0 * pow((double)N * dt / 8.0 / s0, (double)j / (rows - 1));

  const double wm = 2.0 * pi * m / ((double)N * dt);
  void* x = malloc((size_t)N * es);
  for (int64_t n = 0; n < N; ++n) {
    const double v = cos(wm * (double)n * dt);
    if (prec == 64) ((double*)x)[n] = v; else ((float*)x)[n] = (float)v;
  }

  cwt_plan* plan = NULL;
  void *x_dev = NULL, *xhat_dev = NULL, *W_dev = NULL;
  CHECK(cwt_plan_create(&plan, 0, N, prec, rows));
  CHECK(cwt_malloc(0, &x_dev, (size_t)N * es));
  CHECK(cwt_malloc(0, &xhat_dev, (size_t)N * 2 * es));
  CHECK(cwt_malloc(0, &W_dev, (size_t)rows * (size_t)N * 2 * es));
  CHECK(cwt_memcpy_h2d(plan, x_dev, x, (size_t)N * es));
  CHECK(cwt_transform(plan, x_dev, N, CWT_MORLET, f0, dt, scales, rows, xhat_dev, W_dev, N, N));
  CHECK(cwt_plan_sync(plan));

  void* W = malloc((size_t)rows * (size_t)N * 2 * es);
  CHECK(cwt_memcpy_d2h(plan, W, W_dev, (size_t)rows * (size_t)N * 2 * es));

  /* error of a row relative to the largest row of the transform: the fp32 input is the ROUNDED cosine, whose rounding noise
   * (white, 6e-8) passes every filter, so a row with a weak response to w_m is not meaningfully compared with its own size */
  double worst_abs = 0.0, largest = 0.0;
  for (int j = 0; j < rows; ++j) {
    const double s = scales[j], norm = 0.5 * sqrt(2.0 * pi * s / dt);
    const double gp = pow(pi, -0.25) * exp(-0.5 * (s * wm - f0) * (s * wm - f0));
    const double gm = pow(pi, -0.25) * exp(-0.5 * (-s * wm - f0) * (-s * wm - f0));
    for (int64_t n = 0; n < N; ++n) {
      const double c = cos(wm * (double)n * dt), sn = sin(wm * (double)n * dt);
      const double re = norm * (gp + gm) * c, im = norm * (gp - gm) * sn;
      double gr, gi;
      if (prec == 64) { gr = ((double*)W)[2 * ((size_t)j * N + n)]; gi = ((double*)W)[2 * ((size_t)j * N + n) + 1]; }
      else { gr = ((float*)W)[2 * ((size_t)j * N + n)]; gi = ((float*)W)[2 * ((size_t)j * N + n) + 1]; }
      const double d = hypot(gr - re, gi - im), a = hypot(re, im);
      if (d > worst_abs) worst_abs = d;
      if (a > largest) largest = a;
    }
  }
  const double worst = worst_abs / largest;
  const double tol = prec == 64 ? 1e-8 : 1e-4;   /* the library's default accuracy targets are 1e-9 / 3e-5 */
  printf("largest error against the closed form, relative to the largest coefficient: %.3e (tolerance %.0e)\n", worst, tol);

  CHECK(cwt_free(0, W_dev));
  CHECK(cwt_free(0, xhat_dev));
  CHECK(cwt_free(0, x_dev));
  CHECK(cwt_plan_destroy(plan));
  free(W);
  free(x);
  if (!(worst < tol)) { fprintf(stderr, "FAILED\n"); return 1; }
  printf("OK\n");
  return 0;
}
