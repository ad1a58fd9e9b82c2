// times build_row_table (host classification) for BASELINE config 2 / 3 grids -- no GPU needed
#include "plan.hpp"
#include <chrono>
using namespace cwtd;
int main(int argc, char** argv) {
  struct Cfg { const char* name; int mother; double param; int prec; double tol; int logn = 20; int rows = 256; long spec_ld = 0; } cfgs[] = {
    {"c2", 0, 6.0, 64, 1e-9}, {"c3_paul", 1, 4.0, 32, 3e-5}, {"c3_dog", 2, 2.0, 32, 3e-5}, {"paul64", 1, 4.0, 64, 1e-9}, {"c2_roundoff", 0, 6.0, 64, 1e-16},
    {"mc_2^23_77", 0, 6.0, 64, 3.16e-10, 23, 77}, {"mc_2^23_77b", 0, 6.0, 64, 1e-9, 23, 77},
    {"smooth_2^23", 2, 0.0, 64, 3.16e-10, 23, 77, 1L << 23}, {"smooth_2^23b", 2, 0.0, 64, 1e-9, 23, 77, 1L << 23}};
  for (auto& c : cfgs) {
    cwt_plan* p = new cwt_plan();
    p->N = 1 << c.logn; p->logN = c.logn; p->prec = c.prec; p->max_rows = 256; p->log_wg_points = c.prec == 64 ? 13 : 14;
    p->narrow_terms = c.prec == 64 ? 4 : 8; p->narrow_mix = c.prec == 64; p->ols_big = c.prec == 32; p->tolerance = c.tol;
    const int rows = c.rows;
    const double pi = 3.14159265358979323846;
    const double fl = c.mother == 0 ? 4 * pi / (c.param + std::sqrt(2 + c.param * c.param)) : c.mother == 1 ? 4 * pi / (2 * c.param + 1) : 2 * pi / std::sqrt(c.param + 0.5);
    const double s0 = 2.0 / fl, dj = std::log2(double(p->N) / s0) / (rows - 1);
    std::vector<double> a(rows), ar(rows), ai(rows);
    double cre, cim; mother_constant(c.mother, c.param, &cre, &cim);
    const double w1 = 2 * pi / double(p->N);
    double best = 1e9;
    for (int rep = 0; rep < (argc > 1 ? atoi(argv[1]) : 5); ++rep) {
      for (int j = 0; j < rows; ++j) { const double s = s0 * std::pow(2.0, j * dj) * (1 + rep * 1e-13); a[j] = s * w1; const double n = std::sqrt(s * w1 * p->N); ar[j] = n * cre; ai[j] = n * cim; }
      p->rt = &p->slots[rep & 1];
      auto t0 = std::chrono::steady_clock::now();
      int rc = build_row_table(p, c.mother, c.param, a.data(), ar.data(), ai.data(), c.spec_ld, rows, nullptr, nullptr, 0, -1, p->N, p->N);
      double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (rc) { printf("rc %d %s\n", rc, g_err.c_str()); return 1; }
      best = std::min(best, ms);
      if (rep == 0) printf("   first call %.1f ms\n", ms);
    }
    set_split(p);
    printf("%-12s build_row_table %.3f ms  (poly %d ols %d aols %d wide %d) planes %.1f MB in %zu chunks, bands %.1f MB\n", c.name, best, p->rt->n_poly, p->rt->n_ols, p->rt->n_aols, p->rt->n_wide, p->rt->poly_coef_elems * (c.prec == 64 ? 16.0 : 8.0) / 1e6, p->rt->poly_chunks.size(), p->rt->poly_band_elems * (c.prec == 64 ? 16.0 : 8.0) / 1e6);
  }
  return 0;
}
