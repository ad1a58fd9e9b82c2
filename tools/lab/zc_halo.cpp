#include "plan.hpp"
namespace cwtd { int aols_halo_zc(int m, double aN, double c, double w, double eps, int hmax, double* amp); }
using namespace cwtd;
int main() {
  for (double s : {2.5, 5.0, 10.0, 20.0, 30.0, 40.0, 60.0, 80.0, 100.0, 128.0, 160.0}) {
    double amp = 0;
    int h = aols_halo_zc(4, 2 * 3.14159265358979 * s, 4.0, 4.0 / 6, 1e-10, 2048, &amp);
    printf("s %.1f halo %d (%.1f s) amp %.1f\n", s, h, h / s, amp);
  }
}
