// test_mixed_dft.cpp -- the in-register DFTs of scanner_amd/csrc/scn_mixed_dft.h (every length the mixed-radix fused kernels use,
// and every other 5-smooth length up to 32) against the DFT sum evaluated in double, on the CPU: the header is plain arithmetic.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../scanner_amd/csrc/scn_mixed_dft.h"

struct cfh {
  float x, y;
};
struct cdh {  // the double-precision instantiation (pass 3 of the sizes beyond 10000)
  double x, y;
};

static int g_fail = 0;

template <int R>
static void check() {
  double worst = 0;
  for (int trial = 0; trial < 4; trial++) {
    cfh v[R];
    double xr[R], xi[R];
    for (int n = 0; n < R; n++) {
      xr[n] = trial == 0 ? (n == 1) : (double)rand() / RAND_MAX - 0.5;   // trial 0: a unit impulse at n = 1 -> X[k] = W_R^k exactly
      xi[n] = trial == 0 ? 0.0 : (double)rand() / RAND_MAX - 0.5;
      v[n] = cfh{(float)xr[n], (float)xi[n]};
      xr[n] = v[n].x;
      xi[n] = v[n].y;
    }
    scn_dft<R>(v);
    double scale = 0;
    for (int n = 0; n < R; n++) scale += std::sqrt(xr[n] * xr[n] + xi[n] * xi[n]);
    for (int k = 0; k < R; k++) {
      double sr = 0, si = 0;
      for (int n = 0; n < R; n++) {
        const double a = -2.0 * M_PI * ((n * k) % R) / R;
        sr += xr[n] * std::cos(a) - xi[n] * std::sin(a);
        si += xr[n] * std::sin(a) + xi[n] * std::cos(a);
      }
      const double e = std::hypot(v[k].x - sr, v[k].y - si) / scale;
      if (e > worst) worst = e;
    }
  }
  printf("dft<%2d>: max error %.2e of the input's l1 norm\n", R, worst);
  if (!(worst < 4e-7)) {
    printf("  FAILED\n");
    g_fail++;
  }
}

template <int R>
static void check_double() {
  cdh v[R];
  double xr[R], xi[R], worst = 0, scale = 0;
  for (int n = 0; n < R; n++) {
    xr[n] = (double)rand() / RAND_MAX - 0.5;
    xi[n] = (double)rand() / RAND_MAX - 0.5;
    v[n] = cdh{xr[n], xi[n]};
    scale += std::hypot(xr[n], xi[n]);
  }
  scn_dft<R>(v);
  for (int k = 0; k < R; k++) {
    long double sr = 0, si = 0;
    for (int n = 0; n < R; n++) {
      const long double a = -2.0L * 3.14159265358979323846264338327950288L * ((n * k) % R) / R;
      sr += xr[n] * cosl(a) - xi[n] * sinl(a);
      si += xr[n] * sinl(a) + xi[n] * cosl(a);
    }
    const double e = std::hypot(v[k].x - (double)sr, v[k].y - (double)si) / scale;
    if (e > worst) worst = e;
  }
  printf("dft<%2d> in double: max error %.2e of the input's l1 norm\n", R, worst);
  if (!(worst < 1e-15)) {
    printf("  FAILED\n");
    g_fail++;
  }
}

int main() {
  check_double<16>(); check_double<20>(); check_double<24>(); check_double<25>(); check_double<32>();
  check<2>(); check<3>(); check<4>(); check<5>(); check<6>(); check<8>(); check<9>(); check<10>(); check<12>(); check<15>();
  check<16>(); check<18>(); check<20>(); check<24>(); check<25>(); check<27>(); check<30>(); check<32>();
  if (g_fail) return 1;
  printf("mixed dft tests ok\n");
  return 0;
}
