/* Host restatement of sf_atan2_core (sf_kernels.hip): the table-step atan2 of the step kernel's hot path.  Same
 * operations in the same order (FMAs where the kernel has them); the hardware's v_rcp_f64 seed is modelled as the
 * reciprocal rounded to FLOAT -- a worse seed than the instruction's -- so that the two Newton steps are what carries
 * the precision.  Reports the largest distance from the host libm's atan2 in ulps over random arguments and over
 * the lattice / axis cases the game produces; the kernel's contract is "within 1e-15 rad" (what is decided by the last
 * bit goes through sf_atan2's exact forms).   gcc -O2 -ffp-contract=off atan2_core.c -lm && ./a.out 4000000 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double atab[17];
static double recip(double d) {
  double r = (double)(float)(1.0 / d);
  r = fma(fma(-d, r, 1.0), r, r);
  return fma(fma(-d, r, 1.0), r, r);
}
static double core(double y, double x) {
  const double ax = fabs(x), ay = fabs(y);
  const double u = fmax(ax, ay), v = fmin(ax, ay);
  const double ru = recip(u);
  double q = v * ru;
  q = fma(fma(-u, q, v), ru, q);
  const double k = rint(q * 16.0), c = k * 0.0625;
  const double den = fma(q, c, 1.0), num = q - c;
  const double rd = recip(den);
  double t = num * rd;
  t = fma(fma(-den, t, num), rd, t);
  const double s = t * t;
  double p = fma(s, 1.0 / 9.0, -1.0 / 7.0);
  p = fma(s, p, 0.2);
  p = fma(s, p, -1.0 / 3.0);
  double a = atab[(int)k] + fma(t, p * s, t);
  a = ay > ax ? 1.5707963267948966 - a : a;
  a = x < 0 ? 3.141592653589793 - a : a;
  return copysign(a, y);
}
static uint64_t s_ = 88172645463325252ull;
static inline uint64_t rnd(void) { s_ ^= s_ << 13; s_ ^= s_ >> 7; s_ ^= s_ << 17; return s_; }
static inline double u01(void) { return (rnd() >> 11) * (1.0 / 9007199254740992.0); }
static double ulps(double a, double b) {
  if (a == b) return 0;
  const double ulp = nextafter(fabs(b), INFINITY) - fabs(b);
  return fabs(a - b) / ulp;
}
int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 4000000;
  for (int k = 0; k <= 16; k++) atab[k] = atan(k / 16.0);
  double worst = 0;
  long bad = 0;
  for (long i = 0; i < n; i++) {
    double x, y;
    switch (rnd() % 4) {
      case 0: x = (u01() - 0.5) * 800; y = (u01() - 0.5) * 700; break;                 /* ship - fortress */
      case 1: x = (u01() - 0.5) * 12; y = (u01() - 0.5) * 12; break;                   /* velocities */
      case 2: x = (double)((int)(rnd() % 400) - 200); y = (double)((int)(rnd() % 360) - 180); break; /* spawn lattice */
      default: x = (u01() - 0.5) * 800; y = x * (double)((int)(rnd() % 5) - 2) * 0.5; break;          /* q on the table's knots */
    }
    if (x == 0 && y == 0) continue;
    const double g = atan2(y, x), c = core(y, x);
    const double e = ulps(c, g);
    if (e > worst) worst = e;
    if (fabs(c - g) > 1e-15) {
      if (bad < 10) printf("y=%a x=%a libm=%a core=%a (%.2f ulps)\n", y, x, g, c, e);
      bad++;
    }
  }
  /* exact cases: the axes and the diagonals give the libm's values bit for bit */
  const double ex[8][2] = {{0, 5}, {5, 0}, {0, -5}, {-5, 0}, {3, 3}, {3, -3}, {-3, 3}, {-3, -3}};
  int exact = 0;
  for (int i = 0; i < 8; i++) {
    const double g = atan2(ex[i][0], ex[i][1]), c = core(ex[i][0], ex[i][1]);
    exact += ulps(c, g) <= 1.0;
  }
  printf("worst=%.3f ulps, off by more than 1e-15 rad: %ld of %ld, axes/diagonals within an ulp: %d of 8\n", worst, bad, n, exact);
  return bad != 0 || exact != 8;
}
