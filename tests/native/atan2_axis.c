/* Host check of the near-axis form of atan2 the step kernel uses (sf_kernels.hip: sf_atan2): next to the y axis
 * +-pi/2 - x/y, next to the negative x axis +-pi + y/x, pi in two doubles, one rounding -- must be glibc's atan2 bit
 * for bit (the reference engine is glibc's; the device libm differs from it in the last bit now and then, and on
 * these two axes that bit decides the fortress sector).  usage: atan2_axis [samples]; exit status 1 on a mismatch. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t s = 88172645463325252ull;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline double u01(void) { return (rnd() >> 11) * (1.0 / 9007199254740992.0); }

static double axis(double y, double x) {
  const double ax = fabs(x), ay = fabs(y);
  if (ax * 0x1p27 < ay) {
    const double t = x / y;
    return copysign(1.5707963267948966, y) + (copysign(6.123233995736766e-17, y) - t);
  }
  if (x < 0 && ay * 0x1p27 < ax) {
    const double t = y / x;
    return copysign(3.141592653589793, y) + (copysign(1.2246467991473532e-16, y) + t);
  }
  return NAN;
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 4000000L;
  long bad = 0, tested = 0;
  for (long i = 0; i < n; i++) {
    const double big = (1.0 + u01() * 700.0) * ((rnd() & 1) ? 1 : -1);
    const int e = -(int)(rnd() % 30) - 28; /* 2^-28 .. 2^-57 of the other coordinate */
    double small = (i & 1) ? ldexp(u01() + 0.5, e) * fabs(big) * ((rnd() & 2) ? 1 : -1)
                           : ldexp((double)(rnd() % 4096), e - 3) * fabs(big) / 700.0 * ((rnd() & 1) ? 1 : -1); /* incl. 0 */
    double y, x;
    if (i & 2) { y = big; x = small; } else { x = -fabs(big); y = small; }
    const double r = atan2(y, x), p = axis(y, x);
    if (isnan(p)) continue;
    tested++;
    if (memcmp(&r, &p, 8) != 0) {
      if (bad < 8) printf("y=%a x=%a glibc=%a axis=%a\n", y, x, r, p);
      bad++;
    }
  }
  printf("tested %ld mismatches %ld\n", tested, bad);
  return bad != 0 || tested < n / 2;
}
