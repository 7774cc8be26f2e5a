/* Host check of the three-operation constant division the step kernel uses (sf_kernels.hip: sf_div_const):
 * for every divisor on the kernel's path, q' = fma(fma(-c, RN(a*rc), a), rc, RN(a*rc)) with rc = RN(1/c) must be
 * the IEEE quotient a / c, bit for bit.  Operands: the value ranges the kernel feeds it, and random significands over
 * 120 binades.  Prints the number of mismatches per divisor; exit status 1 if any.  usage: div_const [samples] */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t s = 88172645463325252ull;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline double divc(double a, double c, double rc) {
  double q = a * rc, rem = fma(-c, q, a);
  return fma(rem, rc, q);
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 4000000L;
  const double cs[] = {M_PI, 10.0, 20.0, 80.0, 90.0, 92.0, 180.0, 360.0, 5294.0};
  const double lo[] = {-6.4, -400, 0, -100, -100, -100, -700, -400, -6000};
  const double hi[] = {6.4, 400, 21, 400, 800, 800, 700, 400, 6000};
  long total = 0;
  for (unsigned k = 0; k < sizeof(cs) / sizeof(cs[0]); k++) {
    volatile double one = 1.0;
    const double c = cs[k], rc = one / c;
    long bad = 0;
    for (long i = 0; i < n; i++) {
      double a;
      if (i & 1) {
        a = lo[k] + (hi[k] - lo[k]) * ((rnd() >> 11) * (1.0 / 9007199254740992.0));
      } else {
        uint64_t b = rnd();
        b = (b & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(1023 - 60 + (rnd() % 120)) << 52);
        memcpy(&a, &b, 8);
      }
      volatile double va = a;
      const double ref = va / c, got = divc(a, c, rc);
      if (memcmp(&ref, &got, 8) != 0) bad++;
    }
    printf("c=%.17g mismatches=%ld of %ld\n", c, bad, n);
    total += bad;
  }
  return total != 0;
}
