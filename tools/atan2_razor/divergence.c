/* How often can the device's autoturn heading differ from the reference's?  (tools/, diagnostic; not on the product path.)
 *
 * sf_atan2 (sf_kernels.hip) is correctly rounded within 1e-9 degrees of an integer degree; glibc 2.35's atan2 is not
 * always (0.50x-ulp errors), so on such an argument the two can differ in the last bit, and when the bit decides
 * ceil(bearing in degrees) the autoturn heading (SRC/game.cpp:318-319) -- or the fortress sector (:205) -- differs by one
 * step.  This program PLAYS the game (the plain-C restatement of the engine, oracle/sf_oracle.c, which calls the
 * host's libm like the reference does) under a play pattern and counts, over all env-steps,
 *   razor     heading / sector arguments within 1e-9 degrees of an integer degree (where the last bit matters),
 *   bits      of those: glibc's atan2 and the correctly rounded value differ,
 *   heading   of those: ceil() of the two differs -- a step at which device and reference would part ways.
 *
 *   gcc -std=gnu99 -O2 -ffp-contract=off -fno-builtin-sin -fno-builtin-cos -I oracle -I spacefortress_amd/csrc \
 *       -o /tmp/divergence tools/atan2_razor/divergence.c oracle/sf_oracle.c -lm
 *   /tmp/divergence autoturn charger 2048 20000 [seed]
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sf_oracle.h"
#define __device__
#include "sf_deg_dd.h"

static uint64_t s_ = 88172645463325252ull;
static inline uint64_t rnd(void) { s_ ^= s_ << 13; s_ ^= s_ >> 7; s_ ^= s_ << 17; return s_; }

/* the kernel's correctly rounded form: theta = phi_k + N / D (tools/atan2_razor/validate.c) */
static double razor(double y, double x, int k) {
  const double ay = fabs(y);
  const double ph = kDegDD[k][0], pl = kDegDD[k][1], ch = kDegDD[k][2], cl = kDegDD[k][3], sh = kDegDD[k][4], sl = kDegDD[k][5];
  double p1 = ay * ch, e1 = fma(ay, ch, -p1);
  double p2 = x * sh, e2 = fma(x, sh, -p2);
  double d = p1 - p2;
  double bb = d - p1, err = (p1 - (d - bb)) + (-p2 - bb);
  double lo = err + (e1 - e2) + (ay * cl - x * sl);
  double N = d + lo, D = x * ch + ay * sh;
  return copysign(ph + (pl + N / D), y);
}

static long n_arg, n_razor, n_bits, n_flip;
/* one bearing argument (y, x) whose rad2deg is ceil()-ed to a multiple of `step` degrees */
static void probe(double y, double x, double step) {
  n_arg++;
  if (y == 0.0) return;
  const double g = atan2(y, x);
  const double deg = fabs(g) / M_PI * 180.0;
  if (fabs(deg - rint(deg)) >= 1e-9) return;
  const int k = (int)rint(deg);
  if (k < 0 || k > 180) return;
  n_razor++;
  const double r = razor(y, x, k);
  if (memcmp(&g, &r, 8) == 0) return;
  n_bits++;
  double a = g, b = r;
  if (a < 0) a += 2 * M_PI;
  if (b < 0) b += 2 * M_PI;
  if (ceil(a / M_PI * 180 / step) != ceil(b / M_PI * 180 / step)) n_flip++;
}

int main(int argc, char** argv) {
  const char* gametype = argc > 1 ? argv[1] : "autoturn";
  const char* pattern = argc > 2 ? argv[2] : "charger";
  const int n = argc > 3 ? atoi(argv[3]) : 1024;
  const long T = argc > 4 ? atol(argv[4]) : 20000;
  const unsigned seed = argc > 5 ? (unsigned)atol(argv[5]) : 0u; /* another stretch of the spawn stream and of the actions */
  const int autoturn = strstr(gametype, "autoturn") != NULL;
  s_ += 0x9E3779B97F4A7C15ull * seed;
  long steps = 0;
  for (int e = 0; e < n; e++) {
    sfo_env* env = sfo_env_new(gametype, 1, 0, 1, 1 + 3 * e + 7919 * (int)seed);
    if (!env) return 2;
    const int na = sfo_env_n_actions(env);
    double obs[32];
    sfo_env_reset(env, obs);
    sfo_snapshot sn;
    for (long t = 0; t < T; t++) {
      sfo_env_snapshot(env, &sn);
      if (sn.ship_alive) {
        /* the arguments of this tick's two ceil()-ed bearings.  Heading: angleTo(ship, fortress) BEFORE the move
         * (SRC/game.cpp:318); sector: the bearing of the ship AFTER it (:195) -- probed on the pre-move position of
         * the next tick, which is the same point */
        if (autoturn) probe(315.0 - sn.ship_y, 355.0 - sn.ship_x, 1.0);
        probe(sn.ship_y - 315.0, sn.ship_x - 355.0, 10.0);
      }
      int a = (int)(rnd() % (unsigned)na);
      if (!strcmp(pattern, "charger") && (rnd() & 1)) a = 2; /* THRUST half of the time: tools/soak.py `charger` */
      int r, d, i;
      sfo_env_step(env, a, obs, &r, &d, &i);
      if (d) sfo_env_reset(env, obs);
      steps++;
    }
    sfo_env_free(env);
  }
  printf("%s %s: %ld env-steps, %ld bearing arguments, %ld within 1e-9 deg of an integer degree (%.3g per env-step), "
         "%ld of them where glibc's atan2 is not the correctly rounded value (%.3g %%), %ld where ceil() differs "
         "(%.3g per env-step)\n",
         gametype, pattern, steps, n_arg, n_razor, (double)n_razor / steps, n_bits, n_razor ? 100.0 * n_bits / n_razor : 0.0,
         n_flip, (double)n_flip / steps);
  return 0;
}
