/* Runs the C restatement (oracle/sf_oracle.c) under -fsanitize=address,undefined (tests/test_sanitizers.py):
 * single envs of all four presets through full episodes, and a vec env with auto-reset.  SURVEY 5: the
 * reference has no sanitizer coverage; the restatement must be clean of the undefined behaviour listed there. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "sf_oracle.h"

int main(void) {
  const char* presets[4] = {"youturn", "autoturn", "test-youturn", "test-autoturn"};
  long total = 0;
  for (int p = 0; p < 4; p++)
    for (int as = 0; as <= 1; as++) {
      sfo_env* e = sfo_env_new(presets[p], as, 0, 1, p);
      if (!e) return 1;
      total += sfo_rollout(e, 12000, 1234u + (unsigned)p);
      sfo_env_free(e);
    }
  enum { N = 37 };
  sfo_vec_env* v = sfo_vec_create("youturn", N, 1, 1, 1, 0, 3);
  if (!v) return 2;
  const int dim = sfo_vec_obs_dim(v);
  double* obs = (double*)malloc(sizeof(double) * N * dim);
  int32_t act[N], rew[N];
  uint8_t done[N], info[N];
  sfo_vec_reset(v, obs);
  unsigned lcg = 7;
  for (int t = 0; t < 6000; t++) {
    for (int i = 0; i < N; i++) {
      lcg = lcg * 1664525u + 1013904223u;
      act[i] = (int32_t)((lcg >> 16) % 5u);
    }
    sfo_vec_step(v, act, obs, rew, done, info);
    for (int i = 0; i < N; i++) total += rew[i];
  }
  free(obs);
  sfo_vec_destroy(v);
  printf("ok %ld\n", total);
  return 0;
}
