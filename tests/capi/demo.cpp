// A caller of libsfmi.so that knows nothing of Python or PyTorch: plain HIP buffers through the C ABI of
// include/sfmi.h.  Built and run by tests/test_gpu_capi_native.py, which replays the same actions through
// the Python host side and expects the same numbers.
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "sfmi.h"

#define CHECK(x)                                                        \
  do {                                                                  \
    int rc_ = (x);                                                      \
    if (rc_ != 0) {                                                     \
      fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, sf_last_error());     \
      return 1;                                                         \
    }                                                                   \
  } while (0)

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1000, steps = argc > 2 ? atoi(argv[2]) : 300;
  sf_create_params p = {"youturn", n, 0, 1, SF_OBS_FEATURES, 0, 1, 0, 1, 0};
  sf_batch* b = nullptr;
  CHECK(sf_create(&p, &b));
  const int dim = sf_obs_dim(b), n_act = sf_n_actions(b);
  uint8_t *d_act, *d_done, *d_info;
  int32_t* d_rew;
  float* d_obs;
  if (hipMalloc((void**)&d_act, n) || hipMalloc((void**)&d_done, n) || hipMalloc((void**)&d_info, n) ||
      hipMalloc((void**)&d_rew, 4 * (size_t)n) || hipMalloc((void**)&d_obs, 4 * (size_t)n * dim))
    return 2;
  CHECK(sf_reset(b, d_obs, nullptr));
  std::vector<uint8_t> act(n);
  std::vector<int32_t> rew(n);
  std::vector<float> obs((size_t)n * dim);
  long long reward_sum = 0, kills = 0;
  uint32_t lcg = 12345u;
  for (int t = 0; t < steps; t++) {
    for (int i = 0; i < n; i++) {
      lcg = lcg * 1664525u + 1013904223u;
      act[i] = (uint8_t)((lcg >> 16) % (unsigned)n_act);
    }
    if (hipMemcpy(d_act, act.data(), n, hipMemcpyHostToDevice)) return 3;
    CHECK(sf_step(b, d_act, SF_ACT_U8, d_obs, d_rew, d_done, d_info, nullptr));
    if (hipMemcpy(rew.data(), d_rew, 4 * (size_t)n, hipMemcpyDeviceToHost)) return 3;
    for (int i = 0; i < n; i++) reward_sum += rew[i];
  }
  CHECK(sf_check_actions(b, nullptr));
  if (hipMemcpy(obs.data(), d_obs, 4 * (size_t)n * dim, hipMemcpyDeviceToHost)) return 3;
  double obs_sum = 0;
  for (size_t i = 0; i < obs.size(); i++) obs_sum += obs[i];
  std::vector<double> x(n);
  std::vector<int32_t> stats(13 * (size_t)n);
  CHECK(sf_get_field(b, sf_field_id("ship_x"), x.data(), 8 * (size_t)n));
  CHECK(sf_get_field(b, sf_field_id("stats"), stats.data(), 4 * stats.size()));
  double x_sum = 0;
  long long shots = 0;
  for (int i = 0; i < n; i++) {
    x_sum += x[i];
    shots += stats[7 * (size_t)n + i];
  }
  printf("reward_sum=%lld shots=%lld ship_x_sum=%.17g obs_sum=%.17g dim=%d n_act=%d\n", reward_sum, shots, x_sum, obs_sum, dim,
         n_act);
  CHECK(sf_destroy(b));
  return 0;
}
