// sf_norm_capi.cpp -- C ABI of the device-side VecNormalize (include/sfmi.h, sf_normalize.hip).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "sf_internal.h"

struct sf_normalizer {
  sf_normalizer_params p;
  int parity;
  double* d_ret;       // per-env discounted return (VecNormalize.ret)
  double* d_partials;  // SF_NORM_GROUPS x 2 * (D + 1), then the ticket counter
  double* d_stats[2];  // 2 * D + 4
};

namespace {
#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      sf_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return SF_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)

struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    if (changed) (void)hipSetDevice(prev);
  }
};

int n_stats(const sf_normalizer* z) { return 2 * z->p.obs_dim + 4; }
int n_sums(const sf_normalizer* z) { return 2 * (z->p.obs_dim + 1); }
}  // namespace

extern "C" int sf_normalizer_create(const sf_normalizer_params* p, sf_normalizer** out) {
  if (!p || !out || p->n_envs <= 0 || p->obs_dim <= 0 || p->obs_dim > 24) {
    sf_set_error("sf_normalizer_create: need n_envs > 0 and 0 < obs_dim <= 24");
    return SF_ERR_ARG;
  }
  *out = nullptr;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
    sf_set_error("sf_normalizer_create: no HIP device available; libsfmi has no CPU path");
    return SF_ERR_NO_DEVICE;
  }
  if (p->device_id < 0 || p->device_id >= n_dev) {
    sf_set_error("sf_normalizer_create: device_id %d out of range", p->device_id);
    return SF_ERR_ARG;
  }
  DeviceGuard guard(p->device_id);
  sf_normalizer* z = new sf_normalizer();
  memset(z, 0, sizeof(*z));
  z->p = *p;
  // RunningMeanStd.__init__: mean 0, var 1, count 1e-4
  std::vector<double> st(n_stats(z), 0.0);
  for (int f = 0; f < p->obs_dim; f++) st[p->obs_dim + f] = 1.0;
  st[2 * p->obs_dim + 1] = 1.0;
  st[2 * p->obs_dim + 2] = st[2 * p->obs_dim + 3] = 1e-4;
#define TRY_FREE(expr)                                                                 \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      sf_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      sf_normalizer_destroy(z);                                                        \
      return SF_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)
  TRY_FREE(hipMalloc((void**)&z->d_ret, sizeof(double) * p->n_envs));
  TRY_FREE(hipMemset(z->d_ret, 0, sizeof(double) * p->n_envs));
  // rows of partial sums: SF_NORM_GROUPS from the stand-alone reduction, or one per wave of the step kernel
  // (sf_step_normalize; the batch is padded to whole workgroups of 256 envs)
  const size_t rows = (size_t)((p->n_envs + 255) / 256) * 4;
  const size_t prow = rows > SF_NORM_GROUPS ? rows : SF_NORM_GROUPS;
  TRY_FREE(hipMalloc((void**)&z->d_partials, sizeof(double) * n_sums(z) * prow));
  TRY_FREE(hipMemset(z->d_partials, 0, sizeof(double) * n_sums(z) * prow));
  for (int k = 0; k < 2; k++) {
    TRY_FREE(hipMalloc((void**)&z->d_stats[k], sizeof(double) * n_stats(z)));
    TRY_FREE(hipMemcpy(z->d_stats[k], st.data(), sizeof(double) * n_stats(z), hipMemcpyHostToDevice));
  }
#undef TRY_FREE
  *out = z;
  return SF_OK;
}

extern "C" int sf_normalizer_destroy(sf_normalizer* z) {
  if (!z) return SF_OK;
  DeviceGuard guard(z->p.device_id);
  if (z->d_ret) (void)hipFree(z->d_ret);
  if (z->d_partials) (void)hipFree(z->d_partials);
  for (int k = 0; k < 2; k++) {
    if (z->d_stats[k]) (void)hipFree(z->d_stats[k]);
  }
  delete z;
  return SF_OK;
}

extern "C" int sf_normalize(sf_normalizer* z, const void* obs_dev, void* obs_out_dev, const int32_t* reward_dev,
                            float* reward_out_dev, int frozen, void* stream) {
  if (!z || (!obs_dev && !reward_dev)) {
    sf_set_error("sf_normalize: null normalizer or nothing to normalise");
    return SF_ERR_ARG;
  }
  if ((obs_dev == nullptr) != (obs_out_dev == nullptr) || (reward_dev == nullptr) != (reward_out_dev == nullptr)) {
    sf_set_error("sf_normalize: obs / reward inputs and outputs come in pairs");
    return SF_ERR_ARG;
  }
  DeviceGuard guard(z->p.device_id);
  const int k = z->parity;
  const int do_ob = (z->p.ob && obs_dev && !frozen) ? 1 : 0, do_ret = (z->p.ret && reward_dev && !frozen) ? 1 : 0;
  HIP_TRY(sf_launch_normalize(z->p.ob ? obs_dev : nullptr, z->p.ob ? obs_out_dev : nullptr, z->p.obs_f64,
                              z->p.ret ? reward_dev : nullptr, z->p.ret ? reward_out_dev : nullptr, z->d_ret, z->p.n_envs,
                              z->p.obs_dim, z->p.gamma, z->p.epsilon, z->p.clipob, z->p.cliprew, do_ob, do_ret,
                              z->d_partials, z->d_stats[k], z->d_stats[k ^ 1], (hipStream_t)stream));
  if (do_ob || do_ret) z->parity = k ^ 1;
  return SF_OK;
}

extern "C" int sf_normalizer_get_state(sf_normalizer* z, double* host, double* ret_host, void* stream) {
  if (!z || !host) return SF_ERR_ARG;
  DeviceGuard guard(z->p.device_id);
  HIP_TRY(hipMemcpyAsync(host, z->d_stats[z->parity], sizeof(double) * n_stats(z), hipMemcpyDeviceToHost, (hipStream_t)stream));
  if (ret_host)
    HIP_TRY(hipMemcpyAsync(ret_host, z->d_ret, sizeof(double) * z->p.n_envs, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return SF_OK;
}

extern "C" int sf_normalizer_set_state(sf_normalizer* z, const double* host, const double* ret_host, void* stream) {
  if (!z || !host) return SF_ERR_ARG;
  DeviceGuard guard(z->p.device_id);
  HIP_TRY(hipMemcpyAsync(z->d_stats[z->parity], host, sizeof(double) * n_stats(z), hipMemcpyHostToDevice, (hipStream_t)stream));
  if (ret_host)
    HIP_TRY(hipMemcpyAsync(z->d_ret, ret_host, sizeof(double) * z->p.n_envs, hipMemcpyHostToDevice, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return SF_OK;
}

extern "C" int sf_step_normalize(sf_batch* b, sf_normalizer* z, const void* actions_dev, int act_type, void* obs_dev,
                                 int32_t* reward_dev, uint8_t* done_dev, uint8_t* info_dev, float* reward_out_dev, int frozen,
                                 void* stream) {
  if (!b || !z || !obs_dev || !reward_dev || !reward_out_dev) {
    sf_set_error("sf_step_normalize: null batch, normalizer, obs, reward or reward_out");
    return SF_ERR_ARG;
  }
  if (sf_n_envs(b) != z->p.n_envs || sf_obs_dim(b) != z->p.obs_dim) {
    sf_set_error("sf_step_normalize: the normalizer was made for %d envs x %d features", z->p.n_envs, z->p.obs_dim);
    return SF_ERR_ARG;
  }
  if (frozen) {
    int rc = sf_step(b, actions_dev, act_type, obs_dev, reward_dev, done_dev, info_dev, stream);
    if (rc != SF_OK) return rc;
    return sf_normalize(z, obs_dev, obs_dev, reward_dev, reward_out_dev, 1, stream);
  }
  DeviceGuard guard(z->p.device_id);
  int rows = 0;
  int rc = sf_step_with_norm_partials(b, actions_dev, act_type, obs_dev, reward_dev, done_dev, info_dev, z->d_partials,
                                      z->p.ret ? z->d_ret : nullptr, z->p.gamma, &rows, stream);
  if (rc != SF_OK) return rc;
  const int k = z->parity;
  HIP_TRY(sf_launch_normalize_after_step(z->p.ob ? obs_dev : nullptr, z->p.ob ? obs_dev : nullptr, z->p.obs_f64,
                                         z->p.ret ? reward_dev : nullptr, z->p.ret ? reward_out_dev : nullptr, z->p.n_envs,
                                         z->p.obs_dim, z->p.epsilon, z->p.clipob, z->p.cliprew, z->p.ob, z->p.ret, z->d_partials,
                                         rows, z->d_stats[k], z->d_stats[k ^ 1], (hipStream_t)stream));
  z->parity = k ^ 1;
  return SF_OK;
}
