// sf_rollout_ops.hip -- the trainer-side arithmetic that sits between two env steps, on the device
// (SURVEY 8f rank 3), so that policy forward -> sf_step -> bookkeeping -> storage never leaves HBM:
//
//   sf_record_kernel   rl/train.py:82-88: reward -> float, masks = 1 - done, episode_rewards /
//                      final_rewards bookkeeping; one lane per env, one launch per step (the reference
//                      does this with five host-side tensor ops per step after a .cpu() round trip)
//   sf_returns_kernel  RolloutStorage.compute_returns, rl/storage.py:50-63: the backward scan over the
//                      T steps of a rollout (GAE or plain discounted returns); one lane per env walks
//                      its column of the [T+1][N] arrays -- coalesced across envs, one launch instead of
//                      5*T small ones.
// float32 throughout with the reference's operation order (torch evaluates `gamma * V * mask` left to
// right in float32; `gamma * tau` is a Python double product rounded to float32 once), compiled with
// -ffp-contract=off: results are bit-identical to the reference's (tests/golden/trainer/*.npz).
#include <hip/hip_runtime.h>

#include "sf_internal.h"

namespace {

template <typename R>
__global__ __launch_bounds__(256) void sf_record_kernel(int n, const R* reward, const uint8_t* done, float* reward_out,
                                                        float* mask_out, float* episode_rewards, float* final_rewards,
                                                        const void* actions, int act_type, int64_t* actions_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (actions_out)  // rollouts.actions[step].copy_(action): a LongTensor in the reference (rl/storage.py:22-23)
    actions_out[i] = act_type == 1 ? (int64_t) reinterpret_cast<const uint8_t*>(actions)[i]
                   : act_type == 4 ? (int64_t) reinterpret_cast<const int32_t*>(actions)[i]
                                   : reinterpret_cast<const int64_t*>(actions)[i];
  const float r = (float)reward[i];
  const float mask = done[i] ? 0.0f : 1.0f;
  if (reward_out) reward_out[i] = r;
  if (mask_out) mask_out[i] = mask;
  if (episode_rewards) {
    float ep = episode_rewards[i] + r;                     // episode_rewards += reward
    if (final_rewards) {
      float fin = final_rewards[i] * mask;                 // final_rewards *= masks
      fin = fin + (1.0f - mask) * ep;                      // final_rewards += (1 - masks) * episode_rewards
      final_rewards[i] = fin;
    }
    episode_rewards[i] = ep * mask;                        // episode_rewards *= masks
  }
}

__global__ __launch_bounds__(256) void sf_returns_kernel(int T, int n, const float* rewards, float* value_preds,
                                                         const float* masks, const float* next_value, float* returns,
                                                         int use_gae, float gamma, float gamma_tau) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const size_t N = (size_t)n;
  if (use_gae) {
    float v_next = next_value[i];
    value_preds[(size_t)T * N + i] = v_next;  // self.value_preds[-1] = next_value
    float gae = 0.0f;
    bool first = true;
    for (int t = T - 1; t >= 0; t--) {
      const float m = masks[(size_t)(t + 1) * N + i], v = value_preds[(size_t)t * N + i];
      // delta = rewards[t] + gamma * value_preds[t+1] * masks[t+1] - value_preds[t]
      const float delta = (rewards[(size_t)t * N + i] + (gamma * v_next) * m) - v;
      // gae = delta + gamma * tau * masks[t+1] * gae     (gae starts as the Python int 0)
      gae = first ? delta + (gamma_tau * m) * 0.0f : delta + (gamma_tau * m) * gae;
      first = false;
      returns[(size_t)t * N + i] = gae + v;
      v_next = v;
    }
  } else {
    float ret = next_value[i];
    returns[(size_t)T * N + i] = ret;  // self.returns[-1] = next_value
    for (int t = T - 1; t >= 0; t--) {
      // returns[t] = returns[t+1] * gamma * masks[t+1] + rewards[t]
      ret = ((ret * gamma) * masks[(size_t)(t + 1) * N + i]) + rewards[(size_t)t * N + i];
      returns[(size_t)t * N + i] = ret;
    }
  }
}

}  // namespace

extern "C" int sf_record_step(int n, const int32_t* reward_dev, const uint8_t* done_dev, float* reward_out, float* mask_out,
                              float* episode_rewards, float* final_rewards, const void* actions_dev, int act_type,
                              int64_t* actions_out, void* stream) {
  if (n <= 0 || !reward_dev || !done_dev) {
    sf_set_error("sf_record_step: need n > 0, reward_dev and done_dev");
    return SF_ERR_ARG;
  }
  if (actions_out && (!actions_dev || (act_type != SF_ACT_U8 && act_type != SF_ACT_I32 && act_type != SF_ACT_I64))) {
    sf_set_error("sf_record_step: actions_out needs actions_dev and act_type 1, 4 or 8");
    return SF_ERR_ARG;
  }
  hipLaunchKernelGGL(sf_record_kernel<int32_t>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, reward_dev,
                     done_dev, reward_out, mask_out, episode_rewards, final_rewards, actions_dev, act_type, actions_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    sf_set_error("sf_record_step: %s", hipGetErrorString(e));
    return SF_ERR_HIP;
  }
  return SF_OK;
}

// the same bookkeeping on rewards that are already float: what the trainer sees behind VecNormalize
extern "C" int sf_record_step_f32(int n, const float* reward_dev, const uint8_t* done_dev, float* reward_out, float* mask_out,
                                  float* episode_rewards, float* final_rewards, const void* actions_dev, int act_type,
                                  int64_t* actions_out, void* stream) {
  if (n <= 0 || !reward_dev || !done_dev) {
    sf_set_error("sf_record_step_f32: need n > 0, reward_dev and done_dev");
    return SF_ERR_ARG;
  }
  if (actions_out && (!actions_dev || (act_type != SF_ACT_U8 && act_type != SF_ACT_I32 && act_type != SF_ACT_I64))) {
    sf_set_error("sf_record_step_f32: actions_out needs actions_dev and act_type 1, 4 or 8");
    return SF_ERR_ARG;
  }
  hipLaunchKernelGGL(sf_record_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, reward_dev, done_dev,
                     reward_out, mask_out, episode_rewards, final_rewards, actions_dev, act_type, actions_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    sf_set_error("sf_record_step_f32: %s", hipGetErrorString(e));
    return SF_ERR_HIP;
  }
  return SF_OK;
}

extern "C" int sf_compute_returns(int num_steps, int n, const float* rewards, float* value_preds, const float* masks,
                                  const float* next_value, float* returns, int use_gae, double gamma, double tau,
                                  void* stream) {
  if (num_steps <= 0 || n <= 0 || !rewards || !masks || !next_value || !returns || (use_gae && !value_preds)) {
    sf_set_error("sf_compute_returns: bad argument");
    return SF_ERR_ARG;
  }
  // torch turns the Python scalars into float32 when they meet a float tensor: gamma, and the double
  // product gamma * tau (rl/storage.py:56)
  hipLaunchKernelGGL(sf_returns_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, num_steps, n, rewards,
                     value_preds, masks, next_value, returns, use_gae, (float)gamma, (float)(gamma * tau));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    sf_set_error("sf_compute_returns: %s", hipGetErrorString(e));
    return SF_ERR_HIP;
  }
  return SF_OK;
}
