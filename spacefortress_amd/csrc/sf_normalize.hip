// sf_normalize.hip -- VecNormalize on the device.
//
// The trainer wraps the vec-env into gym_vecenv.VecNormalize whenever the observation is 1-D
// (rl/train.py:35-36).  gym-vecenv==1.0 (requirements.txt:4) is a third-party package that is not in
// /root/reference; it is OpenAI baselines' vec_normalize.py + running_mean_std.py of early 2018, whose
// published algorithm is, per step:
//     ret  = ret * gamma + rews
//     ob_rms.update(obs);  obs  = clip((obs - ob_rms.mean) / sqrt(ob_rms.var + eps), -clipob, clipob)
//     ret_rms.update(ret); rews = clip(rews / sqrt(ret_rms.var + eps), -cliprew, cliprew)
// with RunningMeanStd.update(x) = the parallel-variance merge of (mean, var, count) with the batch's
// (mean over envs, population variance over envs, n_envs); mean 0, var 1, count 1e-4 initially.
// Parity is pinned only to the numpy restatement of that algorithm (oracle/vecnorm_np.py).
//
// Two launches per step, both tiny and HBM-streaming:
//   sf_norm_reduce_kernel  a wave stages 64 rows of observations through LDS (coalesced reads), lane f
//                          sums feature f and its square in float64; lane-per-env return update and
//                          its two sums; one float64 atomic per feature and wave;
//   sf_norm_apply_kernel   every workgroup merges the running statistics with the batch sums (19 values:
//                          cheaper than a third launch), then normalises and clips its rows; workgroup
//                          0 stores the merged statistics.  Statistics and sums are double-buffered by
//                          step parity, so nothing is read while it is written.
#include <hip/hip_runtime.h>

#include "sf_internal.h"

namespace {

constexpr int kMaxDim = 24;  // 19 / 17 / 10 on this path; bounds the LDS staging buffer

template <typename T>
__global__ __launch_bounds__(256) void sf_norm_reduce_kernel(const T* obs, const int32_t* rew, double* ret, int n, int dim,
                                                             double gamma, int do_ob, int do_ret, double* sums) {
  __shared__ T stage[4][64 * kMaxDim];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long n_chunks = ((long)n + 63) / 64;
  for (long chunk = (long)blockIdx.x * 4 + wave; chunk < n_chunks; chunk += (long)gridDim.x * 4) {
    const long row0 = chunk * 64;
    const int rows = (int)min((long)64, (long)n - row0);
    if (do_ob) {
      const T* src = obs + row0 * dim;
      for (int i = lane; i < rows * dim; i += 64) stage[wave][i] = src[i];
      __builtin_amdgcn_wave_barrier();
      if (lane < dim) {
        double s = 0, q = 0;
        for (int r = 0; r < rows; r++) {
          const double v = (double)stage[wave][r * dim + lane];
          s += v;
          q += v * v;
        }
        atomicAdd(&sums[lane], s);
        atomicAdd(&sums[dim + 1 + lane], q);
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (do_ret && rew) {
      double v = 0;
      if (lane < rows) {
        v = ret[row0 + lane] * gamma + (double)rew[row0 + lane];
        ret[row0 + lane] = v;
      }
      double s = v, q = v * v;
      for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        q += __shfl_xor(q, o);
      }
      if (lane == 0) {
        atomicAdd(&sums[dim], s);
        atomicAdd(&sums[2 * dim + 1], q);
      }
    }
  }
}

// RunningMeanStd.update with batch (sum, sumsq, n): returns merged mean/var/count
__device__ __forceinline__ void merge(double mean, double var, double count, double sum, double sumsq, double n,
                                      double* nmean, double* nvar, double* ncount) {
  const double bmean = sum / n;
  double bvar = sumsq / n - bmean * bmean;  // population variance of the batch (np.var)
  bvar = bvar < 0 ? 0 : bvar;
  const double delta = bmean - mean, tot = count + n;
  *nmean = mean + delta * n / tot;
  const double m2 = var * count + bvar * n + delta * delta * count * n / tot;
  *nvar = m2 / tot;
  *ncount = tot;
}

template <typename T>
__global__ __launch_bounds__(256) void sf_norm_apply_kernel(const T* obs, T* obs_out, const int32_t* rew, float* rew_out,
                                                            int n, int dim, double eps, double clipob, double cliprew,
                                                            int do_ob, int do_ret, const double* sums, double* sums_next,
                                                            const double* stats, double* stats_next) {
  __shared__ double s_mean[kMaxDim + 1], s_inv[kMaxDim + 1];
  const int t = threadIdx.x;
  // stats: [0,D) mean, [D,2D) var, [2D] ret mean, [2D+1] ret var, [2D+2] ob count, [2D+3] ret count
  if (t <= dim) {
    const bool is_ret = t == dim;
    const bool upd = is_ret ? (do_ret && rew) : do_ob;
    double mean = is_ret ? stats[2 * dim] : stats[t], var = is_ret ? stats[2 * dim + 1] : stats[dim + t];
    double count = stats[2 * dim + 2 + (is_ret ? 1 : 0)];
    if (upd) merge(mean, var, count, sums[t], sums[dim + 1 + t], (double)n, &mean, &var, &count);
    s_mean[t] = mean;
    s_inv[t] = 1.0 / sqrt(var + eps);
    if (blockIdx.x == 0) {
      if (is_ret) {
        stats_next[2 * dim] = mean;
        stats_next[2 * dim + 1] = var;
        stats_next[2 * dim + 3] = count;
      } else {
        stats_next[t] = mean;
        stats_next[dim + t] = var;
        if (t == 0) stats_next[2 * dim + 2] = count;
      }
      // the other parity's sums are idle now: clear them for the next step's reduction
      sums_next[t] = 0;
      sums_next[dim + 1 + t] = 0;
    }
  }
  __syncthreads();
  const long total = (long)n * dim;
  if (obs_out)
    for (long i = (long)blockIdx.x * 256 + t; i < total; i += (long)gridDim.x * 256) {
      const int f = (int)(i % dim);
      double v = ((double)obs[i] - s_mean[f]) * s_inv[f];
      v = v < -clipob ? -clipob : (v > clipob ? clipob : v);
      obs_out[i] = (T)v;
    }
  if (rew && rew_out)
    for (long i = (long)blockIdx.x * 256 + t; i < n; i += (long)gridDim.x * 256) {
      double v = (double)rew[i] * s_inv[dim];  // rews / sqrt(ret_rms.var + eps): no mean subtraction
      v = v < -cliprew ? -cliprew : (v > cliprew ? cliprew : v);
      rew_out[i] = (float)v;
    }
}

}  // namespace

hipError_t sf_launch_normalize(const void* obs, void* obs_out, int obs_f64, const int32_t* rew, float* rew_out, double* ret,
                               int n, int dim, double gamma, double eps, double clipob, double cliprew, int do_ob,
                               int do_ret, double* sums, double* sums_next, const double* stats, double* stats_next,
                               hipStream_t stream) {
  if (n <= 0 || dim <= 0 || dim > kMaxDim) return hipErrorInvalidValue;
  const int chunks = (n + 63) / 64;
  const int g1 = min((chunks + 3) / 4, 1024), g2 = (int)min(((long)n * dim + 255) / 256, (long)2048);
  if (obs_f64) {
    hipLaunchKernelGGL(sf_norm_reduce_kernel<double>, dim3(g1), dim3(256), 0, stream, (const double*)obs, rew, ret, n, dim,
                       gamma, do_ob, do_ret, sums);
    hipLaunchKernelGGL(sf_norm_apply_kernel<double>, dim3(g2), dim3(256), 0, stream, (const double*)obs, (double*)obs_out, rew,
                       rew_out, n, dim, eps, clipob, cliprew, do_ob, do_ret, sums, sums_next, stats, stats_next);
  } else {
    hipLaunchKernelGGL(sf_norm_reduce_kernel<float>, dim3(g1), dim3(256), 0, stream, (const float*)obs, rew, ret, n, dim,
                       gamma, do_ob, do_ret, sums);
    hipLaunchKernelGGL(sf_norm_apply_kernel<float>, dim3(g2), dim3(256), 0, stream, (const float*)obs, (float*)obs_out, rew,
                       rew_out, n, dim, eps, clipob, cliprew, do_ob, do_ret, sums, sums_next, stats, stats_next);
  }
  return hipGetLastError();
}
