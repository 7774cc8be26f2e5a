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
// Three small launches per step:
//   sf_norm_reduce_kernel  256 workgroups; a wave stages 64 rows of observations through LDS (coalesced
//                          reads), three groups of 19 lanes sum feature f and its square in float64 over
//                          every third row; lane-per-env return update and its two sums; one row of
//                          partial sums per workgroup (no atomics on the sums: 1024 waves adding into the
//                          same 40 addresses serialise in L2 -- measured 45 us);
//   sf_norm_merge_kernel   one workgroup per feature (and one for the returns) adds its two columns up, merges
//                          them into the running statistics (RunningMeanStd.update) and stores those for the
//                          other step parity;
//   sf_norm_apply_kernel   reads the 40 merged values, normalises and clips.
// Statistics are double-buffered by step parity so that nothing is read while it is written; sums are
// added in a fixed order: run-to-run deterministic.
#include <hip/hip_runtime.h>

#include "sf_internal.h"

namespace {

constexpr int kMaxDim = 24;  // 19 / 17 / 10 on this path; bounds the LDS staging buffer

static_assert(SF_NORM_GROUPS == 256, "sf_internal.h");
constexpr int kReduceGroups = SF_NORM_GROUPS;  // workgroups of the reduction = rows of partial sums

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

// stats: [0,D) mean, [D,2D) var, [2D] ret mean, [2D+1] ret var, [2D+2] ob count, [2D+3] ret count
// partials: [kReduceGroups][2 * (dim + 1)] doubles -- per workgroup: sum per feature, sum of returns, sum of
// squares per feature, sum of squared returns
template <typename T>
__global__ __launch_bounds__(256) void sf_norm_reduce_kernel(const T* obs, const int32_t* rew, double* ret, int n, int dim,
                                                             double gamma, int do_ob, int do_ret, double* partials) {
  __shared__ T stage[4][64 * kMaxDim];
  __shared__ double wsum[4][2 * (kMaxDim + 1)];
  __shared__ double fold[4][2][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long n_chunks = ((long)n + 63) / 64;
  const int groups = 64 / dim;
  double s = 0, q = 0;    // lane g*dim + f: feature f, rows g, g + groups, ...
  double rs = 0, rq = 0;  // every lane: its envs' returns
  for (long chunk = (long)blockIdx.x * 4 + wave; chunk < n_chunks; chunk += (long)gridDim.x * 4) {
    const long row0 = chunk * 64;
    const int rows = (int)min((long)64, (long)n - row0);
    if (do_ob) {
      const T* src = obs + row0 * dim;
      for (int i = lane; i < rows * dim; i += 64) stage[wave][i] = src[i];
      __builtin_amdgcn_wave_barrier();
      if (lane < groups * dim)
        for (int r = lane / dim; r < rows; r += groups) {
          const double v = (double)stage[wave][r * dim + lane % dim];
          s += v;
          q += v * v;
        }
      __builtin_amdgcn_wave_barrier();
    }
    if (do_ret && lane < rows) {
      const double v = ret[row0 + lane] * gamma + (double)rew[row0 + lane];
      ret[row0 + lane] = v;
      rs += v;
      rq += v * v;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    rs += __shfl_xor(rs, o);
    rq += __shfl_xor(rq, o);
  }
  fold[wave][0][lane] = s;
  fold[wave][1][lane] = q;
  __builtin_amdgcn_wave_barrier();
  if (lane < dim) {
    double ts = 0, tq = 0;
    for (int g = 0; g < groups; g++) {
      ts += fold[wave][0][g * dim + lane];
      tq += fold[wave][1][g * dim + lane];
    }
    wsum[wave][lane] = ts;
    wsum[wave][dim + 1 + lane] = tq;
  }
  if (lane == 0) {
    wsum[wave][dim] = rs;
    wsum[wave][2 * dim + 1] = rq;
  }
  __syncthreads();
  const int t = threadIdx.x, width = 2 * (dim + 1);
  // column-major: partials[c][workgroup], so that the merge reads each column as one contiguous row
  if (t < width) partials[(size_t)t * kReduceGroups + blockIdx.x] = (wsum[0][t] + wsum[1][t]) + (wsum[2][t] + wsum[3][t]);
}

// RunningMeanStd.update per feature and for the returns.  (Tried and dropped: folding this into the reduction
// behind a "last workgroup" ticket -- 256 atomics on one address cost 11 us; a single workgroup walking 1024 rows
// of every column -- its 160 dependent loads per thread cost 50 us.)
__global__ __launch_bounds__(256) void sf_norm_merge_kernel(const double* partials, int rows, int n, int dim, int do_ob,
                                                            int do_ret, const double* stats, double* stats_next) {
  // workgroup f < dim: feature f; workgroup dim: the returns.  Its two columns (sum, sum of squares) are `rows`
  // contiguous doubles each (256 from sf_norm_reduce_kernel, one per wave of the step kernel otherwise): every
  // thread adds its share with all loads in flight, a butterfly and four LDS words finish -- a fixed order
  __shared__ double part[2][4];
  const int f = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const bool is_ret = f == dim;
  const double* cs = partials + (size_t)f * rows;
  const double* cq = partials + (size_t)(dim + 1 + f) * rows;
  // the running statistics this workgroup merges into: fetched now, with the column loads, not after the
  // reduction (thread 0 would wait a memory round trip of its own for them)
  const double mean0 = is_ret ? stats[2 * dim] : stats[f], var0 = is_ret ? stats[2 * dim + 1] : stats[dim + f];
  const double count0 = stats[2 * dim + 2 + (is_ret ? 1 : 0)];
  double s = 0, q = 0;
#pragma unroll 4
  for (int r = t; r < rows; r += 256) {
    s += cs[r];
    q += cq[r];
  }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    q += __shfl_xor(q, o);
  }
  if (lane == 0) {
    part[0][wave] = s;
    part[1][wave] = q;
  }
  __syncthreads();
  if (t != 0) return;
  const double sum = (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]);
  const double sumsq = (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]);
  double mean = mean0, var = var0, count = count0;
  if (is_ret ? do_ret != 0 : do_ob != 0) merge(mean, var, count, sum, sumsq, (double)n, &mean, &var, &count);
  if (is_ret) {
    stats_next[2 * dim] = mean;
    stats_next[2 * dim + 1] = var;
    stats_next[2 * dim + 3] = count;
  } else {
    stats_next[f] = mean;
    stats_next[dim + f] = var;
    if (f == 0) stats_next[2 * dim + 2] = count;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void sf_norm_apply_kernel(const T* obs, T* obs_out, const int32_t* rew, float* rew_out,
                                                            int n, int dim, double eps, double clipob, double cliprew,
                                                            const double* stats) {
  __shared__ double s_mean[kMaxDim + 1], s_inv[kMaxDim + 1];
  const int t = threadIdx.x;
  if (t <= dim) {
    const bool is_ret = t == dim;
    s_mean[t] = is_ret ? stats[2 * dim] : stats[t];
    s_inv[t] = 1.0 / sqrt((is_ret ? stats[2 * dim + 1] : stats[dim + t]) + eps);
  }
  __syncthreads();
  if (obs_out) {
    // a thread walks a column of the flattened [n][dim] array in steps of the grid size, which is a
    // multiple of dim: its feature index never changes (no modulo in the loop)
    const long total = (long)n * dim, stride = (long)gridDim.x * 256;
    const long first = (long)blockIdx.x * 256 + t;
    const int f = (int)(first % dim);
    const double mean = s_mean[f], inv = s_inv[f];
    for (long i = first; i < total; i += stride) {
      double v = ((double)obs[i] - mean) * inv;
      v = v < -clipob ? -clipob : (v > clipob ? clipob : v);
      obs_out[i] = (T)v;
    }
  }
  if (rew && rew_out)
    for (long i = (long)blockIdx.x * 256 + t; i < n; i += (long)gridDim.x * 256) {
      double v = (double)rew[i] * s_inv[dim];  // rews / sqrt(ret_rms.var + eps): no mean subtraction
      v = v < -cliprew ? -cliprew : (v > cliprew ? cliprew : v);
      rew_out[i] = (float)v;
    }
}

}  // namespace

// the tail of sf_step_normalize: the step kernel has already left `rows` rows of partial sums (one per wave)
hipError_t sf_launch_normalize_after_step(const void* obs, void* obs_out, int obs_f64, const int32_t* rew, float* rew_out, int n,
                                          int dim, double eps, double clipob, double cliprew, int do_ob, int do_ret,
                                          const double* partials, int rows, const double* stats, double* stats_next,
                                          hipStream_t stream) {
  if (n <= 0 || dim <= 0 || dim > kMaxDim) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sf_norm_merge_kernel, dim3(dim + 1), dim3(256), 0, stream, partials, rows, n, dim, do_ob, do_ret, stats, stats_next);
  long g2 = ((long)n * dim + 255) / 256;
  g2 = g2 > 2048 ? 2048 : g2;
  g2 = (g2 + dim - 1) / dim * dim;
  if (obs_f64)
    hipLaunchKernelGGL(sf_norm_apply_kernel<double>, dim3((unsigned)g2), dim3(256), 0, stream, (const double*)obs,
                       (double*)obs_out, rew, rew_out, n, dim, eps, clipob, cliprew, stats_next);
  else
    hipLaunchKernelGGL(sf_norm_apply_kernel<float>, dim3((unsigned)g2), dim3(256), 0, stream, (const float*)obs,
                       (float*)obs_out, rew, rew_out, n, dim, eps, clipob, cliprew, stats_next);
  return hipGetLastError();
}

hipError_t sf_launch_normalize(const void* obs, void* obs_out, int obs_f64, const int32_t* rew, float* rew_out, double* ret,
                               int n, int dim, double gamma, double eps, double clipob, double cliprew, int do_ob,
                               int do_ret, double* partials, const double* stats, double* stats_next,
                               hipStream_t stream) {
  if (n <= 0 || dim <= 0 || dim > kMaxDim) return hipErrorInvalidValue;
  const bool update = do_ob || do_ret;
  const double* use = update ? stats_next : stats;  // frozen: normalise with the statistics as they are
  // the apply grid is a multiple of dim workgroups so that (grid * 256) % dim == 0
  long g2 = ((long)n * dim + 255) / 256;
  g2 = g2 > 2048 ? 2048 : g2;
  g2 = (g2 + dim - 1) / dim * dim;
  if (obs_f64) {
    if (update) {
      hipLaunchKernelGGL(sf_norm_reduce_kernel<double>, dim3(kReduceGroups), dim3(256), 0, stream, (const double*)obs, rew,
                         ret, n, dim, gamma, do_ob, do_ret, partials);
      hipLaunchKernelGGL(sf_norm_merge_kernel, dim3(dim + 1), dim3(256), 0, stream, partials, kReduceGroups, n, dim, do_ob, do_ret, stats, stats_next);
    }
    hipLaunchKernelGGL(sf_norm_apply_kernel<double>, dim3((unsigned)g2), dim3(256), 0, stream, (const double*)obs,
                       (double*)obs_out, rew, rew_out, n, dim, eps, clipob, cliprew, use);
  } else {
    if (update) {
      hipLaunchKernelGGL(sf_norm_reduce_kernel<float>, dim3(kReduceGroups), dim3(256), 0, stream, (const float*)obs, rew, ret,
                         n, dim, gamma, do_ob, do_ret, partials);
      hipLaunchKernelGGL(sf_norm_merge_kernel, dim3(dim + 1), dim3(256), 0, stream, partials, kReduceGroups, n, dim, do_ob, do_ret, stats, stats_next);
    }
    hipLaunchKernelGGL(sf_norm_apply_kernel<float>, dim3((unsigned)g2), dim3(256), 0, stream, (const float*)obs,
                       (float*)obs_out, rew, rew_out, n, dim, eps, clipob, cliprew, use);
  }
  return hipGetLastError();
}
