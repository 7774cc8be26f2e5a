"""TEST INFRASTRUCTURE ONLY -- numpy restatement of gym_vecenv.VecNormalize.

The trainer applies it whenever the observation is 1-D (rl/train.py:35-36).  gym-vecenv==1.0
(requirements.txt:4, upstream named in README.md:38) is NOT in /root/reference and not installed here;
it is a repackaging of OpenAI baselines' common/vec_env/vec_normalize.py and common/running_mean_std.py
as of early 2018, whose published algorithm is restated below.  PARITY UNPINNED against the package
itself (no source, no fixtures); the device implementation (sf_normalize.hip) is pinned to this file.
"""
import numpy as np


class RunningMeanStd:
    def __init__(self, epsilon=1e-4, shape=()):
        self.mean = np.zeros(shape, np.float64)
        self.var = np.ones(shape, np.float64)
        self.count = epsilon

    def update(self, x):
        batch_mean, batch_var, batch_count = np.mean(x, axis=0), np.var(x, axis=0), x.shape[0]
        delta = batch_mean - self.mean
        tot = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot
        m2 = self.var * self.count + batch_var * batch_count + np.square(delta) * self.count * batch_count / tot
        self.mean, self.var, self.count = new_mean, m2 / tot, tot


class VecNormalize:
    """Filter form: feed it what the wrapped vec-env returned."""

    def __init__(self, num_envs, obs_shape, ob=True, ret=True, clipob=10., cliprew=10., gamma=0.99, epsilon=1e-8):
        self.ob_rms = RunningMeanStd(shape=obs_shape) if ob else None
        self.ret_rms = RunningMeanStd(shape=()) if ret else None
        self.clipob, self.cliprew, self.gamma, self.epsilon = clipob, cliprew, gamma, epsilon
        self.ret = np.zeros(num_envs)

    def _obfilt(self, obs):
        if self.ob_rms:
            self.ob_rms.update(obs)
            return np.clip((obs - self.ob_rms.mean) / np.sqrt(self.ob_rms.var + self.epsilon), -self.clipob, self.clipob)
        return obs

    def step(self, obs, rews):
        self.ret = self.ret * self.gamma + rews
        obs = self._obfilt(obs)
        if self.ret_rms:
            self.ret_rms.update(self.ret)
            rews = np.clip(rews / np.sqrt(self.ret_rms.var + self.epsilon), -self.cliprew, self.cliprew)
        return obs, rews

    def reset(self, obs):
        return self._obfilt(obs)
