"""DeviceRollout -- RolloutStorage (rl/storage.py:9-64) plus the trainer's per-step bookkeeping
(rl/train.py:74-98), resident on the device.

The reference moves every step through the host: `action.cpu().numpy()` -> SubprocVecEnv pipes ->
`torch.from_numpy(reward)`, five small tensor ops for the episode bookkeeping, `rollouts.insert(...)`.
Here `sf_step` writes the next observation straight into `observations[step + 1]`, one helper launch
(`sf_record_step`) produces `rewards[step]`, `masks[step + 1]` and the episode / final reward
accumulators, and `compute_returns` is one backward-scan kernel (`sf_compute_returns`, bit-identical to
the reference's float32 arithmetic).  Tensor names and shapes are RolloutStorage's.
"""
import ctypes as C

import torch

from . import _lib


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class DeviceRollout:
    def __init__(self, env, num_steps, state_size=1):
        self.env = env
        self._L = _lib.lib()
        T, n, dev = int(num_steps), env.num_envs, env.device
        self.num_steps = T
        obs_shape = tuple(env.obs_shape)
        self.observations = torch.zeros((T + 1, n) + obs_shape, dtype=env.obs_dtype, device=dev)
        self.states = torch.zeros(T + 1, n, state_size, device=dev)
        self.rewards = torch.zeros(T, n, 1, device=dev)
        self.value_preds = torch.zeros(T + 1, n, 1, device=dev)
        self.returns = torch.zeros(T + 1, n, 1, device=dev)
        self.action_log_probs = torch.zeros(T, n, 1, device=dev)
        self.actions = torch.zeros(T, n, 1, dtype=torch.long, device=dev)  # Discrete (rl/storage.py:17-23)
        self.masks = torch.ones(T + 1, n, 1, device=dev)
        self.episode_rewards = torch.zeros(n, 1, device=dev)  # rl/train.py:64-65
        self.final_rewards = torch.zeros(n, 1, device=dev)
        self.num_destruction = torch.zeros((), dtype=torch.long, device=dev)
        self._rew = torch.empty(n, dtype=torch.int32, device=dev)
        self._done = torch.empty(n, dtype=torch.uint8, device=dev)
        self._info = torch.empty(n, dtype=torch.uint8, device=dev)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.env.device).cuda_stream)

    def reset(self):
        """obs = envs.reset(); rollouts.observations[0].copy_(obs) (rl/train.py:60-62)."""
        e = self.env
        _lib.check(e._L.sf_reset(e._h, _p(self.observations[0]), e._stream()))
        return self.observations[0]

    def step(self, step, action, value_pred=None, action_log_prob=None, state=None):
        """envs.step(action) + the bookkeeping + rollouts.insert(...) of rl/train.py:79-98.
        `action`: [N] or [N, 1] integer tensor on the device."""
        e = self.env
        a = action.reshape(-1)
        if a.dtype not in (torch.uint8, torch.int32, torch.int64):
            a = a.long()
        a = a.contiguous()
        e.step_tensors(a, out=(self.observations[step + 1], self._rew, self._done, self._info))
        _lib.check(self._L.sf_record_step(e.num_envs, _p(self._rew), _p(self._done), _p(self.rewards[step]),
                                          _p(self.masks[step + 1]), _p(self.episode_rewards), _p(self.final_rewards),
                                          self._stream()))
        self.num_destruction += self._info.sum()  # num_destruction += sum(info), rl/train.py:81
        self.actions[step].copy_(a.view(-1, 1))
        if value_pred is not None:
            self.value_preds[step].copy_(value_pred)
        if action_log_prob is not None:
            self.action_log_probs[step].copy_(action_log_prob)
        if state is not None:
            self.states[step + 1].copy_(state)
        return self.observations[step + 1], self.rewards[step], self.masks[step + 1]

    def compute_returns(self, next_value, use_gae, gamma, tau):
        """rl/storage.py:50-63 in one launch."""
        n = self.env.num_envs
        nv = next_value.reshape(n).float().contiguous()
        _lib.check(self._L.sf_compute_returns(self.num_steps, n, _p(self.rewards), _p(self.value_preds), _p(self.masks),
                                              _p(nv), _p(self.returns), int(bool(use_gae)), float(gamma), float(tau),
                                              self._stream()))

    def after_update(self):
        """rl/storage.py:45-48."""
        self.observations[0].copy_(self.observations[-1])
        self.states[0].copy_(self.states[-1])
        self.masks[0].copy_(self.masks[-1])
